/*
 * sea_tables.c -- host-side construction of the engine's constant tables.
 *
 * MUST be compiled as C with gcc -O2 -ffp-contract=off (csrc/Makefile does): the values have to
 * be bit-identical to what the reference's init code computes, and that depends on which
 * sub-expressions are float and which are double.  Each builder cites the reference code whose
 * arithmetic it reproduces; the layouts (lane-major, padded, bit-reversed) are this engine's own.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "sea_tables.h"

static const double kTwoPi = 6.28318530717958647692; /* PIx2, etsi/cpp/ParmInterface.h:45 */
static const double kPi = 3.14159265358979323846;    /* M_PI, etsi/cpp/rfft.h:16; PI CompCeps.c:31 */

/* ------------------------------------------------------------------------------------------
 * Split-radix schedule for N = 256 (etsi/cpp/rfft.c:45-180)
 * ---------------------------------------------------------------------------------------- */
static unsigned bitrev(unsigned v, int bits)
{
    unsigned r = 0;
    int b;
    for (b = 0; b < bits; b++)
        if (v & (1u << b)) r |= 1u << (bits - 1 - b);
    return r;
}

/* Calls visit(base, ctx) for every block start the reference's "is/id" double loop selects for
 * block length n2 at FFT size n (rfft.c:107-130 and its twins). */
static void for_each_block(int n, int n2, int first_is_adjust, void (*visit)(int, void *), void *ctx)
{
    int is = 0, id = n2 << 1, i;
    while (is < n - first_is_adjust) {
        for (i = is; i < n; i += id) visit(i, ctx);
        is = (id << 1) - n2;
        id <<= 2;
    }
}

typedef struct {
    unsigned char mark[SEA_NFFT];
} mark_ctx;
static void mark_visit(int i, void *ctx) { ((mark_ctx *)ctx)->mark[i] = 1; }

typedef struct {
    int base[64], n;
} list_ctx;
static void list_visit(int i, void *ctx)
{
    list_ctx *l = (list_ctx *)ctx;
    l->base[l->n++] = i;
}

static void build_fft(sea_fft_tables *f)
{
    mark_ctx len2, len4;
    int lane, s;
    memset(f, 0, sizeof *f);
    memset(&len2, 0, sizeof len2);
    memset(&len4, 0, sizeof len4);
    /* length-two butterflies: rfft.c:82-96 (is=0,id=4; is=2id-2) == blocks of "n2 = 2" */
    for_each_block(SEA_NFFT, 2, 1, mark_visit, &len2);
    /* first L level, n2 = 4: only the plain butterfly exists (n4 == 1): rfft.c:100-113 */
    for_each_block(SEA_NFFT, 4, 0, mark_visit, &len4);
    for (lane = 0; lane < SEA_LANES; lane++) {
        int g = 4 * (int)bitrev((unsigned)lane, 6);
        f->fftFlags[lane] = (len2.mark[g] ? 1u : 0u) | (len2.mark[g + 2] ? 2u : 0u) | (len4.mark[g] ? 4u : 0u);
    }
    /* levels n2 = 8 .. 256: one work item per lane, twiddle items first */
    for (s = 0; s < SEA_FFT_LSTAGES; s++) {
        int n2 = 8 << s, n4 = n2 >> 2, n8 = n2 >> 3, b, j, slot = 0;
        float e = (float)((kPi * 2) / n2); /* rfft.c:105 */
        list_ctx blocks;
        blocks.n = 0;
        for_each_block(SEA_NFFT, n2, 0, list_visit, &blocks);
        for (j = 1; j < n8; j++) {
            float a = j * e, a3 = 3 * a; /* rfft.c:133-138: float angles, double cos/sin */
            float cc1 = (float)cos((double)a), ss1 = (float)sin((double)a);
            float cc3 = (float)cos((double)a3), ss3 = (float)sin((double)a3);
            for (b = 0; b < blocks.n; b++, slot++) {
                unsigned i1 = (unsigned)(blocks.base[b] + j), i5 = (unsigned)(blocks.base[b] + n4 - j);
                f->fftItem[s][slot] = ((unsigned)SEA_BF_TWIDDLE << 16) | (i5 << 8) | i1;
                f->fftTw[s][0][slot] = cc1;
                f->fftTw[s][1][slot] = ss1;
                f->fftTw[s][2][slot] = cc3;
                f->fftTw[s][3][slot] = ss3;
            }
        }
        /* plain / pi4 items use only i1 (+ multiples of n4); b repeats a so that the kernel's
         * unconditional operand fetch stays inside the frame */
        for (b = 0; b < blocks.n; b++, slot++) {
            unsigned i1 = (unsigned)blocks.base[b];
            f->fftItem[s][slot] = ((unsigned)SEA_BF_PLAIN << 16) | (i1 << 8) | i1;
        }
        for (b = 0; b < blocks.n; b++, slot++) {
            unsigned i1 = (unsigned)(blocks.base[b] + n8);
            f->fftItem[s][slot] = ((unsigned)SEA_BF_PI4 << 16) | (i1 << 8) | i1;
        }
        if (slot > SEA_LANES) abort();
        /* half-wave packing: twiddle items, then one PAIR item (plain + pi/4) per block */
        slot = 0;
        for (j = 1; j < n8; j++) {
            float a = j * e, a3 = 3 * a;
            float cc1 = (float)cos((double)a), ss1 = (float)sin((double)a);
            float cc3 = (float)cos((double)a3), ss3 = (float)sin((double)a3);
            for (b = 0; b < blocks.n; b++, slot++) {
                unsigned i1 = (unsigned)(blocks.base[b] + j), i5 = (unsigned)(blocks.base[b] + n4 - j);
                if (slot >= 32) abort();
                f->fft2Item[s][slot] = ((unsigned)SEA_BF_TWIDDLE << 16) | (i5 << 8) | i1;
                f->fft2Tw[s][0][slot] = cc1;
                f->fft2Tw[s][1][slot] = ss1;
                f->fft2Tw[s][2][slot] = cc3;
                f->fft2Tw[s][3][slot] = ss3;
            }
        }
        for (b = 0; b < blocks.n; b++, slot++) {
            unsigned i1 = (unsigned)blocks.base[b], ip = (unsigned)(blocks.base[b] + n8);
            if (slot >= 32) abort();
            f->fft2Item[s][slot] = ((unsigned)SEA_BF_PAIR << 16) | (ip << 8) | i1;
        }
        /* swizzled byte addresses of the eight operands of every full-wave item */
        for (j = 0; j < SEA_LANES; j++) {
            unsigned it = f->fftItem[s][j], a = it & 255u, bb = (it >> 8) & 255u, k;
            for (k = 0; k < 4; k++) {
                unsigned e0 = (k < 2 ? a : bb) + (unsigned)((2 * k) & 3) * (unsigned)n4;
                unsigned e1 = (k < 2 ? a : bb) + (unsigned)((2 * k + 1) & 3) * (unsigned)n4;
                if (e0 > 255u || e1 > 255u) abort();
                f->fftAddr[s][k][j] = (sea_fft_swizzle(e0) * 4u) | ((sea_fft_swizzle(e1) * 4u) << 16);
            }
        }
        /* swizzled byte addresses of the eight operands of every half-wave item (idle slots: item 0 ->
         * element 0, fetched and ignored) */
        for (j = 0; j < 32; j++) {
            unsigned it = f->fft2Item[s][j], a = it & 255u, bb = (it >> 8) & 255u, k;
            for (k = 0; k < 4; k++) {
                unsigned e0 = (k < 2 ? a : bb) + (unsigned)((2 * k) & 3) * (unsigned)n4;
                unsigned e1 = (k < 2 ? a : bb) + (unsigned)((2 * k + 1) & 3) * (unsigned)n4;
                if (e0 > 255u || e1 > 255u) abort();
                f->fft2Addr[s][k][j] = (sea_fft_swizzle(e0) * 4u) | ((sea_fft_swizzle(e1) * 4u) << 16);
            }
        }
    }
    for (lane = 0; lane < SEA_LANES; lane++) {
        unsigned g = 4u * bitrev((unsigned)lane, 6), l = (unsigned)lane;
        f->fft2Head[0][lane] = (sea_fft_swizzle(g) * 4u) | ((sea_fft_swizzle(g + 1) * 4u) << 16);
        f->fft2Head[1][lane] = (sea_fft_swizzle(g + 2) * 4u) | ((sea_fft_swizzle(g + 3) * 4u) << 16);
        f->fft2Psd[0][lane] = (sea_fft_swizzle(2 * l) * 4u) | ((sea_fft_swizzle(2 * l + 1) * 4u) << 16);
        f->fft2Psd[1][lane] = (sea_fft_swizzle(255 - 2 * l) * 4u) | ((sea_fft_swizzle((256 - 2 * l) & 255u) * 4u) << 16);
    }
    f->fft2Nyq = sea_fft_swizzle(128) * 4u;
    {   /* eight positions per lane: flags of the three register-resident stages and the store addresses */
        mark_ctx len8;
        memset(&len8, 0, sizeof len8);
        for_each_block(SEA_NFFT, 8, 0, mark_visit, &len8);
        for (lane = 0; lane < SEA_LANES; lane++) {
            unsigned g = 8u * bitrev((unsigned)lane & 31u, 5), q, fl = 0;
            for (q = 0; q < 4; q++) {
                if (len2.mark[g + 2 * q]) fl |= 1u << q;
                f->fft8Addr[q][lane] = (sea_fft_swizzle(g + 2 * q) * 4u) | ((sea_fft_swizzle(g + 2 * q + 1) * 4u) << 16);
            }
            if (len4.mark[g]) fl |= 16u;
            if (len4.mark[g + 4]) fl |= 32u;
            if (len8.mark[g]) fl |= 64u;
            f->fft8Flags[lane] = fl;
        }
    }
    {   /* the n2 = 16 level on registers: see sea_tables.h, fft16* */
        static const unsigned evenPos[8] = {0, 4, 8, 12, 2, 6, 10, 14}, oddPos[8] = {1, 5, 9, 13, 3, 7, 11, 15};
        mark_ctx len16;
        memset(&len16, 0, sizeof len16);
        for_each_block(SEA_NFFT, 16, 0, mark_visit, &len16);
        for (lane = 0; lane < SEA_LANES; lane++) {
            unsigned k = bitrev((unsigned)lane & 31u, 5), blk = 16u * (k >> 1), q;
            const unsigned *pos = (k & 1u) ? oddPos : evenPos;
            if ((k & 1u) != (((unsigned)lane >> 4) & 1u)) abort(); /* the second half of a 16-block sits 16 lanes up */
            f->fft16Flags[lane] = f->fft8Flags[lane] | (len16.mark[blk] ? 128u : 0u);
            for (q = 0; q < 4; q++)
                f->fft16Addr[q][lane] = (sea_fft_swizzle(blk + pos[2 * q]) * 4u) | ((sea_fft_swizzle(blk + pos[2 * q + 1]) * 4u) << 16);
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * Mel filter banks
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int start, len;
    float w[32];
} band_t;

static float hz_to_mel(float hz) { return (float)(2595.0 * log10(1.0 + hz / 700.0)); }

/* mel-spaced FFT bin for fraction num/den of the way from lo_mel to hi_mel:
 * MelProc.c:147-150 and :421-424 */
static int mel_bin(float lo_mel, float hi_mel, int num, int den, int nfft, float fs)
{
    float mel = lo_mel + (float)num / den * (hi_mel - lo_mel);
    float hz = (float)(700 * (pow(10, mel / 2595.0) - 1.0));
    return (int)(nfft * hz / fs + 0.5);
}

/* 25 normalised triangles for the Wiener filter design: InitMelFBwindows(.., 0.0, 8000, 128, 25, 1)
 * called at NoiseSup.c:988; arithmetic of MelProc.c:128-230 */
static void wiener_bands(band_t *B)
{
    const float fs = 8000.0f;
    int c[SEA_NMEL], i, j;
    float lo = hz_to_mel(0.0f), hi = hz_to_mel(fs / 2);
    for (i = 0; i < SEA_NMEL; i++) c[i] = mel_bin(lo, hi, i, SEA_NMEL - 1, 128, fs);
    for (i = 0; i < SEA_NMEL; i++) {
        band_t *b = &B[i];
        float area = 0.0f;
        int rise = (i > 0) ? c[i] - c[i - 1] : 0;
        int fall = (i < SEA_NMEL - 1) ? c[i + 1] - c[i] : 0;
        int n = 0;
        memset(b, 0, sizeof *b);
        if (i == 0) { /* falling edge only, starts at its own centre */
            b->start = c[0];
            for (j = 0; j < fall; j++) b->w[n++] = (float)(1.0 - (float)j / (float)fall);
        } else {
            b->start = c[i - 1] + 1;
            for (j = 0; j < rise; j++) b->w[n++] = (float)(j + 1) / (float)rise;
            for (j = 0; j < fall - 1; j++) b->w[n++] = (float)(1.0 - (j + 1) / (float)fall);
        }
        b->len = n;
        for (j = 0; j < n; j++) area += b->w[j];
        for (j = 0; j < n; j++) b->w[j] /= area;
    }
}

/* 23 un-normalised triangles for the cepstrum: InitFFTWindows(.., 64.0, 8000, 256, 23) +
 * ComputeTriangle, called at CompCeps.c:283-286; arithmetic of MelProc.c:402-522 */
static void cepstral_bands(band_t *B)
{
    const float fs = 8000.0f;
    float lo = hz_to_mel(64.0f), hi = hz_to_mel(fs / 2);
    int i, j, top_prev = 0;
    for (i = 0; i < SEA_CC_NCHAN; i++) {
        memset(&B[i], 0, sizeof B[i]);
        B[i].start = mel_bin(lo, hi, i, SEA_CC_NCHAN + 1, 256, fs);
        B[i].len = mel_bin(lo, hi, i + 2, SEA_CC_NCHAN + 1, 256, fs) - B[i].start + 1;
    }
    for (i = 0; i < SEA_CC_NCHAN; i++) {
        band_t *b = &B[i];
        int up = ((i + 1 < SEA_CC_NCHAN) ? B[i + 1].start : top_prev) - b->start + 1;
        int down = b->len - up + 1;
        for (j = 0; j < up; j++) b->w[j] = (float)(j + 1) / up;
        for (j = 1; j < down; j++) b->w[up + j - 1] = (float)(down - j) / down;
        top_prev = b->start + b->len - 1;
    }
}

/* mel-warped inverse DCT basis: InitMelIDCTbasis(.., 25, 8000, 128), MelProc.c:283-337 */
static void wiener_idct(const band_t *B, float basis[SEA_NMEL][SEA_NMEL])
{
    const int fs = 8000;
    const float step = fs / (float)128;
    float centre[SEA_NMEL], width[SEA_NMEL];
    int f, t, i;
    for (f = 0; f < SEA_NMEL; f++) {
        if (f == 0)
            centre[f] = B[f].start * step;
        else if (f == SEA_NMEL - 1)
            centre[f] = (B[f].start + B[f].len - 1) * step;
        else {
            float origin = B[f].start * step, mass = 0.0f, moment = 0.0f;
            for (i = 0; i < B[f].len; i++) {
                moment += B[f].w[i] * (origin + i * step);
                mass += B[f].w[i];
            }
            centre[f] = moment / mass;
        }
    }
    for (f = 0; f < SEA_NMEL; f++) {
        int a = (f == 0) ? 0 : f - 1, b = (f == SEA_NMEL - 1) ? f : f + 1;
        width[f] = (centre[b] - centre[a]) / fs;
    }
    for (t = 0; t < SEA_NMEL; t++)
        for (f = 0; f < SEA_NMEL; f++) basis[t][f] = (float)(width[f] * cos(kTwoPi * t * centre[f] / fs));
}

static float hanning(int i, int n)
{ /* NoiseSup.c:975 / :979 */
    return (float)(0.5 - 0.5 * cos((kTwoPi * ((float)i + 0.5)) / (float)(short)n));
}

static float hamming_half(int i)
{ /* CompCeps.c:92-93, i < 100 */
    return (float)(0.54 - 0.46 * cos(kTwoPi * (i + 0.5) / (short)SEA_WIN));
}

/* ------------------------------------------------------------------------------------------
 * Public builders
 * ---------------------------------------------------------------------------------------- */
void sea_build_ns_tables(sea_ns_tables *t)
{
    band_t B[SEA_NMEL];
    float basis[SEA_NMEL][SEA_NMEL];
    int lane, k, f, j;
    memset(t, 0, sizeof *t);
    build_fft(&t->fft);
    wiener_bands(B);
    wiener_idct(B, basis);
    for (lane = 0; lane < SEA_LANES; lane++)
        for (k = 0; k < 4; k++) {
            int i = lane + 64 * k;
            t->win[k][lane] = (i < SEA_WIN) ? hanning(i, SEA_WIN) : 0.0f;
        }
    for (lane = 0; lane < SEA_LANES; lane++)
        for (k = 0; k < 8; k++) {
            int i = (lane & 31) + 32 * (int)bitrev((unsigned)k, 3);
            t->win8[k][lane] = (i < SEA_WIN) ? hanning(i, SEA_WIN) : 0.0f;
        }
    for (f = 0; f < SEA_NMEL; f++) {
        if (B[f].len > SEA_MEL_TAPS) abort();
        t->melStart[f] = B[f].start;
        t->melLen[f] = B[f].len;
        for (j = 0; j < B[f].len; j++) t->melW[j][f] = B[f].w[j];
    }
    for (f = 0; f < SEA_NMEL; f++)
        for (lane = 0; lane <= 8; lane++) t->idct[f][lane] = basis[lane][f];
    for (lane = 0; lane <= 8; lane++) t->irWin[lane] = hanning(8 + lane, SEA_NTAP);
    t->eps = (float)exp(-10.0); /* NS_EPS, NoiseSup.h:32 */
}

/* lane map of the tiled kernels' mel pass (sea_tables.h, melLaneBase): items = (frame h of the pair, band b); item i may sit in
 * lane group g = 0 / 1 (lanes 0-31 / 32-63: the two halves a ds_read_b64 is served in) with its first bin at an even
 * base <= start(b), as long as base + SEA_CC_TAPS2 still covers the band; the pair of banks it then reads is
 * ((base + SEA_CC_PWROW h) / 2) mod 32 (+ i for tap pair i: the same shift for every lane).  Augmenting-path matching of the
 * items onto the 2 x 32 (group, bank pair) slots; aborts if the 46 items do not all find one (they do: checked at build). */
typedef struct { int h, b, slot, base; } mel_item_t;
static int mel_try(const band_t *B, mel_item_t *it, int n, int i, int *owner, char *seen)
{
    int base, g;
    for (base = B[it[i].b].start & ~1; base >= 0 && B[it[i].b].start - base + B[it[i].b].len <= SEA_CC_TAPS2; base -= 2)
        for (g = 0; g < 2; g++) {
            int slot = 32 * g + ((base + SEA_CC_PWROW * it[i].h) / 2) % 32;
            if (seen[slot]) continue;
            seen[slot] = 1;
            if (owner[slot] < 0 || mel_try(B, it, n, owner[slot], owner, seen)) {
                owner[slot] = i;
                it[i].slot = slot;
                it[i].base = base;
                return 1;
            }
        }
    return 0;
}

static void cc_mel_lanes(const band_t *B, sea_cc_tables *t)
{
    mel_item_t it[2 * SEA_CC_NCHAN];
    int owner[64], n = 0, i, j, g, fill[2] = {0, 0};
    char seen[64];
    for (i = 0; i < 64; i++) owner[i] = -1;
    for (i = 0; i < 2; i++)
        for (j = 0; j < SEA_CC_NCHAN; j++) { it[n].h = i; it[n].b = j; it[n].slot = -1; it[n].base = 0; n++; }
    for (i = 0; i < n; i++) {
        memset(seen, 0, sizeof seen);
        if (!mel_try(B, it, n, i, owner, seen)) abort();
    }
    for (i = 0; i < SEA_LANES; i++) { t->melLaneBase[i] = 0; t->melLaneFb[i] = -1; }
    for (i = 0; i < n; i++) {
        const band_t *b = &B[it[i].b];
        int lane, lead = b->start - it[i].base;
        g = it[i].slot / 32;
        if (fill[g] >= 32 || (SEA_CC_PWROW & 1)) abort();
        lane = 32 * g + fill[g]++;
        t->melLaneBase[lane] = it[i].base + SEA_CC_PWROW * it[i].h;
        t->melLaneFb[lane] = 24 * it[i].h + it[i].b;
        for (j = 0; j < b->len; j++) t->melLaneW[lead + j][lane] = b->w[j];
    }
    for (g = 0; g < 2; g++) { /* no two lanes of a group on one pair of banks */
        memset(seen, 0, sizeof seen);
        for (i = 32 * g; i < 32 * g + fill[g]; i++) {
            int bank = (t->melLaneBase[i] / 2) % 32;
            if (seen[bank]) abort();
            seen[bank] = 1;
        }
    }
}

void sea_build_cc_tables(sea_cc_tables *t)
{
    band_t B[SEA_CC_NCHAN];
    int lane, k, f, j, i;
    memset(t, 0, sizeof *t);
    build_fft(&t->fft);
    cepstral_bands(B);
    for (lane = 0; lane < SEA_LANES; lane++)
        for (k = 0; k < 4; k++) {
            i = lane + 64 * k;
            t->win[k][lane] = (i < SEA_WIN) ? hamming_half(i < SEA_WIN / 2 ? i : SEA_WIN - 1 - i) : 0.0f;
        }
    for (f = 0; f < SEA_CC_NCHAN; f++) {
        if (B[f].len > SEA_CC_TAPS) abort();
        t->melStart[f] = B[f].start;
        t->melLen[f] = B[f].len;
        for (j = 0; j < B[f].len; j++) t->melW[j][f] = B[f].w[j];
    }
    /* InitDCTMatrix(13, 23): CompCeps.c:153-173 */
    for (i = 1; i <= 12; i++)
        for (j = 0; j < SEA_CC_NCHAN; j++)
            t->dct[j][i - 1] = (float)cos(kPi * (float)i / (float)SEA_CC_NCHAN * ((float)j + 0.5));
    t->floorFB = (float)exp((double)-10.0); /* CompCeps.c:405-406 */
    t->floorE = (float)exp((double)-50.0);
    for (lane = 0; lane < SEA_LANES; lane++)
        for (k = 0; k < 8; k++) {
            i = (lane & 31) + 32 * (int)bitrev((unsigned)k, 3);
            t->win8[k][lane] = (i < SEA_WIN) ? hamming_half(i < SEA_WIN / 2 ? i : SEA_WIN - 1 - i) : 0.0f;
        }
    for (j = 0; j < SEA_CC_NCHAN; j++) {
        for (i = 0; i < 12; i++) t->dctT[j][i] = t->dct[j][i];
        t->dctT[j][12] = 1.0f;
    }
    cc_mel_lanes(B, t);
}

void sea_ns_plain_tables(float *sigWindow200, float *irWindow17, float *idct25x25, int *melStart25,
                         int *melLen25, float *melData)
{
    band_t B[SEA_NMEL];
    float basis[SEA_NMEL][SEA_NMEL];
    int i, f;
    wiener_bands(B);
    wiener_idct(B, basis);
    for (i = 0; i < SEA_WIN; i++) sigWindow200[i] = hanning(i, SEA_WIN);
    for (i = 0; i < SEA_NTAP; i++) irWindow17[i] = hanning(i, SEA_NTAP);
    memcpy(idct25x25, basis, sizeof basis);
    for (f = 0; f < SEA_NMEL; f++) {
        melStart25[f] = B[f].start;
        melLen25[f] = B[f].len;
        for (i = 0; i < B[f].len && i < 16; i++) melData[f * 16 + i] = B[f].w[i];
    }
}

void sea_cc_plain_tables(float *hamming100, float *dct12x23, int *melStart23, int *melLen23, float *melData)
{
    band_t B[SEA_CC_NCHAN];
    int i, j, f;
    cepstral_bands(B);
    for (i = 0; i < SEA_WIN / 2; i++) hamming100[i] = hamming_half(i);
    for (i = 1; i <= 12; i++)
        for (j = 0; j < SEA_CC_NCHAN; j++)
            dct12x23[(i - 1) * SEA_CC_NCHAN + j] = (float)cos(kPi * (float)i / (float)SEA_CC_NCHAN * ((float)j + 0.5));
    for (f = 0; f < SEA_CC_NCHAN; f++) {
        melStart23[f] = B[f].start;
        melLen23[f] = B[f].len;
        for (i = 0; i < B[f].len && i < 32; i++) melData[f * 32 + i] = B[f].w[i];
    }
}

/* ------------------------------------------------------------------------------------------
 * Gammatone bank: resyth_64sub_ori/cpp/extractwav.cpp:41-54 (channel constants),
 * :133-166 + :258-278 (BS3383 middle-ear curve), :176-183 (filter coefficients),
 * :95-107 (raised-cosine overlap-add halves).  The reference is C++: exp/cos/sin of float
 * arguments resolve to the float overloads, pow() to double.
 * ---------------------------------------------------------------------------------------- */
static const float kEarF[29] = {20.0,  25.0,  31.5,   40.0,   50.0,   63.0,   80.0,   100.0,  125.0,  160.0,
                                200.0, 250.0, 315.0,  400.0,  500.0,  630.0,  800.0,  1000.0, 1250.0, 1600.0,
                                2000.0, 2500.0, 3150.0, 4000.0, 5000.0, 6300.0, 8000.0, 10000.0, 12500.0};
static const float kEarA[29] = {2.347, 2.190, 2.050, 1.879, 1.724, 1.579, 1.512, 1.466, 1.426, 1.394,
                                1.372, 1.344, 1.304, 1.256, 1.203, 1.135, 1.062, 1.000, 0.967, 0.943,
                                0.932, 0.933, 0.937, 0.952, 0.974, 1.027, 1.135, 1.266, 1.501};
static const float kEarB[29] = {0.00561,  0.00527,  0.00481,  0.00404,  0.00383, 0.00286, 0.00259, 0.00257,
                                0.00256,  0.00255,  0.00254,  0.00248,  0.00229, 0.00201, 0.00162, 0.00111,
                                0.00052,  0.00000,  -0.00039, -0.00067, -0.00092, -0.00105, -0.00104,
                                -0.00088, -0.00055, 0.00000,  0.00089,  0.00211, 0.00488};
static const float kEarT[29] = {74.3, 65.0, 56.3, 48.4, 41.7, 35.5, 29.8, 25.1, 20.7, 16.8, 13.8, 11.2, 8.9, 7.2, 6.0,
                                5.0,  4.4,  4.2,  3.7,  2.6,  1.0,  -1.2, -3.6, -3.9, -1.1, 6.6,  15.3, 16.4, 11.6};

static float lerp_tab(const float *tab, int hi, float frac) { return tab[hi - 1] + frac * (tab[hi] - tab[hi - 1]); }

static float phons_at(float hz)
{
    int hi = 0;
    float frac, a, b, t;
    while (kEarF[hi] < hz) hi++;
    frac = (hz - kEarF[hi - 1]) / (kEarF[hi] - kEarF[hi - 1]);
    a = lerp_tab(kEarA, hi, frac);
    b = lerp_tab(kEarB, hi, frac);
    t = lerp_tab(kEarT, hi, frac);
    return (float)(4.2 + a * (60.0 - t) / (1.0 + b * (60.0 - t)));
}

void sea_build_gt_tables(sea_gt_tables *t)
{
    const double kPiHW = 3.1415926535897932384626433832795; /* HuWang.h:7 */
    float erbLo = (float)(21.4 * log10(50 * 0.00437 + 1.0));
    float erbHi = (float)(21.4 * log10(8000 * 0.00437 + 1.0));
    float erbStep = (erbHi - erbLo) / (SEA_GT_NCHAN - 1);
    float dt = 1 / (float)16000;
    float twoPiT = (float)(2 * kPiHW * dt);
    int c, n;
    memset(t, 0, sizeof *t);
    for (c = 0; c < SEA_GT_NCHAN; c++) {
        float cf = (float)((pow(10, (erbLo + c * erbStep) / 21.4) - 1) / 0.00437);
        float bw = (float)(24.7 * (cf * 0.00437 + 1.0) * 1.019);
        float phon = (float)(phons_at(cf) - 60.0);
        float ear = (float)pow(10, (double)(phon / 20));
        float z = expf(-twoPiT * bw);
        t->cf[c] = cf;
        t->bw[c] = bw;
        t->midEar[c] = ear;
        t->gain[c] = (float)(ear * pow((double)(twoPiT * bw), 4.0) / 3.0);
        t->f1[c] = cosf(cf * twoPiT) * z;
        t->f2[c] = sinf(cf * twoPiT) * z;
    }
    for (n = 0; n < 160; n++) {
        t->olaUp[n] = 0.5 * (1.0 + cos(n * kPiHW / (160) + kPiHW));
        t->olaDown[n] = 0.5 * (1.0 + cos(n * kPiHW / (160)));
    }
}

/* ------------------------------------------------------------------------------------------
 * The 16 k-native NoiseSup variant: function/20141106_speech_enhancement/aurora_etsi/
 * NoiseSup.cpp:1034-1079 (windows), MelProc.cpp:269-341 (InitGammawindows), :464-503 (InitGammaIDCTbasis),
 * :505-513 (ERB scale), rfft.cpp:46-181 called with (512, 8).  These files are C++: cos / sin of a float there are
 * the float overloads (cosf / sinf below); everything else promotes as in C.
 * ---------------------------------------------------------------------------------------- */
static float hz_to_erb(float hz) { return (float)(21.4 * log10(hz * 0.00437 + 1.0)); }
static float erb_to_hz(float r) { return (float)((pow(10, r / 21.4) - 1) / 0.00437); }

static void gamma_windows(int start[SEA16_NGAM], float w[SEA16_NGAM][SEA16_GLEN])
{ /* InitGammawindows (First, 80.0, 16000.0f, 2 * (129 - 1), 25, 1) */
    const float st = 80.0f, smpl = 16000.0f;
    const int nfft = 2 * (SEA16_NSPEC - 1);
    float cf[SEA16_NGAM], erb[SEA16_NGAM];
    const float lo = hz_to_erb(st), hi = hz_to_erb(smpl / 2);
    const float step = (hi - lo) / (SEA16_NGAM - 1);
    int i, j;
    for (i = 0; i < SEA16_NGAM; i++) {
        cf[i] = erb_to_hz(lo + step * i);
        erb[i] = (float)(1.019 * 24.7 * (4.37 * cf[i] / 1000 + 1));
    }
    for (i = 0; i < SEA16_NGAM; i++) {
        float norm = 0.0f;
        start[i] = (int)((int)cf[i] * nfft / smpl);
        for (j = 0; j < SEA16_GLEN; j++) {
            w[i][j] = (float)(1.0 / pow((1.0 + (j * 1.0 / nfft * smpl - cf[i]) / erb[i] * (j * 1.0 / nfft * smpl - cf[i]) / erb[i]), 2));
            norm += w[i][j];
        }
        for (j = 0; j < SEA16_GLEN; j++) w[i][j] /= norm;
    }
}

static void gamma_idct(const int start[SEA16_NGAM], float basis[SEA16_NGAM][SEA16_NGAM])
{ /* InitGammaIDCTbasis (basis, First, 25, 16000, 256) */
    const int fs = 16000;
    const float lin = fs / (float)(2 * (SEA16_NSPEC - 1));
    float cf[SEA16_NGAM], df[SEA16_NGAM];
    int i, j;
    for (j = 0; j < SEA16_NGAM; j++) cf[j] = start[j] * lin;
    for (j = 0; j < SEA16_NGAM; j++) {
        if (j == 0)
            df[j] = (cf[1] - cf[0]) / fs;
        else if (j == SEA16_NGAM - 1)
            df[j] = (cf[j] - cf[j - 1]) / fs;
        else
            df[j] = (cf[j + 1] - cf[j - 1]) / fs;
    }
    for (i = 0; i < SEA16_NGAM; i++)
        for (j = 0; j < SEA16_NGAM; j++) basis[i][j] = (float)(df[j] * cos(kTwoPi * i * cf[j] / fs));
}

typedef struct {
    sea_ns16k_tables *t;
    int pass, n8, kind;
    int count[4]; /* per kind, this pass */
} item_ctx;
static void slot_put(item_ctx *c, int kind, unsigned entry)
{
    /* pass 0: LEN2 fills slots 0, 1, 2; pass 1: PLAIN fills slots 0, 1; later passes: one slot per kind */
    const int n = c->count[kind]++;
    int slot, lane = n & 63;
    if (c->pass == 0) slot = n >> 6;
    else if (c->pass == 1) slot = n >> 6;
    else slot = (kind == SEA16_BF_PLAIN ? 0 : kind == SEA16_BF_PI4 ? 1 : 2) + 3 * (n >> 6); /* >= 3: does not fit */
    if (slot >= SEA16_FFT_SLOTS || (c->pass == 1 && (kind != SEA16_BF_PLAIN || slot > 1))) abort();
    c->t->fftSlot[c->pass][slot][lane] = 0x80000000u | entry;
}
static void item_visit(int i, void *ctx)
{
    item_ctx *c = (item_ctx *)ctx;
    int j;
    if (c->kind == SEA16_BF_LEN2) {
        slot_put(c, SEA16_BF_LEN2, (unsigned)i);
        return;
    }
    slot_put(c, SEA16_BF_PLAIN, (unsigned)i);
    if (c->n8 >= 1) slot_put(c, SEA16_BF_PI4, (unsigned)i); /* n4 != 1 */
    for (j = 1; j < c->n8; j++) slot_put(c, SEA16_BF_TWIDDLE, ((unsigned)j << 16) | (unsigned)i);
}


/* ---- tables of the PIPELINED kernel's transform wave (csrc/ns16k_pipe_kernel.hip) ----------------------------------
 * rfft (x, 512, 8) with a register-resident start: lane l owns the places 8l .. 8l+7 of the digit-reversed order, i.e.
 * the input elements bitrev6(l) + 64 bitrev3(j) (rev[] is the 9-bit reversal), so that the length-2 butterflies, the
 * n2 = 4 and the n2 = 8 levels touch one lane's registers only, gated by the reference's is/id schedule; the levels
 * n2 = 16 .. 256 go through LDS, ONE work item per lane and level -- a PAIR (plain butterfly on i + k n4, pi/4 butterfly
 * on i + n8 + k n4) or a TWIDDLED butterfly -- dealt alternately to the two 32-lane halves, on a work area that is
 * XOR-swizzled per 32-word block (SEA16_SWZ, found by tools/ns16k_swizzle_search.py: 800 -> 204 bank passes per
 * transform, floor 192). */
static const unsigned char kSwz16[16] = SEA16_SWZ;
static unsigned swz16(unsigned word) { return (word ^ kSwz16[(word >> 5) & 15]) * 4u; } /* byte offset */
typedef struct {
    unsigned char mark[SEA16_NFFT];
} mark16_ctx;
static void mark16_visit(int i, void *ctx) { ((mark16_ctx *)ctx)->mark[i] = 1; }
typedef struct {
    int base[64], n;
} list16_ctx;
static void list16_visit(int i, void *ctx)
{
    list16_ctx *l = (list16_ctx *)ctx;
    if (l->n >= 64) abort();
    l->base[l->n++] = i;
}

static void build_ns16k_pipe(sea_ns16k_tables *t)
{
    sea_ns16k_pipe_tables *p = &t->pipe;
    mark16_ctx len2, len4, len8;
    int l, j, q, s;
    memset(&len2, 0, sizeof len2);
    memset(&len4, 0, sizeof len4);
    memset(&len8, 0, sizeof len8);
    for_each_block(SEA16_NFFT, 2, 1, mark16_visit, &len2);
    for_each_block(SEA16_NFFT, 4, 0, mark16_visit, &len4);
    for_each_block(SEA16_NFFT, 8, 0, mark16_visit, &len8);
    for (l = 0; l < SEA_LANES; l++) {
        unsigned fl = 0;
        for (j = 0; j < 8; j++) {
            const unsigned e = bitrev((unsigned)l, 6) + 64u * bitrev((unsigned)j, 3);
            if (t->rev[e] != 8 * l + j) abort(); /* the head's element map IS the digit-reverse counter's */
            p->src8[j][l] = (unsigned short)e;
            p->win8[j][l] = t->sigWindow[e]; /* 0 beyond 479 (the kernel writes literal zeros there, as the reference pads) */
        }
        for (q = 0; q < 4; q++) {
            if (len2.mark[8 * l + 2 * q]) fl |= 1u << q;
            p->head8Addr[q][l] = swz16(8u * l + 2u * q) | (swz16(8u * l + 2u * q + 1u) << 16);
        }
        if (len4.mark[8 * l]) fl |= 16u;
        if (len4.mark[8 * l + 4]) fl |= 32u;
        if (len8.mark[8 * l]) fl |= 64u;
        p->head8Flags[l] = fl;
    }
    for (s = 0; s < SEA16_PIPE_LEVELS; s++) {
        const int k = s + 3, n2 = 2 << k, n4 = n2 >> 2, n8 = n2 >> 3;
        list16_ctx blocks;
        int b, item = 0, op;
        unsigned words[64][8];
        blocks.n = 0;
        for_each_block(SEA16_NFFT, n2, 0, list16_visit, &blocks);
        for (l = 0; l < SEA_LANES; l++) p->kind[s][l] = SEA_BF_NONE;
        for (b = 0; b < blocks.n; b++)
            for (j = 1; j < n8; j++, item++) {
                const int lane = (item & 1) * 32 + (item >> 1), i = blocks.base[b];
                if (item >= 64) abort();
                p->kind[s][lane] = SEA_BF_TWIDDLE;
                for (op = 0; op < 4; op++) {
                    words[lane][op] = (unsigned)(i + j + op * n4);
                    words[lane][4 + op] = (unsigned)(i + n4 - j + op * n4);
                    p->tw[s][op][lane] = t->fftTw[k][j][op];
                }
            }
        for (b = 0; b < blocks.n; b++, item++) {
            const int lane = (item & 1) * 32 + (item >> 1), i = blocks.base[b];
            if (item >= 64) abort();
            p->kind[s][lane] = SEA_BF_PAIR;
            for (op = 0; op < 4; op++) {
                words[lane][op] = (unsigned)(i + op * n4);
                words[lane][4 + op] = (unsigned)(i + n8 + op * n4);
            }
        }
        for (l = 0; l < SEA_LANES; l++) {
            /* an idle lane fetches what the first lane of its half fetches (identical addresses broadcast: no bank pass of
             * its own) and stores nothing */
            const int src = (p->kind[s][l] == SEA_BF_NONE) ? (l & 32) : l;
            if (p->kind[s][src] == SEA_BF_NONE) abort();
            for (q = 0; q < 4; q++) p->addr[s][q][l] = swz16(words[src][2 * q]) | (swz16(words[src][2 * q + 1]) << 16);
        }
    }
    /* FFTtoPSD (NoiseSup.cpp:240-261): value b = lane + 64 h needs x[2b], x[2b+1], x[512-2b] (b = 0: unused, element 0),
     * x[511-2b]; value 128 is x[256] squared */
    for (l = 0; l < SEA_LANES; l++)
        for (q = 0; q < 2; q++) {
            const unsigned b = (unsigned)l + 64u * (unsigned)q;
            p->psd[q][0][l] = swz16(2 * b) | (swz16(2 * b + 1) << 16);
            p->psd[q][1][l] = swz16((SEA16_NFFT - 2 * b) % SEA16_NFFT) | (swz16(SEA16_NFFT - 1 - 2 * b) << 16);
        }
    p->nyq = swz16(256);
}

/* rfft (x, 512, 8) walked on the host through the PIPELINED kernel's tables exactly as its transform wave walks them
 * (tests): x512 in natural order in, the reference's output order out. */
void sea_ns16k_pipe_fft_host(float *x512)
{
    static sea_ns16k_tables t;
    static int ready = 0;
    float work[SEA16_NFFT], out[SEA16_NFFT];
    int l, j, s, q;
    if (!ready) {
        sea_build_ns16k_tables(&t);
        ready = 1;
    }
    {
        const sea_ns16k_pipe_tables *p = &t.pipe;
        for (l = SEA_LANES - 1; l >= 0; l--) {
            float e[8];
            const unsigned fl = p->head8Flags[l];
            for (j = 0; j < 8; j++) e[j] = x512[p->src8[j][l]];
            for (q = 0; q < 4; q++)
                if (fl & (1u << q)) {
                    const float a = e[2 * q], b = e[2 * q + 1];
                    e[2 * q] = a + b;
                    e[2 * q + 1] = a - b;
                }
            for (q = 0; q < 2; q++)
                if (fl & (16u << q)) {
                    const float g0 = e[4 * q], g2 = e[4 * q + 2], g3 = e[4 * q + 3], t1 = g3 + g2;
                    e[4 * q + 3] = g3 - g2;
                    e[4 * q + 2] = g0 - t1;
                    e[4 * q] = g0 + t1;
                }
            if (fl & 64u) {
                const float x1 = e[0], x3 = e[4], x4 = e[6], x5 = e[1], x6 = e[3], x7 = e[5], x8 = e[7], t1 = x4 + x3;
                const float u1 = (float)((double)(x7 + x8) / 1.41421356237309504880), u2 = (float)((double)(x7 - x8) / 1.41421356237309504880);
                e[6] = x4 - x3;
                e[4] = x1 - t1;
                e[0] = x1 + t1;
                e[7] = x6 - u1;
                e[5] = -x6 - u1;
                e[3] = x5 - u2;
                e[1] = x5 + u2;
            }
            for (q = 0; q < 4; q++) {
                work[(p->head8Addr[q][l] & 0xffffu) / 4] = e[2 * q];
                work[(p->head8Addr[q][l] >> 16) / 4] = e[2 * q + 1];
            }
        }
        for (s = 0; s < SEA16_PIPE_LEVELS; s++)
            for (l = SEA_LANES - 1; l >= 0; l--) {
                unsigned a[8];
                float x[8], o[8];
                if (p->kind[s][l] == SEA_BF_NONE) continue;
                for (q = 0; q < 4; q++) a[2 * q] = (p->addr[s][q][l] & 0xffffu) / 4, a[2 * q + 1] = (p->addr[s][q][l] >> 16) / 4;
                for (q = 0; q < 8; q++) x[q] = work[a[q]];
                if (p->kind[s][l] == SEA_BF_TWIDDLE) {
                    const float cc1 = p->tw[s][0][l], ss1 = p->tw[s][1][l], cc3 = p->tw[s][2][l], ss3 = p->tw[s][3][l];
                    float t1 = x[2] * cc1 + x[6] * ss1, t2 = x[6] * cc1 - x[2] * ss1, t3 = x[3] * cc3 + x[7] * ss3, t4 = x[7] * cc3 - x[3] * ss3;
                    const float t5 = t1 + t3, t6 = t2 + t4;
                    t3 = t1 - t3;
                    t4 = t2 - t4;
                    o[2] = t6 - x[5], o[7] = x[5] + t6, o[6] = -x[1] - t3, o[3] = x[1] - t3;
                    o[5] = x[0] - t5, o[0] = x[0] + t5, o[4] = x[4] - t4, o[1] = x[4] + t4;
                } else {
                    const float t1 = x[3] + x[2];
                    const float u1 = (float)((double)(x[6] + x[7]) / 1.41421356237309504880), u2 = (float)((double)(x[6] - x[7]) / 1.41421356237309504880);
                    o[3] = x[3] - x[2], o[2] = x[0] - t1, o[0] = x[0] + t1, o[1] = x[1];
                    o[7] = x[5] - u1, o[6] = -x[5] - u1, o[5] = x[4] - u2, o[4] = x[4] + u2;
                }
                for (q = 0; q < 8; q++) work[a[q]] = o[q];
            }
    }
    for (l = 0; l < SEA16_NFFT; l++) out[l] = work[swz16((unsigned)l) / 4];
    memcpy(x512, out, sizeof out);
}

void sea_build_ns16k_tables(sea_ns16k_tables *t)
{
    int start[SEA16_NGAM], i, j, k, n2;
    static float w[SEA16_NGAM][SEA16_GLEN], basis[SEA16_NGAM][SEA16_NGAM];
    item_ctx ctx;
    memset(t, 0, sizeof *t);
    for (i = 0; i < SEA16_WIN; i++) /* NoiseSup.cpp:1034-1037 */
        t->sigWindow[i] = (float)(0.5 - 0.5 * cos((kTwoPi * ((float)i + 0.5)) / (float)(short)SEA16_WIN));
    for (j = 0; j <= 8; j++) /* :1040-1043, the half DoFilterWindowing reads (:716-725) */
        t->irWin[j] = (float)(0.5 - 0.5 * cos((kTwoPi * ((float)(8 + j) + 0.5)) / (float)(short)SEA_NTAP));
    gamma_windows(start, w);
    gamma_idct(start, basis);
    for (i = 0; i < SEA16_GLEN; i++)
        for (j = 0; j < SEA16_NGAM; j++) t->gammaT[i][j] = w[j][i];
    for (j = 0; j < SEA16_NGAM; j++)
        for (i = 0; i <= 8; i++) t->idctT[j][i] = basis[i][j];
    t->eps = (float)exp(-10.0);
    /* the digit-reverse counter of rfft.cpp:57-79, run on the indices */
    {
        unsigned short pos[SEA16_NFFT];
        int jj = 0, kk;
        for (i = 0; i < SEA16_NFFT; i++) pos[i] = (unsigned short)i; /* pos[p] = input element now at place p */
        for (i = 0; i < SEA16_NFFT - 1; i++) {
            if (i < jj) {
                unsigned short x = pos[jj];
                pos[jj] = pos[i];
                pos[i] = x;
            }
            kk = SEA16_NFFT >> 1;
            while (kk <= jj) {
                jj -= kk;
                kk >>= 1;
            }
            jj += kk;
        }
        for (i = 0; i < SEA16_NFFT; i++) t->rev[pos[i]] = (unsigned short)i;
    }
    /* pass 0: length-2 butterflies (:82-96); passes 1..7: levels n2 = 4..256 (:99-179 with m = 8) */
    memset(&ctx, 0, sizeof ctx);
    ctx.t = t;
    ctx.pass = 0;
    ctx.kind = SEA16_BF_LEN2;
    ctx.n8 = 0;
    {
        int is = 0, id = 4;
        while (is < SEA16_NFFT - 1) {
            for (i = is; i < SEA16_NFFT; i += id) item_visit(i, &ctx);
            is = (id << 1) - 2;
            id <<= 2;
        }
    }
    for (k = 1, n2 = 2; k < 8; k++) {
        float e;
        n2 <<= 1;
        ctx.pass = k;
        ctx.kind = SEA16_BF_PLAIN;
        ctx.n8 = n2 >> 3;
        memset(ctx.count, 0, sizeof ctx.count);
        for_each_block(SEA16_NFFT, n2, 0, item_visit, &ctx);
        (void)e;
    }
    /* the twiddles (rfft.cpp:135-144, float overloads of cos / sin) are generated CONSTANTS, not calls of the host's
     * cosf / sinf at initialisation: the result must not depend on the box's libm (tools/gen_ns16k_twiddles.py) */
    {
        static const struct {
            int pass, j;
            float v[4];
        } kTw[] = {
#include "ns16k_twiddles.inc"
        };
        unsigned q;
        for (q = 0; q < sizeof kTw / sizeof kTw[0]; q++)
            for (i = 0; i < 4; i++) t->fftTw[kTw[q].pass][kTw[q].j][i] = kTw[q].v[i];
    }
    build_ns16k_pipe(t);
}

void sea_ns16k_plain_tables(float *sigWindow480, float *irWindow17, int *gammaStart25, float *gamma25x128, float *idct25x25)
{
    static float w[SEA16_NGAM][SEA16_GLEN], basis[SEA16_NGAM][SEA16_NGAM];
    int i;
    for (i = 0; i < SEA16_WIN; i++) sigWindow480[i] = (float)(0.5 - 0.5 * cos((kTwoPi * ((float)i + 0.5)) / (float)(short)SEA16_WIN));
    for (i = 0; i < SEA_NTAP; i++) irWindow17[i] = (float)(0.5 - 0.5 * cos((kTwoPi * ((float)i + 0.5)) / (float)(short)SEA_NTAP));
    gamma_windows(gammaStart25, w);
    gamma_idct(gammaStart25, basis);
    memcpy(gamma25x128, w, sizeof w);
    memcpy(idct25x25, basis, sizeof basis);
}

/* The table-driven schedule run on the host, item by item as ns16k_kernel.hip's ns16_fft runs it (items of one pass
 * in any order, passes in order): lets a CPU test check rev / fftItem / fftTw against the reference's loop nest. */
void sea_ns16k_fft_host(float *x512)
{
    static sea_ns16k_tables t;
    static int ready = 0;
    float x[SEA16_NFFT];
    int pass, i;
    unsigned r;
    if (!ready) {
        sea_build_ns16k_tables(&t);
        ready = 1;
    }
    for (i = 0; i < SEA16_NFFT; i++) x[t.rev[i]] = x512[i];
    for (pass = 0; pass < SEA16_FFT_PASSES; pass++) {
        const int n4 = (pass == 0) ? 0 : (1 << (pass - 1)), n8 = n4 >> 1;
        for (r = SEA16_FFT_SLOTS * SEA_LANES; r-- > 0;) { /* backwards: the order within a pass must not matter */
            const unsigned it = t.fftSlot[pass][r / SEA_LANES][r % SEA_LANES];
            const int slot = (int)(r / SEA_LANES), j = (int)((it >> 16) & 0xffu);
            const int kind = (pass == 0) ? SEA16_BF_LEN2 : (pass == 1) ? SEA16_BF_PLAIN : (slot == 0 ? SEA16_BF_PLAIN : slot == 1 ? SEA16_BF_PI4 : SEA16_BF_TWIDDLE);
            if (!(it >> 31)) continue;
            i = (int)(it & 0xffffu);
            if (kind == SEA16_BF_LEN2) {
                const float a0 = x[i], a1 = x[i + 1];
                x[i] = a0 + a1;
                x[i + 1] = a0 - a1;
            } else if (kind == SEA16_BF_PLAIN) {
                const int i3 = i + 2 * n4, i4 = i + 3 * n4;
                const float x1 = x[i], x3 = x[i3], x4 = x[i4], t1 = x4 + x3;
                x[i4] = x4 - x3;
                x[i3] = x1 - t1;
                x[i] = x1 + t1;
            } else if (kind == SEA16_BF_PI4) {
                const int i1 = i + n8, i2 = i1 + n4, i3 = i2 + n4, i4 = i3 + n4;
                const float x1 = x[i1], x2 = x[i2], x3 = x[i3], x4 = x[i4];
                const float t1 = (float)((double)(x3 + x4) / 1.41421356237309504880);
                const float t2 = (float)((double)(x3 - x4) / 1.41421356237309504880);
                x[i4] = x2 - t1;
                x[i3] = -x2 - t1;
                x[i2] = x1 - t2;
                x[i1] = x1 + t2;
            } else {
                const float cc1 = t.fftTw[pass][j][0], ss1 = t.fftTw[pass][j][1], cc3 = t.fftTw[pass][j][2], ss3 = t.fftTw[pass][j][3];
                const int i1 = i + j, i2 = i1 + n4, i3 = i2 + n4, i4 = i3 + n4;
                const int i5 = i + n4 - j, i6 = i5 + n4, i7 = i6 + n4, i8 = i7 + n4;
                const float x1 = x[i1], x2 = x[i2], x3 = x[i3], x4 = x[i4], x5 = x[i5], x6 = x[i6], x7 = x[i7], x8 = x[i8];
                float t1 = x3 * cc1 + x7 * ss1, t2 = x7 * cc1 - x3 * ss1, t3 = x4 * cc3 + x8 * ss3, t4 = x8 * cc3 - x4 * ss3;
                const float t5 = t1 + t3, t6 = t2 + t4;
                t3 = t1 - t3;
                t4 = t2 - t4;
                x[i3] = t6 - x6;
                x[i8] = x6 + t6;
                x[i7] = -x2 - t3;
                x[i4] = x2 - t3;
                x[i6] = x1 - t5;
                x[i1] = x1 + t5;
                x[i5] = x5 - t4;
                x[i2] = x5 + t4;
            }
        }
    }
    memcpy(x512, x, sizeof x);
}

/* ---- rfft (x, n, m), any size (sea_tables.h) ------------------------------------------------------------------- */
typedef struct {
    unsigned *w;
    unsigned long n;
} words_ctx;
static void words_visit(int i, void *ctx)
{
    words_ctx *c = (words_ctx *)ctx;
    c->w[c->n++] = (unsigned)i;
}
static unsigned float_bits(float f)
{
    unsigned u;
    memcpy(&u, &f, sizeof u);
    return u;
}

unsigned *sea_rfft_schedule(int n, int m, unsigned long *n_words)
{
    unsigned *w;
    unsigned long cap, at;
    int i, j, k, n2;
    words_ctx c;
    if (n_words) *n_words = 0;
    if (n < 2 || n > SEA_RFFT_MAXN || (n & (n - 1)) != 0 || m < 1 || m > 14 || (1 << m) > n) return NULL;
    /* header + rev + length-two list + per level (block list <= n / 4, twiddles <= n / 2 words): generous bound */
    cap = SEA_RFFT_HDR + (unsigned long)n + (unsigned long)n / 2 + (unsigned long)m * ((unsigned long)n / 2 + (unsigned long)n / 2 + 8) + 64;
    w = (unsigned *)calloc(cap, sizeof *w);
    if (!w) return NULL;
    w[0] = (unsigned)n;
    w[1] = (unsigned)m;
    at = SEA_RFFT_HDR;
    /* the digit-reverse counter (rfft.c:57-79) run on the indices: pos[p] = input element that ends at place p */
    w[2] = (unsigned)at;
    {
        unsigned *pos = (unsigned *)malloc((size_t)n * sizeof *pos);
        int jj = 0, kk;
        if (!pos) {
            free(w);
            return NULL;
        }
        for (i = 0; i < n; i++) pos[i] = (unsigned)i;
        for (i = 0; i < n - 1; i++) {
            if (i < jj) {
                const unsigned x = pos[jj];
                pos[jj] = pos[i];
                pos[i] = x;
            }
            kk = n >> 1;
            while (kk <= jj && kk > 0) { /* (kk > 0 only guards the n = 2 corner, where the reference's loop ends with i) */
                jj -= kk;
                kk >>= 1;
            }
            jj += kk;
        }
        for (i = 0; i < n; i++) w[at + pos[i]] = (unsigned)i;
        free(pos);
        at += (unsigned long)n;
    }
    /* length-two butterflies (:82-96) */
    w[3] = (unsigned)at;
    c.w = w + at;
    c.n = 0;
    for_each_block(n, 2, 1, words_visit, &c);
    w[4] = (unsigned)c.n;
    at += c.n;
    /* L-shaped levels (:99-179) */
    for (k = 1, n2 = 2; k < m; k++) {
        float e;
        int n8;
        n2 <<= 1;
        n8 = n2 >> 3;
        w[5 + 3 * k] = (unsigned)at;
        c.w = w + at;
        c.n = 0;
        for_each_block(n, n2, 0, words_visit, &c);
        w[6 + 3 * k] = (unsigned)c.n;
        at += c.n;
        w[7 + 3 * k] = (unsigned)at;
        e = (float)((kPi * 2) / n2); /* :105 */
        for (j = 1; j < n8; j++) {
            const float a = j * e, a3 = 3 * a; /* :133-138: float angles, double cos / sin, stored as float */
            w[at + 4 * (unsigned long)j + 0] = float_bits((float)cos((double)a));
            w[at + 4 * (unsigned long)j + 1] = float_bits((float)sin((double)a));
            w[at + 4 * (unsigned long)j + 2] = float_bits((float)cos((double)a3));
            w[at + 4 * (unsigned long)j + 3] = float_bits((float)sin((double)a3));
        }
        at += 4ul * (unsigned long)(n8 > 0 ? n8 : 0);
    }
    if (at > cap) { /* cannot happen (bound above); never hand out an overrun table */
        free(w);
        return NULL;
    }
    if (n_words) *n_words = at;
    return w;
}

/* The schedule walked on the host exactly as rfft_any_kernel walks it (items of a level in reverse order: the order
 * within a level must not matter): lets a CPU test check rev / block lists / twiddles against the reference's loop nest
 * for every size.  Returns 0, or 1 for a size sea_rfft_schedule refuses (x untouched). */
int sea_rfft_schedule_host(float *x, int n, int m)
{
    unsigned long words = 0;
    unsigned *w = sea_rfft_schedule(n, m, &words);
    float *xs;
    int i, k, n2;
    long it;
    if (!w) return 1;
    xs = (float *)malloc((size_t)n * sizeof *xs);
    if (!xs) {
        free(w);
        return 1;
    }
    for (i = 0; i < n; i++) xs[w[w[2] + i]] = x[i];
    for (it = (long)w[4] - 1; it >= 0; it--) {
        const int i0 = (int)w[w[3] + it];
        const float a0 = xs[i0], a1 = xs[i0 + 1];
        xs[i0] = a0 + a1;
        xs[i0 + 1] = a0 - a1;
    }
    for (k = 1, n2 = 2; k < m; k++) {
        int n4, n8, per;
        const unsigned *blk;
        const unsigned *tw;
        n2 <<= 1;
        n4 = n2 >> 2, n8 = n2 >> 3, per = n8 > 0 ? n8 : 1;
        blk = w + w[5 + 3 * k];
        tw = w + w[7 + 3 * k];
        for (it = (long)w[6 + 3 * k] * per - 1; it >= 0; it--) {
            const int b = (int)blk[it / per], j = (int)(it % per);
            if (j == 0) {
                {
                    const int i1 = b, i3 = i1 + 2 * n4, i4 = i3 + n4;
                    const float x1 = xs[i1], x3 = xs[i3], x4 = xs[i4], t1 = x4 + x3;
                    xs[i4] = x4 - x3;
                    xs[i3] = x1 - t1;
                    xs[i1] = x1 + t1;
                }
                if (n4 != 1) {
                    const int i1 = b + n8, i2 = i1 + n4, i3 = i2 + n4, i4 = i3 + n4;
                    const float x1 = xs[i1], x2 = xs[i2], x3 = xs[i3], x4 = xs[i4];
                    const float t1 = (float)((double)(x3 + x4) / 1.41421356237309504880);
                    const float t2 = (float)((double)(x3 - x4) / 1.41421356237309504880);
                    xs[i4] = x2 - t1;
                    xs[i3] = -x2 - t1;
                    xs[i2] = x1 - t2;
                    xs[i1] = x1 + t2;
                }
            } else {
                float cc1, ss1, cc3, ss3, t1, t2, t3, t4, t5, t6;
                const int i1 = b + j, i2 = i1 + n4, i3 = i2 + n4, i4 = i3 + n4;
                const int i5 = b + n4 - j, i6 = i5 + n4, i7 = i6 + n4, i8 = i7 + n4;
                const float x1 = xs[i1], x2 = xs[i2], x3 = xs[i3], x4 = xs[i4], x5 = xs[i5], x6 = xs[i6], x7 = xs[i7], x8 = xs[i8];
                memcpy(&cc1, &tw[4 * j], 4), memcpy(&ss1, &tw[4 * j + 1], 4), memcpy(&cc3, &tw[4 * j + 2], 4), memcpy(&ss3, &tw[4 * j + 3], 4);
                t1 = x3 * cc1 + x7 * ss1, t2 = x7 * cc1 - x3 * ss1, t3 = x4 * cc3 + x8 * ss3, t4 = x8 * cc3 - x4 * ss3;
                t5 = t1 + t3, t6 = t2 + t4;
                t3 = t1 - t3;
                t4 = t2 - t4;
                xs[i3] = t6 - x6;
                xs[i8] = x6 + t6;
                xs[i7] = -x2 - t3;
                xs[i4] = x2 - t3;
                xs[i6] = x1 - t5;
                xs[i1] = x1 + t5;
                xs[i5] = x5 - t4;
                xs[i2] = x5 + t4;
            }
        }
    }
    memcpy(x, xs, (size_t)n * sizeof *xs);
    free(xs);
    free(w);
    return 0;
}
