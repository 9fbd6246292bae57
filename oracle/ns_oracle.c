/*
 * oracle/ns_oracle.c -- TEST INFRASTRUCTURE ONLY (see sea_oracle.h).
 *
 * Plain-C restatement of the ETSI ES 202 050 two-stage Wiener noise suppressor as the reference
 * runs it (8 kHz framing on whatever it is fed: hop 80, window 200, FFT 256), its real
 * split-radix FFT and the CompCeps front-end.  Flat tables, no linked lists, no per-frame malloc.
 * Every float/double promotion of the reference C (compiled as C, gnu dialect, no FMA) is kept,
 * so results are bit-identical to oracle/_ref/libetsi_ref.so; tests/test_oracle_vs_ref.py and
 * the committed golden vectors check exactly that.
 *
 * Build: gcc -O2 -ffp-contract=off (oracle/Makefile).  Do NOT add -march / -ffast-math.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "sea_oracle.h"

#define PI2_D 6.28318530717958647692 /* etsi/cpp/ParmInterface.h:45 */
#define PI_D 3.14159265358979323846  /* etsi/cpp/rfft.h:16 */
#define SQRT2_D 1.41421356237309504880

enum { HOP = 80, WIN = 200, NFFT = 256, NSPEC = 65, NMEL = 25, NBUF = 320, NTAP = 17, HALF = 8 };

/* ------------------------------------------------------------------------------------------
 * rfft: etsi/cpp/rfft.c:45-180.  Restated as an explicit butterfly schedule: a bit-reversal
 * permutation, the length-2 pass, then per split-radix level three kinds of independent
 * butterflies (plain 4-point, the pi/4 4-point, the general twiddled 8-point).  Butterflies of
 * one level touch disjoint elements, which is what the GPU kernel exploits; the arithmetic inside
 * each butterfly is the reference's, operation for operation.
 * ---------------------------------------------------------------------------------------- */
static void bf_plain(float *x, int i1, int i2, int i3, int i4)
{ /* rfft.c:110-113 */
    float t1 = x[i4] + x[i3];
    x[i4] -= x[i3];
    x[i3] = x[i1] - t1;
    x[i1] += t1;
    (void)i2;
}

static void bf_pi4(float *x, int i1, int i2, int i3, int i4)
{ /* rfft.c:120-125: the division by sqrt(2) is a double division of a float sum */
    float t1 = (float)((double)(x[i3] + x[i4]) / SQRT2_D);
    float t2 = (float)((double)(x[i3] - x[i4]) / SQRT2_D);
    x[i4] = x[i2] - t1;
    x[i3] = -x[i2] - t1;
    x[i2] = x[i1] - t2;
    x[i1] = x[i1] + t2;
}

static void bf_twiddle(float *x, int i, int j, int n4, float cc1, float ss1, float cc3, float ss3)
{ /* rfft.c:145-174 */
    int i1 = i + j, i2 = i1 + n4, i3 = i2 + n4, i4 = i3 + n4;
    int i5 = i + n4 - j, i6 = i5 + n4, i7 = i6 + n4, i8 = i7 + n4;
    float t1 = x[i3] * cc1 + x[i7] * ss1;
    float t2 = x[i7] * cc1 - x[i3] * ss1;
    float t3 = x[i4] * cc3 + x[i8] * ss3;
    float t4 = x[i8] * cc3 - x[i4] * ss3;
    float t5 = t1 + t3, t6 = t2 + t4;
    t3 = t1 - t3;
    t4 = t2 - t4;
    t2 = x[i6] + t6;
    x[i3] = t6 - x[i6];
    x[i8] = t2;
    t2 = x[i2] - t3;
    x[i7] = -x[i2] - t3;
    x[i4] = t2;
    t1 = x[i1] + t5;
    x[i6] = x[i1] - t5;
    x[i1] = t1;
    t1 = x[i5] + t4;
    x[i5] = x[i5] - t4;
    x[i2] = t1;
}

/* n: the length the digit-reverse counter and the index patterns run on; m: the number of levels (the reference's
 * callers pass m = log2 n, except the 16 k-native variant: (512, 8)); cxx: the twiddles as the C++ build of the same
 * text computes them (cos / sin of a float are the float overloads there: aurora_etsi/rfft.cpp) */
static void rfft_core(float *x, int n, int m, int cxx)
{
    int i, j, k, is, id, n2, n4, n8, bits = 0;

    /* bit reversal (rfft.c:57-79 implements the classic in-place bit-reversed swap over log2 n bits) */
    while ((1 << bits) < n) bits++;
    for (i = 0; i < n; i++) {
        int r = 0;
        for (k = 0; k < bits; k++) r |= ((i >> k) & 1) << (bits - 1 - k);
        if (i < r) { float t = x[i]; x[i] = x[r]; x[r] = t; }
    }
    /* length-two butterflies on the split-radix index pattern (rfft.c:82-96) */
    for (is = 0, id = 4; is < n - 1; is = 2 * id - 2, id *= 4)
        for (i = is; i < n; i += id) {
            float a0 = x[i];
            x[i] = a0 + x[i + 1];
            x[i + 1] = a0 - x[i + 1];
        }
    /* L-shaped levels (rfft.c:99-178) */
    for (k = 1, n2 = 2; k < m; k++) {
        float e;
        n2 <<= 1;
        n4 = n2 >> 2;
        n8 = n2 >> 3;
        e = (float)((PI_D * 2) / n2);
        for (is = 0, id = n2 << 1; is < n; is = 2 * id - n2, id *= 4)
            for (i = is; i < n; i += id) {
                bf_plain(x, i, i + n4, i + 2 * n4, i + 3 * n4);
                if (n4 != 1) bf_pi4(x, i + n8, i + n8 + n4, i + n8 + 2 * n4, i + n8 + 3 * n4);
            }
        for (j = 1; j < n8; j++) {
            float a = j * e, a3 = 3 * a;
            float cc1 = cxx ? cosf(a) : (float)cos((double)a), ss1 = cxx ? sinf(a) : (float)sin((double)a);
            float cc3 = cxx ? cosf(a3) : (float)cos((double)a3), ss3 = cxx ? sinf(a3) : (float)sin((double)a3);
            for (is = 0, id = n2 << 1; is < n; is = 2 * id - n2, id *= 4)
                for (i = is; i < n; i += id) bf_twiddle(x, i, j, n4, cc1, ss1, cc3, ss3);
        }
    }
}

void ora_rfft(float *x, int n, int m) { rfft_core(x, n, m, 0); }   /* etsi/cpp/rfft.c, compiled as C */
void ora16_rfft(float *x, int n, int m) { rfft_core(x, n, m, 1); } /* aurora_etsi/rfft.cpp, compiled as C++ */

/* ------------------------------------------------------------------------------------------
 * Constant tables
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int start, len;
    float w[32];
} mel_band;

typedef struct {
    int ready;
    float sigWindow[WIN];      /* NoiseSup.c:974-975 */
    float irWindow[NTAP];      /* NoiseSup.c:978-979 */
    mel_band mel[NMEL];        /* MelProc.c:128-230, (0 Hz, 8000, 128, 25, normalised) */
    float idct[NMEL][NMEL];    /* MelProc.c:283-337 */
    float hamming[WIN / 2];    /* CompCeps.c:87-94 */
    float dct[12][23];         /* CompCeps.c:153-173 */
    mel_band ccmel[23];        /* MelProc.c:402-522, (64 Hz, 8000, 256, 23) */
    float eps;                 /* NS_EPS, NoiseSup.h:32 */
} ns_tables;

static ns_tables T;

static float mel_of(float hz) { return (float)(2595.0 * log10(1.0 + hz / 700.0)); }

static void build_ns_mel(void)
{
    /* InitMelFBwindows(FirstWin, 0.0, 8000.0f, 128, 25, 1): MelProc.c:128-230 */
    const float smpl = 8000.0f;
    const int nfft = 128, nch = NMEL;
    int c[NMEL], i, j, k;
    float start_mel = mel_of((float)0.0f);
    float top_mel = (float)(2595.0 * log10(1.0 + (smpl / 2) / 700.0));
    for (i = 0; i < nch; i++) {
        float frac = start_mel + (float)i / (nch - 1) * (top_mel - start_mel);
        float freq = (float)(700 * (pow(10, frac / 2595.0) - 1.0));
        c[i] = (int)(nfft * freq / smpl + 0.5);
    }
    for (i = 0; i < nch; i++) {
        mel_band *b = &T.mel[i];
        float norm = 0.0f;
        if (i == 0) {
            b->start = c[0];
            b->len = c[1] - c[0];
            for (j = 0; j < b->len; j++) {
                b->w[j] = (float)(1.0 - (float)j / (float)b->len);
                norm += b->w[j];
            }
        } else if (i < nch - 1) {
            int up = c[i] - c[i - 1];
            b->start = c[i - 1] + 1;
            b->len = c[i + 1] - c[i - 1] - 1;
            for (j = 0; j < up; j++) {
                b->w[j] = (float)(j + 1) / (float)up;
                norm += b->w[j];
            }
            for (j = up, k = 0; j < b->len; j++, k++) {
                b->w[j] = (float)(1.0 - (k + 1) / (float)(c[i + 1] - c[i]));
                norm += b->w[j];
            }
        } else {
            b->start = c[nch - 2] + 1;
            b->len = c[nch - 1] - c[nch - 2];
            for (j = 0; j < b->len; j++) {
                b->w[j] = (float)(j + 1) / (float)b->len;
                norm += b->w[j];
            }
        }
        for (j = 0; j < b->len; j++) b->w[j] /= norm;
    }
}

static void build_ns_idct(void)
{
    /* InitMelIDCTbasis(.., 25, 8000, 128): MelProc.c:283-337 */
    const int fs = 8000;
    float lin = fs / (float)128;
    float cf[NMEL], df[NMEL];
    int i, j;
    for (j = 0; j < NMEL; j++) {
        const mel_band *b = &T.mel[j];
        if (j == 0)
            cf[j] = b->start * lin;
        else if (j == NMEL - 1)
            cf[j] = (b->start + b->len - 1) * lin;
        else {
            float st = b->start * lin, sum = 0.0f;
            cf[j] = 0.0f;
            for (i = 0; i < b->len; i++) {
                cf[j] += b->w[i] * (st + i * lin);
                sum += b->w[i];
            }
            cf[j] /= sum;
        }
    }
    for (j = 0; j < NMEL; j++) {
        if (j == 0)
            df[j] = (cf[1] - cf[0]) / fs;
        else if (j == NMEL - 1)
            df[j] = (cf[j] - cf[j - 1]) / fs;
        else
            df[j] = (cf[j + 1] - cf[j - 1]) / fs;
    }
    for (i = 0; i < NMEL; i++)
        for (j = 0; j < NMEL; j++) T.idct[i][j] = (float)(df[j] * cos(PI2_D * i * cf[j] / fs));
}

static void build_cc_tables(void)
{
    const float stf = 64.0f, smpl = 8000.0f; /* ParmInterface.h:38, CCX->SamplingFrequency */
    const int nfft = 256, nch = 23;
    float start_mel = mel_of(stf);
    float top_mel = (float)(2595.0 * log10(1.0 + (smpl / 2) / 700.0));
    int i, j, prev_top = 0;
    for (i = 0; i < WIN / 2; i++) T.hamming[i] = (float)(0.54 - 0.46 * cos(PI2_D * (i + 0.5) / (short)WIN));
    for (i = 1; i <= 12; i++)
        for (j = 0; j < nch; j++)
            T.dct[i - 1][j] = (float)cos(PI_D * (float)i / (float)nch * ((float)j + 0.5));
    /* InitFFTWindows: MelProc.c:402-463 */
    for (i = 0; i < nch; i++) {
        float lo = start_mel + (float)i / (nch + 1) * (top_mel - start_mel);
        float hi = start_mel + (float)(i + 2) / (nch + 1) * (top_mel - start_mel);
        float f_lo = (float)(700 * (pow(10, lo / 2595.0) - 1.0));
        float f_hi = (float)(700 * (pow(10, hi / 2595.0) - 1.0));
        int s = (int)(nfft * f_lo / smpl + 0.5);
        T.ccmel[i].start = s;
        T.ccmel[i].len = (int)(nfft * f_hi / smpl + 0.5) - s + 1;
    }
    /* ComputeTriangle: MelProc.c:482-522 */
    for (i = 0; i < nch; i++) {
        mel_band *b = &T.ccmel[i];
        int low = (i < nch - 1) ? T.ccmel[i + 1].start - b->start + 1 : prev_top - b->start + 1;
        int hgh = b->len - low + 1;
        for (j = 0; j < low; j++) b->w[j] = (float)(j + 1) / low;
        for (j = 1; j < hgh; j++) b->w[low + j - 1] = (float)(hgh - j) / hgh;
        prev_top = b->start + b->len - 1;
    }
}

static void tables_init(void)
{
    int i;
    if (T.ready) return;
    for (i = 0; i < WIN; i++)
        T.sigWindow[i] = (float)(0.5 - 0.5 * cos((PI2_D * ((float)i + 0.5)) / (float)(short)WIN));
    for (i = 0; i < NTAP; i++)
        T.irWindow[i] = (float)(0.5 - 0.5 * cos((PI2_D * ((float)i + 0.5)) / (float)(short)NTAP));
    build_ns_mel();
    build_ns_idct();
    build_cc_tables();
    T.eps = (float)exp(-10.0);
    T.ready = 1;
}

void ora_ns_tables(float *sigWindow200, float *irWindow17, float *idct25x25, int *melStart25,
                   int *melLen25, float *melData)
{
    int i, j;
    tables_init();
    memcpy(sigWindow200, T.sigWindow, sizeof T.sigWindow);
    memcpy(irWindow17, T.irWindow, sizeof T.irWindow);
    memcpy(idct25x25, T.idct, sizeof T.idct);
    for (j = 0; j < NMEL; j++) {
        melStart25[j] = T.mel[j].start;
        melLen25[j] = T.mel[j].len;
        for (i = 0; i < T.mel[j].len && i < 16; i++) melData[j * 16 + i] = T.mel[j].w[i];
    }
}

void ora_cc_tables(float *hamming100, float *dct12x23, int *melStart23, int *melLen23, float *melData)
{
    int i, j;
    tables_init();
    memcpy(hamming100, T.hamming, sizeof T.hamming);
    memcpy(dct12x23, T.dct, sizeof T.dct);
    for (j = 0; j < 23; j++) {
        melStart23[j] = T.ccmel[j].start;
        melLen23[j] = T.ccmel[j].len;
        for (i = 0; i < T.ccmel[j].len && i < 32; i++) melData[j * 32 + i] = T.ccmel[j].w[i];
    }
}

/* DoMelFB: MelProc.c:82-104 -- in place, band sums in tap order */
static void mel_fb(float *spec, const mel_band *bands, int nb)
{
    float sum[NMEL];
    int j, i;
    for (j = 0; j < nb; j++) {
        sum[j] = 0.0f;
        for (i = 0; i < bands[j].len; i++) sum[j] += spec[bands[j].start + i] * bands[j].w[i];
    }
    for (j = 0; j < nb; j++) spec[j] = sum[j];
}

/* ------------------------------------------------------------------------------------------
 * NoiseSup state and per-frame step: NoiseSup.c:49-143 (state), :1061-1440 (DoNoiseSup)
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    float buf[2][NBUF];        /* First/SecondStageInFloatBuffer */
    int nIn1, nIn2, nOut2;     /* nbFramesInFirstStage / InSecondStage / OutSecondStage */
    float nSig[2][NSPEC], noise[2][NSPEC], den[2][NSPEC];
    float psdPrev[2][NSPEC];   /* the other slot of PSDMeanBuffer == previous frame's PSD */
    float dcX, dcY;            /* prevSamples */
    float denEn[3], lowSNRtrack, alfaGF;
    int nbFrame[2];
    short flagVAD, hangOver, nbSpeechFrames;
    float meanEn;
    /* VAD for frame dropping (struct vad_data_fd, NoiseSup.c:72-83; zeroed by DoNoiseSupInit :935-943) */
    float fdMelMean, fdVarMean, fdAccTest, fdSpecMean, fdMelValues[2], fdSpecValues, fdSpeechInVADQ;
    /* FEParamsX flags the frame-dropping VAD reads (ParmInterface.h:85-89) */
    int speechFoundVar, speechFoundSpec, speechFoundMel, speechFoundVADNS, frameCounter;
} ns_state;

static void ns_init(ns_state *s)
{ /* DoNoiseSupInit: NoiseSup.c:884-968 */
    int i, st;
    memset(s, 0, sizeof *s);
    s->alfaGF = (float)0.8;
    for (st = 0; st < 2; st++)
        for (i = 0; i < NSPEC; i++) s->noise[st][i] = T.eps;
}

static void ns_vad(ns_state *s, int st, const float *frame)
{ /* VAD: NoiseSup.c:359-430 */
    int nb = s->nbFrame[st], i;
    float frameEn, lambdaLTE, meanEn = s->meanEn;
    short flagVAD = s->flagVAD, hangOver = s->hangOver, nbSpeech = s->nbSpeechFrames;
    if (nb < 2147483647) nb++;
    s->nbFrame[st] = nb;
    if (st == 1) return;

    lambdaLTE = (nb < 10) ? 1 - 1 / (float)nb : (float)0.97;
    frameEn = 64.0f;
    for (i = 0; i < HOP; i++) frameEn += frame[i] * frame[i];
    frameEn = (float)(0.5 + (log(frameEn / 64.0) / log(2.0)) * 16.0);

    if (((frameEn - meanEn) < (short)20) || (nb < (short)10)) {
        if ((frameEn < meanEn) || (nb < (short)10))
            meanEn += (1 - lambdaLTE) * (frameEn - meanEn);
        else
            meanEn += (1 - (float)0.99) * (frameEn - meanEn);
        if (meanEn < (float)80.0) meanEn = (float)80.0;
    }
    if (nb > 4) {
        if ((frameEn - meanEn) > (short)15) {
            flagVAD = 1;
            nbSpeech++;
        } else {
            if (nbSpeech > (short)4) hangOver = 15;
            nbSpeech = 0;
            if (hangOver != 0) {
                hangOver--;
                flagVAD = 1;
            } else
                flagVAD = 0;
        }
    }
    s->meanEn = meanEn;
    s->flagVAD = flagVAD;
    s->hangOver = hangOver;
    s->nbSpeechFrames = nbSpeech;
}

static void ns_filter_calc(ns_state *s, int st, float *P, float *W)
{ /* FilterCalc: NoiseSup.c:449-563.  The frame counter is narrowed to int16 there. */
    float *nSig = s->nSig[st], *noise = s->noise[st], *den = s->den[st];
    short nb = (short)s->nbFrame[st];
    const float beta = (float)0.98, rsbMin = (float)0.079432823;
    float lambda;
    int i;

    if (st == 1) {
        for (i = 0; i < NSPEC; i++) noise[i] *= noise[i];
        if (nb < 11) {
            lambda = 1 - 1 / (float)nb;
            for (i = 0; i < NSPEC; i++) noise[i] = lambda * noise[i] + (1 - lambda) * P[i];
        } else {
            for (i = 0; i < NSPEC; i++) {
                float upd = (float)(0.9 + 0.1 * (P[i] / (P[i] + noise[i])) *
                                              (1.0 + 1.0 / (1.0 + 0.1 * (P[i] / noise[i]))));
                noise[i] *= upd;
            }
        }
        for (i = 0; i < NSPEC; i++) {
            noise[i] = (float)sqrt((double)noise[i]);
            if (noise[i] < T.eps) noise[i] = T.eps;
        }
    }
    for (i = 0; i < NSPEC; i++) {
        nSig[i] = (float)sqrt((double)nSig[i]);
        P[i] = (float)sqrt((double)P[i]);
    }
    if (st == 0) {
        lambda = (nb < (short)100) ? 1 - 1 / (float)nb : (float)0.99;
        if (s->flagVAD == 0)
            for (i = 0; i < NSPEC; i++) {
                noise[i] = lambda * noise[i] + (1 - lambda) * P[i];
                if (noise[i] < T.eps) noise[i] = T.eps;
            }
    }
    for (i = 0; i < NSPEC; i++) {
        float post = (P[i] / noise[i]) - 1;
        float prio = beta * (den[i] / noise[i]) + (1 - beta) * ((0 > post) ? 0 : post);
        W[i] = prio / (1 + prio);
        prio = W[i] * P[i] / noise[i];
        prio = (prio > rsbMin) ? prio : rsbMin;
        W[i] = prio / (1 + prio);
        den[i] = W[i] * nSig[i];
    }
}

static void ns_gain_fact(ns_state *s, int st, float *W)
{ /* DoGainFact: NoiseSup.c:581-642 */
    int i;
    if (st == 0) {
        const float *den = s->den[0];
        s->denEn[0] = s->denEn[1];
        s->denEn[1] = s->denEn[2];
        s->denEn[2] = 0.0f;
        for (i = 0; i < NSPEC; i++) s->denEn[2] += den[i];
    } else {
        const float *noise = s->noise[1];
        float noiseEn = 0.0f, averSNR, lambdaSNR;
        for (i = 0; i < NSPEC; i++) noiseEn += noise[i];
        averSNR = (s->denEn[0] * s->denEn[1] * s->denEn[2]) / (noiseEn * noiseEn * noiseEn);
        if (averSNR > 0.00001)
            averSNR = (float)((20 * log10((double)averSNR)) / 3.0);
        else
            averSNR = (float)(-100.0 / 3.0);

        if (((averSNR - s->lowSNRtrack) < 10.0) || (s->nbFrame[1] < (short)10)) {
            if (s->nbFrame[1] < (short)10)
                lambdaSNR = (float)(1.0 - 1.0 / (float)s->nbFrame[1]);
            else
                lambdaSNR = (averSNR < s->lowSNRtrack) ? (float)0.95 : (float)0.99;
            s->lowSNRtrack = (float)(s->lowSNRtrack + (1.0 - lambdaSNR) * (averSNR - s->lowSNRtrack));
        }
        if (s->denEn[2] > 100) {
            if (averSNR < (s->lowSNRtrack + 3.5)) {
                s->alfaGF = (float)(s->alfaGF + 0.15);
                if (s->alfaGF > 0.8) s->alfaGF = (float)0.8;
            } else {
                s->alfaGF = (float)(s->alfaGF - 0.3);
                if (s->alfaGF < 0.1) s->alfaGF = (float)0.1;
            }
        }
        for (i = 0; i < NMEL; i++) W[i] = (float)(s->alfaGF * W[i] + (1.0 - s->alfaGF) * 1.0);
    }
}

/* ---- VAD for frame dropping: measures taken inside the first stage (NoiseSup.c:672-839) --------
 * Arithmetic follows the reference's promotions: float state, double literals (1.1, 1.5, 0.8 ...)
 * promote the expression to double, the result is rounded to float on assignment; the frame
 * counter is narrowed to int16 first (X_INT16 FrameCounter = nbFrame[0]). */
#define FMAXF(a, b) (((a) > (b)) ? (a) : (b))

static int speech_q_var(ns_state *s, const float *W)
{ /* SpeechQVar, NoiseSup.c:798-839: variance of the first 64 Wiener gains */
    const short ssize = 64, fc = (short)s->nbFrame[0];
    float var = 0.0f, mean = 0.0f, specVar;
    int i;
    for (i = 0; i < ssize; i++) {
        mean += W[i];
        var += W[i] * W[i];
    }
    specVar = (var / ssize) - mean * mean / (ssize * ssize);
    if (fc < 15) s->fdVarMean = FMAXF(s->fdVarMean, specVar);
    if (specVar < s->fdVarMean * 1.5 && specVar > s->fdVarMean * 0.85)
        s->fdVarMean = (s->fdVarMean * 0.8 + specVar * 0.2);
    if (specVar <= s->fdVarMean * 0.25) s->fdVarMean = (s->fdVarMean * 0.97 + specVar * 0.03);
    return (specVar > s->fdVarMean * 1.65) ? 1 : 0;
}

static int speech_q_spec(ns_state *s)
{ /* SpeechQSpec, NoiseSup.c:686-727 */
    const short fc = (short)s->nbFrame[0];
    if (fc == 1) s->fdSpecMean = s->fdSpecValues;
    if (fc < 15) {
        float acceleration;
        s->fdAccTest = 1.1 * (s->fdAccTest * (float)(fc - 1) + s->fdSpecValues) / (float)(fc);
        acceleration = s->fdSpecValues / s->fdAccTest;
        if (acceleration > 2.5) s->fdSpeechInVADQ = 1;
        if (s->fdSpeechInVADQ == 0) s->fdSpecMean = FMAXF(s->fdSpecMean, s->fdSpecValues);
    }
    if (s->fdSpecValues < s->fdSpecMean * 1.5 && s->fdSpecValues > s->fdSpecMean * 0.75)
        s->fdSpecMean = (s->fdSpecMean * 0.8 + s->fdSpecValues * 0.2);
    if (s->fdSpecValues <= s->fdSpecMean * 0.5) s->fdSpecMean = (s->fdSpecMean * 0.97 + s->fdSpecValues * 0.03);
    return (s->fdSpecValues > s->fdSpecMean * 1.65) ? 1 : 0;
}

static int speech_q_mel(ns_state *s)
{ /* SpeechQMel, NoiseSup.c:744-781 */
    const short fc = (short)s->nbFrame[0];
    float smoothMel = 0.75 * s->fdMelValues[1] + 0.25 * s->fdMelValues[0];
    s->fdMelValues[0] = s->fdMelValues[1];
    if (fc < 15) s->fdMelMean = FMAXF(s->fdMelMean, smoothMel);
    if (smoothMel < s->fdMelMean * 1.5 && smoothMel > s->fdMelMean * 0.75)
        s->fdMelMean = (s->fdMelMean * 0.8 + smoothMel * 0.2);
    if (smoothMel <= s->fdMelMean * 0.5) s->fdMelMean = (s->fdMelMean * 0.97 + smoothMel * 0.03);
    return (smoothMel > s->fdMelMean * 3.25) ? 1 : 0;
}

/* One stage of the two-stage filter on buffer st; writes 80 filtered samples to dst. */
static void ns_stage(ns_state *s, int st, float *dst)
{
    float work[NFFT], W[NSPEC], P[NSPEC], h[NMEL], fir[NTAP];
    const float *buf = s->buf[st];
    float *nSig = s->nSig[st];
    int i, j, t, f;

    /* window + zero pad (DoSigWindowing, NoiseSup.c:218-231) on buf[60..259] */
    for (i = 0; i < WIN; i++) work[i] = buf[60 + i] * T.sigWindow[i];
    for (i = WIN; i < NFFT; i++) work[i] = 0.0f;
    ora_rfft(work, NFFT, 8);

    /* FFTtoPSD: NoiseSup.c:249-270 */
    work[0] = work[0] * work[0];
    for (i = 1, j = NFFT - 1; i < NFFT / 2; i++, j--) work[i] = work[i] * work[i] + work[j] * work[j];
    work[i] = work[i] * work[i];
    for (i = 0, j = 0; i < NSPEC - 1; i++, j += 2) nSig[i] = (float)((work[j] + work[j + 1]) / 2.0);
    nSig[i] = work[j];

    /* PSDMean: NoiseSup.c:289-303 */
    for (i = 0; i < NSPEC; i++) {
        P[i] = (s->psdPrev[st][i] + nSig[i]) / (short)2;
        s->psdPrev[st][i] = nSig[i];
    }
    ns_vad(s, st, buf + HOP);
    ns_filter_calc(s, st, P, W);
    if (st == 0) s->speechFoundVar = speech_q_var(s, W); /* NoiseSup.c:1255-1258 */
    mel_fb(W, T.mel, NMEL);
    if (st == 0) { /* NoiseSup.c:1268-1281 */
        float tempEn = 0.0f;
        for (i = 0; i < NMEL; i++) tempEn += (float)W[i];
        s->fdSpecValues = tempEn * tempEn - 3.0;
        s->speechFoundSpec = speech_q_spec(s);
        s->fdMelValues[1] = (float)(W[1] + W[2] + W[3]) / 3.0;
        s->speechFoundMel = speech_q_mel(s);
    }
    ns_gain_fact(s, st, W);

    /* DoMelIDCT (MelProc.c:357-378); only taps 0..8 are consumed (NoiseSup.c:660-669) */
    for (t = 0; t <= HALF; t++) {
        h[t] = 0.0f;
        for (f = 0; f < NMEL; f++) h[t] += W[f] * T.idct[t][f];
    }
    for (j = 0; j <= HALF; j++) {
        fir[HALF + j] = h[j] * T.irWindow[HALF + j];
        fir[HALF - j] = fir[HALF + j];
    }
    /* ApplyWF: NoiseSup.c:324-340 -- taps j = -8..8 in that order over buf[80+i-j] */
    for (i = 0; i < HOP; i++) {
        float acc = 0.0f;
        for (j = -HALF; j <= HALF; j++) acc += fir[j + HALF] * buf[HOP + i - j];
        dst[i] = acc;
    }
}

/* DoNoiseSup: NoiseSup.c:1061-1440.  Returns 1 when out[80] was produced. */
static int ns_step(ns_state *s, const float *in, float *out)
{
    int i;
    for (i = 0; i < HOP; i++) s->buf[0][240 + i] = in[i];
    s->nIn1++;
    if (s->nIn1 - s->nIn2 > 2) {
        ns_stage(s, 0, s->buf[1] + 240);
        s->nIn2++;
        s->speechFoundVADNS = (s->nbSpeechFrames > 4) ? 1 : 0; /* NoiseSup.c:1359-1365 */
        s->frameCounter = s->nbFrame[0];                        /* :1366 */
    }
    if (s->nIn2 - s->nOut2 > 2) {
        ns_stage(s, 1, out);
        s->nOut2++;
        s->frameCounter = s->nbFrame[0];
    }
    if (s->nIn1) memmove(s->buf[0], s->buf[0] + HOP, 240 * sizeof(float));
    if (s->nIn2) memmove(s->buf[1], s->buf[1] + HOP, 240 * sizeof(float));
    if (s->nOut2 > 0) {
        /* DCOffsetFil: NoiseSup.c:182-198 */
        for (i = 0; i < HOP; i++) {
            float x = out[i];
            out[i] = (float)((out[i] - s->dcX) + 0.9990234375 * s->dcY);
            s->dcX = x;
            s->dcY = out[i];
        }
        return 1;
    }
    return 0;
}

/* float -> int16 as the reference's (FILE_TYPE) cast compiles on x86-64: truncate to int32, keep
 * the low 16 bits (ParmInterface.c:266) */
static short cast_i16(float v)
{
    if (!(v > -2147483648.0f && v < 2147483648.0f)) return 0;
    return (short)(int)v;
}

/* ------------------------------------------------------------------------------------------
 * CompCeps: CompCeps.c:368-549
 * ---------------------------------------------------------------------------------------- */
static void compceps(const float *cur /* cur[-1] valid */, float *coef14)
{
    float fb[NFFT], logE;
    const float floorFB = (float)exp((double)-10.0), floorE = (float)exp((double)-50.0);
    int i, j;

    logE = 0.0f;
    for (i = 0; i < WIN; i++) logE += cur[i] * cur[i];
    if (logE < floorE)
        logE = (float)-50.0;
    else
        logE = (float)log((double)logE);

    for (i = 0; i < WIN; i++) fb[i] = (float)(cur[i] - 0.90 * cur[i - 1]);
    for (i = 0; i < WIN / 2; i++) fb[i] *= T.hamming[i];
    for (i = WIN / 2; i < WIN; i++) fb[i] *= T.hamming[WIN - 1 - i];
    for (i = WIN; i < NFFT; i++) fb[i] = 0.0f;
    ora_rfft(fb, NFFT, 8);

    fb[0] = (float)((double)fb[0] * fb[0]);
    for (i = 1; i < NFFT / 2; i++)
        fb[i] = (float)((double)fb[i] * (double)fb[i] + (double)fb[NFFT - i] * (double)fb[NFFT - i]);
    fb[NFFT / 2] = (float)((double)fb[NFFT / 2] * fb[NFFT / 2]);

    mel_fb(fb, T.ccmel, 23);
    for (i = 0; i < 23; i++) fb[i] = (fb[i] < floorFB) ? (float)-10.0 : (float)log((double)fb[i]);

    /* DCT: CompCeps.c:203-227 */
    for (i = 1; i <= 12; i++) {
        fb[23 + i - 1] = 0.0f;
        for (j = 0; j < 23; j++) fb[23 + i - 1] += fb[j] * T.dct[i - 1][j];
    }
    fb[23 + 12] = 0.0f;
    for (i = 0; i < 23; i++) fb[23 + 12] += fb[i];
    fb[23 + 13] = logE; /* Noc0 == 0 */
    for (i = 0; i < 14; i++) coef14[i] = fb[23 + i];
}

void ora_compceps_frame(const float *data201, float *coef14)
{
    tables_init();
    compceps(data201 + 1, coef14);
}

/* ------------------------------------------------------------------------------------------
 * Utterance drivers: ParmInterface.c:208-330 + AdvFrontEnd.c:125-210
 * ---------------------------------------------------------------------------------------- */
/* DoNoiseSup on float frames, no zero-frame gate (the frame-level plugin entry, NoiseSup.c:1061):
 * out gets 80 floats per TRUE return, produced[f] = the return value.  Returns the number of outputs. */
long ora_ns_stream_f32(const float *in, long nframes, float *out, int *produced)
{
    ns_state *s = (ns_state *)malloc(sizeof *s);
    long f, nout = 0;
    tables_init();
    ns_init(s);
    for (f = 0; f < nframes; f++) {
        float y[HOP];
        int ok = ns_step(s, in + f * HOP, y);
        produced[f] = ok ? 1 : 0;
        if (ok) {
            memcpy(out + nout * HOP, y, sizeof y);
            nout++;
        }
    }
    free(s);
    return nout;
}

long ora_ns_trace(const short *in, long n, short *out_i16, float *den_f32, float *ceps,
                  float *scal, float *spec, long *counts)
{
    ns_state *s = (ns_state *)malloc(sizeof *s);
    float hist[241]; /* denoisedBuf: BufInAlloc(200 + 200%80 + 1), ParmInterface.c:178-181 */
    short last[HOP];
    long nfr = n / HOP, f, nout = 0, nceps = 0;
    int onset = 0, i;

    tables_init();
    ns_init(s);
    memset(hist, 0, sizeof hist);
    memset(last, 0, sizeof last);

    for (f = 0; f < nfr; f++) {
        float cur[HOP];
        int any = 0;
        memmove(hist, hist + HOP, (241 - HOP) * sizeof(float)); /* BufInShiftToPut */
        for (i = 0; i < HOP; i++) {
            cur[i] = (float)in[f * HOP + i];
            any |= (in[f * HOP + i] != 0);
        }
        /* zero-frame gate (ParmInterface.c:244-251): (int)Σx² != 0  <=>  some sample != 0 */
        if (any || onset) {
            onset = 1;
            if (ns_step(s, cur, hist + 241 - HOP)) {
                for (i = 0; i < HOP; i++) last[i] = cast_i16(hist[241 - HOP + i]);
                if (den_f32) memcpy(den_f32 + nout * HOP, hist + 241 - HOP, HOP * sizeof(float));
                nout++;
                if (ceps && nout >= 3) {
                    compceps(hist + 1, ceps + nceps * 14);
                    nceps++;
                }
            }
        }
        if (out_i16) memcpy(out_i16 + f * HOP, last, sizeof last);
        if (scal) {
            float *p = scal + f * ORA_TRACE_NSCAL;
            p[0] = (float)s->nbFrame[0];
            p[1] = (float)s->nbFrame[1];
            p[2] = (float)s->flagVAD;
            p[3] = (float)s->hangOver;
            p[4] = (float)s->nbSpeechFrames;
            p[5] = s->meanEn;
            p[6] = s->alfaGF;
            p[7] = s->lowSNRtrack;
            p[8] = s->denEn[0];
            p[9] = s->denEn[1];
            p[10] = s->denEn[2];
            p[11] = s->dcX;
            p[12] = s->dcY;
            p[13] = (float)s->nIn1;
            p[14] = (float)s->nIn2;
            p[15] = (float)s->nOut2;
        }
        if (spec) {
            float *p = spec + f * 4 * NSPEC;
            memcpy(p, s->noise[0], NSPEC * sizeof(float));
            memcpy(p + NSPEC, s->noise[1], NSPEC * sizeof(float));
            memcpy(p + 2 * NSPEC, s->den[0], NSPEC * sizeof(float));
            memcpy(p + 3 * NSPEC, s->den[1], NSPEC * sizeof(float));
        }
    }
    free(s);
    if (counts) {
        counts[0] = nout;
        counts[1] = nceps;
    }
    return nfr;
}

/* ------------------------------------------------------------------------------------------
 * the chain after NoiseSup that the reference author commented out
 * (ParmInterface.c:274-311): WaveProc -> CompCeps -> PostProc -> VAD, FlushAdvProcess at the end.
 * ---------------------------------------------------------------------------------------- */

/* DoWaveProc, WaveProc.c:397-455 with TeagerEng :216-226, GetTeagerFilter :244-330,
 * GetMaximaPositions :102-190 (sort_it :60-84 yields ascending positions).  d[200] in place. */
static void waveproc(float *d)
{
    enum { N = 200 };
    float tw[N], energy = 0.0f, lowVal, highVal;
    const float eps = (float)0.2; /* WP_EPS stored in an X_FLOAT32 field */
    int q[N], sm[N], pos[24], R[12], Lf[12];
    int i, j, k, p0 = 0, found = 0, cR = 0, cL = 0, nom, best = 0;

    for (i = 0; i < N; i++) energy += (d[i] * d[i]);
    if (!(energy >= 100.0)) return;

    tw[0] = fabs(d[0] * d[0] - d[0] * d[1]);
    for (i = 1; i < N - 1; i++) tw[i] = fabs(d[i] * d[i] - d[i - 1] * d[i + 1]);
    tw[N - 1] = fabs(d[N - 1] * d[N - 1] - d[N - 2] * d[N - 1]);

    /* 9-point running sum of (int)floor(0.25 T + 0.5) with the end values repeated; int32 wraps */
    for (i = 0; i < N; i++) q[i] = (int)floor(tw[i] * 0.25 + 0.5);
    for (i = 0; i < N; i++) {
        unsigned acc = 0;
        for (k = -4; k <= 4; k++) {
            j = i + k;
            j = j < 0 ? 0 : (j > N - 1 ? N - 1 : j);
            acc += (unsigned)q[j];
        }
        sm[i] = (int)acc;
    }

    /* global maximum (first strict one), then neighbours 25..79 samples away, last of equals */
    for (i = 0; i < N; i++)
        if (sm[i] > best) {
            best = sm[i];
            p0 = i;
            found = 1;
        }
    nom = 0;
    if (found) {
        R[0] = Lf[0] = p0;
        while ((R[cR] + 25) < N && found) {
            int m = 0;
            found = 0;
            for (i = 25; i < 80; i++)
                if (R[cR] + i < N && sm[R[cR] + i] >= m) {
                    found = 1;
                    m = sm[R[cR] + i];
                    R[cR + 1] = R[cR] + i;
                }
            if (found) cR++;
        }
        found = 1;
        while ((Lf[cL] - 25) > 0 && found) {
            int m = 0;
            found = 0;
            for (i = 25; i < 80; i++)
                if (Lf[cL] - i > -1 && sm[Lf[cL] - i] >= m) {
                    found = 1;
                    m = sm[Lf[cL] - i];
                    Lf[cL + 1] = Lf[cL] - i;
                }
            if (found) cL++;
        }
        for (i = cL; i >= 1; i--) pos[nom++] = Lf[i]; /* ascending: left ones, centre, right ones */
        for (i = 0; i <= cR; i++) pos[nom++] = R[i];
    }

    lowVal = (1 - eps) / 2.0;
    highVal = (1 + eps) / 2.0;
    for (i = 0; i < N; i++) tw[i] = lowVal;
    if (nom > 1) {
        for (i = 0; i < nom - 1; i++)
            for (j = pos[i] - 4; j < (pos[i] - 4) + ((80 * (pos[i + 1] - pos[i]) + 99) / 100); j++)
                if (j >= 0) tw[j] = highVal;
        for (j = pos[nom - 1] - 4; j < (pos[nom - 1] - 4) + ((80 * (pos[nom - 1] - pos[nom - 2]) + 99) / 100); j++)
            if (j < N) tw[j] = highVal;
    }
    for (i = 0; i < N - 1; i++) d[i] *= (tw[i] + tw[i + 1]);
    d[N - 1] *= (tw[N - 1] + tw[N - 1]);
}

/* DoPostProc (blind LMS equalisation of c1..c12), PostProc.c:123-149; w[12] is the state */
static void postproc(float *coef14, float *w)
{
    static const float target[12] = {(float)-6.618909, (float)0.198269, (float)-0.740308, (float)0.055132,
                                     (float)-0.227086, (float)0.144280, (float)-0.112451, (float)-0.146940,
                                     (float)-0.327466, (float)0.134571, (float)0.027884,  (float)-0.114905};
    const float lambda = (float)0.0087890625;
    float wp = (coef14[13] * (float)64 - (float)211) / (float)64; /* Noc0 == 0: logE is Coef[13] */
    int i;
    if (wp < 0)
        wp = 0;
    else if (wp > 1)
        wp = lambda;
    else
        wp *= lambda;
    for (i = 0; i < 12; i++) {
        float dif = ((coef14[i] - w[i]) - target[i]);
        coef14[i] = coef14[i] - w[i];
        w[i] += dif * wp;
    }
}

/* DoVADProc / DoVADFlush, VAD.c:219-317 / :342-433 */
typedef struct {
    int focus, hangOver, hCount, vCount, flushFocus;
    float buf[7][15];
} vad_state;

static void vad_init(vad_state *v)
{
    memset(v, 0, sizeof *v);
    v->hangOver = 23;
    v->flushFocus = -1;
}

static int vad_focal(int focus, int off)
{
    int t = focus + off;
    if (t > 6) t -= 7;
    if (t < 0) t += 7;
    return t;
}

/* the decision shared by DoVADProc and DoVADFlush once FrameCounter > BUFFER_SIZE + 3 */
static void vad_decide(vad_state *v, int focus, int frameCounter, float *feat15)
{
    int i, sum = 0, trigger = 0, r;
    for (i = 0; i < 7; i++) {
        r = vad_focal(focus, i + 1);
        if (v->buf[r][14])
            sum++;
        else {
            if (sum > trigger) trigger = sum;
            sum = 0;
        }
    }
    if (sum > trigger) trigger = sum;
    if (trigger >= 4) {
        v->hCount = v->hangOver;
        if (frameCounter <= 35) v->hangOver = 50;
    }
    if (v->hCount && trigger < 3) v->hCount--;
    if (trigger >= 3) v->vCount = 5;
    if (v->vCount && trigger < 3) v->vCount--;
    r = vad_focal(focus, 1);
    for (i = 0; i < 15; i++) feat15[i] = v->buf[r][i];
    feat15[14] = (v->vCount || v->hCount || trigger >= 3) ? 1.0f : 0.0f;
}

static int vad_proc(vad_state *v, float *feat15, const ns_state *s)
{
    int i, focus = v->focus + 1;
    if (focus == 7) focus = 0;
    for (i = 0; i < 14; i++) v->buf[focus][i] = feat15[i];
    v->buf[focus][14] =
        (s->speechFoundSpec || s->speechFoundMel || s->speechFoundVar || s->speechFoundVADNS) ? 1.0f : 0.0f;
    v->focus = focus;
    if (s->frameCounter > 7 + 3) {
        vad_decide(v, focus, s->frameCounter, feat15);
        return 1;
    }
    return 0;
}

static int vad_flush(vad_state *v, float *feat15, ns_state *s)
{
    int focus = v->focus;
    if (v->flushFocus == -1) v->flushFocus = focus;
    focus++;
    if (focus == 7) focus = 0;
    if (focus == v->flushFocus) return 0;
    s->frameCounter++;
    if (s->frameCounter > 7 + 3) vad_decide(v, focus, s->frameCounter, feat15);
    v->focus = focus;
    return 1; /* TRUE even when the counter gate kept FeatureBuffer untouched (VAD.c:421-428) */
}

/* Same contract as ref_afe_trace (oracle/ref_driver.c). */
long ora_afe_trace(const short *in, long n, int *flags, float *feat_cc, float *feat_pp, float *vad_out,
                   long *counts)
{
    ns_state *s = (ns_state *)malloc(sizeof *s);
    vad_state vad;
    float hist[241], lms[12], feat[15];
    long nfr = n / HOP, f, nout = 0, nceps = 0, nvad = 0;
    int onset = 0, i;

    tables_init();
    ns_init(s);
    vad_init(&vad);
    memset(hist, 0, sizeof hist);
    memset(lms, 0, sizeof lms);
    memset(feat, 0, sizeof feat);

    for (f = 0; f < nfr; f++) {
        float cur[HOP];
        int any = 0;
        memmove(hist, hist + HOP, (241 - HOP) * sizeof(float));
        for (i = 0; i < HOP; i++) {
            cur[i] = (float)in[f * HOP + i];
            any |= (in[f * HOP + i] != 0);
        }
        if (any || onset) {
            onset = 1;
            if (ns_step(s, cur, hist + 241 - HOP)) {
                nout++;
                if (nout >= 3) {
                    float frame[241];
                    memcpy(frame, hist, sizeof frame);
                    waveproc(frame + 1);
                    compceps(frame + 1, feat);
                    memcpy(feat_cc + nceps * 14, feat, 14 * sizeof(float));
                    postproc(feat, lms);
                    memcpy(feat_pp + nceps * 14, feat, 14 * sizeof(float));
                    nceps++;
                    if (vad_proc(&vad, feat, s)) {
                        memcpy(vad_out + nvad * 15, feat, 15 * sizeof(float));
                        nvad++;
                    }
                }
            }
        } else { /* null MFCC vector, ParmInterface.c:314-329 */
            for (i = 0; i < 14; i++) feat[i] = 0.0f;
            memcpy(vad_out + nvad * 15, feat, 14 * sizeof(float));
            vad_out[nvad * 15 + 14] = 0.0f;
            nvad++;
        }
        if (flags) {
            int *p = flags + f * 5;
            p[0] = s->speechFoundVar;
            p[1] = s->speechFoundSpec;
            p[2] = s->speechFoundMel;
            p[3] = s->speechFoundVADNS;
            p[4] = s->frameCounter;
        }
    }
    while (vad_flush(&vad, feat, s)) {
        memcpy(vad_out + nvad * 15, feat, 15 * sizeof(float));
        nvad++;
    }
    free(s);
    counts[0] = nout;
    counts[1] = nceps;
    counts[2] = nvad;
    return nfr;
}

int ora_etsi_denoise(const short *in, short *out, long n)
{
    ora_ns_trace(in, n, out, NULL, NULL, NULL, NULL, NULL);
    return 0;
}
