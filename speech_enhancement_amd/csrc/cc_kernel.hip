/*
 * cc_kernel.hip -- batched rfft and the CompCeps cepstral front-end, gfx950 (MI355X).
 *
 * Both are recursion-free per frame: a wave takes two frames (rfft) or a tile of 16 frames (CompCeps) at a time.
 *   rfft256_kernel           etsi/cpp/rfft.c:45-180 on [nframes][256] floats, in place or not
 *   compceps_frames_kernel   DoCompCeps (etsi/cpp/CompCeps.c:309-318 -> WI8CompCeps :368-549) on
 *                            caller-supplied frames of 201 floats (Data[-1..199])
 *   compceps_kernel          the same, reading frames straight out of the float NoiseSup stream:
 *                            frame j of an utterance = denoised[80j-1 .. 80j+199], available once 3
 *                            NoiseSup outputs exist (the commented-out driver block
 *                            etsi/cpp/ParmInterface.c:275-293)
 */
#include "ns_core.h" /* the kernels' own double log and its guard (ns_ln, ns_near_float_boundary, ns_ln_cr) */

namespace sea {

/* Streaming form: a wave transforms TWO frames at a time (lanes 0..31 / 32..63, the dual transform of
 * sea_device.h: swizzled work area, five LDS round trips), 2 KB in + 2 KB out per pass.  Budget per frame at the
 * HBM rate: ~240 clk per CU, i.e. ~960 SIMD-clk and ~240 LDS-array clk; the dual transform needs ~180 vector
 * issue slots and ~144 LDS clk per frame (the one-frame-per-wave form: ~300 and ~320 -- LDS-bound).  Each lane
 * gathers its eight inputs n0 + 32 bitrev3(j) straight from global memory (per instruction the 32 lanes of a frame
 * read one contiguous 128-byte line), one pair ahead; results leave as one float4 per lane and frame. */
#ifndef SEA_RFFT_ADDR_LDS
#define SEA_RFFT_ADDR_LDS 0 /* operand addresses in VGPRs: 4.45 TB/s; in LDS (76 VGPRs, five waves per SIMD): 4.04 */
#endif
#ifndef SEA_RFFT_WAVES
#define SEA_RFFT_WAVES 4
#endif
#ifndef SEA_RFFT_NT
#define SEA_RFFT_NT 1
#endif
__global__ __launch_bounds__(64, SEA_RFFT_WAVES) void rfft256_kernel(const float *in, float *out, long long nframes,
                                                                     const sea_fft_tables *t)
{
    __shared__ __attribute__((aligned(16))) float work[512];
    __shared__ uint4 addrLds[SEA_RFFT_ADDR_LDS ? SEA_FFT_LSTAGES * 64 : 1];
    const int lane = threadIdx.x;
    Fft2Regs R;
    load_fft2_regs<SEA_RFFT_ADDR_LDS != 0>(R, t, lane, addrLds);
    wave_sync();
    const int n0 = lane & 31, h = lane >> 5;
    unsigned oa[4]; /* where elements 4l..4l+3 of the reference's order sit in a (swizzled) work area */
#pragma unroll
    for (int q = 0; q < 4; ++q) oa[q] = fft_swz(4u * (unsigned)lane + (unsigned)q);
    const long long npair = (nframes + 1) >> 1;
    auto load = [&](long long p, float(&e)[8]) {
        const long long f = 2 * p + h;
        const float *x = in + (f < nframes ? f : 0) * 256 + n0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            constexpr int kRev3[8] = {0, 4, 2, 6, 1, 5, 3, 7};
            const float v = SEA_RFFT_NT ? __builtin_nontemporal_load(x + 32 * kRev3[k]) : x[32 * kRev3[k]];
            e[k] = (f < nframes) ? v : 0.0f;
        }
    };
    float cur[8], nxt[8];
    long long p = blockIdx.x;
    if (p < npair) load(p, cur);
    for (; p < npair; p += gridDim.x) {
        const long long pn = p + gridDim.x;
#pragma unroll
        for (int k = 0; k < 8; ++k) nxt[k] = 0.0f;
        if (pn < npair) load(pn, nxt);
        rfft256_dual<SEA_RFFT_ADDR_LDS != 0>(cur, work, R);
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const long long f = 2 * p + hh;
            if (f < nframes) {
                const float *w = work + 256 * hh;
                float4 v;
                v.x = fft_at(w, oa[0]);
                v.y = fft_at(w, oa[1]);
                v.z = fft_at(w, oa[2]);
                v.w = fft_at(w, oa[3]);
                typedef float v4f __attribute__((ext_vector_type(4)));
                if (SEA_RFFT_NT)
                    __builtin_nontemporal_store(v4f{v.x, v.y, v.z, v.w}, reinterpret_cast<v4f *>(out + f * 256 + 4 * lane));
                else
                    *reinterpret_cast<float4 *>(out + f * 256 + 4 * lane) = v;
            }
        }
        wave_sync();
#pragma unroll
        for (int k = 0; k < 8; ++k) cur[k] = nxt[k];
    }
}

/* rfft (x, n, m) for every size the reference's routine takes (etsi/cpp/rfft.c:45-180): ONE workgroup per transform walks
 * the schedule sea_rfft_schedule() unrolled on the host -- digit reversal, the length-two butterflies, then level by
 * level the plain, pi/4 and twiddled butterflies of all blocks, which touch disjoint elements within a level -- with the
 * frame in LDS (n floats, dynamic).  A convenience path behind the drop-in symbol (the hot path's only size, (256, 8),
 * keeps rfft256_kernel; the 16 k-native variant's (512, 8) keeps its own register schedule in ns16k_kernel.hip): written
 * for exactness -- the reference's operations in the reference's order per butterfly -- not for speed. */
__global__ __launch_bounds__(256) void rfft_any_kernel(float *x, const unsigned *sched, long long nframes)
{
    extern __shared__ __attribute__((aligned(16))) float xs[];
    const int n = (int)sched[0], m = (int)sched[1];
    const unsigned *rev = sched + sched[2], *len2 = sched + sched[3];
    const int nlen2 = (int)sched[4], tid = threadIdx.x;
    for (long long f = blockIdx.x; f < nframes; f += gridDim.x) {
        float *xf = x + f * n;
        for (int i = tid; i < n; i += 256) xs[rev[i]] = xf[i];
        __syncthreads();
        for (int w = tid; w < nlen2; w += 256) { /* :82-96 */
            const int i0 = (int)len2[w];
            const float a0 = xs[i0], a1 = xs[i0 + 1];
            xs[i0] = a0 + a1;
            xs[i0 + 1] = a0 - a1;
        }
        __syncthreads();
        int n2 = 2;
        for (int k = 1; k < m; ++k) {
            n2 <<= 1;
            const int n4 = n2 >> 2, n8 = n2 >> 3, per = n8 > 0 ? n8 : 1;
            const unsigned *blk = sched + sched[5 + 3 * k];
            const float *tw = reinterpret_cast<const float *>(sched + sched[7 + 3 * k]);
            const int items = (int)sched[6 + 3 * k] * per;
            for (int w = tid; w < items; w += 256) {
                const int i = (int)blk[w / per], j = w % per;
                if (j == 0) {
                    { /* :108-115 */
                        const int i1 = i, i3 = i1 + 2 * n4, i4 = i3 + n4;
                        const float x1 = xs[i1], x3 = xs[i3], x4 = xs[i4];
                        const float t1 = x4 + x3;
                        xs[i4] = x4 - x3;
                        xs[i3] = x1 - t1;
                        xs[i1] = x1 + t1;
                    }
                    if (n4 != 1) { /* :117-128; the division by sqrt 2 in double == this multiplication for every float
                                      (sea_device.h, swept over all 2^32 by sea_selftest_pi4) */
                        const int i1 = i + n8, i2 = i1 + n4, i3 = i2 + n4, i4 = i3 + n4;
                        const float x1 = xs[i1], x2 = xs[i2], x3 = xs[i3], x4 = xs[i4];
                        const float t1 = (float)((double)(x3 + x4) * 0.70710678118654752440);
                        const float t2 = (float)((double)(x3 - x4) * 0.70710678118654752440);
                        xs[i4] = x2 - t1;
                        xs[i3] = -x2 - t1;
                        xs[i2] = x1 - t2;
                        xs[i1] = x1 + t2;
                    }
                } else { /* :145-174 */
                    const float cc1 = tw[4 * j], ss1 = tw[4 * j + 1], cc3 = tw[4 * j + 2], ss3 = tw[4 * j + 3];
                    const int i1 = i + j, i2 = i1 + n4, i3 = i2 + n4, i4 = i3 + n4;
                    const int i5 = i + n4 - j, i6 = i5 + n4, i7 = i6 + n4, i8 = i7 + n4;
                    const float x1 = xs[i1], x2 = xs[i2], x3 = xs[i3], x4 = xs[i4], x5 = xs[i5], x6 = xs[i6], x7 = xs[i7], x8 = xs[i8];
                    float t1 = x3 * cc1 + x7 * ss1;
                    float t2 = x7 * cc1 - x3 * ss1;
                    float t3 = x4 * cc3 + x8 * ss3;
                    float t4 = x8 * cc3 - x4 * ss3;
                    const float t5 = t1 + t3, t6 = t2 + t4;
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
            __syncthreads();
        }
        for (int i = tid; i < n; i += 256) xf[i] = xs[i];
        __syncthreads();
    }
}

/* ==================================================================================================
 * Tiled CompCeps: one wave owns a TILE of kCcT consecutive frames.
 *
 * A one-frame-per-wave form (round 1: 2.5 ms for 810 511 frames) spends most of its time on work that every one of
 * its 64 lanes repeats: the frame's 200-term in-order energy sum (CompCeps.c:413-423) and its double-precision log.
 * Here the tile's samples are staged in LDS once and
 *   * the energy sums run LANE = FRAME (lane f adds the 200 squares of frame f in order; the hop
 *     blocks sit 81 words apart -- one pad word per 80 samples -- so the 16 lanes hit 16 banks),
 *     and the log of the sum is evaluated once per frame, again lane = frame;
 *   * two frames at a time go through the dual transform (lanes 0..31 / 32..63, swizzled work area,
 *     five LDS round trips instead of six), their power spectra and the 23 mel triangles (lane =
 *     (frame, band));
 *   * the 23 log energies of all frames are taken lane = (frame, band) flattened (6 evaluations of the
 *     double log per tile instead of 16), the DCT lane = (frame, coefficient) flattened (4 passes).
 * Arithmetic per value is WI8CompCeps' (CompCeps.c:368-549), operation by operation: pre-emphasis in double
 * (:427-429), power spectrum products and sum in double (:451-459), taps / DCT terms in their order.  Zero-weight taps stand in for the band length test (acc + p * 0 == acc: p is a
 * finite power, acc >= +0) and c0's plain sum is a DCT row of ones (x * 1.0f == x).
 * ================================================================================================ */
namespace {

#ifndef SEA_CC_TILE
#define SEA_CC_TILE 16
#endif
constexpr int kCcT = SEA_CC_TILE;

#ifndef SEA_CC_FASTLOG
#define SEA_CC_FASTLOG 1
#endif
#ifndef SEA_CC_WAVES
#define SEA_CC_WAVES 3 /* waves per SIMD the register allocation leaves room for (LDS allows twelve waves per CU) */
#endif

/* (float)log((double)v) for a positive normal float v, as CompCeps.c:423 / :511 take it.  The library's double log
 * costs ~150 instructions; the kernels' own table-driven one (ns_core.h, ns_ln: within ~1.1 ulp) gives the same float
 * unless the double lands within a few ulps of a float ROUNDING BOUNDARY -- there (probability ~2^-27 per call) the
 * logarithm is redone in double-double arithmetic and rounded once, exactly like the two NoiseSup sites
 * (ns_near_float_boundary; window 4 ulps: ours 1.1 + glibc's 0.52, doubled). */
__device__ __forceinline__ float cc_logf(float v)
{
#if SEA_CC_FASTLOG
    const double l = ns_ln<false>((double)v);
    if (__builtin_expect(ns_near_float_boundary(l, 4), 0)) return (float)ns_ln_cr((double)v);
    return (float)l;
#else
    return (float)log((double)v);
#endif
}

template <bool SHARED, int T = kCcT>
struct CcGeom {
    /* SHARED: frames of one utterance, 80 samples apart, share their samples; word x of the span (x = 0 is
     * Data[-1] of the tile's first frame) sits at x + x / 80.  Otherwise: kCcT separate frames of 201 floats. */
    static constexpr int FS = SHARED ? 81 : 201;
    static constexpr int SPAN = SHARED ? 81 * (T - 1) + 204 : 201 * T;
};

template <bool SHARED, int T = kCcT>
struct __attribute__((aligned(16))) CcTileLds {
    float span[(CcGeom<SHARED, T>::SPAN + 3) & ~3];
    float work[512];                  /* the dual transform's work area; after the tile's last pair, its T x 14 output rows */
    float pw[2][SEA_CC_PWROW];        /* 129 power bins per frame, zeros behind (the mel taps read past 128) */
    float fb[T][24];
    float dctT[SEA_CC_NCHAN * 16];
};

struct CcTileConst {
    Fft2Regs fft;
    float win8[8];
    int qd[8], qm[8];                 /* word offsets of Data[idx], Data[idx-1] from the frame's base; qd < 0: idx >= 200 */
    int pwAB, pwCD;                   /* words from the pair's first power row: bins j (+ 64) and 64 - j (+ 64) of this lane's
                                         last-level item (rfft256_dual_keep_last), row of its transform */
    bool pairLane;                    /* the item with bins 0, 64, 128 | 32, 96 */
    int melBase, melFb;               /* this lane's (frame, band) of the mel pass: sea_tables.h, melLaneBase */
    float melW[SEA_CC_TAPS2];
    float floorFB, floorE;
};

template <bool SHARED>
__device__ __forceinline__ int cc_q(int x) /* word offset of Data[x-1] within its frame, x = 0..200 */
{
    return SHARED ? x + (x >= 80 ? 1 : 0) + (x >= 160 ? 1 : 0) : x;
}

template <bool SHARED, int T = kCcT>
__device__ __forceinline__ void load_cc_tile_const(CcTileConst &C, CcTileLds<SHARED, T> &L, const sea_cc_tables *t, int lane)
{
    load_fft2_regs<false>(C.fft, &t->fft, lane, nullptr);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        constexpr int kRev3[8] = {0, 4, 2, 6, 1, 5, 3, 7};
        const int idx = (lane & 31) + 32 * kRev3[k];
        C.win8[k] = t->win8[k][lane];
        C.qd[k] = (idx < SEA_WIN) ? cc_q<SHARED>(idx + 1) : -1;
        C.qm[k] = (idx < SEA_WIN) ? cc_q<SHARED>(idx) : 0;
    }
    {
        const unsigned item = t->fft.fft2Item[SEA_FFT_LSTAGES - 1][lane & 31];
        C.pairLane = (item >> 16) == SEA_BF_PAIR;
        const int ja = (int)(item & 255u), jc = C.pairLane ? (int)((item >> 8) & 255u) : ja; /* j, j | 0, 32 */
        C.pwAB = SEA_CC_PWROW * (lane >> 5) + ja;
        C.pwCD = SEA_CC_PWROW * (lane >> 5) + 64 - jc;
    }
    C.melBase = t->melLaneBase[lane];
    C.melFb = t->melLaneFb[lane];
#pragma unroll
    for (int i = 0; i < SEA_CC_TAPS2; ++i) C.melW[i] = t->melLaneW[i][lane];
    C.floorFB = t->floorFB;
    C.floorE = t->floorE;
    for (int i = lane; i < SEA_CC_NCHAN * 16; i += kLanes) L.dctT[i] = t->dctT[i >> 4][i & 15];
    for (int i = lane; i < 2 * SEA_CC_PWROW; i += kLanes) (&L.pw[0][0])[i] = 0.0f;
    wave_sync();
}

/* timing-only diagnostic (-DSEA_CC_TIMING, tools/cc_phases.py): shader clocks workgroup 0 of compceps_kernel spends per step of a tile */
#ifdef SEA_CC_TIMING
__device__ unsigned long long g_cc_ck[8];
extern "C" int sea_cc_timing(unsigned long long *out8, int reset)
{
    if (reset) {
        unsigned long long z[8] = {};
        return hipMemcpyToSymbol(HIP_SYMBOL(g_cc_ck), z, sizeof z) != hipSuccess;
    }
    return hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_cc_ck), 8 * sizeof(unsigned long long)) != hipSuccess;
}
__device__ unsigned g_cc_wave[16384 * 4]; /* per wave of compceps_kernel: start, end (constant 100 MHz counter), HW_ID, XCC_ID */
extern "C" int sea_cc_waves(unsigned *out, int n) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_cc_wave), (size_t)n * 4 * sizeof(unsigned)) != hipSuccess; }
#define CC_CK_START unsigned long long cck_ = clock64()
#define CC_CK(k) do { const unsigned long long c_ = clock64(); if (SHARED && blockIdx.x == 0 && threadIdx.x == 0) g_cc_ck[k] += c_ - cck_; cck_ = c_; } while (0)
#else
#define CC_CK_START
#define CC_CK(k)
#endif

/* the staged tile -> nv rows of 14 coefficients at dst */
template <bool SHARED, int T = kCcT>
__device__ __forceinline__ void cc_tile(CcTileLds<SHARED, T> &L, const CcTileConst &C, int nv, float *dst, int lane)
{
    constexpr int FS = CcGeom<SHARED, T>::FS;
    /* logE (CompCeps.c:413-423): lane f sums the squares of frame f in sample order */
    float logE; /* three ranges of the walk, each with a constant pad */
    CC_CK_START;
    {
        const float *p = L.span + FS * (lane & (T - 1));
        float acc = 0.0f;
        if (lane < T) {
            if (SHARED) {
#pragma unroll 8
                for (int x = 1; x < 80; ++x) { const float v = p[x]; acc += v * v; }
#pragma unroll 8
                for (int x = 80; x < 160; ++x) { const float v = p[x + 1]; acc += v * v; }
#pragma unroll 8
                for (int x = 160; x < 201; ++x) { const float v = p[x + 2]; acc += v * v; }
            } else {
#pragma unroll 8
                for (int x = 1; x < 201; ++x) { const float v = p[x]; acc += v * v; }
            }
        }
        logE = (acc < C.floorE) ? (float)-50.0 : cc_logf(acc);
    }
    CC_CK(1);
    const int npair = (nv + 1) >> 1;
    for (int pr = 0; pr < npair; ++pr) {
        const int h = lane >> 5;
        const int f = 2 * pr + h;
        const bool act = f < nv;
        const float *p = L.span + FS * f;
        /* pre-emphasis in double (:427-429), symmetric Hamming (:115-125), zero padding (:439-440) */
        float e[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float v = 0.0f;
            if (act && C.qd[k] >= 0) {
                const float d = p[C.qd[k]], dm1 = p[C.qm[k]];
                v = (float)((double)d - 0.90 * (double)dm1) * C.win8[k];
            }
            e[k] = v;
        }
        /* the transform's last level stays in registers: every lane holds four complete bins of its frame (one lane per frame
         * five), whose power -- products and sum in double (:451-459) -- goes straight to the frame's row */
        float o[8];
        rfft256_dual_keep_last<false>(e, L.work, C.fft, o);
        {
            const bool pl = C.pairLane;
            const float ia = pl ? 0.0f : o[7], ib = pl ? o[3] : o[6], ic = pl ? o[7] : o[3], id = pl ? o[6] : o[2];
            auto power = [](float re, float im) { return (float)((double)re * (double)re + (double)im * (double)im); };
            float *rowAB = &L.pw[0][0] + C.pwAB, *rowCD = &L.pw[0][0] + C.pwCD;
            rowAB[0] = power(o[0], ia);   /* bin j | 0: (float)(re * re + 0.0) == (float)(re * re) */
            rowAB[64] = power(o[1], ib);  /* bin 64 + j | 64 */
            rowCD[0] = power(o[4], ic);   /* bin 64 - j | 32 */
            rowCD[64] = power(o[5], id);  /* bin 128 - j | 96 */
            if (pl) rowAB[128] = power(o[2], 0.0f); /* bin 128 */
        }
        wave_sync();
        /* 23 mel triangles (DoMelFB, MelProc.c:82-104): lane = one (frame, band) of the pair, dealt so that the aligned
         * pairs the lanes of a group read lie on different banks (round 4: 3-way conflicts on every tap before) */
        if (C.melFb >= 0 && 2 * pr + (C.melFb >= 24 ? 1 : 0) < nv) {
            const float2 *q = reinterpret_cast<const float2 *>(&L.pw[0][0] + C.melBase);
            float acc = 0.0f;
#pragma unroll
            for (int i = 0; i < SEA_CC_TAPS2 / 2; ++i) {
                const float2 v = q[i];
                acc = acc + v.x * C.melW[2 * i];
                acc = acc + v.y * C.melW[2 * i + 1];
            }
            (&L.fb[2 * pr][0])[C.melFb] = acc;
        }
        wave_sync();
    }
    CC_CK(2);
    /* natural log with floor (:509-513): lane = (frame, band) flattened */
    for (int idx = lane; idx < nv * SEA_CC_NCHAN; idx += kLanes) {
        const int f = idx / SEA_CC_NCHAN, b = idx - f * SEA_CC_NCHAN;
        const float v = L.fb[f][b];
        L.fb[f][b] = (v < C.floorFB) ? (float)-10.0 : cc_logf(v);
    }
    wave_sync();
    CC_CK(3);
    /* DCT (:203-227): lane = (frame, coefficient) flattened; c = 12 is c0, logE goes to c = 13 */
    for (int idx = lane; idx < nv * 13; idx += kLanes) {
        const int f = idx / 13, c = idx - f * 13;
        float acc = 0.0f;
#pragma unroll
        for (int j = 0; j < SEA_CC_NCHAN; ++j) acc += L.fb[f][j] * L.dctT[j * 16 + c];
        L.work[f * SEA_CC_NCEP + c] = acc;
    }
    if (lane < nv) L.work[lane * SEA_CC_NCEP + 13] = logE;
    wave_sync();
    CC_CK(4);
    for (int idx = lane; idx < nv * SEA_CC_NCEP; idx += kLanes) dst[idx] = L.work[idx];
    wave_sync();
    CC_CK(5);
}

} // namespace

__global__ __launch_bounds__(64) void compceps_frames_kernel(const float *data201, float *coef14,
                                                             long long nframes, const sea_cc_tables *t)
{
    __shared__ CcTileLds<false> L;
    const int lane = threadIdx.x;
    CcTileConst C;
    load_cc_tile_const<false>(C, L, t, lane);
    const long long ntile = (nframes + kCcT - 1) / kCcT;
    for (long long tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        const long long f0 = tile * kCcT;
        const int nv = (int)((nframes - f0 < kCcT) ? nframes - f0 : kCcT);
        const float *src = data201 + f0 * 201;
        constexpr int kBatch = 13, kIter = (kCcT * 201 + kLanes - 1) / kLanes;
        for (int b0 = 0; b0 < kIter; b0 += kBatch) { /* requests in batches before their stores (see compceps_kernel) */
            float sv[kBatch];
#pragma unroll
            for (int k = 0; k < kBatch; ++k) {
                const int i = lane + kLanes * (b0 + k);
                sv[k] = (i < nv * 201) ? src[i] : 0.0f;
            }
#pragma unroll
            for (int k = 0; k < kBatch; ++k) {
                const int i = lane + kLanes * (b0 + k);
                if (i < nv * 201) L.span[i] = sv[k];
            }
        }
        wave_sync();
        cc_tile<false>(L, C, nv, coef14 + f0 * SEA_CC_NCEP, lane);
    }
}

#ifndef SEA_CC_MINW
#define SEA_CC_MINW 3 /* 168 VGPRs, no spilled vector register; left to itself the allocator takes 193 = two waves per SIMD */
#endif
__global__ __launch_bounds__(64, SEA_CC_MINW) void compceps_kernel(CepsArgs a)
{
    __shared__ CcTileLds<true> L;
    const int lane = threadIdx.x;
#ifdef SEA_CC_TIMING
    if (lane == 0 && blockIdx.x < 16384) {
        g_cc_wave[4 * blockIdx.x] = (unsigned)wall_clock64();
        g_cc_wave[4 * blockIdx.x + 2] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
        g_cc_wave[4 * blockIdx.x + 3] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);
    }
#endif
    CcTileConst C;
    load_cc_tile_const<true>(C, L, a.tables, lane);
    /* tile slots: utterance u owns slots [ceps_cum[u] / T + u, ceps_cum[u+1] / T + u + 1), at least
     * ceil(capacity / T) of them; slot k of an utterance covers its cepstral frames kT .. kT + T - 1 */
    const long long nslot = a.ceps_cum[a.n_utt] / kCcT + a.n_utt;
    for (long long s = blockIdx.x; s < nslot; s += gridDim.x) {
        int lo = 0, hi = a.n_utt; /* largest u with base(u) <= s */
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (a.ceps_cum[mid] / kCcT + mid <= s) lo = mid; else hi = mid;
        }
        const int u = lo;
        const long long c0 = a.ceps_cum[u], cap = a.ceps_cum[u + 1] - c0;
        const long long j0 = (s - (c0 / kCcT + u)) * kCcT;
        if (j0 >= cap) continue; /* spare slot */
        const int f0 = a.first_out[u];
        const long long nfr = a.lengths[u] / SEA_HOP;
        const long long nout = (f0 >= 0) ? nfr - f0 : 0;
        const long long nceps = (nout >= 3) ? nout - 2 : 0;
        if (j0 == 0 && lane == 0 && a.n_ceps) a.n_ceps[u] = (int)nceps;
        const int nrow = (int)((cap - j0 < kCcT) ? cap - j0 : kCcT);
        long long left = nceps - j0;
        const int nv = (int)(left < 0 ? 0 : (left > nrow ? nrow : left));
        float *dst = a.ceps + (c0 + j0) * SEA_CC_NCEP;
        if (nv > 0) {
#ifdef SEA_CC_TIMING
            constexpr bool SHARED = true; /* (CC_CK's switch) */
#endif
            CC_CK_START;
            /* span word x = Data[x-1] of frame j0: the float NoiseSup stream from sample 80 (f0 + j0) - 1 on;
             * Data[-1] of the utterance's very first cepstral frame is the zero before the first output */
            const float *cur0 = a.den_f32 + a.offsets[u] + (f0 + j0) * SEA_HOP;
            const int nword = SEA_HOP * (nv - 1) + SEA_WIN + 1;
            /* all of the tile's words are requested before the first is stored: written as a load-store loop the
             * compiler waits for each of the 22 requests in turn -- ~22 HBM latencies per tile, most of the kernel's time */
            constexpr int kReq = (SEA_HOP * (kCcT - 1) + SEA_WIN + 1 + kLanes - 1) / kLanes; /* 22 */
#ifndef SEA_CC_STAGE_BATCH
#define SEA_CC_STAGE_BATCH 11 /* round 3: 22 at once two waves per SIMD 0.77 ms, 8: three waves 0.71; round 4 (three waves per SIMD forced, 168 VGPRs
                                * either way): 8 0.610 ms, 11 = two equal batches 0.58-0.60, 12 0.60, 22 (32 spilled registers) not run */
#endif
#pragma unroll 1
            for (int b0 = 0; b0 < kReq; b0 += SEA_CC_STAGE_BATCH) {
                float sv[SEA_CC_STAGE_BATCH];
#pragma unroll
                for (int k = 0; k < SEA_CC_STAGE_BATCH; ++k) {
                    const int x = lane + kLanes * (b0 + k);
                    sv[k] = (x < nword && !(x == 0 && j0 == 0)) ? cur0[x - 1] : 0.0f;
                }
#pragma unroll
                for (int k = 0; k < SEA_CC_STAGE_BATCH; ++k) {
                    const int x = lane + kLanes * (b0 + k);
                    if (x < nword) L.span[x + x / SEA_HOP] = sv[k];
                }
            }
            wave_sync();
            CC_CK(0);
#ifdef SEA_CC_TIMING
            if (blockIdx.x == 0 && lane == 0) g_cc_ck[7] += 1;
#endif
            cc_tile<true>(L, C, nv, dst, lane);
        }
        for (int idx = nv * SEA_CC_NCEP + lane; idx < nrow * SEA_CC_NCEP; idx += kLanes) dst[idx] = 0.0f;
    }
#ifdef SEA_CC_TIMING
    if (lane == 0 && blockIdx.x < 16384) g_cc_wave[4 * blockIdx.x + 1] = (unsigned)wall_clock64();
#endif
}

/* ==================================================================================================
 * the chain after NoiseSup that the reference has commented out
 * (etsi/cpp/ParmInterface.c:274-311): WaveProc -> CompCeps -> PostProc -> VAD, FlushAdvProcess.
 * WaveProc and CompCeps depend on the frame only -> one wave per cepstral frame (afe_ceps_kernel);
 * PostProc (an LMS recurrence over frames) and the frame-dropping VAD (a 7-frame ring and two
 * hangover counters) are serial per utterance and tiny -> one wave per utterance (afe_vad_kernel).
 * ================================================================================================ */
namespace {

#ifndef SEA_WP_ROWS
#define SEA_WP_ROWS 1 /* 1: the peak searches of four frames side by side, one per row of 16 lanes (feature pass 4.19 -> 3.82 ms);
                         0: one frame at a time, wave-wide.  Dealing the Teager / smoothing / window steps of the four frames
                         to the lanes as 800 samples as well (13 rounds, one sync per group) measured SLOWER, 4.39 ms: per-lane
                         frame index, divisions by 200 and a divergent loop over each frame's own peak list */
#endif

struct __attribute__((aligned(16))) WpLds { /* scratch of DoWaveProc: four frames in flight */
    float tw[200];
    int q[200];
    int sm[4][200];
    int pos[4][24];
    int nom[4];
};

/* wave-wide maximum of a signed 32-bit value in 6 DPP steps (row_shr 1/2/4/8 within rows of 16
 * lanes, then row_bcast 15 / 31 across rows: the total lands in lane 63); a ds_bpermute butterfly
 * costs an LDS round trip per step instead */
__device__ __forceinline__ int wave_max_i32(int v)
{
    constexpr int kMin = -2147483647 - 1;
    auto mx = [](int a, int b) { return a > b ? a : b; };
    v = mx(v, __builtin_amdgcn_update_dpp(kMin, v, 0x111, 0xf, 0xf, false)); /* row_shr:1 */
    v = mx(v, __builtin_amdgcn_update_dpp(kMin, v, 0x112, 0xf, 0xf, false)); /* row_shr:2 */
    v = mx(v, __builtin_amdgcn_update_dpp(kMin, v, 0x114, 0xf, 0xf, false)); /* row_shr:4 */
    v = mx(v, __builtin_amdgcn_update_dpp(kMin, v, 0x118, 0xf, 0xf, false)); /* row_shr:8 */
    v = mx(v, __builtin_amdgcn_update_dpp(kMin, v, 0x142, 0xa, 0xf, false)); /* row_bcast:15 -> rows 1, 3 */
    v = mx(v, __builtin_amdgcn_update_dpp(kMin, v, 0x143, 0xc, 0xf, false)); /* row_bcast:31 -> rows 2, 3 */
    return __builtin_amdgcn_readlane(v, 63);
}

/* wave-wide arg-max of (value >= 0, index < 256) pairs; ties go to the LOWER index if lowWins, else
 * the higher.  Entries with valid == false never win.  Returns the winning index, -1 if none. */
__device__ __forceinline__ int wave_argmax(int value, int index, bool valid, bool lowWins)
{
    const int m = wave_max_i32(valid ? value : -1);
    if (m < 0) return -1;
    const int code = (valid && value == m) ? (lowWins ? 255 - index : index) : -1;
    const int c = wave_max_i32(code);
    return lowWins ? 255 - c : c;
}

/* maximum over each ROW of 16 lanes, left in every lane of the row: an xor butterfly in four DPP steps
 * (quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror) */
__device__ __forceinline__ int row_max_i32(int v)
{
    auto mx = [](int a, int b) { return a > b ? a : b; };
    v = mx(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false));
    v = mx(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false));
    v = mx(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xf, 0xf, false));
    v = mx(v, __builtin_amdgcn_update_dpp(v, v, 0x140, 0xf, 0xf, false));
    return v;
}

/* wave_argmax per row of 16 lanes (every lane of a row gets its row's answer) */
__device__ __forceinline__ int row_argmax(int value, int index, bool valid, bool lowWins)
{
    const int m = row_max_i32(valid ? value : -1);
    const int code = (valid && value == m) ? (lowWins ? 255 - index : index) : -1;
    const int c = row_max_i32(code);
    return (m < 0) ? -1 : (lowWins ? 255 - c : c);
}

/* DoWaveProc (WaveProc.c:397-455) on a frame d[0..199] whose low-energy check (:423-427: in-order sum of squares >= 100,
 * evaluated by the caller lane = frame) has passed, in three steps:
 *   wp_smooth   Teager energy (:216-226) and its 9-point integer smoothing                      -> W.sm[slot]
 *   wp_peaks    maxima 25..79 samples apart (:102-190)                                          -> W.pos[slot], W.nom[slot]
 *   wp_window   a two-level window around them (:244-330), applied in place
 * Each ends with wave_sync(). */
__device__ __forceinline__ void wp_smooth(WpLds &W, int slot, const float *d, int lane)
{
    constexpr int N = 200;
    /* Teager energy and its integer quarter, (int)floor(T * 0.25 + 0.5) in double */
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = lane + 64 * k;
        if (i < N) {
            const float a = d[i], l = d[i > 0 ? i - 1 : 0], r = d[i < N - 1 ? i + 1 : N - 1];
            /* ends: |d0*d0 - d0*d1| and |dN-1*dN-1 - dN-2*dN-1| (the missing neighbour is the sample itself) */
            const float t = (i == 0) ? fabsf(a * a - a * r) : ((i == N - 1) ? fabsf(a * a - l * a) : fabsf(a * a - l * r));
            W.q[i] = (int)floor((double)t * 0.25 + 0.5);
        }
    }
    wave_sync();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = lane + 64 * k;
        if (i < N) {
            unsigned acc = 0;
#pragma unroll
            for (int j = -4; j <= 4; ++j) {
                int idx = i + j;
                idx = idx < 0 ? 0 : (idx > N - 1 ? N - 1 : idx);
                acc += (unsigned)W.q[idx];
            }
            W.sm[slot][i] = (int)acc;
        }
    }
    wave_sync();
}

/* one frame, wave-wide */
__device__ __forceinline__ void wp_peaks(WpLds &W, int slot, int lane)
{
    constexpr int N = 200;
    const int *sm = W.sm[slot];
    /* global maximum: first index of the largest value, which must exceed 0 */
    int nom = 0;
    {
        int bv = 0, bi = -1;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = lane + 64 * k;
            if (i < N && sm[i] > bv) { /* ascending i per lane: strict > keeps the first */
                bv = sm[i];
                bi = i;
            }
        }
        const int p0 = wave_argmax(bv, bi, bi >= 0, true);
        if (p0 >= 0) {
            int R[10], Lf[10], cR = 0, cL = 0;
            R[0] = Lf[0] = p0;
            bool found = true;
#pragma unroll 1
            while (R[cR] + 25 < N && found) { /* to the right: last of equals = the higher index */
                const int idx = R[cR] + 25 + lane;
                const bool in = lane < 55 && idx < N;
                const int v = in ? sm[idx] : -1;
                const int nx = wave_argmax(v, idx, in && v >= 0, false);
                found = nx >= 0;
                if (found) {
#pragma unroll
                    for (int c = 0; c < 9; ++c)
                        if (c == cR) R[c + 1] = nx;
                    cR++;
                }
            }
            found = true;
#pragma unroll 1
            while (Lf[cL] - 25 > 0 && found) { /* to the left: last of equals in scan order = the lower index */
                const int idx = Lf[cL] - 25 - lane;
                const bool in = lane < 55 && idx > -1;
                const int v = in ? sm[idx] : -1;
                const int nx = wave_argmax(v, idx, in && v >= 0, true);
                found = nx >= 0;
                if (found) {
#pragma unroll
                    for (int c = 0; c < 9; ++c)
                        if (c == cL) Lf[c + 1] = nx;
                    cL++;
                }
            }
            /* ascending: left ones (farthest first), centre, right ones */
            if (lane == 0) {
#pragma unroll
                for (int c = 9; c >= 1; --c)
                    if (c <= cL) W.pos[slot][nom++] = Lf[c];
#pragma unroll
                for (int c = 0; c < 10; ++c)
                    if (c <= cR) W.pos[slot][nom++] = R[c];
            }
            nom = cL + cR + 1;
        }
    }
    if (lane == 0) W.nom[slot] = nom;
    wave_sync();
}

/* the same search for up to four frames at once, frame `slot` = row `slot` of 16 lanes (mask: bit slot = that frame takes
 * part): the searches are short dependent chains of wave-wide reductions, so four of them side by side cost what one does */
__device__ __forceinline__ void wp_peaks4(WpLds &W, unsigned mask, int lane)
{
    constexpr int N = 200;
    const int row = lane >> 4, l = lane & 15;
    const int *sm = W.sm[row];
    const bool on = (mask >> row) & 1u;
    int bv = 0, bi = -1;
#pragma unroll
    for (int k = 0; k < 13; ++k) {
        const int i = l + 16 * k;
        if (i < N) {
            const int v = sm[i];
            if (v > bv) { /* ascending i per lane: strict > keeps the first */
                bv = v;
                bi = i;
            }
        }
    }
    const int p0 = row_argmax(bv, bi, on && bi >= 0, true);
    int nom = 0;
    int R[10], Lf[10], cR = 0, cL = 0;
    R[0] = Lf[0] = p0;
    int cur = p0;
    bool go = p0 >= 0 && cur + 25 < N;
#pragma unroll 1
    while (__ballot(go) != 0ull) { /* to the right: last of equals = the higher index */
        int v = -1, vi = -1;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int off = l + 16 * k, idx = cur + 25 + off;
            if (go && off < 55 && idx < N) {
                const int x = sm[idx];
                if (x >= v) { /* ascending idx per lane: >= keeps the last */
                    v = x;
                    vi = idx;
                }
            }
        }
        const int nx = row_argmax(v, vi, go && v >= 0, false);
        if (go) {
            if (nx >= 0) {
#pragma unroll
                for (int c = 0; c < 9; ++c)
                    if (c == cR) R[c + 1] = nx;
                cR++;
                cur = nx;
                go = cur + 25 < N;
            } else
                go = false;
        }
    }
    cur = p0;
    go = p0 >= 0 && cur - 25 > 0;
#pragma unroll 1
    while (__ballot(go) != 0ull) { /* to the left: last of equals in scan order = the lower index */
        int v = -1, vi = -1;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int off = l + 16 * k, idx = cur - 25 - off;
            if (go && off < 55 && idx > -1) {
                const int x = sm[idx];
                if (x >= v) { /* descending idx per lane: >= keeps the lowest */
                    v = x;
                    vi = idx;
                }
            }
        }
        const int nx = row_argmax(v, vi, go && v >= 0, true);
        if (go) {
            if (nx >= 0) {
#pragma unroll
                for (int c = 0; c < 9; ++c)
                    if (c == cL) Lf[c + 1] = nx;
                cL++;
                cur = nx;
                go = cur - 25 > 0;
            } else
                go = false;
        }
    }
    if (p0 >= 0) {
        if (l == 0) { /* ascending: left ones (farthest first), centre, right ones */
#pragma unroll
            for (int c = 9; c >= 1; --c)
                if (c <= cL) W.pos[row][nom++] = Lf[c];
#pragma unroll
            for (int c = 0; c < 10; ++c)
                if (c <= cR) W.pos[row][nom++] = R[c];
        }
        nom = cL + cR + 1;
    }
    if (l == 0) W.nom[row] = nom;
    wave_sync();
}

__device__ __forceinline__ void wp_window(WpLds &W, int slot, float *d, int lane)
{
    constexpr int N = 200;
    constexpr int kMaxPeaks = 12; /* maxima are at least 25 samples apart: at most 8 in 200 samples */
    const int nom = W.nom[slot];
    const int *pos = W.pos[slot];
    const float eps = (float)0.2;
    const float lowVal = (float)((double)(1 - eps) / 2.0), highVal = (float)((double)(1 + eps) / 2.0);
    /* the peak list once into registers (one LDS round trip instead of one per peak and round); entries beyond the
     * list sit past every sample */
    int pk[kMaxPeaks];
#pragma unroll
    for (int c = 0; c < kMaxPeaks; ++c) pk[c] = (c < nom) ? pos[c] : (1 << 20);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int j = lane + 64 * k;
        if (j < N) {
            /* the raised segments [pos_i - 4, pos_i - 4 + ceil(0.8 gap_i)) are ordered and disjoint: the only
             * one that can hold j is the last one starting at or before j */
            bool high = false;
            if (nom > 1) {
                int cnt = 0;
#pragma unroll
                for (int c = 0; c < kMaxPeaks; ++c) cnt += (pk[c] - 4 <= j) ? 1 : 0;
                if (cnt > 0) {
                    const int i = cnt - 1;
                    const int gap = (i < nom - 1) ? (pos[i + 1] - pos[i]) : (pos[nom - 1] - pos[nom - 2]);
                    high = j < pos[i] - 4 + (80 * gap + 99) / 100;
                }
            }
            W.tw[j] = high ? highVal : lowVal;
        }
    }
    wave_sync();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = lane + 64 * k;
        if (i < N) d[i] *= (W.tw[i] + W.tw[i < N - 1 ? i + 1 : N - 1]);
    }
    wave_sync();
}

} // namespace

/* WaveProc + CompCeps of the restored feature chain, tiled like compceps_kernel: a wave owns 16 consecutive
 * cepstral frames of one utterance as SEPARATE 201-float frames in LDS (WaveProc reshapes each frame in place, so
 * they cannot share samples).  The low-energy check's in-order sum of squares runs lane = frame; the frames that
 * pass go through DoWaveProc one after the other (wave-wide peak search), then the tile through cc_tile(). */
/* frames per tile of afe_ceps_kernel: 8 (same-box A/B of the feature pass: 16 frames 3.71 ms, 8 frames 3.05 ms -- half the
 * LDS per wave, 15.6 instead of 25 KB, lets the CU hold the eight waves its registers allow instead of six; compceps_kernel
 * itself is fastest with 16: 0.715 ms against 0.77-0.79 with 8 and 1.42 with 4) */
#ifndef SEA_AFE_TILE
#define SEA_AFE_TILE 8
#endif
constexpr int kAfeT = SEA_AFE_TILE;

/* timing-only diagnostic (-DSEA_AFE_TIMING, tools/afe_phases.py): shader clocks workgroup 0 spends per step of a tile */
#ifdef SEA_AFE_TIMING
__device__ unsigned long long g_afe_ck[8];
extern "C" int sea_afe_timing(unsigned long long *out8, int reset)
{
    if (reset) {
        unsigned long long z[8] = {};
        return hipMemcpyToSymbol(HIP_SYMBOL(g_afe_ck), z, sizeof z) != hipSuccess;
    }
    return hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_afe_ck), 8 * sizeof(unsigned long long)) != hipSuccess;
}
#define AFE_CK_START unsigned long long ck_ = clock64()
#define AFE_CK(k) do { const unsigned long long c_ = clock64(); if (blockIdx.x == 0 && threadIdx.x == 0) g_afe_ck[k] += c_ - ck_; ck_ = c_; } while (0)
#else
#define AFE_CK_START
#define AFE_CK(k)
#endif

#ifndef SEA_AFE_MINW
#define SEA_AFE_MINW 1
#endif
__global__ __launch_bounds__(64, SEA_AFE_MINW) void afe_ceps_kernel(AfeArgs a)
{
    __shared__ CcTileLds<false, kAfeT> L;
    __shared__ WpLds W;
    const int lane = threadIdx.x;
    CcTileConst C;
    load_cc_tile_const<false, kAfeT>(C, L, a.tables, lane);
    const long long nslot = a.ceps_cum[a.n_utt] / kAfeT + a.n_utt; /* tile slots as in compceps_kernel */
    for (long long s = blockIdx.x; s < nslot; s += gridDim.x) {
        int lo = 0, hi = a.n_utt;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (a.ceps_cum[mid] / kAfeT + mid <= s) lo = mid; else hi = mid;
        }
        const int u = lo;
        const long long c0 = a.ceps_cum[u], cap = a.ceps_cum[u + 1] - c0;
        const long long j0 = (s - (c0 / kAfeT + u)) * kAfeT;
        if (j0 >= cap) continue;
        const int f0 = a.first_out[u];
        const long long nfr = a.lengths[u] / SEA_HOP;
        const long long nout = (f0 >= 0) ? nfr - f0 : 0;
        const long long nceps = (nout >= 3) ? nout - 2 : 0;
        if (j0 == 0 && lane == 0 && a.n_ceps) a.n_ceps[u] = (int)nceps;
        const int nrow = (int)((cap - j0 < kAfeT) ? cap - j0 : kAfeT);
        long long left = nceps - j0;
        const int nv = (int)(left < 0 ? 0 : (left > nrow ? nrow : left));
        float *dst = a.feat_cc + (c0 + j0) * SEA_CC_NCEP;
        AFE_CK_START;
        if (nv > 0) {
            /* frameBuf of ParmInterface.c:281 for cepstral frame j: Data[-1..199] = the float NoiseSup stream from
             * sample 80 (f0 + j) - 1 on; Data[-1] of the utterance's first cepstral frame is 0 */
            const float *cur0 = a.den_f32 + a.offsets[u] + (f0 + j0) * SEA_HOP;
#ifndef SEA_AFE_BATCH
#define SEA_AFE_BATCH 17 /* 1: 4.03 ms for the feature pass, 6: 3.78, 13-17: 3.66 (outer loop kept rolled) */
#endif
            /* requests in batches before their stores, the outer loop kept rolled (fully unrolled the allocator went to
             * 256 VGPRs + 95 AGPRs, one wave per SIMD: 3.96 -> 5.9 ms) */
            constexpr int kIter = (kAfeT * 201 + kLanes - 1) / kLanes; /* 51 */
#pragma unroll 1
            for (int b0 = 0; b0 < kIter; b0 += SEA_AFE_BATCH) {
                float sv[SEA_AFE_BATCH];
#pragma unroll
                for (int k = 0; k < SEA_AFE_BATCH; ++k) {
                    const int i = lane + kLanes * (b0 + k);
                    const int f = i / 201, x = i - f * 201;
                    sv[k] = (i < nv * 201 && !(x == 0 && f == 0 && j0 == 0)) ? cur0[SEA_HOP * f + x - 1] : 0.0f;
                }
#pragma unroll
                for (int k = 0; k < SEA_AFE_BATCH; ++k) {
                    const int i = lane + kLanes * (b0 + k);
                    if (i < nv * 201) L.span[i] = sv[k];
                }
            }
            wave_sync();
            AFE_CK(0);
            float energy = 0.0f; /* WaveProc.c:423-427, lane = frame */
            if (lane < nv) {
                const float *p = L.span + 201 * lane;
#pragma unroll 8
                for (int x = 1; x < 201; ++x) {
                    const float v = p[x];
                    energy += v * v;
                }
            }
            const unsigned long long pass = __ballot(lane < nv && (double)energy >= 100.0);
            AFE_CK(1);
#if SEA_WP_ROWS
            for (int g = 0; g < nv; g += 4) { /* four frames at a time: their peak searches run side by side */
                const unsigned m4 = (unsigned)(pass >> g) & 0xfu;
                if (m4 == 0) continue;
                for (int r = 0; r < 4; ++r)
                    if ((m4 >> r) & 1u) wp_smooth(W, r, L.span + 201 * (g + r) + 1, lane);
                AFE_CK(2);
                wp_peaks4(W, m4, lane);
                AFE_CK(3);
                for (int r = 0; r < 4; ++r)
                    if ((m4 >> r) & 1u) wp_window(W, r, L.span + 201 * (g + r) + 1, lane);
                AFE_CK(4);
            }
#else
            for (int f = 0; f < nv; ++f)
                if ((pass >> f) & 1ull) {
                    wp_smooth(W, 0, L.span + 201 * f + 1, lane);
                    wp_peaks(W, 0, lane);
                    wp_window(W, 0, L.span + 201 * f + 1, lane);
                }
#endif
            wave_sync();
            cc_tile<false, kAfeT>(L, C, nv, dst, lane);
            AFE_CK(5);
#ifdef SEA_AFE_TIMING
            if (blockIdx.x == 0 && threadIdx.x == 0) g_afe_ck[7] += 1;
#endif
        }
        for (int idx = nv * SEA_CC_NCEP + lane; idx < nrow * SEA_CC_NCEP; idx += kLanes) dst[idx] = 0.0f;
    }
}

/* DoPostProc (PostProc.c:123-149), DoVADProc (VAD.c:219-317), DoVADFlush (:342-433) and the null
 * feature frames of the all-zero lead (ParmInterface.c:314-329), in emission order.  lane = feature
 * index (0..13 cepstra/energies, 14 the VAD flag). */
__global__ __launch_bounds__(64) void afe_vad_kernel(AfeArgs a)
{
    __shared__ float ring[7][16];
    const int lane = threadIdx.x;
    const int u = blockIdx.x;
    const int f0 = a.first_out[u];
    const long long nfr = a.lengths[u] / SEA_HOP;
    const long long nout = (f0 >= 0) ? nfr - f0 : 0;
    const long long nceps = (nout >= 3) ? nout - 2 : 0;
    long long nnull = a.onset[u];
    nnull = nnull < nfr ? nnull : nfr;
    float *out = a.feat15 + a.feat_cum[u] * 15;
    const float *cc = a.feat_cc + a.ceps_cum[u] * SEA_CC_NCEP;
    float *pp = a.feat_pp ? a.feat_pp + a.ceps_cum[u] * SEA_CC_NCEP : nullptr;
    const unsigned char *flg = a.flags + a.offsets[u] / 8;
    long long nemit = 0;

    for (long long k = 0; k < nnull; ++k) { /* null MFCC vectors, VAD = NON_SPEECH */
        if (lane < 15) out[nemit * 15 + lane] = 0.0f;
        nemit++;
    }
    if (lane < 16)
#pragma unroll
        for (int r = 0; r < 7; ++r) ring[r][lane] = 0.0f;
    wave_sync();

    static const float target[12] = {(float)-6.618909, (float)0.198269, (float)-0.740308, (float)0.055132,
                                     (float)-0.227086, (float)0.144280, (float)-0.112451, (float)-0.146940,
                                     (float)-0.327466, (float)0.134571, (float)0.027884,  (float)-0.114905};
    const float tgt = (lane < 12) ? target[lane] : 0.0f;
    const float lambda = (float)0.0087890625;
    float wLMS = 0.0f;   /* weightLMS[lane] */
    float feat = 0.0f;   /* FeatureBuffer[lane]: persists between calls like the reference's buffer */
    int focus = 0, hangOver = 23, hCount = 0, vCount = 0, frameCounter = 0;

    /* trigger = longest run of speech-flagged frames in the ring, scanned from focus+1 round to focus */
    auto decide = [&](int fc) {
        int sum = 0, trigger = 0;
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            int r = focus + i + 1;
            r = r > 6 ? r - 7 : r;
            if (ring[r][14] != 0.0f)
                sum++;
            else {
                trigger = sum > trigger ? sum : trigger;
                sum = 0;
            }
        }
        trigger = sum > trigger ? sum : trigger;
        if (trigger >= 4) {
            hCount = hangOver;
            if (fc <= 35) hangOver = 50;
        }
        if (hCount && trigger < 3) hCount--;
        if (trigger >= 3) vCount = 5;
        if (vCount && trigger < 3) vCount--;
        int r = focus + 1;
        r = r > 6 ? r - 7 : r;
        feat = (lane < 15) ? ring[r][lane] : 0.0f;
        if (lane == 14) feat = (vCount || hCount || trigger >= 3) ? 1.0f : 0.0f;
    };

    /* rows and flag bytes are requested kAhead frames before use: the loop body is a few dozen
     * instructions, an HBM round trip a few thousand clocks */
    constexpr int kAhead = 8;
    float rowQ[kAhead];
    int bitQ[kAhead];
    auto fetch = [&](long long j, float &row, int &bits) {
        const long long jj = j < nceps ? j : (nceps > 0 ? nceps - 1 : 0);
        row = (lane < 14 && nceps > 0) ? cc[jj * SEA_CC_NCEP + lane] : 0.0f;
        bits = (nceps > 0) ? (int)flg[10 * (f0 + jj + 2)] : 0;
    };
#pragma unroll
    for (int q = 0; q < kAhead; ++q) fetch(q, rowQ[q], bitQ[q]);
    for (long long j0 = 0; j0 < nceps; j0 += kAhead) {
#pragma unroll
      for (int q = 0; q < kAhead; ++q) {
        const long long j = j0 + q;
        if (j >= nceps) break;
        /* PostProc on c1..c12; the weighting comes from logE = Coef[13] (Noc0 == 0) */
        const float c = rowQ[q];
        const int bits = bitQ[q];
        fetch(j + kAhead, rowQ[q], bitQ[q]);
        const float logE = __shfl(c, 13, 64);
        float wp = (logE * (float)64 - (float)211) / (float)64;
        wp = (wp < 0) ? 0.0f : ((wp > 1) ? lambda : wp * lambda);
        float v = c;
        if (lane < 12) {
            const float dif = ((c - wLMS) - tgt);
            v = c - wLMS;
            wLMS += dif * wp;
        }
        if (pp && lane < 14) pp[j * SEA_CC_NCEP + lane] = v;
        if (lane < 14) feat = v;
        /* DoVADProc */
        frameCounter = (int)j + 5; /* nbFrame[0] when NoiseSup output j+3 appears */
        focus = (focus + 1 == 7) ? 0 : focus + 1;
        if (lane < 14) ring[focus][lane] = feat;
        if (lane == 14) ring[focus][14] = bits ? 1.0f : 0.0f;
        wave_sync();
        if (frameCounter > 10) {
            decide(frameCounter);
            if (lane < 15) out[nemit * 15 + lane] = feat;
            nemit++;
        }
        wave_sync();
      }
    }
    /* FlushAdvProcess until DoVADFlush returns FALSE */
    {
        const int flushFocus = focus;
        for (;;) {
            int nf = focus + 1;
            nf = (nf == 7) ? 0 : nf;
            if (nf == flushFocus) break;
            focus = nf;
            frameCounter++;
            if (frameCounter > 10) decide(frameCounter);
            if (lane < 15) out[nemit * 15 + lane] = feat;
            nemit++;
        }
    }
    if (lane == 0) a.n_feat[u] = (int)nemit;
}

} // namespace sea

