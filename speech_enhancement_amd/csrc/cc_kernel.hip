/*
 * cc_kernel.hip -- batched rfft and the CompCeps cepstral front-end, gfx950 (MI355X).
 *
 * Both are recursion-free per frame, so the launch is one 64-lane wavefront per frame.
 *   rfft256_kernel           etsi/cpp/rfft.c:45-180 on [nframes][256] floats, in place or not
 *   compceps_frames_kernel   DoCompCeps (etsi/cpp/CompCeps.c:309-318 -> WI8CompCeps :368-549) on
 *                            caller-supplied frames of 201 floats (Data[-1..199])
 *   compceps_kernel          the same, reading frames straight out of the float NoiseSup stream:
 *                            frame j of an utterance = denoised[80j-1 .. 80j+199], available once 3
 *                            NoiseSup outputs exist (the commented-out driver block
 *                            etsi/cpp/ParmInterface.c:275-293)
 */
#include "sea_device.h"
#include "sea_kernels.h"

namespace sea {

__global__ __launch_bounds__(64) void rfft256_kernel(const float *in, float *out, long long nframes,
                                                     const sea_fft_tables *t)
{
    __shared__ __attribute__((aligned(16))) float work[256];
    const int lane = threadIdx.x;
    FftRegs R;
    load_fft_regs(R, t, lane);
    for (long long f = blockIdx.x; f < nframes; f += gridDim.x) {
        const float *x = in + f * 256;
        rfft256(x[lane], x[lane + 64], x[lane + 128], x[lane + 192], work, R, lane);
        const float4 v = *reinterpret_cast<const float4 *>(work + 4 * lane);
        *reinterpret_cast<float4 *>(out + f * 256 + 4 * lane) = v;
        wave_sync();
    }
}

namespace {

struct __attribute__((aligned(16))) CcLds {
    float work[256];
    float sq[200];
    float pw[132];   /* 129 power bins */
    float fb[24];    /* 23 log mel energies */
};

struct CcConst {
    FftRegs fft;
    float win[4];
    int melStart, melLen;
    float melW[SEA_CC_TAPS];
    float dct[SEA_CC_NCHAN];
    float floorFB, floorE;
};

__device__ __forceinline__ void load_cc_const(CcConst &C, const sea_cc_tables *t, int lane)
{
    load_fft_regs(C.fft, &t->fft, lane);
#pragma unroll
    for (int k = 0; k < 4; ++k) C.win[k] = t->win[k][lane];
    C.melStart = t->melStart[lane];
    C.melLen = t->melLen[lane];
#pragma unroll
    for (int i = 0; i < SEA_CC_TAPS; ++i) C.melW[i] = t->melW[i][lane];
#pragma unroll
    for (int j = 0; j < SEA_CC_NCHAN; ++j) C.dct[j] = t->dct[j][lane];
    C.floorFB = t->floorFB;
    C.floorE = t->floorE;
}

/* cur points at Data[0]; Data[-1] is passed separately (prev) so the caller can substitute the
 * zero that precedes the very first denoised sample.  Writes 14 floats. */
__device__ __forceinline__ void compceps_frame(const float *cur, float prev, float *coef, CcLds &L,
                                               const CcConst &C, int lane)
{
    /* gather Data[i] and Data[i-1] for i = lane + 64k */
    float d[4], dm1[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = lane + 64 * k;
        const bool in = i < SEA_WIN;
        d[k] = in ? cur[i] : 0.0f;
        dm1[k] = in ? ((i == 0) ? prev : cur[i - 1]) : 0.0f;
    }
    /* logE terms (CompCeps.c:413-423): sum of squares in sample order */
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = lane + 64 * k;
        if (i < SEA_WIN) L.sq[i] = d[k] * d[k];
    }
    /* pre-emphasis in double (:427-429), symmetric Hamming (:115-125), zero padding (:439-440) */
    float e[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = lane + 64 * k;
        const float pre = (float)((double)d[k] - 0.90 * (double)dm1[k]);
        e[k] = (i < SEA_WIN) ? pre * C.win[k] : 0.0f;
    }
    rfft256(e[0], e[1], e[2], e[3], L.work, C.fft, lane);

    float logE = serial_sum<200>(L.sq, 0.0f);
    logE = (logE < C.floorE) ? (float)-50.0 : (float)log((double)logE);

    /* power spectrum, products and sum in double (:451-459) */
    {
        const int i0 = lane, i1 = lane + 64;
        const double r0 = (double)L.work[i0], r1 = (double)L.work[i1];
        const double m0 = (lane > 0) ? (double)L.work[256 - i0] : 0.0, m1 = (double)L.work[256 - i1];
        L.pw[i0] = (lane > 0) ? (float)(r0 * r0 + m0 * m0) : (float)(r0 * r0);
        L.pw[i1] = (float)(r1 * r1 + m1 * m1);
        if (lane == 0) {
            const double ny = (double)L.work[128];
            L.pw[128] = (float)(ny * ny);
        }
    }
    wave_sync();
    /* 23 mel triangles (DoMelFB, MelProc.c:82-104), natural log with floor (:509-513) */
    if (lane < SEA_CC_NCHAN) {
        float acc = 0.0f;
#pragma unroll
        for (int i = 0; i < SEA_CC_TAPS; ++i) {
            const int idx = C.melStart + i;
            const float t = acc + L.pw[idx < 129 ? idx : 128] * C.melW[i];
            acc = (i < C.melLen) ? t : acc;
        }
        L.fb[lane] = (acc < C.floorFB) ? (float)-10.0 : (float)log((double)acc);
    }
    wave_sync();
    /* DCT (:203-227): lanes 0..11 -> c1..c12, lane 12 -> c0, lane 13 -> logE */
    if (lane < 14) {
        float acc = 0.0f;
        if (lane < 12) {
#pragma unroll
            for (int j = 0; j < SEA_CC_NCHAN; ++j) acc += L.fb[j] * C.dct[j];
        } else if (lane == 12) {
#pragma unroll
            for (int j = 0; j < SEA_CC_NCHAN; ++j) acc += L.fb[j];
        } else
            acc = logE;
        coef[lane] = acc;
    }
    wave_sync();
}

} // namespace

__global__ __launch_bounds__(64) void compceps_frames_kernel(const float *data201, float *coef14,
                                                             long long nframes, const sea_cc_tables *t)
{
    __shared__ CcLds L;
    const int lane = threadIdx.x;
    CcConst C;
    load_cc_const(C, t, lane);
    for (long long f = blockIdx.x; f < nframes; f += gridDim.x) {
        const float *p = data201 + f * 201;
        compceps_frame(p + 1, p[0], coef14 + f * SEA_CC_NCEP, L, C, lane);
    }
}

__global__ __launch_bounds__(64) void compceps_kernel(CepsArgs a)
{
    __shared__ CcLds L;
    const int lane = threadIdx.x;
    CcConst C;
    load_cc_const(C, a.tables, lane);
    const long long total = a.ceps_cum[a.n_utt];
    for (long long g = blockIdx.x; g < total; g += gridDim.x) {
        /* locate the utterance: largest u with ceps_cum[u] <= g */
        int lo = 0, hi = a.n_utt;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (a.ceps_cum[mid] <= g) lo = mid; else hi = mid;
        }
        const int u = lo;
        const long long j = g - a.ceps_cum[u];
        const int f0 = a.first_out[u];
        const long long nfr = a.lengths[u] / SEA_HOP;
        const long long nout = (f0 >= 0) ? nfr - f0 : 0;
        const long long nceps = (nout >= 3) ? nout - 2 : 0;
        if (j == 0 && lane == 0 && a.n_ceps) a.n_ceps[u] = (int)nceps;
        float *dst = a.ceps + g * SEA_CC_NCEP;
        if (j < nceps) {
            const float *cur = a.den_f32 + a.offsets[u] + (f0 + j) * SEA_HOP;
            const float prev = (j == 0) ? 0.0f : cur[-1];
            compceps_frame(cur, prev, dst, L, C, lane);
        } else if (lane < SEA_CC_NCEP) {
            dst[lane] = 0.0f;
        }
    }
}

} // namespace sea
