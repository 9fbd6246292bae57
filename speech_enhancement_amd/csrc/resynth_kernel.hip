/*
 * resynth_kernel.hip -- Hu-Wang 64-channel gammatone analysis/synthesis resynthesis, gfx950.
 *
 * One workgroup (three 64-lane wavefronts) owns one utterance; LANE = CHANNEL (64 channels = one
 * wave), so the 4th-order complex one-pole cascade of every channel advances one sample per step
 * in lock-step.  A wave that is alone on its SIMD issues one vector instruction every 4 cycles, so
 * the ~49 operations of one cascade step bound a one-wave recurrence at ~200 cycles per sample.  The
 * cascade is therefore cut in two feed-forward halves that run as a software pipeline, 16-sample
 * tiles through LDS, one s_barrier per tile:
 *
 *   wave R1  cascade stages 0-1 (state p0,q0,p1,q1)          -> (p1,q1) per sample
 *   wave R2  cascade stages 2-3 (state p2,q2,p3,q3; it re-derives x1,y1 from the previous
 *            (p1,q1), which it already holds)                -> filter output per sample
 *   wave H   everything per-sample but not recursive: divisions by the middle-ear gain, the
 *            overlap-add weights, channel sums, casts, HBM stores
 *
 *   resynth_fwd_kernel : analysis pass; H writes g1[n][c]/midEar[c] to HBM as rows of 64 floats
 *                        (256 B per step, fully coalesced).  288 GB of HBM is what makes keeping the
 *                        whole [L][64] intermediate of a 1024-utterance batch (~17 GB) resident
 *                        feasible.
 *   resynth_bwd_kernel : R1 reads the rows in reverse time order (8 in flight per lane); H divides
 *                        again, evaluates the mask-weighted raised-cosine overlap-add weight of
 *                        that sample on the fly (at most two overlapping frames per sample),
 *                        multiplies, and sums the 64 channels IN CHANNEL ORDER through a padded LDS
 *                        transpose (lane = sample), then truncates to int16.  No second
 *                        intermediate is written.
 *
 * Reference reproduced: resyth_64sub_ori/cpp/extractwav.cpp:55-121 (resynth body; hairCell is dead
 * code there), :167-211 (gammaToneFilter); resyth_64sub_IBM/cpp/extractwav.cpp:97-99
 * (binary mask).  Float arithmetic order is the reference's (-ffp-contract=off).
 */
#include "sea_device.h"
#include "sea_kernels.h"

namespace sea {

namespace {

struct GtState {
    float p0, p1, p2, p3, q0, q1, q2, q3;
};

/* one sample of gammaToneFilter (extractwav.cpp:188-210); returns output[n] = p[3]*gain taken
 * BEFORE the update */
__device__ __forceinline__ float gt_step(GtState &s, float in, float f1, float f2, float gain)
{
    const float out = s.p3 * gain;
    const float x0 = f1 * s.p0 - f2 * s.q0, y0 = f2 * s.p0 + f1 * s.q0;
    const float x1 = f1 * s.p1 - f2 * s.q1, y1 = f2 * s.p1 + f1 * s.q1;
    const float x2 = f1 * s.p2 - f2 * s.q2, y2 = f2 * s.p2 + f1 * s.q2;
    const float x3 = f1 * s.p3 - f2 * s.q3, y3 = f2 * s.p3 + f1 * s.q3;
    s.p0 = in * f1 + x0;
    s.q0 = in * f2 + y0;
    s.p1 = s.p0 + x1;
    s.q1 = s.q0 + y1;
    s.p2 = s.p1 + x1 + x2;
    s.q2 = s.q1 + y1 + y2;
    s.p3 = s.p2 + x1 + 2 * x2 + x3;
    s.q3 = s.q2 + y1 + 2 * y2 + y3;
    return out;
}

/* The same step cut in two.  Stages 0-1: consumes the input sample, returns the NEW (p1,q1). */
struct GtLo {
    float p0, q0, p1, q1;
};
__device__ __forceinline__ float2 gt_step_lo(GtLo &s, float in, float f1, float f2)
{
    const float x0 = f1 * s.p0 - f2 * s.q0, y0 = f2 * s.p0 + f1 * s.q0;
    const float x1 = f1 * s.p1 - f2 * s.q1, y1 = f2 * s.p1 + f1 * s.q1;
    s.p0 = in * f1 + x0;
    s.q0 = in * f2 + y0;
    s.p1 = s.p0 + x1;
    s.q1 = s.q0 + y1;
    return make_float2(s.p1, s.q1);
}
/* Stages 2-3: needs the new (p1,q1) and x1,y1 -- the rotation of the OLD (p1,q1), which this wave
 * kept from the previous sample, so the same two products and one add/sub reproduce them exactly. */
struct GtHi {
    float p1old, q1old, p2, q2, p3, q3;
};
__device__ __forceinline__ float gt_step_hi(GtHi &s, float2 pq1, float f1, float f2, float gain)
{
    const float out = s.p3 * gain;
    const float x1 = f1 * s.p1old - f2 * s.q1old, y1 = f2 * s.p1old + f1 * s.q1old;
    const float x2 = f1 * s.p2 - f2 * s.q2, y2 = f2 * s.p2 + f1 * s.q2;
    const float x3 = f1 * s.p3 - f2 * s.q3, y3 = f2 * s.p3 + f1 * s.q3;
    s.p2 = pq1.x + x1 + x2;
    s.q2 = pq1.y + y1 + y2;
    s.p3 = s.p2 + x1 + 2 * x2 + x3;
    s.q3 = s.q2 + y1 + 2 * y2 + y3;
    s.p1old = pq1.x;
    s.q1old = pq1.y;
    return out;
}

constexpr int kTile = 16;        /* time steps per hand-over between the pipelined waves */
constexpr int kTileStride = 65;  /* 64 channels + 1 pad: conflict-free row reads by lane = sample */

/* workgroup barrier for the role-specialised waves (same count, different program counters);
 * LDS-only fences: HBM loads and stores stay in flight across it */
__device__ __forceinline__ void tile_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

struct __attribute__((aligned(16))) RsLds {
    float2 pq[2][kTile][64]; /* R1 -> R2 */
    float g[2][kTile][64];   /* R2 -> H  */
};

} // namespace

__global__ __launch_bounds__(192) void resynth_fwd_kernel(ResynthArgs a)
{
    __shared__ RsLds S;
    __shared__ __attribute__((aligned(16))) float xs[kTile];
    const int lane = threadIdx.x & 63;
    const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int u = a.order ? a.order[blockIdx.x] : (int)blockIdx.x;
    const long long off = a.offsets[u], L = a.lengths[u];
    const long long ntile = (L + kTile - 1) / kTile, niter = ntile + 2;
    if (role == 0) {
        const int16_t *in = a.in + off;
        const float f1 = a.tables->f1[lane], f2 = a.tables->f2[lane];
        GtLo s = {0, 0, 0, 0};
        float xnext = (lane < kTile && lane < L) ? (float)in[lane] : 0.0f; /* extractwav.cpp:55-58 */
        for (long long j = 0; j < niter; ++j) {
            if (j < ntile) {
                if (lane < kTile) xs[lane] = xnext;
                const long long nn = (j + 1) * kTile + lane;
                if (lane < kTile) xnext = (nn < L) ? (float)in[nn] : 0.0f;
                wave_sync();
                float2(*o)[64] = S.pq[j & 1];
#pragma unroll
                for (int t = 0; t < kTile; ++t) o[t][lane] = gt_step_lo(s, xs[t], f1, f2);
                wave_sync();
            }
            tile_sync();
        }
    } else if (role == 1) {
        const float f1 = a.tables->f1[lane], f2 = a.tables->f2[lane], gain = a.tables->gain[lane];
        GtHi s = {0, 0, 0, 0, 0, 0};
        for (long long j = 0; j < niter; ++j) {
            const long long jt = j - 1;
            if (jt >= 0 && jt < ntile) {
                const float2(*i)[64] = S.pq[jt & 1];
                float(*o)[64] = S.g[jt & 1];
#pragma unroll
                for (int t = 0; t < kTile; ++t) o[t][lane] = gt_step_hi(s, i[t][lane], f1, f2, gain);
            }
            tile_sync();
        }
    } else {
        /* reverse[...] = gOut / midEar (extractwav.cpp:86-87), streamed to HBM in natural time order */
        float *rin = a.inter + off * 64 + lane;
        const float ear = a.tables->midEar[lane];
        for (long long j = 0; j < niter; ++j) {
            const long long jt = j - 2;
            if (jt >= 0 && jt < ntile) {
                const long long n0 = jt * kTile;
                const int cnt = (L - n0 < kTile) ? (int)(L - n0) : kTile;
                const float(*g)[64] = S.g[jt & 1];
                if (cnt == kTile) {
#pragma unroll
                    for (int t = 0; t < kTile; ++t) rin[(n0 + t) * 64] = g[t][lane] / ear;
                } else {
                    for (int t = 0; t < cnt; ++t) rin[(n0 + t) * 64] = g[t][lane] / ear;
                }
            }
            tile_sync();
        }
    }
}

/* gammaToneFilter() for one channel of the bank (HuWang.h:49): a serial recurrence, one lane. */
__global__ __launch_bounds__(64) void gammatone_kernel(const float *in, float *out, int chan, long long L,
                                                       const sea_gt_tables *t)
{
    if (threadIdx.x != 0) return;
    const float f1 = t->f1[chan], f2 = t->f2[chan], gain = t->gain[chan];
    GtState s = {0, 0, 0, 0, 0, 0, 0, 0};
    for (long long n = 0; n < L; ++n) out[n] = gt_step(s, in[n], f1, f2, gain);
}

__global__ __launch_bounds__(192) void resynth_bwd_kernel(ResynthArgs a)
{
    __shared__ RsLds S;
    __shared__ float prod[kTile * kTileStride];
    __shared__ double olaUp[160], olaDown[160];
    const int lane = threadIdx.x & 63;
    const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int u = a.order ? a.order[blockIdx.x] : (int)blockIdx.x;
    const long long off = a.offsets[u], L = a.lengths[u];
    if (L < 320) return; /* no mask frame fits (wave-uniform exit before any barrier) */
    const long long ntile = (L + kTile - 1) / kTile, niter = ntile + 2;
    for (int i = threadIdx.x; i < 160; i += 192) {
        olaUp[i] = a.tables->olaUp[i];
        olaDown[i] = a.tables->olaDown[i];
    }
    tile_sync();

    if (role == 0) {
        /* second pass over the time-reversed signal (extractwav.cpp:88), stages 0-1 */
        const float *rin = a.inter + off * 64 + lane;
        const float f1 = a.tables->f1[lane], f2 = a.tables->f2[lane];
        GtLo s = {0, 0, 0, 0};
        constexpr int kAhead = 8;
        float cur[kAhead], nxt[kAhead];
#pragma unroll
        for (int k = 0; k < kAhead; ++k) cur[k] = (k < L) ? rin[(L - 1 - k) * 64] : 0.0f;
        for (long long j = 0; j < niter; ++j) {
            if (j < ntile) {
                float2(*o)[64] = S.pq[j & 1];
#pragma unroll
                for (int t0 = 0; t0 < kTile; t0 += kAhead) {
#pragma unroll
                    for (int k = 0; k < kAhead; ++k) { /* time runs backwards: row index decreases */
                        const long long nn = j * kTile + t0 + kAhead + k;
                        nxt[k] = (nn < L) ? rin[(L - 1 - nn) * 64] : 0.0f;
                    }
#pragma unroll
                    for (int k = 0; k < kAhead; ++k) o[t0 + k][lane] = gt_step_lo(s, cur[k], f1, f2);
#pragma unroll
                    for (int k = 0; k < kAhead; ++k) cur[k] = nxt[k];
                }
            }
            tile_sync();
        }
    } else if (role == 1) {
        const float f1 = a.tables->f1[lane], f2 = a.tables->f2[lane], gain = a.tables->gain[lane];
        GtHi s = {0, 0, 0, 0, 0, 0};
        for (long long j = 0; j < niter; ++j) {
            const long long jt = j - 1;
            if (jt >= 0 && jt < ntile) {
                const float2(*i)[64] = S.pq[jt & 1];
                float(*o)[64] = S.g[jt & 1];
#pragma unroll
                for (int t = 0; t < kTile; ++t) o[t][lane] = gt_step_hi(s, i[t][lane], f1, f2, gain);
            }
            tile_sync();
        }
    } else {
        const long long F = (L - 320) / 160 + 1;
        const float *mask = a.mask + a.mask_offsets[u] * 64 + lane;
        int16_t *out = a.out + off;
        const float ear = a.tables->midEar[lane];
        const bool binary = a.binary != 0;
        /* mask value of row h as the weight code sees it: the IBM variant turns > 0.5 into 1.0 and
         * skips everything else (resyth_64sub_IBM/cpp/extractwav.cpp:97-99); skipped == 0 here */
        auto mask_row = [&](long long h) -> float {
            if (h < 0 || h >= F) return 0.0f;
            const float v = mask[h * 64];
            return binary ? ((v > 0.5f) ? 1.0f : 0.0f) : v;
        };
        /* output sample m = L-1-n runs backwards through hops of 160: r = m % 160 counts down.
         * mh / mh1: mask rows of hop h (falling half) and h+1 (rising half); mhPrev: row h-1,
         * fetched one hop ahead so that its latency is never exposed. */
        long long h = (L - 1) / 160;
        int r = (int)((L - 1) - h * 160);
        float mh = mask_row(h), mh1 = mask_row(h + 1), mhPrev = mask_row(h - 1);
        for (long long j = 0; j < niter; ++j) {
            const long long jt = j - 2;
            if (jt >= 0 && jt < ntile) {
                const long long n0 = jt * kTile;
                const int cnt = (L - n0 < kTile) ? (int)(L - n0) : kTile;
                const float(*g)[64] = S.g[jt & 1];
                if (cnt == kTile && r >= kTile - 1) {
                    /* whole tile inside one hop (9 tiles out of 10): branch-free, 16 independent
                     * steps for the scheduler.  float(double(0.0f) + x) == float(x), so the first
                     * accumulation needs no add. */
                    const double mhD = (double)mh, mh1D = (double)mh1;
                    const bool useH = mh > 0.0f, useH1 = mh1 > 0.0f;
#pragma unroll
                    for (int t = 0; t < kTile; ++t) {
                        const float v = g[t][lane] / ear; /* :89-90, the value landing on sample m */
                        const float w1 = (float)(olaDown[r - t] * mhD);
                        float w = useH ? w1 : 0.0f;
                        const float w2 = (float)((double)w + olaUp[r - t] * mh1D);
                        w = useH1 ? w2 : w;
                        prod[t * kTileStride + lane] = w * v; /* :108-112 term of this channel */
                    }
                    r -= kTile;
                    if (r < 0) { /* step into hop h-1 */
                        r += 160;
                        h--;
                        mh1 = mh;
                        mh = mhPrev;
                        mhPrev = mask_row(h - 1);
                    }
                } else {
                    for (int t = 0; t < cnt; ++t) {
                        const float v = g[t][lane] / ear;
                        float w = 0.0f; /* :91-107: falling half of frame h, then rising half of h+1 */
                        if (mh > 0.0f) w = (float)((double)w + olaDown[r] * (double)mh);
                        if (mh1 > 0.0f) w = (float)((double)w + olaUp[r] * (double)mh1);
                        prod[t * kTileStride + lane] = w * v;
                        if (--r < 0) { /* step into hop h-1 */
                            r = 159;
                            h--;
                            mh1 = mh;
                            mh = mhPrev;
                            mhPrev = mask_row(h - 1);
                        }
                    }
                }
                wave_sync();
                if (lane < cnt) { /* channel sum in order 0..63 for sample t = lane, (short) cast :120-121 */
                    float acc = 0.0f;
                    const float *row = prod + lane * kTileStride;
#pragma unroll 16
                    for (int c = 0; c < 64; ++c) acc += row[c];
                    out[L - 1 - (n0 + lane)] = (int16_t)cast_i16(acc);
                }
                wave_sync();
            }
            tile_sync();
        }
    }
}

/* ---- subbband() -- the analysis half on its own --------------------------------
 * gammatone -> Meddis hair cell -> (short) cast -> 64 int16 streams per utterance
 * (enhancement_extract_test/cpp/extractwav.cpp:40-101; hairCell resyth_64sub_ori/cpp/extractwav.cpp:
 * 212-257, constants HuWang.h:36-44).  Output is 64x the input: 128 B written per input sample, the
 * one genuinely HBM-write-heavy kernel of the path.  Workgroup = one utterance = five waves,
 * lane = channel, 16-sample tiles:
 *   R1, R2  the two halves of the gammatone cascade (as in the resynthesis kernels)
 *   K       the hair cell's input-only permeability kt (a double division per sample)
 *   HC      the q/c/w recurrence, output hdt*c truncated to int16 into a padded LDS tile
 *   W       transposes the tile (lane = 16 samples x 4 channels) and streams 32-byte runs of each
 *           channel's row; consecutive tiles complete the 128-byte lines in L2. */
namespace {

struct HairCell { /* Meddis 1988 constants as the reference's C++ evaluates them (float dt, double literals) */
    float ymdt, xdt, ydt, lplusrdt, rdt, gdt, hdt;
    float q, c, w;
};

__device__ __forceinline__ void haircell_init(HairCell &h)
{
    const double Y = 5.05, G = 2000.0, Lc = 2500.0, R = 6580.0, X = 66.31, A = 3.0, B = 300.0, H = 48000.0, M = 1.0;
    const float dt = 1 / (float)16000;
    h.ymdt = (float)(Y * M * dt);
    h.xdt = (float)(X * dt);
    h.ydt = (float)(Y * dt);
    h.lplusrdt = (float)((Lc + R) * dt);
    h.rdt = (float)(R * dt);
    h.gdt = (float)(G * dt);
    h.hdt = (float)H;
    const float kt = (float)(G * A / (A + B));
    h.c = (float)(M * Y * kt / (Lc * kt + Y * (Lc + R)));
    h.q = (float)(h.c * (Lc + R) / kt);
    h.w = (float)(h.c * R / X);
}

/* permeability of one sample (extractwav.cpp:237): depends on the input only */
__device__ __forceinline__ float haircell_kt(const HairCell &h, float in)
{
    const double s = (double)in + 3.0;
    return (s > 0.0) ? (float)((double)h.gdt * s / (s + 300.0)) : 0.0f;
}

/* the recurrence (extractwav.cpp:239-255); returns output[n] = hdt * c */
__device__ __forceinline__ float haircell_step(HairCell &h, float kt)
{
    const float replenish = ((double)h.q < 1.0) ? (h.ymdt - h.ydt * h.q) : 0.0f;
    const float eject = kt * h.q;
    const float reuptakeandloss = h.lplusrdt * h.c;
    const float reuptake = h.rdt * h.c;
    const float reprocess = h.xdt * h.w;
    h.q = h.q + replenish - eject + reprocess;
    if (h.q < 0.0f) h.q = 0.0f;
    h.c = h.c + eject - reuptakeandloss;
    if (h.c < 0.0f) h.c = 0.0f;
    h.w = h.w + reuptake - reprocess;
    if (h.w < 0.0f) h.w = 0.0f;
    return h.hdt * h.c;
}

constexpr int kSbStride = 66; /* int16 per tile row: 64 channels + 2 pad (33 words: conflict-free) */

} // namespace

__global__ __launch_bounds__(320) void subband_kernel(SubbandArgs a)
{
    __shared__ RsLds S;
    __shared__ __attribute__((aligned(16))) float xs[kTile];
    __shared__ __attribute__((aligned(16))) float ktile[2][kTile][64];
    __shared__ __attribute__((aligned(16))) short otile[2][kTile * kSbStride];
    const int lane = threadIdx.x & 63;
    const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int u = a.order ? a.order[blockIdx.x] : (int)blockIdx.x;
    const long long off = a.offsets[u], L = a.lengths[u];
    const long long ntile = (L + kTile - 1) / kTile, niter = ntile + 4;
    if (role == 0) {
        const int16_t *in = a.in + off;
        const float f1 = a.tables->f1[lane], f2 = a.tables->f2[lane];
        GtLo s = {0, 0, 0, 0};
        float xnext = (lane < kTile && lane < L) ? (float)in[lane] : 0.0f;
        for (long long j = 0; j < niter; ++j) {
            if (j < ntile) {
                if (lane < kTile) xs[lane] = xnext;
                const long long nn = (j + 1) * kTile + lane;
                if (lane < kTile) xnext = (nn < L) ? (float)in[nn] : 0.0f;
                wave_sync();
                float2(*o)[64] = S.pq[j & 1];
#pragma unroll
                for (int t = 0; t < kTile; ++t) o[t][lane] = gt_step_lo(s, xs[t], f1, f2);
                wave_sync();
            }
            tile_sync();
        }
    } else if (role == 1) {
        const float f1 = a.tables->f1[lane], f2 = a.tables->f2[lane], gain = a.tables->gain[lane];
        GtHi s = {0, 0, 0, 0, 0, 0};
        for (long long j = 0; j < niter; ++j) {
            const long long jt = j - 1;
            if (jt >= 0 && jt < ntile) {
                const float2(*i)[64] = S.pq[jt & 1];
                float(*o)[64] = S.g[jt & 1];
#pragma unroll
                for (int t = 0; t < kTile; ++t) o[t][lane] = gt_step_hi(s, i[t][lane], f1, f2, gain);
            }
            tile_sync();
        }
    } else if (role == 2) {
        /* K: the hair cell's permeability, input-only (16 independent double divisions per tile) */
        HairCell h;
        haircell_init(h);
        for (long long j = 0; j < niter; ++j) {
            const long long jt = j - 2;
            if (jt >= 0 && jt < ntile) {
                const float(*g)[64] = S.g[jt & 1];
                float(*o)[64] = ktile[jt & 1];
#pragma unroll
                for (int t = 0; t < kTile; ++t) o[t][lane] = haircell_kt(h, g[t][lane]);
            }
            tile_sync();
        }
    } else if (role == 3) {
        /* HC: the q/c/w recurrence and the (short) cast of hOut (extractwav.cpp:85-88) */
        HairCell h;
        haircell_init(h);
        for (long long j = 0; j < niter; ++j) {
            const long long jt = j - 3;
            if (jt >= 0 && jt < ntile) {
                const float(*k)[64] = ktile[jt & 1];
                short *o = otile[jt & 1];
#pragma unroll
                for (int t = 0; t < kTile; ++t) o[t * kSbStride + lane] = (short)cast_i16(haircell_step(h, k[t][lane]));
            }
            tile_sync();
        }
    } else {
        /* W: lane = sample t (0..15) of channel group cg (0..3): 16 passes cover the 64 channels */
        int16_t *out = a.out + off * 64;
        const long long Lp = (L + 7) & ~7LL; /* row pitch of this utterance's [64][Lp] block */
        const int t = lane & 15, cg = lane >> 4;
        for (long long j = 0; j < niter; ++j) {
            const long long jt = j - 4;
            if (jt >= 0 && jt < ntile) {
                const long long n = jt * kTile + t;
                const short *o = otile[jt & 1] + t * kSbStride;
                if (n < L) {
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        const int c = 4 * k + cg;
                        out[c * Lp + n] = o[c];
                    }
                }
            }
            tile_sync();
        }
    }
}

} // namespace sea
