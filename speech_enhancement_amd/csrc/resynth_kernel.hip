/*
 * resynth_kernel.hip -- Hu-Wang 64-channel gammatone analysis/synthesis resynthesis, gfx950.
 *
 * One 64-lane wavefront owns one utterance; LANE = CHANNEL (64 channels = one wave), so the
 * 4th-order complex one-pole cascade of every channel advances one sample per step in lock-step.
 *
 *   resynth_fwd_kernel : analysis pass.  g1[n][c] written to HBM as rows of 64 floats (256 B per
 *                        step, fully coalesced).  288 GB of HBM is what makes keeping the whole
 *                        [L][64] intermediate of a 1024-utterance batch (~17 GB) resident feasible.
 *   resynth_bwd_kernel : reads the rows in reverse time order, divides by the middle-ear gain,
 *                        re-filters, divides again, evaluates the mask-weighted raised-cosine
 *                        overlap-add weight of that sample on the fly (at most two overlapping
 *                        frames per sample), multiplies, and sums the 64 channels IN CHANNEL ORDER
 *                        through a padded LDS transpose (64 samples at a time, lane = sample), then
 *                        truncates to int16.  No second intermediate is written.
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

constexpr int kTileStride = 65; /* 64 channels + 1 pad: conflict-free row reads by lane = sample */

} // namespace

__global__ __launch_bounds__(64) void resynth_fwd_kernel(ResynthArgs a)
{
    __shared__ float xs[64];
    const int lane = threadIdx.x;
    const int u = a.order ? a.order[blockIdx.x] : (int)blockIdx.x;
    const long long off = a.offsets[u], L = a.lengths[u];
    const int16_t *in = a.in + off;
    float *g1 = a.inter + off * 64;
    const float f1 = a.tables->f1[lane], f2 = a.tables->f2[lane], gain = a.tables->gain[lane];
    GtState s = {0, 0, 0, 0, 0, 0, 0, 0};

    for (long long n0 = 0; n0 < L; n0 += 64) {
        const int cnt = (L - n0 < 64) ? (int)(L - n0) : 64;
        wave_sync();
        xs[lane] = (lane < cnt) ? (float)in[n0 + lane] : 0.0f; /* extractwav.cpp:55-58 */
        wave_sync();
        if (cnt == 64) {
#pragma unroll 8
            for (int t = 0; t < 64; ++t) g1[(n0 + t) * 64 + lane] = gt_step(s, xs[t], f1, f2, gain);
        } else {
            for (int t = 0; t < cnt; ++t) g1[(n0 + t) * 64 + lane] = gt_step(s, xs[t], f1, f2, gain);
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

__global__ __launch_bounds__(64) void resynth_bwd_kernel(ResynthArgs a)
{
    __shared__ float tile[64 * kTileStride];
    __shared__ double olaUp[160], olaDown[160];
    const int lane = threadIdx.x;
    const int u = a.order ? a.order[blockIdx.x] : (int)blockIdx.x;
    const long long off = a.offsets[u], L = a.lengths[u];
    if (L < 320) return; /* no mask frame fits: the reference would index before the array */
    const long long F = (L - 320) / 160 + 1;
    const float *g1 = a.inter + off * 64;
    const float *mask = a.mask + a.mask_offsets[u] * 64;
    int16_t *out = a.out + off;
    const float f1 = a.tables->f1[lane], f2 = a.tables->f2[lane], gain = a.tables->gain[lane];
    const float ear = a.tables->midEar[lane];
    const bool binary = a.binary != 0;
    for (int i = lane; i < 160; i += 64) {
        olaUp[i] = a.tables->olaUp[i];
        olaDown[i] = a.tables->olaDown[i];
    }
    wave_sync();

    GtState s = {0, 0, 0, 0, 0, 0, 0, 0};
    /* mask rows of the hop that contains output sample m (h) and of the next hop (h+1) */
    long long hop = -1;
    float mh = 0.0f, mh1 = 0.0f;

    for (long long n0 = 0; n0 < L; n0 += 64) {
        const int cnt = (L - n0 < 64) ? (int)(L - n0) : 64;
        for (int t = 0; t < cnt; ++t) {
            const long long n = n0 + t, m = L - 1 - n;
            /* reverse[n] = g1[L-1-n] / midEar (extractwav.cpp:86-87), second pass (:88) */
            const float rin = g1[m * 64 + lane] / ear;
            const float g2 = gt_step(s, rin, f1, f2, gain);
            /* reverse[L-1-n] = g2[n] / midEar (:89-90): the value that lands on output sample m */
            const float v = g2 / ear;
            /* overlap-add weight of sample m (:91-107): falling half of frame h, then rising half
             * of frame h+1, each added as float(double(w) + half * mask) */
            const long long h = m / 160;
            const int r = (int)(m - h * 160);
            if (h != hop) {
                hop = h;
                mh = (h < F) ? mask[h * 64 + lane] : 0.0f;
                mh1 = (h + 1 < F) ? mask[(h + 1) * 64 + lane] : 0.0f;
                if (binary) { /* IBM: > 0.5 becomes 1.0, everything else is skipped */
                    mh = (mh > 0.5f) ? 1.0f : 0.0f;
                    mh1 = (mh1 > 0.5f) ? 1.0f : 0.0f;
                }
            }
            float w = 0.0f;
            if (mh > 0.0f) w = (float)((double)w + olaDown[r] * (double)mh);
            if (mh1 > 0.0f) w = (float)((double)w + olaUp[r] * (double)mh1);
            tile[t * kTileStride + lane] = w * v; /* :108-112 term of this channel */
        }
        wave_sync();
        /* channel sum in order 0..63 for sample t = lane, then the (short) cast (:120-121) */
        if (lane < cnt) {
            float acc = 0.0f;
            const float *row = tile + lane * kTileStride;
#pragma unroll 16
            for (int c = 0; c < 64; ++c) acc += row[c];
            out[L - 1 - (n0 + lane)] = (int16_t)cast_i16(acc);
        }
        wave_sync();
    }
}

} // namespace sea
