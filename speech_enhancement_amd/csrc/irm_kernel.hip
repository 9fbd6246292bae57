/*
 * irm_kernel.hip -- SURVEY 8(f) rank 2: the ideal-ratio-mask TARGET of make_single_IBM
 * (enhancement_extract_test/cpp/show_IBM.cpp:105-169), gfx950.
 *
 * Input: the 64 int16 subband streams of the clean and of the noise signal of every utterance, exactly as
 * sea::subband_kernel leaves them ([64][pitch] per utterance).  Per stream and frame (320 samples every 160) a
 * 512-point power spectrum; its first 64 bins summed in float in bin order; IRM = sum_pure / (sum_pure +
 * sum_noise) -> one row of 64 floats per frame, the layout the resynthesis kernels take as their mask.
 *
 * Only bins 0..63 of the 512-point spectrum of 320 real samples are needed:  X[k] = E[k] + e^(-2 pi i k / 512) O[k]
 * with E / O the 256-point spectra of the even / odd samples (160 each, zero padded) -- ONE dual transform of
 * sea_device.h per frame and stream (lanes 0..31 the even half, 32..63 the odd half), then lane = bin for the
 * combination and the power, and LANE = FRAME for the in-order 64-bin sums of a tile of 16 frames.
 *
 * PARITY UNPINNED: asdk::SpecInfo (the reference's spectrum routine) is absent third-party code; its analysis
 * window is a parameter here (0 rectangular, 1 Hamming, 2 Hanning).  The oracle (oracle/resynth_oracle.c,
 * ora_irm_target) evaluates the same definition as a direct double-precision DFT; the float transform here agrees
 * with it to ~1e-6 relative (tests/test_gpu_parity.py::test_irm_target_vs_oracle, tolerance 1e-4).
 */
#include "sea_device.h"
#include "sea_kernels.h"

namespace sea {

namespace {
constexpr int kIrmT = 16; /* frames per tile */
}

/* round 2-3 form (SEA_IRM_KERNEL=dual, kept for A/B): one LDS-based dual 256-point transform per frame and stream */
__global__ __launch_bounds__(64) void irm_target_dual_kernel(IrmArgs a)
{
    __shared__ __attribute__((aligned(16))) float work[512];
    __shared__ float pw[2][kIrmT][65];
    const int lane = threadIdx.x;
    const int u = blockIdx.x >> 6, c = blockIdx.x & 63;
    const long long L = a.lengths[u];
    if (L < 320) return;
    const long long F = (L - 320) / 160 + 1, pitch = (L + 7) & ~7LL;
    const int16_t *stream[2] = {a.pure + a.offsets[u] * 64 + c * pitch, a.noise + a.offsets[u] * 64 + c * pitch};
    Fft2Regs R;
    load_fft2_regs<false>(R, a.fft, lane, nullptr);
    /* this lane's eight inputs of its half's transform: element idx = n0 + 32 bitrev3(k) of the even (h = 0) or
     * odd (h = 1) samples, i.e. sample 2 idx + h of the frame, times the analysis window; idx >= 160: zero padding */
    const int n0 = lane & 31, h = lane >> 5;
    float w8[8];
    int so[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        constexpr int kRev3[8] = {0, 4, 2, 6, 1, 5, 3, 7};
        const int idx = n0 + 32 * kRev3[k], s = 2 * idx + h;
        const bool valid = idx < 160;
        double w = 1.0;
        if (a.window == 1) w = 0.54 - 0.46 * cos(2.0 * 3.1415926535897932384626433832795 * s / 319.0);
        if (a.window == 2) w = 0.5 - 0.5 * cos(2.0 * 3.1415926535897932384626433832795 * s / 319.0);
        w8[k] = valid ? (float)w : 0.0f;
        so[k] = valid ? s : 0;
    }
    const float tc = (float)cos(2.0 * 3.1415926535897932384626433832795 * lane / 512.0);
    const float ts = (float)sin(2.0 * 3.1415926535897932384626433832795 * lane / 512.0);
    const unsigned aRe = fft_swz((unsigned)lane), aIm = fft_swz((unsigned)(256 - lane) & 255u);
    float *out = a.irm + a.row_offsets[u] * 64 + c;
    for (long long i0 = 0; i0 < F; i0 += kIrmT) {
        const int nv = (int)((F - i0 < kIrmT) ? F - i0 : kIrmT);
        for (int f = 0; f < nv; ++f)
#pragma unroll
            for (int which = 0; which < 2; ++which) {
                const int16_t *x = stream[which] + (i0 + f) * 160;
                float e[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) e[k] = w8[k] * (float)x[so[k]];
                rfft256_dual<false>(e, work, R);
                const float Ere = fft_at(work, aRe), Ore = fft_at(work + 256, aRe);
                const float Eim = (lane > 0) ? fft_at(work, aIm) : 0.0f, Oim = (lane > 0) ? fft_at(work + 256, aIm) : 0.0f;
                const float xr = Ere + (tc * Ore + ts * Oim), xi = Eim + (tc * Oim - ts * Ore);
                pw[which][f][lane] = xr * xr + xi * xi;
                wave_sync();
            }
        if (lane < nv) { /* lane = frame: bins 0..63 in order (show_IBM.cpp:154-158), then the ratio (:165) */
            float sp = 0.0f, sn = 0.0f;
#pragma unroll 8
            for (int j = 0; j < 64; ++j) {
                sp += pw[0][lane][j];
                sn += pw[1][lane][j];
            }
            out[(i0 + lane) * 64] = sp / (sp + sn);
        }
        wave_sync();
    }
}


/* ---- round 4: the spectrum in REGISTERS, lane = (polyphase component, frame) --------------------------------------------------
 * The dual-transform form above spends 253 vector and 94 LDS instructions per frame and stream on ~4.3 kflop (the LDS pipe is
 * what it waits for: profiles/r03_pmc_sq2.txt).  Only bins 0..63 of the 512-point spectrum are used, so split the frame into its
 * four polyphase components x_r[m] = x[4m + r] (80 samples each):  X[k] = sum_r W512^(rk) G_r[k],  G_r = 128-point DFT of x_r.
 * A LANE takes one component of one frame -- lane = 16 r + f, sixteen frames per wave -- and evaluates its G_r[0..63] with a
 * generated straight-line codelet (tools/gen_rdft_codelet.py -> irm_rdft128.inc: 1189 instructions, no memory, no cross-lane
 * traffic, every lane busy), multiplies by its twiddles and the four rows of the wave are added up with v_permlane16_swap /
 * v_permlane32_swap (gfx950) so that each row ends with a quarter of the bins; powers, the 64-bin sum, the ratio.
 * The samples of a 16-frame tile (17 half-frames = 2720 int16, one coalesced read) are staged in LDS as four polyphase planes:
 * lane (r, f) reads 160 contiguous bytes of plane r -- ten ds_read_b128, conflict-free within each row (80 f bytes apart: sixteen
 * different 16-byte slots of the 256-byte bank row).  The float operation order is the codelet's (parity unpinned, 1e-4). */
namespace {
constexpr int kTileF = 16;                           /* frames per tile */
constexpr int kTileS = 160 * (kTileF + 1);           /* samples per tile: 2720 */
constexpr int kPlane = kTileS / 4 + 8;               /* int16 per polyphase plane (+ pad): 688 -> 1376 B, a multiple of 16 */

__device__ __forceinline__ void rdft128_80(const float (&x)[80], float (&g_re)[64], float (&g_im)[64])
{
#include "irm_rdft128.inc"
}

/* sum over the four rows (lanes l, l + 16, l + 32, l + 48) of 128 values held one per register, leaving each row with a
 * quarter of them: row 0 -> v[0..31] of the first 32 ... see the comment above; returns the lane's 32 sums in v[0..31] */
__device__ __forceinline__ void rows_reduce(float (&re)[64], float (&im)[64], float (&qre)[16], float (&qim)[16])
{
    float ure[32], uim[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) { /* even rows keep bins k, odd rows bins k + 32 (rows 0|1 and 2|3 added) */
        auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(re[k]), __float_as_uint(re[k + 32]), false, false);
        ure[k] = __uint_as_float(a[0]) + __uint_as_float(a[1]);
        auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(im[k]), __float_as_uint(im[k + 32]), false, false);
        uim[k] = __uint_as_float(b[0]) + __uint_as_float(b[1]);
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) { /* lanes 0..31 keep u[k], lanes 32..63 u[k + 16] (rows 0|2 and 1|3 added) */
        auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(ure[k]), __float_as_uint(ure[k + 16]), false, false);
        qre[k] = __uint_as_float(a[0]) + __uint_as_float(a[1]);
        auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(uim[k]), __float_as_uint(uim[k + 16]), false, false);
        qim[k] = __uint_as_float(b[0]) + __uint_as_float(b[1]);
    }
}
__device__ __forceinline__ float rows_sum(float p)
{
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(p), __float_as_uint(p), false, false);
    const float s = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(s), __float_as_uint(s), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
} // namespace

__global__ __launch_bounds__(256, 2) void irm_target_kernel(IrmArgs a)
{
    __shared__ __attribute__((aligned(16))) float wq[4][80];     /* analysis window by polyphase component: w[4m + r] */
    __shared__ __attribute__((aligned(16))) float2 tw[4][64];    /* W512^(rk) = (cos, -sin)(2 pi r k / 512) */
    __shared__ __attribute__((aligned(16))) short planes[4][4][kPlane]; /* [wave][component][position] */
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int u = blockIdx.x >> 6, c = blockIdx.x & 63;
    const long long L = a.lengths[u];
    for (int i = threadIdx.x; i < 320; i += 256) {
        double w = 1.0;
        if (a.window == 1) w = 0.54 - 0.46 * cos(2.0 * 3.1415926535897932384626433832795 * i / 319.0);
        if (a.window == 2) w = 0.5 - 0.5 * cos(2.0 * 3.1415926535897932384626433832795 * i / 319.0);
        wq[i & 3][i >> 2] = (float)w;
    }
    {
        const int r = threadIdx.x >> 6, k = threadIdx.x & 63;
        const double ang = 2.0 * 3.1415926535897932384626433832795 * (double)(r * k) / 512.0;
        tw[r][k] = make_float2((float)cos(ang), (float)-sin(ang));
    }
    __syncthreads();
    if (L < 320) return;
    const long long F = (L - 320) / 160 + 1, pitch = (L + 7) & ~7LL;
    const int16_t *stream[2] = {a.pure + a.offsets[u] * 64 + c * pitch, a.noise + a.offsets[u] * 64 + c * pitch};
    float *out = a.irm + a.row_offsets[u] * 64 + c;
    const int r = lane >> 4, f = lane & 15;
    short(*pl)[kPlane] = planes[wave];
    const long long ntile = (F + kTileF - 1) / kTileF;
    for (long long tile = wave; tile < ntile; tile += 4) {
        const long long i0 = tile * kTileF;
        float sum2[2];
#pragma unroll 1
        for (int which = 0; which < 2; ++which) {
            /* stage the tile: 340 x 16 bytes, sample p of the tile -> plane p & 3, position p >> 2 */
            const int16_t *src = stream[which] + i0 * 160;
            const long long avail = pitch - i0 * 160; /* samples of this channel row from the tile's start on */
            uint4 v[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const int idx = lane + 64 * j; /* 16-byte piece: samples 8 idx .. 8 idx + 7 */
                v[j] = (idx < kTileS / 8 && 8LL * idx + 8 <= avail) ? *reinterpret_cast<const uint4 *>(src + 8 * idx) : make_uint4(0u, 0u, 0u, 0u);
            }
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const int idx = lane + 64 * j;
                if (idx < kTileS / 8) {
                    const uint32_t d0 = v[j].x, d1 = v[j].y, d2 = v[j].z, d3 = v[j].w; /* (s0,s1) (s2,s3) (s4,s5) (s6,s7) */
                    reinterpret_cast<uint32_t *>(pl[0])[idx] = (d0 & 0xffffu) | (d2 << 16);
                    reinterpret_cast<uint32_t *>(pl[1])[idx] = (d0 >> 16) | (d2 & 0xffff0000u);
                    reinterpret_cast<uint32_t *>(pl[2])[idx] = (d1 & 0xffffu) | (d3 << 16);
                    reinterpret_cast<uint32_t *>(pl[3])[idx] = (d1 >> 16) | (d3 & 0xffff0000u);
                }
            }
            wave_sync();
            /* this lane's 80 windowed samples: component r of frame f = positions 40 f .. 40 f + 79 of plane r */
            float x[80];
            {
                const uint4 *p = reinterpret_cast<const uint4 *>(pl[r] + 40 * f);
                const float4 *w4 = reinterpret_cast<const float4 *>(wq[r]);
#pragma unroll
                for (int j = 0; j < 10; ++j) {
                    const uint4 q = p[j];
                    const float4 wa = w4[2 * j], wb = w4[2 * j + 1];
                    x[8 * j + 0] = (float)(short)(q.x & 0xffffu) * wa.x;
                    x[8 * j + 1] = (float)(short)(q.x >> 16) * wa.y;
                    x[8 * j + 2] = (float)(short)(q.y & 0xffffu) * wa.z;
                    x[8 * j + 3] = (float)(short)(q.y >> 16) * wa.w;
                    x[8 * j + 4] = (float)(short)(q.z & 0xffffu) * wb.x;
                    x[8 * j + 5] = (float)(short)(q.z >> 16) * wb.y;
                    x[8 * j + 6] = (float)(short)(q.w & 0xffffu) * wb.z;
                    x[8 * j + 7] = (float)(short)(q.w >> 16) * wb.w;
                }
            }
            wave_sync(); /* the planes may be overwritten by the next stream's staging */
            float g_re[64], g_im[64];
            rdft128_80(x, g_re, g_im);
            /* times W512^(rk) */
#pragma unroll
            for (int k = 0; k < 64; ++k) {
                const float2 t = tw[r][k];
                const float re = g_re[k] * t.x - g_im[k] * t.y, im = g_re[k] * t.y + g_im[k] * t.x;
                g_re[k] = re;
                g_im[k] = im;
            }
            float qre[16], qim[16];
            rows_reduce(g_re, g_im, qre, qim);
            float p = 0.0f;
#pragma unroll
            for (int k = 0; k < 16; ++k) p += qre[k] * qre[k] + qim[k] * qim[k];
            sum2[which] = rows_sum(p); /* sum over bins 0..63 of |X[k]|^2 (show_IBM.cpp:154-158) */
        }
        if (r == 0 && i0 + f < F) out[(i0 + f) * 64] = sum2[0] / (sum2[0] + sum2[1]); /* the ratio (:165) */
    }
}

} // namespace sea
