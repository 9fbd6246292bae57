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

__global__ __launch_bounds__(64) void irm_target_kernel(IrmArgs a)
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

} // namespace sea
