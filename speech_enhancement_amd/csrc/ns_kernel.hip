/*
 * ns_kernel.hip -- two-stage mel-warped Wiener noise suppressor, gfx950 (MI355X).
 *
 * One 64-lane wavefront owns one utterance and walks its 80-sample frames in order (frames of an
 * utterance are serially dependent: SURVEY F6); a launch runs one wavefront per utterance of the
 * batch.  All per-utterance state lives on chip for the whole utterance:
 *   LDS   : the two 320-sample stage buffers, the 256-point FFT workspace, small exchange areas
 *   VGPRs : per-bin spectra (lane l = PSD bin l; bin 64 rides in lane 0), window / mel / IDCT /
 *           FFT-schedule constants (loaded once, lane-major tables from sea_tables.c)
 * HBM traffic per frame is the algorithmic minimum: 160 B int16 in + 160 B int16 out
 * (+320 B when the float stream for CompCeps is requested).
 *
 * Reference path reproduced (results bit-identical up to libm log/log10):
 *   etsi/cpp/AdvFrontEnd.c:125-210   etsi_denoise        utterance loop, output placement
 *   etsi/cpp/ParmInterface.c:208-330 DoAdvProcess        int16<->float, zero-frame gate
 *   etsi/cpp/NoiseSup.c:1061-1440    DoNoiseSup          two stages, latency gates, buffers
 *   etsi/cpp/NoiseSup.c:182-669      DCOffsetFil .. DoFilterWindowing
 *   etsi/cpp/MelProc.c:82-104,357-378 DoMelFB, DoMelIDCT
 *   etsi/cpp/rfft.c:45-180           rfft (sea_device.h)
 */
#include "ns_core.h"

namespace sea {

/* etsi_denoise over a packed batch: one wavefront per utterance. */
__global__ __launch_bounds__(64) void ns_denoise_kernel(NsBatchArgs a)
{
    __shared__ NsLds L;
    const int lane = threadIdx.x;
    const int u = a.order ? a.order[blockIdx.x] : (int)blockIdx.x;
    const long long off = a.offsets[u];
    const long long nfr = a.lengths[u] / SEA_HOP;
    float *outf = a.out_f32 ? a.out_f32 + off : nullptr;

    NsConst C;
    load_ns_const(C, a.tables, lane);
    NsRegs s;
    regs_init(s, C.eps);
    for (int i = lane; i < 2 * kRing; i += kLanes) (&L.ring[0][0])[i] = 0.0f;
    wave_sync();

    int firstOut = -1;
    /* lanes 0..39 carry two consecutive int16 samples each: 160 B per frame, one dword per lane */
    const uint32_t *in32 = reinterpret_cast<const uint32_t *>(a.in + off);
    uint32_t *out32 = reinterpret_cast<uint32_t *>(a.out + off);
    uint32_t nextw = (lane < 40 && nfr > 0) ? in32[lane] : 0u;

    for (long long f = 0; f < nfr; ++f) {
        const uint32_t w = nextw;
        if (f + 1 < nfr && lane < 40) nextw = in32[(f + 1) * 40 + lane]; /* prefetch next frame */

        /* zero-frame gate: (int)sum(x*x) != 0 <=> some sample != 0 (ParmInterface.c:244-251) */
        const bool any = __ballot(w != 0u) != 0ull;
        bool produced = false;
        if (any || s.onset) {
            s.onset = 1;
            const float x0 = (float)(short)(w & 0xFFFFu), x1 = (float)(short)(w >> 16);
            produced = ns_tick(L, s, C, lane, x0, x1);
            if (produced && firstOut < 0) firstOut = (int)f;
        }
        /* what etsi_denoise copies to p_denoised for this frame (AdvFrontEnd.c:186-190): the
         * DenoiseBuffer, i.e. zeros until the first NoiseSup output; float -> int16 is the bare
         * truncating cast of ParmInterface.c:266 */
        if (lane < 40) {
            uint32_t packed = 0u;
            if (produced) {
                const float2 v = *reinterpret_cast<const float2 *>(&L.outb[2 * lane]);
                packed = (uint32_t)cast_i16(v.x) | ((uint32_t)cast_i16(v.y) << 16);
                if (outf) *reinterpret_cast<float2 *>(outf + f * SEA_HOP + 2 * lane) = v;
            }
            out32[f * 40 + lane] = packed;
        }
        wave_sync();
    }
    if (a.first_out && lane == 0) a.first_out[u] = firstOut;
}

/* DoNoiseSup-shaped streaming: state lives in HBM between calls (one blob per stream), frames are
 * float in / float out, no zero-frame gate (that belongs to DoAdvProcess, not DoNoiseSup). */
namespace {

template <bool FD>
__device__ __forceinline__ void ns_stream_body(const NsStreamArgs &a)
{
    __shared__ NsLds L;
    const int lane = threadIdx.x;
    const int b = blockIdx.x;
    float *blob = a.state + (long long)b * kNsStateFloats;
    NsConst C;
    load_ns_const(C, a.tables, lane);
    NsRegs s;
    NsFd fd;
    fd_init(fd);
    int bits = 0; /* the flags persist between ticks (and pushes) like FEParamsX's (ParmInterface.h:86-89) */
    if (a.reset) {
        regs_init(s, C.eps);
        for (int i = lane; i < 2 * kRing; i += kLanes) (&L.ring[0][0])[i] = 0.0f;
    } else
        state_load(blob, L, s, lane, FD ? &fd : nullptr, FD ? &bits : nullptr);
    wave_sync();
    for (int f = 0; f < a.nframes; ++f) {
        const float *x = a.in + ((long long)b * a.nframes + f) * SEA_HOP;
        float *y = a.out + ((long long)b * a.nframes + f) * SEA_HOP;
        float2 v = make_float2(0.0f, 0.0f);
        if (lane < 40) v = *reinterpret_cast<const float2 *>(x + 2 * lane);
        const bool produced = ns_tick<FD>(L, s, C, lane, v.x, v.y, &fd, &bits);
        if (produced && lane < 40)
            *reinterpret_cast<float2 *>(y + 2 * lane) = *reinterpret_cast<const float2 *>(&L.outb[2 * lane]);
        if (lane == 0) {
            a.produced[(long long)b * a.nframes + f] = produced ? 1 : 0;
            if (FD) {
                if (a.flags) a.flags[(long long)b * a.nframes + f] = (unsigned char)bits;
                if (a.frame_counter) a.frame_counter[(long long)b * a.nframes + f] = s.nbFrame[0];
            }
        }
        wave_sync();
    }
    state_store(blob, L, s, lane, FD ? &fd : nullptr, bits);
}

} // namespace

__global__ __launch_bounds__(64) void ns_stream_kernel(NsStreamArgs a) { ns_stream_body<false>(a); }

/* the same with the frame-dropping VAD's inputs (the etsi_denoise_mapping_func_Wiener output shape:
 * function/20141106_speech_enhancement/aurora_etsi/NoiseSupExports.h:19-27) */
__global__ __launch_bounds__(64) void ns_stream_fd_kernel(NsStreamArgs a) { ns_stream_body<true>(a); }

/* ---- device self-tests behind sea_selftest_*(): exhaustive / adversarial checks of the two places
 * where the kernel takes a cheaper route than the reference's literal arithmetic ---- */
__global__ __launch_bounds__(256) void selftest_pi4_kernel(unsigned long long *mismatches)
{
    unsigned long long bad = 0;
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long v = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; v < (1ull << 32); v += stride)
        bad += pi4_identity_holds(__uint_as_float((unsigned)v)) ? 0 : 1;
    if (bad) atomicAdd(mismatches, bad);
}

__global__ __launch_bounds__(64) void selftest_dc_kernel(const float *dif, const float *y0, float *out, int *fellback,
                                                         int ncases)
{
    __shared__ __attribute__((aligned(16))) float d[80], o[80];
    const int lane = threadIdx.x;
    for (int c = blockIdx.x; c < ncases; c += gridDim.x) {
        d[lane] = dif[c * 80 + lane];
        if (lane < 16) d[64 + lane] = dif[c * 80 + 64 + lane];
        wave_sync();
        float y = y0[c];
        const bool fb = dc_filter(d, o, y, lane);
        out[c * 80 + lane] = o[lane];
        if (lane < 16) out[c * 80 + 64 + lane] = o[64 + lane];
        if (lane == 0) fellback[c] = fb ? 1 : 0;
        wave_sync();
    }
}

/* the hot path's own natural log (ns_core.h) on n floats promoted to double */
__global__ __launch_bounds__(256) void selftest_log_kernel(const float *x, double *out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = ns_ln((double)x[i]);
}

/* the slow path's double-double logarithm on n doubles: hi + lo */
__global__ __launch_bounds__(256) void selftest_log_dd_kernel(const double *x, double *hi, double *lo, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const NsDD l = ns_ln_dd(x[i]);
        hi[i] = l.hi;
        lo[i] = l.lo;
    }
}

/* the two guarded call sites, complete (fast log, guard, slow path), on n floats: site1[i] = the VAD frame
 * log-energy for frameSum = x[i] (x >= 64), site2[i] = averSNR for x[i] (x > 1e-5); NaN outside a site's range */
__global__ __launch_bounds__(256) void selftest_log_sites_kernel(const float *x, float *site1, float *site2, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const float v = x[i];
        site1[i] = (v >= 64.0f) ? ns_vad_energy_expr(v) : __uint_as_float(0x7fc00000u);
        site2[i] = ((double)v > 0.00001) ? ns_aversnr_expr(v) : __uint_as_float(0x7fc00000u);
    }
}

/* EVERY float argument a site can see -- site 1: every finite float frameSum >= 64 (the int16 batch path stays below
 * 64 + 80 * 32768^2 = 2^36.3; the float streaming entry points, sea_ns_stream_push / sea_ns_streams_push, can present
 * any float); site 2: every finite float above 1e-5 -- through the site's fast form AND its slow form (double-double log, the
 * reference's literal operation sequence).  stats[0] = arguments, [1] = guard hits, [2] = hits where the slow
 * form changed the float, [3] = hits recorded, [4] = arguments OUTSIDE the guard window on which the two forms
 * disagree (must be 0: that is the guard's claim); the first `cap` hits are recorded as (argument, float of the
 * fast form alone, float returned) so that the host can check them against an arbitrary-precision logarithm. */
__global__ __launch_bounds__(256) void selftest_log_guard_kernel(int site, unsigned long long *stats, float *hits, int cap)
{
    const unsigned lo = (site == 1) ? 0x42800000u /* 64 */ : 0x3727C5ADu /* first float above 1e-5 (double compare) */;
    const unsigned hi = 0x7F7FFFFFu; /* the largest finite float, both sites */
    unsigned long long nhit = 0, nflip = 0, ntest = 0, nmiss = 0;
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long b = (unsigned long long)lo + blockIdx.x * blockDim.x + threadIdx.x; b <= hi; b += stride) {
        const float x = __uint_as_float((unsigned)b);
        if (site == 2 && !((double)x > 0.00001)) continue;
        ntest++;
        bool hit = false;
        float got, fast, slow;
        if (site == 1) {
            got = ns_vad_energy_expr(x, &hit);
            fast = (float)__fma_rn(ns_ln((double)x * 0.015625), 23.083120654223414, 0.5);
            slow = ns_vad_energy_slow(x);
        } else {
            got = ns_aversnr_expr(x, &hit);
            fast = (float)(ns_ln((double)x) * 2.8952965460216789);
            slow = ns_aversnr_slow(x);
        }
        if (hit) {
            nhit++;
            nflip += (__float_as_uint(got) != __float_as_uint(fast)) ? 1 : 0;
            const unsigned long long slot = atomicAdd(stats + 3, 1ull);
            if (slot < (unsigned long long)cap) {
                hits[3 * slot + 0] = x;
                hits[3 * slot + 1] = fast;
                hits[3 * slot + 2] = got;
            }
        } else
            nmiss += (__float_as_uint(got) != __float_as_uint(slow)) ? 1 : 0;
    }
    if (ntest) atomicAdd(stats + 0, ntest);
    if (nhit) atomicAdd(stats + 1, nhit);
    if (nflip) atomicAdd(stats + 2, nflip);
    if (nmiss) atomicAdd(stats + 4, nmiss);
}

/* sea_selftest_nsdiv: ns_div / ns_inv64 (ns_core.h) against the compiler's IEEE division, bit for bit, on
 * pseudo-random and edge-mantissa operands spanning the whole domain ns_back() admits: denominators
 * 2^-30 .. 2^59, numerators 0 or 2^-76 .. 2^49 with an exponent difference within [-106, 80]; the double
 * reciprocal on d = 1 + 0.1 r2, r2 in {0} u [2^-100, 2^80].
 * out[0] = float pairs tested, out[1] = float mismatches, out[2] = double mismatches,
 * out[3] = mismatches of ns_sqrt_fast against sqrtf over every float of its range */
__device__ __forceinline__ unsigned st_hash(unsigned x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ float st_float(unsigned h, int expLo, int expHi)
{
    const unsigned sel = h >> 29; /* mantissa: mostly random, sometimes all zeros / all ones / one bit */
    unsigned man = st_hash(h) & 0x7fffffu;
    if (sel == 0) man = 0;
    if (sel == 1) man = 0x7fffffu;
    if (sel == 2) man = 1u << (h % 23u);
    const int e = expLo + (int)((h >> 8) % (unsigned)(expHi - expLo + 1));
    return __uint_as_float(((unsigned)(e + 127) << 23) | man);
}
__global__ __launch_bounds__(256) void selftest_nsdiv_kernel(unsigned long long *out, int iters)
{
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long bad32 = 0, bad64 = 0;
    for (int it = 0; it < iters; ++it) {
        const unsigned h = st_hash(tid * 2654435761u + (unsigned)it * 40503u + 1u);
        const float b = st_float(st_hash(h ^ 0x9e3779b9u), -30, 58);
        const int eb = (int)(__float_as_uint(b) >> 23) - 127;
        float a = st_float(st_hash(h + 0x85ebca6bu), max(-76, eb - 106), min(48, eb + 80));
        if ((h & 63u) == 0) a = 0.0f;
        const float want = a / b, got = ns_div(a, ns_rcp(b));
        bad32 += (__float_as_uint(want) != __float_as_uint(got)) ? 1 : 0;
        float r2 = st_float(st_hash(h ^ 0x27d4eb2fu), -100, 79);
        if ((h & 127u) == 1) r2 = 0.0f;
        const double d = 1.0 + 0.1 * (double)r2;
        bad64 += (__double_as_longlong(1.0 / d) != __double_as_longlong(ns_inv64(d))) ? 1 : 0;
    }
    /* ns_sqrt_fast against sqrtf on EVERY float of its range, 2^-96 .. 2^126, and on 0 */
    unsigned long long badsq = 0;
    const unsigned lo = 0x0F800000u, hi = 0x7E800000u;
    for (unsigned long long v = (unsigned long long)lo + tid; v <= hi; v += (unsigned long long)gridDim.x * blockDim.x) {
        const float x = __uint_as_float((unsigned)v);
        badsq += (__float_as_uint(sqrtf(x)) != __float_as_uint(ns_sqrt_fast(x))) ? 1 : 0;
    }
    if (tid == 0) badsq += (__float_as_uint(ns_sqrt_fast(0.0f)) != 0u) ? 1 : 0;
    if (bad32) atomicAdd(out + 1, bad32);
    if (bad64) atomicAdd(out + 2, bad64);
    if (badsq) atomicAdd(out + 3, badsq);
    if (tid == 0) out[0] = (unsigned long long)gridDim.x * blockDim.x * (unsigned long long)iters;
}

} // namespace sea
