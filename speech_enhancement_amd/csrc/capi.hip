/*
 * capi.hip -- the C ABI of libsea_mi355x.so (include/sea_mi355x.h): context, launches, and the
 * host-buffer drop-ins that keep the reference's own signatures.
 */
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <new>
#include <vector>

#include "capi_internal.h"

namespace sea_capi {

namespace {
thread_local char g_err[512] = "";
}

int fail(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return 1;
}
const char *last_error() { return g_err; }

namespace {
constexpr int kMaxDev = 64;
DeviceCtx g_ctx[kMaxDev];
std::mutex g_mu;
sea_ns_tables g_ns_host;
sea_cc_tables g_cc_host;
sea_gt_tables g_gt_host;
sea_ns16k_tables g_ns16_host;
bool g_host_ready = false;

void host_tables()
{
    if (g_host_ready) return;
    sea_build_ns_tables(&g_ns_host);
    sea_build_cc_tables(&g_cc_host);
    sea_build_gt_tables(&g_gt_host);
    sea_build_ns16k_tables(&g_ns16_host);
    g_host_ready = true;
}
} // namespace

int ctx(DeviceCtx **out)
{
    int dev = -1;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= kMaxDev) return fail("device index %d out of range", dev);
    std::lock_guard<std::mutex> lk(g_mu);
    DeviceCtx &c = g_ctx[dev];
    if (!c.ready) {
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, dev));
        if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
            return fail("device %d is %s; this library is built for gfx950 (MI355X) only", dev, prop.gcnArchName);
        c.n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        host_tables();
        HIP_TRY(hipMalloc(&c.ns, sizeof(sea_ns_tables)));
        HIP_TRY(hipMalloc(&c.cc, sizeof(sea_cc_tables)));
        HIP_TRY(hipMalloc(&c.gt, sizeof(sea_gt_tables)));
        HIP_TRY(hipMalloc(&c.ns16, sizeof(sea_ns16k_tables)));
        HIP_TRY(hipMemcpy(c.ns, &g_ns_host, sizeof g_ns_host, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(c.cc, &g_cc_host, sizeof g_cc_host, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(c.gt, &g_gt_host, sizeof g_gt_host, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(c.ns16, &g_ns16_host, sizeof g_ns16_host, hipMemcpyHostToDevice));
        c.ready = true;
    }
    *out = &c;
    return 0;
}

/* launch order of the utterance-per-workgroup kernels: longest first, every other row of n_cu reversed
 * (workgroups b, b + n_cu, ... share a CU; see speech_enhancement_amd/engine.py::launch_order) */
void launch_order(const long long *lens, int n, int n_cu, int *order)
{
    std::vector<int> idx(n);
    for (int i = 0; i < n; ++i) idx[i] = i;
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return lens[a] > lens[b]; });
    for (int start = 0, row = 0; start < n; start += n_cu, ++row) {
        const int end = start + n_cu < n ? start + n_cu : n;
        if (row & 1) std::reverse(idx.begin() + start, idx.begin() + end);
    }
    for (int i = 0; i < n; ++i) order[i] = idx[i];
}

void gammatone_host_tables(float *cf64, float *bw64, float *midEar64)
{
    std::lock_guard<std::mutex> lk(g_mu);
    host_tables();
    memcpy(cf64, g_gt_host.cf, sizeof g_gt_host.cf);
    memcpy(bw64, g_gt_host.bw, sizeof g_gt_host.bw);
    memcpy(midEar64, g_gt_host.midEar, sizeof g_gt_host.midEar);
}

/* diagnostic (tools/ns6_perm_sweep.py): wave -> role map of the six-wave forms, three bits per wave, wave 0 lowest; 0 = the
 * kernel's own.  Initialised from SEA_NS6_PERM (octal). */
static std::atomic<int> g_ns6_perm{[] { const char *e = getenv("SEA_NS6_PERM"); return e ? (int)strtol(e, nullptr, 8) : 0; }()};
/* Forms of the same arithmetic (identical results), chosen by how many utterances share a CU:
 *   <= 3 per CU  six waves per utterance: shortest frame period (the run time is one utterance's
 *                chain of frames; 768 utterances: 1.74 ms against 1.85 for the dense form)   SEA_NS_KERNEL=pipe6
 *   <= 4 per CU  six waves per utterance compiled for seven waves per SIMD, so that four workgroups
 *                co-reside on a CU (round 4; ns_pipe6_kernel.hip; 896: 1.89 ms against 2.44)   SEA_NS_KERNEL=pipe6d
 *   more         four waves, tables in LDS: six workgroups per CU     SEA_NS_KERNEL=big
 *   (not chosen) four waves, transform address tables in VGPRs: the form for <= 4 per CU until round 4, 4 % behind the
 *                dense six-wave form there; the time-slice launches of the host pipelines run on it   SEA_NS_KERNEL=pipe
 *   (never)      two utterances per workgroup, lane-sparse phases packed   SEA_NS_KERNEL=pair (experiment, slower)
 *   (never)      one wave per utterance, the roles in sequence, no workgroup barrier (ns_wave_kernel.hip): 363 M frames/s on the
 *                configs[4] shard against 465                           SEA_NS_KERNEL=wave (experiment, slower)
 * SEA_NS_KERNEL=single: one wave per utterance (the streaming plug-in's kernel), for A/B. */
int ns_pick_form(int n_inflight, int n_cu)
{
    const int forced = sea_ns_kernel_form(-1);
    /* (form 5, two utterances per workgroup with their lane-sparse phases packed into one wave -- ns_pipe2_kernel.hip -- has
     * 22 % fewer vector instructions per frame and is slower: 434 against 465 M frames/s on the configs[4] shard; never
     * chosen here, see that file's header) */
    return forced ? forced : (n_inflight <= 3 * n_cu ? 3 : (n_inflight <= 4 * n_cu ? 6 : 4));
}

int ns_launch(const sea::NsBatchArgs &a, int form, hipStream_t stream)
{
    if (a.n_utt <= 0) return 0;
    if (a.state) { /* time slices: the four-wave forms only */
        if (form == 4 || form == 5)
            hipLaunchKernelGGL(sea::ns_denoise_pipe_big_slice_kernel, dim3(a.n_utt), dim3(256), 0, stream, a);
        else
            hipLaunchKernelGGL(sea::ns_denoise_pipe_slice_kernel, dim3(a.n_utt), dim3(256), 0, stream, a);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    if (form == 1)
        hipLaunchKernelGGL(sea::ns_denoise_kernel, dim3(a.n_utt), dim3(64), 0, stream, a);
    else if (form == 3 || form == 6) {
        sea::NsBatchArgs b = a;
        const int perm = g_ns6_perm.load(); /* diagnostic: the wave -> role map of the six-wave form (sea_debug_ns6_perm / SEA_NS6_PERM) */
        if (perm) b.perm6 = perm;
        if (form == 6) hipLaunchKernelGGL(sea::ns_denoise_pipe6_dense_kernel, dim3(a.n_utt), dim3(384), 0, stream, b);
        else hipLaunchKernelGGL(sea::ns_denoise_pipe6_kernel, dim3(a.n_utt), dim3(384), 0, stream, b);
    } else if (form == 4)
        hipLaunchKernelGGL(sea::ns_denoise_pipe_big_kernel, dim3(a.n_utt), dim3(256), 0, stream, a);
    else if (form == 7) {
        const int per = sea::ns_wave_utts_per_block();
        hipLaunchKernelGGL(sea::ns_denoise_wave_kernel, dim3((a.n_utt + per - 1) / per), dim3(64 * per), 0, stream, a);
    } else if (form == 5)
        hipLaunchKernelGGL(sea::ns_denoise_pipe_pair_kernel, dim3((a.n_utt + 1) / 2), dim3(sea::ns_pair_threads()), 0, stream, a);
    else
        hipLaunchKernelGGL(sea::ns_denoise_pipe_kernel, dim3(a.n_utt), dim3(256), 0, stream, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

} // namespace sea_capi

using namespace sea_capi;

extern "C" int sea_ns16k_kernel_form(int form);

extern "C" {

const char *sea_last_error(void) { return last_error(); }
const char *sea_version(void) { return "sea_mi355x 0.1 (gfx950)"; }

int sea_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int sea_init(int device)
{
    if (device >= 0) HIP_TRY(hipSetDevice(device));
    DeviceCtx *c;
    return ctx(&c);
}

int sea_tables_host(float *sigWindow200, float *irWindow17, float *idct25x25, int *melStart25,
                    int *melLen25, float *melData25x16, float *hamming100, float *dct12x23,
                    int *ccStart23, int *ccLen23, float *ccData23x32)
{
    sea_ns_plain_tables(sigWindow200, irWindow17, idct25x25, melStart25, melLen25, melData25x16);
    sea_cc_plain_tables(hamming100, dct12x23, ccStart23, ccLen23, ccData23x32);
    return 0;
}

/* diagnostic (tests/test_host_cpu.py, no GPU): the lane map of the tiled CompCeps kernels' mel pass (sea_tables.h, melLaneBase) */
extern "C" int sea_debug_cc_mel_lanes(int *base64, int *fb64, float *w24x64)
{
    std::lock_guard<std::mutex> lk(g_mu);
    host_tables();
    memcpy(base64, g_cc_host.melLaneBase, sizeof g_cc_host.melLaneBase);
    memcpy(fb64, g_cc_host.melLaneFb, sizeof g_cc_host.melLaneFb);
    memcpy(w24x64, g_cc_host.melLaneW, sizeof g_cc_host.melLaneW);
    return 0;
}

int sea_gammatone_channels(float *cf64, float *bw64, float *midEar64)
{
    gammatone_host_tables(cf64, bw64, midEar64);
    return 0;
}

/* ------------------------------------------------------------------------------------------- */
extern "C" int sea_debug_ns6_perm(int perm)
{
    return g_ns6_perm.exchange(perm);
}

/* kernel form override: 0 = by batch size; initialised from SEA_NS_KERNEL on first use */
static std::atomic<int> g_ns_form{-1};
static int ns_form()
{
    int f = g_ns_form.load();
    if (f < 0) {
        const char *e = getenv("SEA_NS_KERNEL");
        f = 0;
        if (e && !strcmp(e, "single")) f = 1;
        if (e && !strcmp(e, "pipe")) f = 2;
        if (e && !strcmp(e, "pipe6")) f = 3;
        if (e && !strcmp(e, "big")) f = 4;
        if (e && !strcmp(e, "pair")) f = 5;
        if (e && !strcmp(e, "pipe6d")) f = 6;
        if (e && !strcmp(e, "wave")) f = 7;
        g_ns_form.store(f);
    }
    return f;
}

/* diagnostic (tools/): resident workgroups per CU of NoiseSup kernel form f as the runtime computes them (0: unknown form) */
int sea_debug_ns_occupancy(int form)
{
    int n = 0;
    hipError_t e = hipErrorInvalidValue;
    if (form == 2) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, sea::ns_denoise_pipe_kernel, 256, 0);
    if (form == 3) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, sea::ns_denoise_pipe6_kernel, 384, 0);
    if (form == 6) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, sea::ns_denoise_pipe6_dense_kernel, 384, 0);
    if (form == 4) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, sea::ns_denoise_pipe_big_kernel, 256, 0);
    if (form == 5) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, sea::ns_denoise_pipe_pair_kernel, sea::ns_pair_threads(), 0);
    if (form == 16) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, sea::ns16k_pipe_kernel, 512, 0);
    return e == hipSuccess ? n : -1;
}

int sea_ns_kernel_form(int form)
{
    const int prev = ns_form();
    if (form >= 0 && form <= 7) g_ns_form.store(form);
    return prev;
}

int sea_ns_denoise_batch(const short *d_in, short *d_out, float *d_out_f32,
                         const long long *d_offsets, const long long *d_lengths, const int *d_order,
                         int *d_first_out, int n_utt, void *stream)
{
    if (n_utt <= 0) return 0;
    DeviceCtx *c;
    if (ctx(&c)) return 1;
    sea::NsBatchArgs a = {};
    a.in = d_in;
    a.out = d_out;
    a.out_f32 = d_out_f32;
    a.offsets = d_offsets;
    a.lengths = d_lengths;
    a.order = d_order;
    a.first_out = d_first_out;
    a.tables = c->ns;
    a.n_utt = n_utt;
    const int form = ns_pick_form(n_utt, c->n_cu);
    if (form == 2 || form == 3 || form == 6) a.prio_row = (d_order && n_utt > c->n_cu) ? c->n_cu : 0; /* rows of the longest-first launch order */
    return ns_launch(a, form, (hipStream_t)stream);
}

/* One TIME SLICE of a batch: see include/sea_mi355x.h */
int sea_ns_denoise_batch_slice(const short *d_in, short *d_out, float *d_out_f32, const long long *d_offsets,
                               const long long *d_lengths, const int *d_order, int *d_first_out, float *d_state, int n_utt,
                               int frame_base, int resume, void *stream)
{
    if (n_utt <= 0) return 0;
    if (!d_state) return fail("sea_ns_denoise_batch_slice: d_state is required");
    DeviceCtx *c;
    if (ctx(&c)) return 1;
    sea::NsBatchArgs a = {};
    a.in = d_in;
    a.out = d_out;
    a.out_f32 = d_out_f32;
    a.offsets = d_offsets;
    a.lengths = d_lengths;
    a.order = d_order;
    a.first_out = d_first_out;
    a.tables = c->ns;
    a.n_utt = n_utt;
    a.state = d_state;
    a.resume = resume;
    a.frame_base = frame_base;
    const int form = (n_utt <= 4 * c->n_cu) ? 2 : 4;
    if (form == 2) a.prio_row = (d_order && n_utt > c->n_cu) ? c->n_cu : 0;
    return ns_launch(a, form, (hipStream_t)stream);
}

int sea_ns_slice_state_floats(void) { return sea::kNsPipeStateFloats; }

/* ------------------------------------------------------------------------------------------- */
int sea_ns_denoise_batch_fd(const short *d_in, short *d_out, float *d_out_f32,
                            const long long *d_offsets, const long long *d_lengths, const int *d_order,
                            int *d_first_out, unsigned char *d_flags, int *d_onset, int n_utt, void *stream)
{
    if (n_utt <= 0) return 0;
    if (!d_out_f32 || !d_first_out || !d_flags || !d_onset)
        return fail("ns_denoise_batch_fd: the float stream, first_out, flags and onset outputs are all required");
    DeviceCtx *c;
    if (ctx(&c)) return 1;
    sea::NsBatchArgs a = {};
    a.in = d_in;
    a.out = d_out;
    a.out_f32 = d_out_f32;
    a.offsets = d_offsets;
    a.lengths = d_lengths;
    a.order = d_order;
    a.first_out = d_first_out;
    a.tables = c->ns;
    a.n_utt = n_utt;
    a.flags_out = d_flags;
    a.onset_out = d_onset;
    if (n_utt <= 2 * c->n_cu)
        hipLaunchKernelGGL(sea::ns_denoise_pipe6_fd_kernel, dim3(n_utt), dim3(384), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(sea::ns_denoise_pipe_fd_kernel, dim3(n_utt), dim3(256), 0, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

int sea_afe_features_batch(const float *d_den_f32, const unsigned char *d_flags, const long long *d_offsets,
                           const long long *d_lengths, const int *d_first_out, const int *d_onset,
                           const long long *d_ceps_cum, long long total_ceps, float *d_feat_cc, float *d_feat_pp,
                           const long long *d_feat_cum, float *d_feat15, int *d_n_feat, int *d_n_ceps, int n_utt,
                           void *stream)
{
    if (n_utt <= 0) return 0;
    DeviceCtx *c;
    if (ctx(&c)) return 1;
    sea::AfeArgs a = {};
    a.den_f32 = d_den_f32;
    a.flags = d_flags;
    a.offsets = d_offsets;
    a.lengths = d_lengths;
    a.first_out = d_first_out;
    a.onset = d_onset;
    a.ceps_cum = d_ceps_cum;
    a.feat_cc = d_feat_cc;
    a.feat_pp = d_feat_pp;
    a.feat_cum = d_feat_cum;
    a.feat15 = d_feat15;
    a.n_feat = d_n_feat;
    a.n_ceps = d_n_ceps;
    a.tables = c->cc;
    a.n_utt = n_utt;
    if (total_ceps > 0) {
        const long long nslot = total_ceps / 8 + n_utt; /* tile slots of 8 frames (cc_kernel.hip, kAfeT) */
#ifndef SEA_AFE_GRID
#define SEA_AFE_GRID 8192
#endif
        const long long want = nslot < SEA_AFE_GRID ? nslot : SEA_AFE_GRID;
        hipLaunchKernelGGL(sea::afe_ceps_kernel, dim3((unsigned)want), dim3(64), 0, (hipStream_t)stream, a);
        HIP_TRY(hipGetLastError());
    }
    hipLaunchKernelGGL(sea::afe_vad_kernel, dim3(n_utt), dim3(64), 0, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

int sea_rfft256_batch(const float *d_in, float *d_out, long long nframes, void *stream)
{
    if (nframes <= 0) return 0;
    DeviceCtx *c;
    if (ctx(&c)) return 1;
    const long long npair = (nframes + 1) / 2; /* one wave transforms two frames at a time */
    const long long grid = npair < 16LL * c->n_cu ? npair : 16LL * c->n_cu;
    hipLaunchKernelGGL(sea::rfft256_kernel, dim3((unsigned)grid), dim3(64), 0, (hipStream_t)stream, d_in,
                       d_out, nframes, &c->ns->fft);
    HIP_TRY(hipGetLastError());
    return 0;
}

int sea_compceps_frames(const float *d_data201, float *d_coef14, long long nframes, void *stream)
{
    if (nframes <= 0) return 0;
    DeviceCtx *c;
    if (ctx(&c)) return 1;
    const long long ntile = (nframes + 15) / 16; /* one wave per tile of 16 frames (cc_kernel.hip, kCcT) */
    const long long grid = ntile < 8192 ? ntile : 8192;
    hipLaunchKernelGGL(sea::compceps_frames_kernel, dim3((unsigned)grid), dim3(64), 0, (hipStream_t)stream,
                       d_data201, d_coef14, nframes, c->cc);
    HIP_TRY(hipGetLastError());
    return 0;
}

int sea_compceps_batch(const float *d_den_f32, const long long *d_offsets, const long long *d_lengths,
                       const int *d_first_out, const long long *d_ceps_cum, long long total_frames,
                       float *d_ceps, int *d_n_ceps, int n_utt, void *stream)
{
    if (n_utt <= 0 || total_frames <= 0) return 0;
    DeviceCtx *c;
    if (ctx(&c)) return 1;
    sea::CepsArgs a;
    a.den_f32 = d_den_f32;
    a.offsets = d_offsets;
    a.lengths = d_lengths;
    a.first_out = d_first_out;
    a.ceps_cum = d_ceps_cum;
    a.ceps = d_ceps;
    a.n_ceps = d_n_ceps;
    a.tables = c->cc;
    a.n_utt = n_utt;
    const long long nslot = total_frames / 16 + n_utt; /* tile slots of 16 frames (cc_kernel.hip, kCcT) */
#ifndef SEA_CC_GRID
#define SEA_CC_GRID 16384 /* waves of the launch (3072 are resident): 8192 0.63 ms, 13312-24576 0.59-0.61, 3072 (every wave its share of the tiles, all in step) 1.0 */
#endif
    const long long grid = nslot < SEA_CC_GRID ? nslot : SEA_CC_GRID;
    hipLaunchKernelGGL(sea::compceps_kernel, dim3((unsigned)grid), dim3(64), 0, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

int etsi_denoise(short *p_data, short *p_denoised, long i_frame)
{
    const short *in[1] = {p_data};
    short *out[1] = {p_denoised};
    long len[1] = {i_frame};
    if (sea_denoise_utterances(in, out, len, 1)) {
        fprintf(stderr, "ERROR:   etsi_denoise (MI355X): %s\r\n", last_error());
        return 1; /* TRUE == fault, AdvFrontEnd.c:207-209 */
    }
    return 0;
}

int etsi_denoise_synchronization(short *p_data, short *p_denoised, long i_frame)
{ /* AdvFrontEnd.c:213-226: copies out only when etsi_denoise reports a fault */
    std::vector<short> tmp((size_t)(i_frame > 0 ? i_frame : 1));
    (void)p_denoised;
    return etsi_denoise(p_data, tmp.data(), i_frame);
}

int etsi_denoise_16k(short *p_data, short *p_denoised, long i_frame)
{
    (void)p_data;
    (void)p_denoised;
    (void)i_frame;
    fprintf(stderr, "ERROR:   etsi_denoise_16k: the reference's 16 kHz mode is defective (heap over-read) and is not provided\r\n");
    return fail("etsi_denoise_16k is not provided");
}

int etsi_denoise_16k_synchronization(short *p_data, short *p_denoised, long i_frame)
{ /* AdvFrontEnd.c:316-329 calls the 8 kHz-mode etsi_denoise, exactly like the non-16k twin */
    return etsi_denoise_synchronization(p_data, p_denoised, i_frame);
}

/* nframes transforms rfft (x, n, m) in place on device memory: [nframes][n] floats.  (256, 8) -- the hot path's size --
 * runs the streaming kernel; every other size the reference's routine takes (n = 2^q <= 16384, 2^m <= n;
 * etsi/cpp/rfft.c:45-180) the one-workgroup schedule walker.  The schedule of a size is built once per device. */
int sea_rfft_batch(float *d_x, int n, int m, long long nframes, void *stream)
{
    if (nframes <= 0) return 0;
    if (n == 256 && m == 8) return sea_rfft256_batch(d_x, d_x, nframes, stream);
    DeviceCtx *c;
    if (ctx(&c)) return 1;
    struct Sched {
        int device, n, m;
        unsigned *d;
    };
    static std::mutex mu;
    static std::vector<Sched> cache;
    int dev = -1;
    HIP_TRY(hipGetDevice(&dev));
    unsigned *d_sched = nullptr;
    {
        std::lock_guard<std::mutex> lk(mu);
        for (const Sched &e : cache)
            if (e.device == dev && e.n == n && e.m == m) d_sched = e.d;
        if (!d_sched) {
            unsigned long words = 0;
            unsigned *h = sea_rfft_schedule(n, m, &words);
            if (!h) return fail("rfft: size n=%d, m=%d is outside what etsi/cpp/rfft.c:45-180 can take here (n a power of two <= %d, 2^m <= n)", n, m, (int)SEA_RFFT_MAXN);
            hipError_t e = hipMalloc((void **)&d_sched, words * sizeof(unsigned));
            if (e == hipSuccess) e = hipMemcpy(d_sched, h, words * sizeof(unsigned), hipMemcpyHostToDevice);
            free(h);
            if (e != hipSuccess) {
                if (d_sched) (void)hipFree(d_sched);
                return fail("rfft: schedule upload: %s", hipGetErrorString(e));
            }
            cache.push_back(Sched{dev, n, m, d_sched});
        }
    }
    const long long grid = nframes < 8LL * c->n_cu ? nframes : 8LL * c->n_cu;
    hipLaunchKernelGGL(sea::rfft_any_kernel, dim3((unsigned)grid), dim3(256), (size_t)n * sizeof(float), (hipStream_t)stream, d_x,
                       d_sched, nframes);
    HIP_TRY(hipGetLastError());
    return 0;
}

/* etsi/cpp/rfft.h:19.  The reference's routine returns nothing and cannot fail; here a size it could not take either
 * (n not a power of two: its digit-reverse counter does not terminate properly; 2^m > n: it indexes past x) or that does
 * not fit this engine (n > 16384), or a missing device, leaves x UNTOUCHED, prints the reason to stderr and sets
 * sea_last_error() -- it no longer abort()s the caller's process. */
void rfft(float *x, int n, int m)
{
    if (!x || n < 2) return;
    DevBuf<float> d;
    const size_t bytes = (size_t)n * sizeof(float);
    bool ok = d.alloc((size_t)n) == hipSuccess && hipMemcpy(d.p, x, bytes, hipMemcpyHostToDevice) == hipSuccess;
    if (ok) ok = sea_rfft_batch(d.p, n, m, 1, nullptr) == 0;
    else fail("rfft: no usable gfx950 device or allocation failure");
    if (ok) ok = hipDeviceSynchronize() == hipSuccess && hipMemcpy(x, d.p, bytes, hipMemcpyDeviceToHost) == hipSuccess;
    if (!ok) fprintf(stderr, "ERROR:   rfft (MI355X): n=%d, m=%d not transformed: %s\r\n", n, m, last_error());
}

int sea_compceps_frame(const float *Data, float *Coef14)
{
    DevBuf<float> din, dout;
    HIP_TRY(din.alloc(201));
    HIP_TRY(dout.alloc(14));
    HIP_TRY(hipMemcpy(din.p, Data - 1, 201 * sizeof(float), hipMemcpyHostToDevice));
    if (sea_compceps_frames(din.p, dout.p, 1, nullptr)) return 1;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(Coef14, dout.p, 14 * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}

/* ------------------------------------------------------------------------------------------- */
long long sea_resynth_scratch_bytes(long long total_padded_samples, int n_utt)
{ /* [sample][64] floats, plus 8 time steps per utterance for rounding up to whole 16-step tiles */
    return (total_padded_samples + 8LL * (n_utt > 0 ? n_utt : 0)) * 64 * (long long)sizeof(float);
}

int sea_resynth64_batch(const short *d_in, short *d_out, const long long *d_offsets,
                        const long long *d_lengths, const float *d_mask,
                        const long long *d_mask_offsets, float *d_inter, const int *d_order,
                        int n_utt, int binary, void *stream)
{
    if (n_utt <= 0) return 0;
    DeviceCtx *c;
    if (ctx(&c)) return 1;
    sea::ResynthArgs a = {};
    a.in = d_in;
    a.out = d_out;
    a.offsets = d_offsets;
    a.lengths = d_lengths;
    a.mask = d_mask;
    a.mask_offsets = d_mask_offsets;
    a.inter = d_inter;
    a.order = d_order;
    a.tables = c->gt;
    a.n_utt = n_utt;
    a.binary = binary;
    /* default: both passes of an utterance in one workgroup; SEA_RESYNTH=split selects the two-launch form
     * (analysis pass of the whole batch, then synthesis pass; identical results) */
    static const bool split = [] {
        const char *e = getenv("SEA_RESYNTH");
        return e && !strcmp(e, "split");
    }();

    if (split) {
        hipLaunchKernelGGL(sea::resynth_fwd_kernel, dim3(n_utt), dim3(192), 0, (hipStream_t)stream, a);
        HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(sea::resynth_bwd_kernel, dim3(n_utt), dim3(256), 0, (hipStream_t)stream, a);
    } else {
        hipLaunchKernelGGL(sea::resynth_fused_kernel, dim3(n_utt), dim3(256), 0, (hipStream_t)stream, a);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

int sea_resynth64(const short *in, long L, const float *mask, int F, int binary, short *out)
{
    const long minL = (binary & 2) ? 160 : 320;
    if (L < minL) return fail("resynth64: L=%ld is shorter than %ld samples", L, minL);
    if (F != (int)((binary & 2) ? L / 160 : (L - 320) / 160 + 1)) return fail("resynth64: F=%d does not match L=%ld", F, L);
    const long long Lp = align8(L);
    DevBuf<short> din, dout;
    DevBuf<float> dmask, dinter;
    DevBuf<long long> dmeta;
    HIP_TRY(din.alloc((size_t)Lp));
    HIP_TRY(dout.alloc((size_t)Lp));
    HIP_TRY(dmask.alloc((size_t)F * 64));
    HIP_TRY(dinter.alloc((size_t)sea_resynth_scratch_bytes(Lp, 1) / sizeof(float)));
    HIP_TRY(dmeta.alloc(3));
    const long long meta[3] = {0, L, 0};
    HIP_TRY(hipMemcpy(din.p, in, (size_t)L * sizeof(short), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dmask.p, mask, (size_t)F * 64 * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dmeta.p, meta, sizeof meta, hipMemcpyHostToDevice));
    if (sea_resynth64_batch(din.p, dout.p, dmeta.p, dmeta.p + 1, dmask.p, dmeta.p + 2, dinter.p, nullptr, 1,
                            binary, nullptr))
        return 1;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, dout.p, (size_t)L * sizeof(short), hipMemcpyDeviceToHost));
    return 0;
}

int sea_subband64_batch(const short *d_in, short *d_out, const long long *d_offsets, const long long *d_lengths,
                        const int *d_order, int n_utt, void *stream)
{
    if (n_utt <= 0) return 0;
    DeviceCtx *c;
    if (ctx(&c)) return 1;
    sea::SubbandArgs a;
    a.in = d_in;
    a.out = d_out;
    a.offsets = d_offsets;
    a.lengths = d_lengths;
    a.order = d_order;
    a.tables = c->gt;
    a.n_utt = n_utt;
    hipLaunchKernelGGL(sea::subband_kernel, dim3(n_utt), dim3(384), 0, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

int sea_subband64(const short *in, long L, short *out)
{
    if (L <= 0) return fail("subband64: L=%ld", L);
    const long long Lp = align8(L);
    DevBuf<short> din, dout;
    DevBuf<long long> dmeta;
    HIP_TRY(din.alloc((size_t)Lp));
    HIP_TRY(dout.alloc((size_t)Lp * 64));
    HIP_TRY(dmeta.alloc(2));
    const long long meta[2] = {0, L};
    HIP_TRY(hipMemcpy(din.p, in, (size_t)L * sizeof(short), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dmeta.p, meta, sizeof meta, hipMemcpyHostToDevice));
    if (sea_subband64_batch(din.p, dout.p, dmeta.p, dmeta.p + 1, nullptr, 1, nullptr)) return 1;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy2D(out, (size_t)L * sizeof(short), dout.p, (size_t)Lp * sizeof(short), (size_t)L * sizeof(short), 64,
                        hipMemcpyDeviceToHost));
    return 0;
}

int sea_irm_target_batch(const short *d_pure64, const short *d_noise64, const long long *d_offsets, const long long *d_lengths,
                         const long long *d_row_offsets, float *d_irm, int window, int n_utt, void *stream)
{
    if (n_utt <= 0) return 0;
    if (window < 0 || window > 2) return fail("irm_target: window %d (0 rectangular, 1 Hamming, 2 Hanning)", window);
    DeviceCtx *c;
    if (ctx(&c)) return 1;
    sea::IrmArgs a;
    a.pure = d_pure64;
    a.noise = d_noise64;
    a.offsets = d_offsets;
    a.lengths = d_lengths;
    a.row_offsets = d_row_offsets;
    a.irm = d_irm;
    a.fft = &c->ns->fft;
    a.n_utt = n_utt;
    a.window = window;
    static const bool dual = [] {
        const char *e = getenv("SEA_IRM_KERNEL");
        return e && !strcmp(e, "dual");
    }();
    if (dual)
        hipLaunchKernelGGL(sea::irm_target_dual_kernel, dim3((unsigned)n_utt * 64u), dim3(64), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(sea::irm_target_kernel, dim3((unsigned)n_utt * 64u), dim3(256), 0, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

int sea_irm_target(const short *pure64, const short *noise64, long L, int window, float *irm)
{
    if (L < 320) return fail("irm_target: L=%ld is shorter than one 320-sample frame", L);
    const long long Lp = align8(L), F = (L - 320) / 160 + 1;
    DevBuf<short> dp, dn;
    DevBuf<float> dirm;
    DevBuf<long long> dmeta;
    HIP_TRY(dp.alloc((size_t)Lp * 64));
    HIP_TRY(dn.alloc((size_t)Lp * 64));
    HIP_TRY(dirm.alloc((size_t)F * 64));
    HIP_TRY(dmeta.alloc(3));
    const long long meta[3] = {0, L, 0};
    HIP_TRY(hipMemcpy2D(dp.p, (size_t)Lp * sizeof(short), pure64, (size_t)L * sizeof(short), (size_t)L * sizeof(short), 64,
                        hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy2D(dn.p, (size_t)Lp * sizeof(short), noise64, (size_t)L * sizeof(short), (size_t)L * sizeof(short), 64,
                        hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dmeta.p, meta, sizeof meta, hipMemcpyHostToDevice));
    if (sea_irm_target_batch(dp.p, dn.p, dmeta.p, dmeta.p + 1, dmeta.p + 2, dirm.p, window, 1, nullptr)) return 1;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(irm, dirm.p, (size_t)F * 64 * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}

int sea_gammatone_filter(const float *input, float *output, int chan, long sigLength)
{
    if (chan < 0 || chan >= 64) return fail("gammatone: channel %d out of range", chan);
    if (sigLength <= 0) return 0;
    DeviceCtx *c;
    if (ctx(&c)) return 1;
    DevBuf<float> din, dout;
    HIP_TRY(din.alloc((size_t)sigLength));
    HIP_TRY(dout.alloc((size_t)sigLength));
    HIP_TRY(hipMemcpy(din.p, input, (size_t)sigLength * sizeof(float), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(sea::gammatone_kernel, dim3(1), dim3(64), 0, nullptr, din.p, dout.p, chan,
                       (long long)sigLength, c->gt);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(output, dout.p, (size_t)sigLength * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}

/* ------------------------------------------------------------------------------------------- */
struct sea_ns_stream {
    float *state = nullptr;  /* kNsStateFloats, device */
    float *h_io = nullptr;   /* pinned, device-visible: 80 in | 80 out | produced flag -- the kernel reads and writes it
                              * over PCIe (324 B per push), so a push is ONE launch and ONE stream wait, no copy calls */
    float *d_io = nullptr;   /* the device's address of h_io */
    hipStream_t stream = nullptr;
    int fresh = 1;
};

sea_ns_stream *sea_ns_stream_alloc(void)
{ /* DoNoiseSupAlloc, NoiseSup.c:859-868: NULL on allocation failure */
    sea_ns_stream *s = new (std::nothrow) sea_ns_stream();
    if (!s) return nullptr;
    if (hipMalloc(&s->state, sea::kNsStateFloats * sizeof(float)) != hipSuccess ||
        hipHostMalloc((void **)&s->h_io, 164 * sizeof(float), hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer((void **)&s->d_io, s->h_io, 0) != hipSuccess ||
        hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess) {
        fail("sea_ns_stream_alloc: device allocation failed");
        sea_ns_stream_delete(s);
        return nullptr;
    }
    return s;
}

void sea_ns_stream_init(sea_ns_stream *s)
{ /* DoNoiseSupInit, NoiseSup.c:884-1009: the next push starts from the initial state */
    if (s) s->fresh = 1;
}

int sea_ns_stream_push(sea_ns_stream *s, const float *in80, float *out80)
{
    DeviceCtx *c;
    if (!s || ctx(&c)) {
        fprintf(stderr, "ERROR:   DoNoiseSup (MI355X): %s\r\n", s ? last_error() : "NULL stream");
        exit(1); /* the reference's DoNoiseSup path ends the process on failure (NoiseSup.c:983-987: exit(0));
                  * a device fault must not look like success to the caller's shell, so the status is non-zero */
    }
    sea::NsStreamArgs a = {};
    a.in = s->d_io;
    a.out = s->d_io + 80;
    a.produced = reinterpret_cast<int *>(s->d_io + 160);
    a.state = s->state;
    a.tables = c->ns;
    a.nframes = 1;
    a.reset = s->fresh;
    memcpy(s->h_io, in80, 80 * sizeof(float));
    hipLaunchKernelGGL(sea::ns_stream_kernel, dim3(1), dim3(64), 0, s->stream, a);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s->stream) != hipSuccess) {
        fprintf(stderr, "ERROR:   DoNoiseSup (MI355X): device failure\r\n");
        exit(1);
    }
    const int produced = *reinterpret_cast<const volatile int *>(s->h_io + 160);
    if (produced) memcpy(out80, s->h_io + 80, 80 * sizeof(float));
    s->fresh = 0;
    return produced;
}

void sea_ns_stream_delete(sea_ns_stream *s)
{
    if (!s) return;
    if (s->state) (void)hipFree(s->state);
    if (s->h_io) (void)hipHostFree(s->h_io);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
}

/* batched form of the above on device pointers: B streams x nframes frames */
int sea_ns_streams_push(const float *d_in, float *d_out, int *d_produced, float *d_state, int n_streams,
                        int nframes, int reset, void *stream)
{
    if (n_streams <= 0 || nframes <= 0) return 0;
    DeviceCtx *c;
    if (ctx(&c)) return 1;
    sea::NsStreamArgs a = {};
    a.in = d_in;
    a.out = d_out;
    a.produced = d_produced;
    a.state = d_state;
    a.tables = c->ns;
    a.nframes = nframes;
    a.reset = reset;
    hipLaunchKernelGGL(sea::ns_stream_kernel, dim3(n_streams), dim3(64), 0, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

int sea_ns_streams_push_fd(const float *d_in, float *d_out, int *d_produced, unsigned char *d_flags,
                           int *d_frame_counter, float *d_state, int n_streams, int nframes, int reset, void *stream)
{
    if (n_streams <= 0 || nframes <= 0) return 0;
    DeviceCtx *c;
    if (ctx(&c)) return 1;
    sea::NsStreamArgs a = {};
    a.in = d_in;
    a.out = d_out;
    a.produced = d_produced;
    a.state = d_state;
    a.tables = c->ns;
    a.nframes = nframes;
    a.reset = reset;
    a.flags = d_flags;
    a.frame_counter = d_frame_counter;
    hipLaunchKernelGGL(sea::ns_stream_fd_kernel, dim3(n_streams), dim3(64), 0, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

int sea_ns_state_floats(void) { return sea::kNsStateFloats; }

/* ---- the 16 k-native variant (ns16k_kernel.hip) ---- */
int sea_ns16k_streams_push(const float *d_in, float *d_out, int *d_produced, unsigned char *d_flags, int *d_frame_counter,
                           float *d_wiener, float *d_state, int n_streams, int nframes, int reset, void *stream)
{
    if (n_streams <= 0 || nframes <= 0) return 0;
    if (!d_in || !d_out || !d_produced || !d_state) return fail("sea_ns16k_streams_push: NULL buffer");
    DeviceCtx *c;
    if (ctx(&c)) return 1;
    sea::Ns16StreamArgs a = {};
    a.in = d_in;
    a.out = d_out;
    a.produced = d_produced;
    a.flags = d_flags;
    a.frame_counter = d_frame_counter;
    a.wiener = d_wiener;
    a.state = d_state;
    a.tables = c->ns16;
    a.nframes = nframes;
    a.reset = reset;
    a.n_streams = n_streams;
    /* SEA_NS16K_KERNEL=single: round 3's one-wavefront-per-stream form (kept for A/B); default: four pipelined waves per
     * stream (ns16k_pipe_kernel.hip).  Same arithmetic, same state blob: a stream may change forms between two pushes. */
    if (sea_ns16k_kernel_form(-1) == 1) {
        constexpr int G = sea::kNs16StreamsPerGroup;
        hipLaunchKernelGGL(sea::ns16k_stream_kernel, dim3((n_streams + G - 1) / G), dim3(64 * G), 0, (hipStream_t)stream, a);
    } else {
        constexpr int G = sea::kNs16PipeStreamsPerGroup;
        hipLaunchKernelGGL(sea::ns16k_pipe_kernel, dim3((n_streams + G - 1) / G), dim3(256 * G), 0, (hipStream_t)stream, a);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

int sea_ns16k_state_floats(void) { return sea::kNs16StateFloats; }

/* the 16 k-native variant's transform, DoGamma and DoGammaIDCT on the device, piece by piece (host pointers): see
 * ns16k_selftest_kernel; fft512_a / _b: [nfft][512], gamma25: [ngain][25], idct9: [ngain][9] (rows 0..8, before the window) */
int sea_selftest_ns16k_pieces(const float *frames512, int nfft, float *fft512_a, float *fft512_b, const float *gains129, int ngain,
                              float *gamma25, float *idct9)
{
    if (nfft < 0 || ngain < 0) return fail("sea_selftest_ns16k_pieces: negative count");
    DeviceCtx *c;
    if (ctx(&c)) return 1;
    DevBuf<float> dfr, da, db, dg, dgam, did;
    HIP_TRY(dfr.alloc((size_t)(nfft > 0 ? nfft : 1) * 512));
    HIP_TRY(da.alloc((size_t)(nfft > 0 ? nfft : 1) * 512));
    HIP_TRY(db.alloc((size_t)(nfft > 0 ? nfft : 1) * 512));
    HIP_TRY(dg.alloc((size_t)(ngain > 0 ? ngain : 1) * 129));
    HIP_TRY(dgam.alloc((size_t)(ngain > 0 ? ngain : 1) * 25));
    HIP_TRY(did.alloc((size_t)(ngain > 0 ? ngain : 1) * 9));
    if (nfft) HIP_TRY(hipMemcpy(dfr.p, frames512, (size_t)nfft * 512 * sizeof(float), hipMemcpyHostToDevice));
    if (ngain) HIP_TRY(hipMemcpy(dg.p, gains129, (size_t)ngain * 129 * sizeof(float), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(sea::ns16k_selftest_kernel, dim3(1), dim3(64), 0, nullptr, c->ns16, dfr.p, nfft, da.p, db.p, dg.p, ngain, dgam.p, did.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    if (nfft) {
        HIP_TRY(hipMemcpy(fft512_a, da.p, (size_t)nfft * 512 * sizeof(float), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(fft512_b, db.p, (size_t)nfft * 512 * sizeof(float), hipMemcpyDeviceToHost));
    }
    if (ngain) {
        HIP_TRY(hipMemcpy(gamma25, dgam.p, (size_t)ngain * 25 * sizeof(float), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(idct9, did.p, (size_t)ngain * 9 * sizeof(float), hipMemcpyDeviceToHost));
    }
    return 0;
}

/* 0: four pipelined waves per stream (default); 1: one wave per stream (round 3's form; SEA_NS16K_KERNEL=single sets it as
 * the initial value).  form < 0 only reads.  Returns the previous form. */
int sea_ns16k_kernel_form(int form)
{
    static std::atomic<int> g_form{[] {
        const char *e = getenv("SEA_NS16K_KERNEL");
        return (e && !strcmp(e, "single")) ? 1 : 0;
    }()};
    const int prev = g_form.load();
    if (form == 0 || form == 1) g_form.store(form);
    return prev;
}

int sea_ns16k_tables_host(float *sigWindow480, float *irWindow17, int *gammaStart25, float *gamma25x128, float *idct25x25)
{
    sea_ns16k_plain_tables(sigWindow480, irWindow17, gammaStart25, gamma25x128, idct25x25);
    return 0;
}

/* ------------------------------------------------------------------------------------------- */
int sea_selftest_pi4(unsigned long long *n_mismatch)
{
    DeviceCtx *c;
    if (ctx(&c)) return 1;
    DevBuf<unsigned long long> d;
    HIP_TRY(d.alloc(1));
    HIP_TRY(hipMemset(d.p, 0, sizeof(unsigned long long)));
    hipLaunchKernelGGL(sea::selftest_pi4_kernel, dim3(4096), dim3(256), 0, nullptr, d.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(n_mismatch, d.p, sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return 0;
}

int sea_selftest_div(unsigned long long *out2)
{
    DeviceCtx *c;
    if (ctx(&c)) return 1;
    DevBuf<unsigned long long> d;
    HIP_TRY(d.alloc(2));
    HIP_TRY(hipMemset(d.p, 0, 2 * sizeof(unsigned long long)));
    hipLaunchKernelGGL(sea::selftest_div_kernel, dim3(4096), dim3(256), 0, nullptr, c->gt, d.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out2, d.p, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return 0;
}

int sea_selftest_nsdiv(unsigned long long *out4)
{
    DeviceCtx *c;
    if (ctx(&c)) return 1;
    DevBuf<unsigned long long> d;
    HIP_TRY(d.alloc(4));
    HIP_TRY(hipMemset(d.p, 0, 4 * sizeof(unsigned long long)));
    hipLaunchKernelGGL(sea::selftest_nsdiv_kernel, dim3(4096), dim3(256), 0, nullptr, d.p, 1024);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out4, d.p, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return 0;
}

int sea_selftest_dc(const float *dif, const float *y0, float *out, int *fellback, int ncases)
{
    if (ncases <= 0) return 0;
    DeviceCtx *c;
    if (ctx(&c)) return 1;
    DevBuf<float> dd, dy, dout;
    DevBuf<int> df;
    HIP_TRY(dd.alloc((size_t)ncases * 80));
    HIP_TRY(dy.alloc(ncases));
    HIP_TRY(dout.alloc((size_t)ncases * 80));
    HIP_TRY(df.alloc(ncases));
    HIP_TRY(hipMemcpy(dd.p, dif, (size_t)ncases * 80 * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dy.p, y0, ncases * sizeof(float), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(sea::selftest_dc_kernel, dim3(ncases < 1024 ? ncases : 1024), dim3(64), 0, nullptr, dd.p, dy.p,
                       dout.p, df.p, ncases);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, dout.p, (size_t)ncases * 80 * sizeof(float), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(fellback, df.p, ncases * sizeof(int), hipMemcpyDeviceToHost));
    return 0;
}

int sea_selftest_log(const float *x, double *ln_out, int n)
{
    if (n <= 0) return 0;
    DeviceCtx *c;
    if (ctx(&c)) return 1;
    DevBuf<float> dx;
    DevBuf<double> dout;
    HIP_TRY(dx.alloc(n));
    HIP_TRY(dout.alloc(n));
    HIP_TRY(hipMemcpy(dx.p, x, (size_t)n * sizeof(float), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(sea::selftest_log_kernel, dim3((n + 255) / 256), dim3(256), 0, nullptr, dx.p, dout.p, n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(ln_out, dout.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}

int sea_selftest_log_dd(const double *x, double *hi, double *lo, int n)
{
    if (n <= 0) return 0;
    DeviceCtx *c;
    if (ctx(&c)) return 1;
    DevBuf<double> dx, dhi, dlo;
    HIP_TRY(dx.alloc(n));
    HIP_TRY(dhi.alloc(n));
    HIP_TRY(dlo.alloc(n));
    HIP_TRY(hipMemcpy(dx.p, x, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(sea::selftest_log_dd_kernel, dim3((n + 255) / 256), dim3(256), 0, nullptr, dx.p, dhi.p, dlo.p, n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(hi, dhi.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(lo, dlo.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}

int sea_selftest_log_sites(const float *x, float *site1, float *site2, int n)
{
    if (n <= 0) return 0;
    DeviceCtx *c;
    if (ctx(&c)) return 1;
    DevBuf<float> dx, d1, d2;
    HIP_TRY(dx.alloc(n));
    HIP_TRY(d1.alloc(n));
    HIP_TRY(d2.alloc(n));
    HIP_TRY(hipMemcpy(dx.p, x, (size_t)n * sizeof(float), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(sea::selftest_log_sites_kernel, dim3((n + 255) / 256), dim3(256), 0, nullptr, dx.p, d1.p, d2.p, n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(site1, d1.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(site2, d2.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}

int sea_selftest_log_guard(int site, unsigned long long *stats8, float *hits3, int cap)
{
    if (site != 1 && site != 2) return fail("selftest_log_guard: site must be 1 or 2");
    if (cap < 0) cap = 0;
    DeviceCtx *c;
    if (ctx(&c)) return 1;
    DevBuf<unsigned long long> ds;
    DevBuf<float> dh;
    HIP_TRY(ds.alloc(8));
    HIP_TRY(dh.alloc((size_t)3 * (cap > 0 ? cap : 1)));
    HIP_TRY(hipMemset(ds.p, 0, 8 * sizeof(unsigned long long)));
    HIP_TRY(hipMemset(dh.p, 0, (size_t)3 * (cap > 0 ? cap : 1) * sizeof(float)));
    hipLaunchKernelGGL(sea::selftest_log_guard_kernel, dim3(4096), dim3(256), 0, nullptr, site, ds.p, dh.p, cap);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(stats8, ds.p, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    if (cap > 0 && hits3) HIP_TRY(hipMemcpy(hits3, dh.p, (size_t)3 * cap * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}

} // extern "C"
