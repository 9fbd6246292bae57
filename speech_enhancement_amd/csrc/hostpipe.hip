/*
 * hostpipe.hip -- the host-buffer batch entry points as a copy / compute pipeline.
 *
 *   sea_denoise_utterances   what etsi/cpp/main.cpp:43-67 does per file (read, etsi_denoise, write) and
 *                            function/20141106_speech_enhancement/aurora_speech_enhancement/aurora_speech_enhancement.cpp:25-80
 *                            does from a thread pool, for a whole list of utterances in host memory
 *   sea_resynth_utterances   the same for resynth() (resyth_64sub_ori/cpp/main.cpp:84-145)
 *
 * PCIe is the slow link of these entry points (160 B per NoiseSup frame each way against 320 B of HBM traffic
 * on the device), so the list is cut into pieces that travel down a pipeline:
 *
 *   pool threads   pack piece k+1 into pinned staging            | unpack piece k-1 into the caller's buffers
 *   copy engines   H2D piece k+1                                 | D2H piece k-1
 *   device         kernel over piece k
 *
 * NoiseSup (sea_denoise_utterances): the pieces are TIME SLICES of the whole list -- frames [B_k, B_k+1) of every
 * utterance, one launch per slice, the recursion carried in a state blob per utterance (denoise_utterances_slices).
 * A launch over whole utterances lasts as long as its longest utterance's chain of frames whatever its size; a slice lasts
 * as long as its own frames.  Resynthesis (sea_resynth_utterances; its second pass runs backwards over the utterance) and
 * SEA_HOST_MODE=chunks: the pieces are chunks of whole utterances, sorted longest first, one launch per chunk on its own
 * part of ONE device buffer.  Results do not depend on either cut.
 * Staging, device buffers, streams and events are grow-only and belong to the calling host thread (the reference's
 * batch tool calls etsi_denoise from N threads); the packing threads are one pool per device.
 */
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "capi_internal.h"

using namespace sea_capi;

namespace {

/* ---- packing threads ------------------------------------------------------------------------------ */
/* One pool PER DEVICE (round 4): one device keeps eight packing threads busy, so a process-wide pool of
 * eight capped the file drivers' device threads at one device's worth of packing on an 8-GPU node.  A pool is created by
 * the first host thread that runs a pipeline on that device -- never from a static constructor, and without a HIP call:
 * the device index comes from the caller's PipeWs -- and host threads that share a device (SEA_DEVICES=3 on one GPU) share
 * its pool.  SEA_HOST_THREADS = threads per device (default min(8, (cores - 1) / devices in use so far), at least 2). */
class Pool {
  public:
    static constexpr int kMaxDev = 64;
    static Pool &get(int device)
    {
        static std::mutex m;
        static Pool *pools[kMaxDev] = {};
        static int n_pools = 0;
        const int d = (device >= 0 && device < kMaxDev) ? device : 0;
        std::lock_guard<std::mutex> lk(m);
        if (!pools[d]) pools[d] = new Pool(++n_pools); /* lives until process exit: its threads block on the queue */
        return *pools[d];
    }
    int size() const { return (int)th_.size(); }
    void submit(std::function<void()> f)
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            q_.push_back(std::move(f));
        }
        cv_.notify_one();
    }

  private:
    explicit Pool(int n_pools_now)
    {
        int n = 0;
        if (const char *e = getenv("SEA_HOST_THREADS")) n = atoi(e);
        if (n <= 0) {
            const unsigned hw = std::thread::hardware_concurrency();
            n = hw > 2 ? (int)std::min(8u, std::max(2u, (hw - 1) / (unsigned)n_pools_now)) : 1;
        }
        for (int i = 0; i < n; ++i) th_.emplace_back([this] { run(); });
    }
    ~Pool()
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto &t : th_) t.join();
    }
    void run()
    {
        for (;;) {
            std::function<void()> f;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [this] { return stop_ || !q_.empty(); });
                if (q_.empty()) return; /* stop requested and nothing left */
                f = std::move(q_.front());
                q_.pop_front();
            }
            f();
        }
    }
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable cv_;
    std::deque<std::function<void()>> q_;
    bool stop_ = false;
};

/* memcpy for the packing threads: the destination (pinned staging the copy engine reads next, or the caller's output
 * buffer) is not read again by this core, so it is written with non-temporal stores -- no read-for-ownership of the
 * destination lines, a third less memory traffic than memcpy's cached stores at these sizes (pieces of ~16 KB, far
 * below glibc's non-temporal threshold).  SEA_HOST_NT=0 switches back to memcpy. */
#if defined(__x86_64__)
#include <emmintrin.h>
inline bool use_nt()
{
    static const bool v = [] {
        const char *e = getenv("SEA_HOST_NT");
        return !(e && e[0] == '0');
    }();
    return v;
}
inline void copy_stream(void *dst, const void *src, size_t n)
{
    if (n < 256 || !use_nt()) {
        memcpy(dst, src, n);
        return;
    }
    char *d = (char *)dst;
    const char *s = (const char *)src;
    const size_t head = (16 - ((uintptr_t)d & 15)) & 15;
    if (head) {
        memcpy(d, s, head);
        d += head;
        s += head;
        n -= head;
    }
    const size_t blocks = n / 64;
    for (size_t i = 0; i < blocks; ++i, d += 64, s += 64) {
        const __m128i a = _mm_loadu_si128((const __m128i *)s), b = _mm_loadu_si128((const __m128i *)(s + 16));
        const __m128i c = _mm_loadu_si128((const __m128i *)(s + 32)), e = _mm_loadu_si128((const __m128i *)(s + 48));
        _mm_stream_si128((__m128i *)d, a);
        _mm_stream_si128((__m128i *)(d + 16), b);
        _mm_stream_si128((__m128i *)(d + 32), c);
        _mm_stream_si128((__m128i *)(d + 48), e);
    }
    n -= blocks * 64;
    if (n) memcpy(d, s, n);
}
inline void copy_fence() { _mm_sfence(); }
#else
inline void copy_stream(void *dst, const void *src, size_t n) { memcpy(dst, src, n); }
inline void copy_fence() {}
#endif

/* counts outstanding tasks; wait() returns when all are done */
struct Latch {
    std::mutex m;
    std::condition_variable cv;
    int pending = 0;
    void add(int n)
    {
        std::lock_guard<std::mutex> lk(m);
        pending += n;
    }
    void done()
    {
        std::lock_guard<std::mutex> lk(m);
        if (--pending == 0) cv.notify_all();
    }
    bool ready()
    {
        std::lock_guard<std::mutex> lk(m);
        return pending == 0;
    }
    void wait()
    {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [this] { return pending == 0; });
    }
    template <class Rep, class Period>
    bool wait_for(std::chrono::duration<Rep, Period> d)
    {
        std::unique_lock<std::mutex> lk(m);
        return cv.wait_for(lk, d, [this] { return pending == 0; });
    }
};

/* copies [j0, j1) of a job list on the pool (or inline when the job is small), cut into ~1 MB tasks */
template <class Copy>
void run_copies(Pool &pool, int j0, int j1, const long long *bytes_prefix, Latch *chunk, Latch *all, bool inline_, Copy copy)
{
    if (j0 >= j1) return;
    if (inline_) {
        for (int j = j0; j < j1; ++j) copy(j);
        copy_fence();
        return;
    }
    const long long kTask = 1 << 20;
    int a = j0;
    while (a < j1) {
        int b = a + 1;
        while (b < j1 && bytes_prefix[b] - bytes_prefix[a] < kTask) ++b;
        chunk->add(1);
        all->add(1);
        pool.submit([=] {
            for (int j = a; j < b; ++j) copy(j);
            copy_fence(); /* non-temporal stores are visible to the copy engine / the caller before the task counts as done */
            chunk->done();
            all->done();
        });
        a = b;
    }
}

/* ---- per-thread workspace -------------------------------------------------------------------------- */
constexpr int kMaxStreams = 8;
int n_streams()
{
    static const int n = [] {
        const char *e = getenv("SEA_HOST_STREAMS");
        const int v = e ? atoi(e) : 4;
        return v < 1 ? 1 : (v > kMaxStreams ? kMaxStreams : v);
    }();
    return n;
}
#define kStreams n_streams()
constexpr int kMaxChunks = 64;

template <class T>
struct Grow { /* one pinned + one device buffer of the same size, grow-only */
    T *h = nullptr, *d = nullptr;
    size_t cap = 0;
    void release()
    {
        if (h) (void)hipHostFree(h);
        if (d) (void)hipFree(d);
        h = d = nullptr;
        cap = 0;
    }
    hipError_t ensure(size_t n)
    {
        if (n <= cap) return hipSuccess;
        release();
        const size_t want = n + n / 4 + 4096;
        hipError_t e = hipHostMalloc((void **)&h, want * sizeof(T), hipHostMallocDefault); /* (write-combined staging: no gain) */
        if (e == hipSuccess) e = hipMalloc((void **)&d, want * sizeof(T) + 16);
        if (e != hipSuccess) {
            release();
            return e;
        }
        cap = want;
        return hipSuccess;
    }
};

struct PipeWs {
    Grow<short> in, out;
    Grow<float> mask;      /* resynth: mask rows */
    Grow<long long> meta;  /* offsets | lengths | mask offsets, in launch order */
    Grow<float> f32, ceps; /* sea_denoise_ceps_utterances: the float NoiseSup stream (device side used only), the cepstra */
    Grow<int> ints;        /* the same: first_out | n_ceps | order */
    float *d_inter = nullptr; /* resynth scratch */
    size_t inter_bytes = 0;
    float *d_state = nullptr; /* NoiseSup in time slices: the recursion per utterance between two launches */
    size_t state_floats = 0;
    hipStream_t stream[kMaxStreams] = {};
    hipEvent_t ev_meta = nullptr, ev_done[kMaxChunks] = {}, ev_kernel[kMaxChunks] = {}, ev_h2d[kMaxChunks] = {};
    int device = -1;
    ~PipeWs() { release(); }
    void release()
    {
        int cur = -1;
        const bool sw = device >= 0 && hipGetDevice(&cur) == hipSuccess && cur != device;
        if (sw) (void)hipSetDevice(device); /* the buffers belong to `device`, whatever the thread uses now */
        in.release();
        out.release();
        mask.release();
        meta.release();
        f32.release();
        ceps.release();
        ints.release();
        if (d_inter) (void)hipFree(d_inter);
        d_inter = nullptr;
        inter_bytes = 0;
        if (d_state) (void)hipFree(d_state);
        d_state = nullptr;
        state_floats = 0;
        for (auto &s : stream) {
            if (s) (void)hipStreamDestroy(s);
            s = nullptr;
        }
        if (ev_meta) (void)hipEventDestroy(ev_meta);
        ev_meta = nullptr;
        for (auto &e : ev_done) {
            if (e) (void)hipEventDestroy(e);
            e = nullptr;
        }
        for (auto &e : ev_kernel) {
            if (e) (void)hipEventDestroy(e);
            e = nullptr;
        }
        for (auto &e : ev_h2d) {
            if (e) (void)hipEventDestroy(e);
            e = nullptr;
        }
        if (sw) (void)hipSetDevice(cur);
        device = -1;
    }
    hipError_t bind()
    {
        int dev = -1;
        hipError_t e;
        if ((e = hipGetDevice(&dev)) != hipSuccess) return e;
        if (dev != device) { /* the thread moved to another device: nothing of the old one is usable */
            if (device >= 0) release();
            device = dev;
        }
        for (int i = 0; i < kStreams; ++i)
            if (!stream[i] && (e = hipStreamCreateWithFlags(&stream[i], hipStreamNonBlocking)) != hipSuccess) return e;
        if (!ev_meta && (e = hipEventCreateWithFlags(&ev_meta, hipEventDisableTiming)) != hipSuccess) return e;
        for (auto &ev : ev_done)
            if (!ev && (e = hipEventCreateWithFlags(&ev, hipEventDisableTiming)) != hipSuccess) return e;
        for (auto &ev : ev_kernel)
            if (!ev && (e = hipEventCreateWithFlags(&ev, hipEventDisableTiming)) != hipSuccess) return e;
        for (auto &ev : ev_h2d)
            if (!ev && (e = hipEventCreateWithFlags(&ev, hipEventDisableTiming)) != hipSuccess) return e;
        return hipSuccess;
    }
    hipError_t ensure_inter(size_t bytes)
    {
        if (bytes <= inter_bytes) return hipSuccess;
        if (d_inter) (void)hipFree(d_inter);
        d_inter = nullptr;
        inter_bytes = 0;
        hipError_t e = hipMalloc((void **)&d_inter, bytes + 16);
        if (e == hipSuccess) inter_bytes = bytes;
        return e;
    }
    hipError_t ensure_state(size_t floats)
    {
        if (floats <= state_floats) return hipSuccess;
        if (d_state) (void)hipFree(d_state);
        d_state = nullptr;
        state_floats = 0;
        const size_t want = floats + floats / 4;
        hipError_t e = hipMalloc((void **)&d_state, want * sizeof(float));
        if (e == hipSuccess) state_floats = want;
        return e;
    }
    void drain()
    {
        for (auto &s : stream)
            if (s) (void)hipStreamSynchronize(s);
    }
};
thread_local PipeWs t_ws;

/* On every exit path: no pool task may outlive the buffers it copies from / to, no stream may still be copying. */
struct Scope {
    Latch all;
    PipeWs *ws;
    bool ok = false;
    explicit Scope(PipeWs *w) : ws(w) {}
    ~Scope()
    {
        all.wait();
        if (!ok) ws->drain();
    }
};

/* hipEventQuery as a three-way answer: 0 and *done set when the event has completed, 0 and *done clear while the work
 * before it is still running (hipErrorNotReady), 1 after fail() for ANYTHING else -- a failed kernel or copy, an invalid
 * event.  The pipelines below poll events; treating every non-success as "not yet" would spin for ever on a fault instead
 * of returning the reference's fault code.  `query` is hipEventQuery except in the state-machine unit test. */
typedef hipError_t (*EventQueryFn)(hipEvent_t);
EventQueryFn g_event_query = hipEventQuery;
int ev_ready(hipEvent_t e, bool *done, const char *what)
{
    const hipError_t r = g_event_query(e);
    *done = (r == hipSuccess);
    if (r == hipSuccess || r == hipErrorNotReady) return 0;
    (void)hipGetLastError(); /* the sticky copy of an asynchronous fault must not fail the caller's next, unrelated call */
    return fail("%s: %s", what, hipGetErrorString(r));
}
/* 1 done, 0 still running, -1 failed (fail() has the message) */
int evq(hipEvent_t e, const char *what)
{
    bool done = false;
    if (ev_ready(e, &done, what)) return -1;
    return done ? 1 : 0;
}

double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

long long env_mb(const char *name, long long dflt)
{
    if (const char *e = getenv(name)) {
        const long long v = atoll(e);
        if (v > 0) return v;
    }
    return dflt;
}

/* cut the (sorted) list into at most max_chunks contiguous chunks of about `target` weight each */
std::vector<int> cut_chunks(const std::vector<long long> &weight_prefix, int n, long long target, int max_chunks)
{
    const long long total = weight_prefix[n];
    int nchunk = (int)std::min<long long>(max_chunks, std::max<long long>(1, (total + target - 1) / target));
    nchunk = std::min(nchunk, n);
    std::vector<int> cuts(1, 0);
    for (int k = 1; k < nchunk; ++k) {
        const long long want = total * k / nchunk;
        int j = (int)(std::lower_bound(weight_prefix.begin(), weight_prefix.begin() + n + 1, want) - weight_prefix.begin());
        j = std::max(j, cuts.back() + 1);
        if (j >= n) break;
        cuts.push_back(j);
    }
    cuts.push_back(n);
    return cuts;
}

} // namespace

extern "C" {

/* Test hook: from the nth event query of this process on (1-based; 0 switches the hook off) every query reports a device
 * fault instead of asking the runtime.  tests/test_gpu_parity.py::test_host_pipeline_returns_fault_on_event_error checks
 * that all three pipelines then return 1 with their streams drained and work again afterwards. */
static std::atomic<long long> g_fault_after{0}, g_queries{0};
static hipError_t faulty_query(hipEvent_t e)
{
    const long long n = g_fault_after.load();
    if (n > 0 && ++g_queries >= n) return hipErrorLaunchFailure;
    return hipEventQuery(e);
}
int sea_selftest_hostpipe_fault(long long nth_query)
{
    g_queries = 0;
    g_fault_after = nth_query;
    g_event_query = nth_query > 0 ? faulty_query : hipEventQuery;
    return 0;
}

int sea_host_threads(void)
{ /* of the calling thread's current device */
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    return Pool::get(dev).size();
}

/* ---------------------------------------------------------------------------------------------------- */
static int denoise_utterances_slices(const short *const *in, short *const *out, const long *lengths, int n_utt);

/* SEA_HOST_MODE=chunks: the list cut into chunks of whole utterances, one launch per chunk on its own stream (round 3's
 * first pipeline, kept for A/B).  Default: time slices (denoise_utterances_slices below). */
static int denoise_utterances_chunks(const short *const *in, short *const *out, const long *lengths, int n_utt)
{
    if (n_utt <= 0) return 0;
    DeviceCtx *dc;
    if (ctx(&dc)) return 1;
    for (int u = 0; u < n_utt; ++u)
        if (lengths[u] < 0) return fail("negative length for utterance %d", u);
    /* launch order: longest first */
    std::vector<int> idx(n_utt);
    for (int i = 0; i < n_utt; ++i) idx[i] = i;
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return lengths[a] > lengths[b]; });
    std::vector<long long> pre(n_utt + 1, 0); /* packed offset of sorted position j, in samples */
    for (int j = 0; j < n_utt; ++j) pre[j + 1] = pre[j] + align8(lengths[idx[j]]);
    const long long total = pre[n_utt];
    if (total == 0) return 0;

    PipeWs &w = t_ws;
    HIP_TRY(w.bind());
    Pool &pool = Pool::get(w.device);
    HIP_TRY(w.in.ensure((size_t)total));
    HIP_TRY(w.out.ensure((size_t)total));
    HIP_TRY(w.meta.ensure(2 * (size_t)n_utt));
    long long *offs = w.meta.h, *lens = w.meta.h + n_utt;
    for (int j = 0; j < n_utt; ++j) {
        offs[j] = pre[j];
        lens[j] = lengths[idx[j]];
    }
    /* Chunks (small lists: one chunk, packed by the calling thread).  Default: a third of the list each -- measured on
     * the 1024-utterance bench corpus (tools/host_sweep.sh): kernels launched on different streams do overlap, but a
     * launch runs as long as its longest utterance's chain of frames whatever its size and several small launches side
     * by side fill the chip worse than one large one, so few large chunks win (3 chunks 6.3 ms, 6 chunks 7.4 ms,
     * 16 chunks 16.5 ms; one chunk = no overlap 10.4 ms).  SEA_HOST_CHUNK_MB sets the chunk size in MB of int16. */
    const long long chunk_samples = getenv("SEA_HOST_CHUNK_MB") ? env_mb("SEA_HOST_CHUNK_MB", 24) * (1 << 20) / 2
                                                                : std::max<long long>((total + 2) / 3, 1 << 20);
    const bool small = total * 2 < (2 << 20) || n_utt == 1 || pool.size() <= 1;
    const std::vector<int> cuts = small ? std::vector<int>{0, n_utt} : cut_chunks(pre, n_utt, chunk_samples, kMaxChunks);
    const int nchunk = (int)cuts.size() - 1;
    const int form = ns_pick_form(n_utt, dc->n_cu);
    /* SEA_HOST_ZEROCOPY=o (experiment kept for other hosts): the kernel's stores go straight to the pinned output
     * staging over PCIe (160 B per frame and utterance), no D2H phase.  Measured here: such stores reach ~24 GB/s
     * against 52 GB/s of a D2H copy and slow the kernel down by as much as the copy phase they save. */
    const char *zc = getenv("SEA_HOST_ZEROCOPY");
    const bool zc_out = zc ? (strchr(zc, 'o') != nullptr) : false;
    short *d_out = w.out.d;
    if (zc_out) HIP_TRY(hipHostGetDevicePointer((void **)&d_out, w.out.h, 0));
    /* issue order of the chunks (chunk 0 holds the longest utterances; SEA_HOST_ORDER=desc|asc|mix, measured within
     * 5 % of each other, longest first best) */
    std::vector<int> seq(nchunk);
    {
        const char *o = getenv("SEA_HOST_ORDER");
        const int mode = (o && !strcmp(o, "mix")) ? 2 : ((o && !strcmp(o, "asc")) ? 1 : 0); /* default: longest first */
        for (int i = 0, lo = 0, hi = nchunk - 1; i < nchunk; ++i)
            seq[i] = mode == 0 ? i : (mode == 1 ? nchunk - 1 - i : ((i & 1) ? hi-- : lo++));
    }

    std::vector<Latch> packed(nchunk), unpacked(nchunk);
    Scope scope(&w);
    short *h_in = w.in.h, *h_out = w.out.h;
    std::vector<long long> bytes_pre(pre); /* task sizing */
    for (auto &b : bytes_pre) b *= 2;
    const long long *bp = bytes_pre.data();
    const int *ix = idx.data();
    for (int i = 0; i < nchunk; ++i) {
        const int k = seq[i];
        run_copies(pool, cuts[k], cuts[k + 1], bp, &packed[k], &scope.all, small, [=](int j) {
            const long long L = lens[j];
            copy_stream(h_in + offs[j], in[ix[j]], (size_t)L * sizeof(short));
            const long long pad = align8(L) - L;
            if (pad) memset(h_in + offs[j] + L, 0, (size_t)pad * sizeof(short));
        });
    }
    HIP_TRY(hipMemcpyAsync(w.meta.d, w.meta.h, 2 * (size_t)n_utt * sizeof(long long), hipMemcpyHostToDevice, w.stream[0]));
    HIP_TRY(hipEventRecord(w.ev_meta, w.stream[0]));

    const bool trace = getenv("SEA_HOST_TRACE") != nullptr;
    /* timing experiments only: from the second call on skip the PCIe copies (the device works on the first call's input) */
    static std::atomic<int> n_calls{0};
    const bool nocopy = getenv("SEA_HOST_DEBUG_NOCOPY") != nullptr && n_calls++ > 0;
    const double t0 = now_ms();
    std::vector<hipEvent_t> tev; /* trace only: base, then (start, h2d, kernel, d2h) per chunk */
    if (trace) {
        tev.resize(1 + 4 * nchunk);
        for (auto &e : tev) HIP_TRY(hipEventCreate(&e));
        HIP_TRY(hipEventRecord(tev[0], w.stream[0]));
    }
    /* Event-driven: a copy is only handed to the runtime when everything it depends on has finished.  The copy
     * engines' queues are in order, so a D2H copy queued behind its still-running kernel would hold up the H2D copies
     * of later chunks (measured: chunk k+2's upload waited for chunk k's kernel). */
    enum { kWaitPack, kComputing, kDownloading, kUnpacking };
    std::vector<char> state(nchunk, kWaitPack);
    int next = 0, finished = 0;
    while (finished < nchunk) {
        bool progressed = false;
        int r = 0;
        /* one upload at a time: concurrent uploads share the link, and the first kernel should start as early as it can */
        if (next < nchunk && packed[seq[next]].ready() &&
            (next == 0 || (r = evq(w.ev_h2d[seq[next - 1]], "hostpipe: upload event")) != 0)) {
            if (r < 0) return 1; /* Scope waits for the pool and drains the streams */
            const int k = seq[next], i = next++;
            hipStream_t s = w.stream[i % kStreams];
            const int u0 = cuts[k], n = cuts[k + 1] - cuts[k];
            const long long o0 = pre[u0], cnt = pre[cuts[k + 1]] - o0;
            if (trace) fprintf(stderr, "[hostpipe] %7.3f ms  chunk %d packed (%d utterances, %.1f MB), issuing\n", now_ms() - t0, k, n, cnt * 2 / 1048576.0);
            if (i > 0 && i < kStreams) HIP_TRY(hipStreamWaitEvent(s, w.ev_meta, 0));
            if (trace) HIP_TRY(hipEventRecord(tev[1 + 4 * k], s));
            if (!nocopy)
                HIP_TRY(hipMemcpyAsync(w.in.d + o0, h_in + o0, (size_t)cnt * sizeof(short), hipMemcpyHostToDevice, s));
            HIP_TRY(hipEventRecord(w.ev_h2d[k], s));
            if (trace) HIP_TRY(hipEventRecord(tev[2 + 4 * k], s));
            sea::NsBatchArgs a = {};
            a.in = w.in.d;
            a.out = d_out;
            a.offsets = w.meta.d + u0;
            a.lengths = w.meta.d + n_utt + u0;
            a.tables = dc->ns;
            a.n_utt = n;
            if (form == 2 && n_utt > dc->n_cu) { /* issue priority by rows of the whole list, as the one-launch form has it */
                a.prio_row = dc->n_cu;
                a.prio_base = u0 / dc->n_cu;
            }
            if (ns_launch(a, form, s)) return 1;
            if (trace) HIP_TRY(hipEventRecord(tev[3 + 4 * k], s));
            HIP_TRY(hipEventRecord(w.ev_kernel[k], s));
            state[k] = kComputing;
            progressed = true;
        }
        for (int i = 0; i < next; ++i) {
            const int k = seq[i];
            if (state[k] == kComputing && (r = evq(w.ev_kernel[k], "hostpipe: kernel event")) != 0) {
                if (r < 0) return 1; /* Scope waits for the pool and drains the streams */
                hipStream_t s = w.stream[i % kStreams];
                const long long o0 = pre[cuts[k]], cnt = pre[cuts[k + 1]] - o0;
                if (!zc_out && !nocopy) HIP_TRY(hipMemcpyAsync(h_out + o0, w.out.d + o0, (size_t)cnt * sizeof(short), hipMemcpyDeviceToHost, s));
                if (trace) HIP_TRY(hipEventRecord(tev[4 + 4 * k], s));
                HIP_TRY(hipEventRecord(w.ev_done[k], s));
                state[k] = kDownloading;
                progressed = true;
            }
            if (state[k] == kDownloading && (r = evq(w.ev_done[k], "hostpipe: download event")) != 0) {
                if (r < 0) return 1; /* Scope waits for the pool and drains the streams */
                if (trace) fprintf(stderr, "[hostpipe] %7.3f ms  chunk %d done on the device, unpacking\n", now_ms() - t0, k);
                run_copies(pool, cuts[k], cuts[k + 1], bp, &unpacked[k], &scope.all, small, [=](int j) {
                    /* the trailing partial frame stays untouched */
                    copy_stream(out[ix[j]], h_out + offs[j], (size_t)(lens[j] / 80 * 80) * sizeof(short));
                });
                state[k] = kUnpacking;
                finished++;
                progressed = true;
            }
        }
        if (!progressed) {
            if (next < nchunk && !packed[seq[next]].ready()) packed[seq[next]].wait_for(std::chrono::microseconds(30));
            else std::this_thread::sleep_for(std::chrono::microseconds(20));
        }
    }
    scope.all.wait();
    if (trace) {
        fprintf(stderr, "[hostpipe] %7.3f ms  all unpacked\n", now_ms() - t0);
        for (int k = 0; k < nchunk; ++k) {
            float t[4];
            for (int q = 0; q < 4; ++q) (void)hipEventElapsedTime(&t[q], tev[0], tev[1 + 4 * k + q]);
            fprintf(stderr, "[hostpipe] device clock, chunk %d: start %.3f  h2d done %.3f  kernel done %.3f  d2h done %.3f ms\n", k,
                    t[0], t[1], t[2], t[3]);
        }
        for (auto &e : tev) (void)hipEventDestroy(e);
    }
    scope.ok = true;
    return 0;
}

int sea_denoise_utterances(const short *const *in, short *const *out, const long *lengths, int n_utt)
{
    const char *m = getenv("SEA_HOST_MODE");
    if ((m && !strcmp(m, "chunks")) || kStreams < 3) return denoise_utterances_chunks(in, out, lengths, n_utt);
    return denoise_utterances_slices(in, out, lengths, n_utt);
}

/* The list cut along the TIME axis: slice k holds the frames [B_k, B_k+1) of every utterance that has them, packed like a
 * batch of its own, and is one launch over all those utterances (sea_ns_denoise_batch_slice: the recursion travels in a
 * state blob per utterance).  A launch then lasts as long as ITS frames take -- a chunk of whole utterances lasts as long
 * as its longest utterance, whatever its size -- so the pipeline
 *     pool threads  pack slice k+1            | unpack slice k-1
 *     copy engines  H2D slice k+1             | D2H slice k-1
 *     device        kernel over slice k
 * has short fill and drain phases: the first kernel starts after 1/K of the upload, the last download carries 1/K of the
 * output.  Slice boundaries equalise the slices' sample counts (SEA_HOST_SLICES of them, default 8: measured on the
 * 1024-utterance bench corpus, tools/host_slices_sweep.sh, alternating on one box -- chunks of whole utterances 7.4 ms,
 * 4 slices 6.2-6.5, 6: 5.9-6.0, 8: 4.8-5.7, 10: 5.6-5.7, 12: 5.3-5.8, 16: 5.8 ms; what is left is the pool's memcpy
 * rate: 2 x 131 MB packed and unpacked in that time).
 * Three streams, one per job -- uploads, kernels (in order: slice k needs slice k - 1's state), downloads: with one
 * stream per slice, reused round robin, a slice's download ended up queued behind a later slice's kernel (calls of
 * 13-20 ms among the 6 ms ones). */
static int denoise_utterances_slices(const short *const *in, short *const *out, const long *lengths, int n_utt)
{
    if (n_utt <= 0) return 0;
    DeviceCtx *dc;
    if (ctx(&dc)) return 1;
    for (int u = 0; u < n_utt; ++u)
        if (lengths[u] < 0) return fail("negative length for utterance %d", u);
    std::vector<int> idx(n_utt);
    for (int i = 0; i < n_utt; ++i) idx[i] = i;
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return lengths[a] > lengths[b]; });
    std::vector<long long> nfr(n_utt);
    long long total_fr = 0;
    for (int j = 0; j < n_utt; ++j) total_fr += (nfr[j] = lengths[idx[j]] / 80);
    if (total_fr == 0) return 0;
    const long long max_fr = nfr[0];
    const long long total = total_fr * 80;
    PipeWs &w = t_ws;
    HIP_TRY(w.bind());
    Pool &pool = Pool::get(w.device);

    /* slice boundaries in frames: equal shares of the samples */
    const bool small = total * 2 < (2 << 20) || pool.size() <= 1;
    int want = small ? 1 : (int)env_mb("SEA_HOST_SLICES", 8);
    want = (int)std::min<long long>(std::min(want, kMaxChunks), std::max<long long>(1, max_fr / 8));
    auto frames_below = [&](long long f) {
        long long s = 0;
        for (int j = 0; j < n_utt; ++j) s += std::min(nfr[j], f);
        return s;
    };
    std::vector<long long> B(1, 0);
    for (int k = 1; k < want; ++k) {
        long long lo = B.back() + 1, hi = max_fr; /* smallest f with frames_below(f) >= share */
        const long long share = total_fr * k / want;
        while (lo < hi) {
            const long long mid = (lo + hi) / 2;
            if (frames_below(mid) >= share) hi = mid; else lo = mid + 1;
        }
        if (lo >= max_fr) break;
        B.push_back(lo);
    }
    B.push_back(max_fr);
    const int K = (int)B.size() - 1;

    /* per slice: the active prefix of the sorted list, packed offsets (absolute, in samples) and lengths */
    std::vector<int> nact(K);
    std::vector<long long> soff(K + 1, 0);
    std::vector<size_t> mbase(K + 1, 0); /* where slice k's offsets | lengths sit in the meta array */
    for (int k = 0; k < K; ++k) {
        int n = 0;
        while (n < n_utt && nfr[n] > B[k]) ++n;
        nact[k] = n;
        long long smp = 0;
        for (int j = 0; j < n; ++j) smp += 80 * (std::min(nfr[j], B[k + 1]) - B[k]);
        soff[k + 1] = soff[k] + smp;
        mbase[k + 1] = mbase[k] + 2 * (size_t)n;
    }
    HIP_TRY(w.in.ensure((size_t)total));
    HIP_TRY(w.out.ensure((size_t)total));
    HIP_TRY(w.meta.ensure(mbase[K]));
    if (K > 1) HIP_TRY(w.ensure_state((size_t)n_utt * sea::kNsPipeStateFloats));
    std::vector<std::vector<long long>> bpre(K); /* byte prefix per slice: task sizing */
    for (int k = 0; k < K; ++k) {
        long long *offs = w.meta.h + mbase[k], *lens = offs + nact[k];
        bpre[k].assign(nact[k] + 1, 0);
        long long o = soff[k];
        for (int j = 0; j < nact[k]; ++j) {
            const long long L = 80 * (std::min(nfr[j], B[k + 1]) - B[k]);
            offs[j] = o;
            lens[j] = L;
            o += L;
            bpre[k][j + 1] = bpre[k][j] + 2 * L;
        }
    }

    std::vector<Latch> packed(K), unpacked(K);
    Scope scope(&w);
    short *h_in = w.in.h, *h_out = w.out.h;
    const int *ix = idx.data();
    for (int k = 0; k < K; ++k) {
        const long long *offs = w.meta.h + mbase[k], *lens = offs + nact[k];
        const long long b0 = 80 * B[k];
        run_copies(pool, 0, nact[k], bpre[k].data(), &packed[k], &scope.all, small,
                   [=](int j) { copy_stream(h_in + offs[j], in[ix[j]] + b0, (size_t)lens[j] * sizeof(short)); });
    }
    hipStream_t sUp = w.stream[0], sKern = w.stream[1], sDown = w.stream[2];
    HIP_TRY(hipMemcpyAsync(w.meta.d, w.meta.h, mbase[K] * sizeof(long long), hipMemcpyHostToDevice, sUp));

    /* SEA_HOST_TRACE=n: print the timeline of this thread's calls from the n-th list of more than one slice on */
    static std::atomic<int> n_sliced{0};
    const char *tr = getenv("SEA_HOST_TRACE");
    const bool trace = tr && (K == 1 ? atoi(tr) <= 0 : n_sliced++ >= atoi(tr));
    const double t0 = now_ms();
    /* Event-driven, as the chunk pipeline: a copy is only handed to the runtime when what it depends on has finished (the
     * copy engines' queues are in order: a D2H copy queued behind its still-running kernel would hold up the next H2D). */
    enum { kWaitPack, kComputing, kDownloading, kUnpacking };
    std::vector<char> state(K, kWaitPack);
    int next = 0, finished = 0;
    while (finished < K) {
        bool progressed = false;
        int r = 0;
        if (next < K && packed[next].ready() && (next == 0 || (r = evq(w.ev_h2d[next - 1], "hostpipe: upload event")) != 0)) {
            if (r < 0) return 1; /* Scope waits for the pool and drains the streams */
            const int k = next++;
            const long long cnt = soff[k + 1] - soff[k];
            if (trace) fprintf(stderr, "[hostpipe] %7.3f ms  slice %d packed (frames %lld..%lld of %d utterances, %.1f MB), issuing\n",
                               now_ms() - t0, k, B[k], B[k + 1], nact[k], cnt * 2 / 1048576.0);
            /* one stream per job: uploads (after the meta arrays, same stream), kernels (in order: the recursion needs
             * slice k after slice k - 1), downloads -- a slice's download is never queued behind a later slice's work */
            HIP_TRY(hipMemcpyAsync(w.in.d + soff[k], h_in + soff[k], (size_t)cnt * sizeof(short), hipMemcpyHostToDevice, sUp));
            HIP_TRY(hipEventRecord(w.ev_h2d[k], sUp));
            HIP_TRY(hipStreamWaitEvent(sKern, w.ev_h2d[k], 0));
            hipStream_t s = sKern;
            sea::NsBatchArgs a = {};
            a.in = w.in.d;
            a.out = w.out.d;
            a.offsets = w.meta.d + mbase[k];
            a.lengths = w.meta.d + mbase[k] + nact[k];
            a.tables = dc->ns;
            a.n_utt = nact[k];
            int form;
            if (K > 1) {
                a.state = w.d_state;
                a.resume = k > 0;
                a.frame_base = (int)B[k];
                form = (nact[k] <= 4 * dc->n_cu) ? 2 : 4;
            } else
                form = ns_pick_form(nact[k], dc->n_cu);
            if (form == 2 && nact[k] > dc->n_cu) a.prio_row = dc->n_cu;
            if (ns_launch(a, form, s)) return 1;
            HIP_TRY(hipEventRecord(w.ev_kernel[k], s));
            state[k] = kComputing;
            progressed = true;
        }
        for (int k = 0; k < next; ++k) {
            if (state[k] == kComputing && (r = evq(w.ev_kernel[k], "hostpipe: kernel event")) != 0) {
                if (r < 0) return 1; /* Scope waits for the pool and drains the streams */
                const long long cnt = soff[k + 1] - soff[k];
                HIP_TRY(hipMemcpyAsync(h_out + soff[k], w.out.d + soff[k], (size_t)cnt * sizeof(short), hipMemcpyDeviceToHost, sDown));
                HIP_TRY(hipEventRecord(w.ev_done[k], sDown));
                state[k] = kDownloading;
                progressed = true;
                if (trace) fprintf(stderr, "[hostpipe] %7.3f ms  slice %d computed, downloading\n", now_ms() - t0, k);
            }
            if (state[k] == kDownloading && (r = evq(w.ev_done[k], "hostpipe: download event")) != 0) {
                if (r < 0) return 1; /* Scope waits for the pool and drains the streams */
                if (trace) fprintf(stderr, "[hostpipe] %7.3f ms  slice %d on the host, unpacking\n", now_ms() - t0, k);
                const long long *offs = w.meta.h + mbase[k], *lens = offs + nact[k];
                const long long b0 = 80 * B[k];
                run_copies(pool, 0, nact[k], bpre[k].data(), &unpacked[k], &scope.all, small,
                           [=](int j) { copy_stream(out[ix[j]] + b0, h_out + offs[j], (size_t)lens[j] * sizeof(short)); });
                state[k] = kUnpacking;
                finished++;
                progressed = true;
            }
        }
        if (!progressed) {
            if (next < K && !packed[next].ready()) packed[next].wait_for(std::chrono::microseconds(30));
            else std::this_thread::sleep_for(std::chrono::microseconds(20));
        }
    }
    scope.all.wait();
    if (trace) fprintf(stderr, "[hostpipe] %7.3f ms  all unpacked (%d slices)\n", now_ms() - t0, K);
    scope.ok = true;
    return 0;
}

/* ---------------------------------------------------------------------------------------------------- */
/* sea_packed_*: NoiseSup from PINNED staging the CALLER fills and reads (VERDICT r03 #5 ii).
 *
 * sea_denoise_utterances takes the caller's pageable buffers, so every sample is copied twice more than the PCIe transfer
 * needs (pack into pinned staging, unpack out of it): 2 x 131 MB per 1024 utterances, and with the copies the DMA engines do
 * that is ~790 MB of DRAM traffic per call -- the packing threads' aggregate rate (~58 GB/s on the 1-GPU box, no better with
 * more threads) is what paces that entry point at ~4.5 ms per call (profiles/r04_host_pipeline.txt).  A caller that PRODUCES
 * its samples (a file reader) can write them where the copy engine reads them, and read the results where it writes them:
 *
 *   p = sea_packed_create();                       once per reusable staging set (any thread)
 *   sea_packed_plan(p, lengths, n);                lays the list out: time slices as in denoise_utterances_slices
 *   k = sea_packed_segments(p, u, in, out, cnt, max);   utterance u = k pieces in time order: fill in[i][0 .. cnt[i])
 *   sea_packed_denoise(p);                         on a device thread: uploads, launches, downloads, pipelined over the slices
 *   ... read out[i][0 .. cnt[i]) ...               etsi_denoise's result for every whole frame (the trailing lengths[u] % 80
 *                                                  samples are not part of any piece: the caller's, as with etsi_denoise)
 * The pinned staging is portable (any device may run sea_packed_denoise on it); the device side (buffers, streams, the
 * per-utterance state blob) is the calling thread's workspace, as for sea_denoise_utterances. */
} /* extern "C": the staging set's type and its allocator are C++ */
struct sea_packed {
    Grow<short> in, out;      /* only .h is used: pinned, portable */
    Grow<long long> meta;     /* per slice: offsets | lengths (as the kernel reads them); .d filled at run */
    std::vector<int> idx, inv, nact;
    std::vector<long long> nfr, B, soff;
    std::vector<size_t> mbase;
    int n_utt = 0, K = 0;
    long long total = 0;
};

template <class T>
static hipError_t grow_pinned(Grow<T> &g, size_t n)
{ /* host side only, portable across devices */
    if (n <= g.cap) return hipSuccess;
    if (g.h) (void)hipHostFree(g.h);
    g.h = nullptr;
    g.cap = 0;
    const size_t want = n + n / 4 + 4096;
    hipError_t e = hipHostMalloc((void **)&g.h, want * sizeof(T), hipHostMallocPortable);
    if (e == hipSuccess) g.cap = want;
    return e;
}

extern "C" {

sea_packed *sea_packed_create(void) { return new (std::nothrow) sea_packed(); }

void sea_packed_destroy(sea_packed *p)
{
    if (!p) return;
    if (p->in.h) (void)hipHostFree(p->in.h);
    if (p->out.h) (void)hipHostFree(p->out.h);
    if (p->meta.h) (void)hipHostFree(p->meta.h);
    p->in.h = p->out.h = nullptr;
    p->meta.h = nullptr;
    delete p;
}

int sea_packed_plan(sea_packed *p, const long *lengths, int n_utt)
{
    if (!p || n_utt < 0) return fail("sea_packed_plan: bad argument");
    p->n_utt = n_utt;
    p->K = 0;
    p->total = 0;
    if (n_utt == 0) return 0;
    for (int u = 0; u < n_utt; ++u)
        if (lengths[u] < 0) return fail("negative length for utterance %d", u);
    p->idx.resize(n_utt);
    p->inv.resize(n_utt);
    for (int i = 0; i < n_utt; ++i) p->idx[i] = i;
    std::stable_sort(p->idx.begin(), p->idx.end(), [&](int a, int b) { return lengths[a] > lengths[b]; });
    for (int j = 0; j < n_utt; ++j) p->inv[p->idx[j]] = j;
    p->nfr.resize(n_utt);
    long long total_fr = 0;
    for (int j = 0; j < n_utt; ++j) total_fr += (p->nfr[j] = lengths[p->idx[j]] / 80);
    p->total = total_fr * 80;
    if (total_fr == 0) return 0;
    const long long max_fr = p->nfr[0];
    /* slices: SEA_HOST_SLICES, default 10 (tools/host_packed.py on the configs[1] corpus, one box: 4: 4.22 ms, 6: 3.98, 8: 3.84,
     * 10: 3.74, 12: 3.73, 16: 3.82, 24: 3.94 -- without packing to overlap, the optimum is a little finer than the pointer-array
     * entry point's eight) */
    int want = (p->total * 2 < (2 << 20)) ? 1 : (int)env_mb("SEA_HOST_SLICES", 10);
    want = (int)std::min<long long>(std::min(want, kMaxChunks), std::max<long long>(1, max_fr / 8));
    auto frames_below = [&](long long f) {
        long long s = 0;
        for (int j = 0; j < n_utt; ++j) s += std::min(p->nfr[j], f);
        return s;
    };
    p->B.assign(1, 0);
    for (int k = 1; k < want; ++k) {
        long long lo = p->B.back() + 1, hi = max_fr;
        const long long share = total_fr * k / want;
        while (lo < hi) {
            const long long mid = (lo + hi) / 2;
            if (frames_below(mid) >= share) hi = mid; else lo = mid + 1;
        }
        if (lo >= max_fr) break;
        p->B.push_back(lo);
    }
    p->B.push_back(max_fr);
    const int K = p->K = (int)p->B.size() - 1;
    p->nact.assign(K, 0);
    p->soff.assign(K + 1, 0);
    p->mbase.assign(K + 1, 0);
    for (int k = 0; k < K; ++k) {
        int n = 0;
        while (n < n_utt && p->nfr[n] > p->B[k]) ++n;
        p->nact[k] = n;
        long long smp = 0;
        for (int j = 0; j < n; ++j) smp += 80 * (std::min(p->nfr[j], p->B[k + 1]) - p->B[k]);
        p->soff[k + 1] = p->soff[k] + smp;
        p->mbase[k + 1] = p->mbase[k] + 2 * (size_t)n;
    }
    HIP_TRY(grow_pinned(p->in, (size_t)p->total));
    HIP_TRY(grow_pinned(p->out, (size_t)p->total));
    HIP_TRY(grow_pinned(p->meta, p->mbase[K]));
    for (int k = 0; k < K; ++k) {
        long long *offs = p->meta.h + p->mbase[k], *lens = offs + p->nact[k];
        long long o = p->soff[k];
        for (int j = 0; j < p->nact[k]; ++j) {
            const long long L = 80 * (std::min(p->nfr[j], p->B[k + 1]) - p->B[k]);
            offs[j] = o;
            lens[j] = L;
            o += L;
        }
    }
    return 0;
}

int sea_packed_slices(const sea_packed *p) { return p ? p->K : 0; }

int sea_packed_segments(const sea_packed *p, int u, short **in_seg, short **out_seg, long *count, int max_seg)
{
    if (!p || u < 0 || u >= p->n_utt) return 0;
    const int j = p->inv[u];
    int n = 0;
    for (int k = 0; k < p->K && n < max_seg; ++k) {
        if (j >= p->nact[k]) break; /* the list is sorted by length: no later slice holds this utterance either */
        const long long *offs = p->meta.h + p->mbase[k], *lens = offs + p->nact[k];
        if (in_seg) in_seg[n] = p->in.h + offs[j];
        if (out_seg) out_seg[n] = p->out.h + offs[j];
        if (count) count[n] = (long)lens[j];
        ++n;
    }
    return n;
}

int sea_packed_denoise(sea_packed *p)
{
    if (!p) return fail("sea_packed_denoise: NULL");
    if (p->n_utt == 0 || p->K == 0) return 0;
    DeviceCtx *dc;
    if (ctx(&dc)) return 1;
    PipeWs &w = t_ws;
    HIP_TRY(w.bind());
    const int K = p->K, n_utt = p->n_utt;
    HIP_TRY(w.in.ensure((size_t)p->total)); /* (the workspace's own pinned halves stay unused here) */
    HIP_TRY(w.out.ensure((size_t)p->total));
    HIP_TRY(w.meta.ensure(p->mbase[K]));
    if (K > 1) HIP_TRY(w.ensure_state((size_t)n_utt * sea::kNsPipeStateFloats));
    Scope scope(&w);
    hipStream_t sUp = w.stream[0], sKern = w.stream[1], sDown = w.stream[2];
    HIP_TRY(hipMemcpyAsync(w.meta.d, p->meta.h, p->mbase[K] * sizeof(long long), hipMemcpyHostToDevice, sUp));
    /* uploads in order on one stream, kernels in order on another (slice k needs slice k - 1's state), each download handed
     * to the runtime when its kernel HAS finished (the copy engines' queues are in order: a download queued behind its
     * still-running kernel would hold up the next upload) */
    for (int k = 0; k < K; ++k) {
        const long long cnt = p->soff[k + 1] - p->soff[k];
        HIP_TRY(hipMemcpyAsync(w.in.d + p->soff[k], p->in.h + p->soff[k], (size_t)cnt * sizeof(short), hipMemcpyHostToDevice, sUp));
        HIP_TRY(hipEventRecord(w.ev_h2d[k], sUp));
        HIP_TRY(hipStreamWaitEvent(sKern, w.ev_h2d[k], 0));
        sea::NsBatchArgs a = {};
        a.in = w.in.d;
        a.out = w.out.d;
        a.offsets = w.meta.d + p->mbase[k];
        a.lengths = w.meta.d + p->mbase[k] + p->nact[k];
        a.tables = dc->ns;
        a.n_utt = p->nact[k];
        int form;
        if (K > 1) {
            a.state = w.d_state;
            a.resume = k > 0;
            a.frame_base = (int)p->B[k];
            form = (p->nact[k] <= 4 * dc->n_cu) ? 2 : 4;
        } else
            form = ns_pick_form(p->nact[k], dc->n_cu);
        if (form == 2 && p->nact[k] > dc->n_cu) a.prio_row = dc->n_cu;
        if (ns_launch(a, form, sKern)) return 1;
        HIP_TRY(hipEventRecord(w.ev_kernel[k], sKern));
    }
    int next = 0;
    while (next < K) {
        const int r = evq(w.ev_kernel[next], "sea_packed_denoise: kernel event");
        if (r < 0) return 1;
        if (r == 0) {
            std::this_thread::sleep_for(std::chrono::microseconds(20));
            continue;
        }
        const long long cnt = p->soff[next + 1] - p->soff[next];
        HIP_TRY(hipMemcpyAsync(p->out.h + p->soff[next], w.out.d + p->soff[next], (size_t)cnt * sizeof(short), hipMemcpyDeviceToHost, sDown));
        ++next;
    }
    HIP_TRY(hipStreamSynchronize(sDown));
    scope.ok = true;
    return 0;
}

/* ---------------------------------------------------------------------------------------------------- */
/* NoiseSup + CompCeps from host buffers: the explicit chain SURVEY 8(c) describes for the reference (DoNoiseSup into
 * the denoised-sample shift register, DoCompCeps on its last 201 samples from the third output on,
 * etsi/cpp/ParmInterface.c:275-293).  ceps[u] receives n_ceps[u] rows of 14 floats (c1..c12, c0, logE); its capacity
 * must be max(lengths[u]/80 - 6, 0) rows.  One launch each, no chunking (the feature path of the file driver). */
int sea_denoise_ceps_utterances(const short *const *in, short *const *out, float *const *ceps, int *n_ceps,
                                const long *lengths, int n_utt)
{
    if (n_utt <= 0) return 0;
    DeviceCtx *dc;
    if (ctx(&dc)) return 1;
    std::vector<long long> pre(n_utt + 1, 0), cum(n_utt + 1, 0);
    for (int u = 0; u < n_utt; ++u) {
        if (lengths[u] < 0) return fail("negative length for utterance %d", u);
        pre[u + 1] = pre[u] + align8(lengths[u]);
        cum[u + 1] = cum[u] + std::max<long long>(lengths[u] / 80 - 6, 0);
    }
    const long long total = pre[n_utt], total_ceps = cum[n_utt];
    for (int u = 0; u < n_utt; ++u) n_ceps[u] = 0;
    if (total == 0) return 0;
    PipeWs &w = t_ws;
    HIP_TRY(w.bind());
    HIP_TRY(w.in.ensure((size_t)total));
    HIP_TRY(w.out.ensure((size_t)total));
    HIP_TRY(w.meta.ensure(3 * (size_t)n_utt + 1));
    long long *offs = w.meta.h, *lens = w.meta.h + n_utt, *ccum = w.meta.h + 2 * n_utt;
    for (int u = 0; u < n_utt; ++u) {
        offs[u] = pre[u];
        lens[u] = lengths[u];
        memcpy(w.in.h + pre[u], in[u], (size_t)lengths[u] * sizeof(short));
        const long long pad = align8(lengths[u]) - lengths[u];
        if (pad) memset(w.in.h + pre[u] + lengths[u], 0, (size_t)pad * sizeof(short));
    }
    for (int u = 0; u <= n_utt; ++u) ccum[u] = cum[u];
    /* grow-only members of the workspace (round 3 allocated and freed five device buffers per call: an implicit device
     * synchronisation for every other host thread of the process, and the file driver calls this per chunk) */
    const size_t nceps_f = (size_t)std::max<long long>(total_ceps, 1) * 14;
    HIP_TRY(w.f32.ensure((size_t)total));
    HIP_TRY(w.ceps.ensure(nceps_f));
    HIP_TRY(w.ints.ensure(3 * (size_t)n_utt));
    int *d_first = w.ints.d, *d_nceps = w.ints.d + n_utt, *d_order = w.ints.d + 2 * n_utt;
    int *h_nceps = w.ints.h + n_utt, *h_order = w.ints.h + 2 * n_utt;
    launch_order(lens, n_utt, dc->n_cu, h_order);
    hipStream_t s = w.stream[0];
    /* from the first asynchronous call on every exit path drains the stream (Scope: ok stays false on a fault), so no copy
     * into w's pinned staging or kernel over w's device buffers is still in flight when the caller gets its fault code */
    Scope scope(&w);
    HIP_TRY(hipMemcpyAsync(w.in.d, w.in.h, (size_t)total * sizeof(short), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(w.meta.d, w.meta.h, (3 * (size_t)n_utt + 1) * sizeof(long long), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d_order, h_order, (size_t)n_utt * sizeof(int), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemsetAsync(d_nceps, 0, (size_t)n_utt * sizeof(int), s));
    if (sea_ns_denoise_batch(w.in.d, w.out.d, w.f32.d, w.meta.d, w.meta.d + n_utt, n_utt > 1 ? d_order : nullptr, d_first,
                             n_utt, s))
        return 1;
    if (total_ceps > 0 && sea_compceps_batch(w.f32.d, w.meta.d, w.meta.d + n_utt, d_first, w.meta.d + 2 * n_utt, total_ceps,
                                             w.ceps.d, d_nceps, n_utt, s))
        return 1;
    HIP_TRY(hipMemcpyAsync(w.out.h, w.out.d, (size_t)total * sizeof(short), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(w.ceps.h, w.ceps.d, nceps_f * sizeof(float), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(h_nceps, d_nceps, (size_t)n_utt * sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    scope.ok = true;
    for (int u = 0; u < n_utt; ++u) {
        n_ceps[u] = h_nceps[u];
        memcpy(out[u], w.out.h + pre[u], (size_t)(lengths[u] / 80 * 80) * sizeof(short));
        if (n_ceps[u] > 0) memcpy(ceps[u], w.ceps.h + (size_t)cum[u] * 14, (size_t)n_ceps[u] * 14 * sizeof(float));
    }
    return 0;
}

/* ---------------------------------------------------------------------------------------------------- */
/* The [time][64] float intermediate costs 256 B of HBM per sample (~16 MB per 4-s utterance).  The chunks of the
 * pipeline run on kStreams streams, each with a scratch region of its own, sized so that all regions together fit
 * in 60 % of the HBM that is free right now (SEA_RESYNTH_SCRATCH_MB overrides the budget); a chunk is at most what
 * one region holds.  The reference processes one utterance at a time (resyth_64sub_ori/cpp/main.cpp:84-145);
 * results do not depend on the cut. */
int sea_resynth_utterances(const short *const *in, const long *lengths, const float *const *masks, int binary,
                           short *const *out, int n_utt)
{
    if (n_utt <= 0) return 0;
    DeviceCtx *dc;
    if (ctx(&dc)) return 1;
    auto nrows = [&](long L) -> long long { return (binary & 2) ? L / 160 : (L - 320) / 160 + 1; };
    long long largest = 0;
    for (int u = 0; u < n_utt; ++u) {
        if (lengths[u] < ((binary & 2) ? 160 : 320))
            return fail("resynth: utterance %d has %ld samples (too short for one mask frame)", u, lengths[u]);
        largest = std::max(largest, sea_resynth_scratch_bytes(align8(lengths[u]), 1));
    }
    std::vector<int> idx(n_utt);
    for (int i = 0; i < n_utt; ++i) idx[i] = i;
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return lengths[a] > lengths[b]; });
    std::vector<long long> pre(n_utt + 1, 0), rpre(n_utt + 1, 0);
    for (int j = 0; j < n_utt; ++j) {
        pre[j + 1] = pre[j] + align8(lengths[idx[j]]);
        rpre[j + 1] = rpre[j] + nrows(lengths[idx[j]]);
    }
    const long long total = pre[n_utt], rows = rpre[n_utt];

    PipeWs &w = t_ws;
    HIP_TRY(w.bind());
    Pool &pool = Pool::get(w.device);
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    long long budget = (long long)((free_b + w.inter_bytes) / 10 * 6);
    if (const char *e = getenv("SEA_RESYNTH_SCRATCH_MB")) budget = atoll(e) * (1LL << 20);
    /* regions: as many streams as there will be chunks, each at least one utterance */
    long long region = std::max(largest, budget / kStreams);
    /* chunk = at most a region's worth of scratch and about SEA_HOST_CHUNK_MB of (int16 + mask) input */
    std::vector<int> cuts(1, 0);
    {
        /* default: about a quarter of the list per chunk (few large launches fill the chip better than many small ones) */
        const long long in_target = getenv("SEA_HOST_CHUNK_MB") ? env_mb("SEA_HOST_CHUNK_MB", 24) * (1 << 20)
                                                                : std::max<long long>((total * 4 + rows * 256 + 3) / 4, 1 << 20);
        long long run = 0, run_bytes = 0;
        int run_n = 0;
        for (int j = 0; j < n_utt; ++j) {
            const long long Lp = align8(lengths[idx[j]]);
            const long long b = Lp * 4 + nrows(lengths[idx[j]]) * 256;
            const bool over = sea_resynth_scratch_bytes(run + Lp, run_n + 1) > region || run_bytes + b > in_target ||
                              run_n >= 4 * dc->n_cu;
            if (run_n > 0 && over && (int)cuts.size() < kMaxChunks) {
                cuts.push_back(j);
                run = run_bytes = 0;
                run_n = 0;
            }
            run += Lp;
            run_bytes += b;
            run_n++;
        }
        cuts.push_back(n_utt);
    }
    const int nchunk = (int)cuts.size() - 1;
    /* a region is as large as the largest chunk (the last one can exceed the budget share when kMaxChunks is reached:
     * hipMalloc then reports what does not fit) */
    region = 0;
    for (int k = 0; k < nchunk; ++k)
        region = std::max(region, sea_resynth_scratch_bytes(pre[cuts[k + 1]] - pre[cuts[k]], cuts[k + 1] - cuts[k]));
    const int nregion = std::min(nchunk, kStreams);
    HIP_TRY(w.ensure_inter((size_t)region * nregion));
    HIP_TRY(w.in.ensure((size_t)total));
    HIP_TRY(w.out.ensure((size_t)total));
    HIP_TRY(w.mask.ensure((size_t)rows * 64));
    HIP_TRY(w.meta.ensure(3 * (size_t)n_utt));
    long long *offs = w.meta.h, *lens = w.meta.h + n_utt, *moffs = w.meta.h + 2 * n_utt;
    for (int k = 0; k < nchunk; ++k)
        for (int j = cuts[k]; j < cuts[k + 1]; ++j) {
            /* sample offsets are relative to the chunk (the scratch region is indexed by them), buffers are sliced */
            offs[j] = pre[j] - pre[cuts[k]];
            lens[j] = lengths[idx[j]];
            moffs[j] = rpre[j] - rpre[cuts[k]];
        }
    const bool small = (total * 4 + rows * 256) < (2 << 20) || n_utt == 1 || pool.size() <= 1;

    std::vector<Latch> packed(nchunk), unpacked(nchunk);
    Scope scope(&w);
    short *h_in = w.in.h, *h_out = w.out.h;
    float *h_mask = w.mask.h;
    std::vector<long long> bytes_pre(n_utt + 1);
    for (int j = 0; j <= n_utt; ++j) bytes_pre[j] = pre[j] * 2 + rpre[j] * 256;
    const long long *bp = bytes_pre.data(), *prep = pre.data(), *rprep = rpre.data();
    const int *ix = idx.data();
    for (int k = 0; k < nchunk; ++k)
        run_copies(pool, cuts[k], cuts[k + 1], bp, &packed[k], &scope.all, small, [=](int j) {
            const long long L = lens[j];
            copy_stream(h_in + prep[j], in[ix[j]], (size_t)L * sizeof(short));
            const long long pad = align8(L) - L;
            if (pad) memset(h_in + prep[j] + L, 0, (size_t)pad * sizeof(short));
            copy_stream(h_mask + rprep[j] * 64, masks[ix[j]], (size_t)(rprep[j + 1] - rprep[j]) * 64 * sizeof(float));
        });
    HIP_TRY(hipMemcpyAsync(w.meta.d, w.meta.h, 3 * (size_t)n_utt * sizeof(long long), hipMemcpyHostToDevice, w.stream[0]));
    HIP_TRY(hipEventRecord(w.ev_meta, w.stream[0]));

    /* event-driven, as in sea_denoise_utterances: the D2H copy of a chunk is issued when its kernel has finished */
    enum { kWaitPack, kComputing, kDownloading, kUnpacking };
    std::vector<char> state(nchunk, kWaitPack);
    int next = 0, finished = 0;
    while (finished < nchunk) {
        bool progressed = false;
        int r = 0;
        /* a chunk reuses the scratch region and the stream of chunk k - nregion: stream order protects the region */
        if (next < nchunk && packed[next].ready()) {
            const int k = next++, r = k % nregion;
            hipStream_t s = w.stream[r];
            const int u0 = cuts[k], n = cuts[k + 1] - cuts[k];
            const long long o0 = pre[u0], cnt = pre[cuts[k + 1]] - o0, r0 = rpre[u0], rcnt = rpre[cuts[k + 1]] - r0;
            if (k > 0 && k < nregion) HIP_TRY(hipStreamWaitEvent(s, w.ev_meta, 0));
            HIP_TRY(hipMemcpyAsync(w.in.d + o0, h_in + o0, (size_t)cnt * sizeof(short), hipMemcpyHostToDevice, s));
            HIP_TRY(hipMemcpyAsync(w.mask.d + r0 * 64, h_mask + r0 * 64, (size_t)rcnt * 64 * sizeof(float), hipMemcpyHostToDevice, s));
            if (sea_resynth64_batch(w.in.d + o0, w.out.d + o0, w.meta.d + u0, w.meta.d + n_utt + u0, w.mask.d + r0 * 64,
                                    w.meta.d + 2 * n_utt + u0, (float *)((char *)w.d_inter + (size_t)r * region), nullptr, n,
                                    binary, s))
                return 1;
            HIP_TRY(hipEventRecord(w.ev_kernel[k], s));
            state[k] = kComputing;
            progressed = true;
        }
        for (int k = 0; k < next; ++k) {
            if (state[k] == kComputing && (r = evq(w.ev_kernel[k], "hostpipe: kernel event")) != 0) {
                if (r < 0) return 1; /* Scope waits for the pool and drains the streams */
                const long long o0 = pre[cuts[k]], cnt = pre[cuts[k + 1]] - o0;
                hipStream_t s = w.stream[k % nregion];
                HIP_TRY(hipMemcpyAsync(h_out + o0, w.out.d + o0, (size_t)cnt * sizeof(short), hipMemcpyDeviceToHost, s));
                HIP_TRY(hipEventRecord(w.ev_done[k], s));
                state[k] = kDownloading;
                progressed = true;
            }
            if (state[k] == kDownloading && (r = evq(w.ev_done[k], "hostpipe: download event")) != 0) {
                if (r < 0) return 1; /* Scope waits for the pool and drains the streams */
                run_copies(pool, cuts[k], cuts[k + 1], bp, &unpacked[k], &scope.all, small,
                           [=](int j) { copy_stream(out[ix[j]], h_out + prep[j], (size_t)lens[j] * sizeof(short)); });
                state[k] = kUnpacking;
                finished++;
                progressed = true;
            }
        }
        if (!progressed) {
            if (next < nchunk) packed[next].wait_for(std::chrono::microseconds(30));
            else std::this_thread::sleep_for(std::chrono::microseconds(20));
        }
    }
    scope.all.wait();
    scope.ok = true;
    return 0;
}

} // extern "C"
