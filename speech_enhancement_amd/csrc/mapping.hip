/*
 * mapping.hip -- the reference's batch plug-in symbols
 *     function/20141106_speech_enhancement/aurora_etsi/NoiseSupExports.h:35-42
 * as adapters over this engine's streaming entry points: a global instance (sea_init), a per-thread instance (one
 * NoiseSup state blob in HBM + pinned staging) and one call per chunk of samples (sea_ns_streams_push_fd).
 *
 * SEMANTICS.  By default (sm_glb_res == NULL, which is what the reference's caller passes,
 * resyth_64sub_ori/cpp/aurora_etsi_test.cpp:20) the symbols run what the reference builds behind them: the 16 k-native
 * variant (160-sample frames, window 480, gammatone-shaped windows, rfft (x, 512, 8);
 * aurora_etsi/NoiseSup.cpp:1140-1407 -> csrc/ns16k_kernel.hip), including the line of 25 gains per second-stage frame
 * printed to the FILE* argument ("%f " each, then a newline; :1319-1328) when that is not NULL.
 * sm_glb_res is ignored, as the reference ignores it (its cast to DENOISEGlobalImpl is commented out, :915, and
 * SamplingFrequency forced to 16000).  Extension, opt-in through the environment only (SEA_MAPPING_8K=1 at global_init):
 * the etsi/ arithmetic instead (80-sample frames, window 200, rfft 256, mel filter bank: the hot path's kernels; nothing
 * is printed).
 * In both:
 *   - one call consumes dataNum / hop whole frames of inData and advances the per-thread state;
 *   - a frame whose float sum of squares truncates to 0 is skipped entirely (:1160-1171): no state change, its
 *     outData / flag entries are not written;
 *   - outData[hop n ..] is written when the second stage produced a frame (from the 5th processed frame on), after the
 *     DC-offset filter; pSpeechFoundVar / Spec / Mel / VADNS [n] are written when the first stage ran (from the 3rd
 *     processed frame on), pFrameCounter[n] = the first stage's VAD frame counter after the frame;
 *   - global_init / thread_init return 1 on success (thread_init 0 on allocation failure), func_Wiener returns 0.
 * INSTANCE / PINSTANCE / int32s come from the absent aurora/aurora_include.h: void*, void**, int.
 */
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <new>
#include <vector>

#include "capi_internal.h"

using namespace sea_capi;

namespace {

struct MapGlobal {
    int device;
    int sampling_frequency; /* DENOISEGlobalImpl::SamplingFrequency (NoiseSupExports.h:9-12): 16000, or 8000 = etsi/ arithmetic */
};

struct MapThread {
    float *d_state = nullptr;
    /* pinned, grow-only: in | out | produced | flags | frame counter for up to cap frames */
    float *h_in = nullptr, *h_out = nullptr, *d_in = nullptr, *d_out = nullptr;
    int *h_prod = nullptr, *h_cnt = nullptr, *d_prod = nullptr, *d_cnt = nullptr;
    unsigned char *h_flags = nullptr, *d_flags = nullptr;
    float *h_wiener = nullptr, *d_wiener = nullptr; /* 16 k-native variant: 25 gains per frame */
    int hop = 160;                                  /* 160: the 16 k-native variant; 80: the etsi/ arithmetic */
    size_t cap = 0;
    hipStream_t stream = nullptr;
    int device = -1;
    int fresh = 1;
    void release_io()
    {
        if (h_in) (void)hipHostFree(h_in);
        if (h_out) (void)hipHostFree(h_out);
        if (h_prod) (void)hipHostFree(h_prod);
        if (h_cnt) (void)hipHostFree(h_cnt);
        if (h_flags) (void)hipHostFree(h_flags);
        if (h_wiener) (void)hipHostFree(h_wiener);
        if (d_wiener) (void)hipFree(d_wiener);
        if (d_in) (void)hipFree(d_in);
        if (d_out) (void)hipFree(d_out);
        if (d_prod) (void)hipFree(d_prod);
        if (d_cnt) (void)hipFree(d_cnt);
        if (d_flags) (void)hipFree(d_flags);
        h_in = h_out = d_in = d_out = nullptr;
        h_prod = h_cnt = d_prod = d_cnt = nullptr;
        h_flags = d_flags = nullptr;
        h_wiener = d_wiener = nullptr;
        cap = 0;
    }
    hipError_t ensure(size_t frames)
    {
        if (frames <= cap) return hipSuccess;
        release_io();
        const size_t n = frames + frames / 4 + 64;
        hipError_t e;
        if ((e = hipHostMalloc((void **)&h_in, n * hop * sizeof(float), hipHostMallocDefault)) != hipSuccess ||
            (e = hipHostMalloc((void **)&h_out, n * hop * sizeof(float), hipHostMallocDefault)) != hipSuccess ||
            (e = hipHostMalloc((void **)&h_prod, n * sizeof(int), hipHostMallocDefault)) != hipSuccess ||
            (e = hipHostMalloc((void **)&h_cnt, n * sizeof(int), hipHostMallocDefault)) != hipSuccess ||
            (e = hipHostMalloc((void **)&h_flags, n, hipHostMallocDefault)) != hipSuccess ||
            (e = hipMalloc((void **)&d_in, n * hop * sizeof(float))) != hipSuccess ||
            (e = hipMalloc((void **)&d_out, n * hop * sizeof(float))) != hipSuccess ||
            (e = hipMalloc((void **)&d_prod, n * sizeof(int))) != hipSuccess ||
            (e = hipMalloc((void **)&d_cnt, n * sizeof(int))) != hipSuccess ||
            (e = hipMalloc((void **)&d_flags, n)) != hipSuccess ||
            (hop == 160 && ((e = hipHostMalloc((void **)&h_wiener, n * 25 * sizeof(float), hipHostMallocDefault)) != hipSuccess ||
                            (e = hipMalloc((void **)&d_wiener, n * 25 * sizeof(float))) != hipSuccess))) {
            release_io();
            return e;
        }
        cap = n;
        return hipSuccess;
    }
};

/* the calling thread's current device is put back when an entry point returns */
struct DeviceScope {
    int prev = -1;
    bool ok = false;
    explicit DeviceScope(int dev)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = (prev == dev) || hipSetDevice(dev) == hipSuccess;
        if (prev == dev) prev = -1; /* nothing to restore */
    }
    ~DeviceScope()
    {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

/* NoiseSupExports.h:14-27 */
struct esti_denoise_in {
    float *inData;
    int dataNum;
};
struct esti_denoise_out {
    float *outData;
    int *pSpeechFoundVar;
    int *pSpeechFoundSpec;
    int *pSpeechFoundMel;
    int *pSpeechFoundVADNS;
    int *pFrameCounter;
};

} // namespace

extern "C" {

int etsi_denoise_mapping_global_init(void **sm_glb_pins, void *sm_glb_res)
{ /* aurora_etsi/NoiseSup.cpp:913-922 */
    if (!sm_glb_pins) return 0;
    *sm_glb_pins = nullptr;
    if (sea_init(-1)) return 0;
    MapGlobal *g = new (std::nothrow) MapGlobal();
    if (!g) return 0;
    /* The reference ignores sm_glb_res entirely (the cast is commented out and SamplingFrequency forced to 16000,
     * NoiseSup.cpp:913-922), so nothing is read through it here either: a caller passing a real DENOISEGlobalImpl -- or any
     * other resource pointer -- gets the reference's behaviour.  The etsi/ arithmetic on 80-sample frames is an extension
     * the reference cannot trigger: the environment variable SEA_MAPPING_8K=1, read once per global_init. */
    (void)sm_glb_res;
    const char *e8 = getenv("SEA_MAPPING_8K");
    g->sampling_frequency = (e8 && e8[0] == '1') ? 8000 : 16000;
    if (hipGetDevice(&g->device) != hipSuccess) {
        delete g;
        return 0;
    }
    *sm_glb_pins = g;
    return 1;
}

int etsi_denoise_mapping_thread_init(void **sm_thd_pins, void *sm_glb_ins)
{ /* :937-1097: calloc + DoNoiseSupInit; 0 when the allocation fails */
    if (!sm_thd_pins) return 0;
    *sm_thd_pins = nullptr;
    MapGlobal *g = (MapGlobal *)sm_glb_ins;
    if (!g) return 0;
    DeviceScope dev(g->device);
    if (!dev.ok) return 0;
    MapThread *t = new (std::nothrow) MapThread();
    if (!t) return 0;
    t->device = g->device;
    t->hop = (g->sampling_frequency == 8000) ? 80 : 160;
    const size_t state_floats = (t->hop == 80) ? (size_t)sea::kNsStateFloats : (size_t)sea::kNs16StateFloats;
    if (hipMalloc((void **)&t->d_state, state_floats * sizeof(float)) != hipSuccess ||
        hipStreamCreateWithFlags(&t->stream, hipStreamNonBlocking) != hipSuccess) {
        if (t->d_state) (void)hipFree(t->d_state);
        delete t;
        fail("etsi_denoise_mapping_thread_init: device allocation failed");
        return 0;
    }
    *sm_thd_pins = t;
    return 1;
}

int etsi_denoise_mapping_func_Wiener(void *sm_glb_ins, void *sm_thd_ins, void *in_ins, void *out_ins, void *fp_Wiener)
{ /* :1140-1407 */
    (void)sm_glb_ins;
    MapThread *t = (MapThread *)sm_thd_ins;
    const esti_denoise_in *in = (const esti_denoise_in *)in_ins;
    const esti_denoise_out *out = (const esti_denoise_out *)out_ins;
    if (!t || !in || !out) return fail("etsi_denoise_mapping_func_Wiener: NULL instance");
    const int hop = t->hop;
    const long nfr = in->dataNum / hop;
    if (nfr <= 0) return 0;
    DeviceScope dev(t->device);
    if (!dev.ok) return fail("etsi_denoise_mapping_func_Wiener: cannot select device %d", t->device);
    HIP_TRY(t->ensure((size_t)nfr));
    std::vector<long> kept;
    kept.reserve((size_t)nfr);
    if (hop == 80) {
        /* the zero-frame gate (:1160-1171) on the host: float sum of squares in sample order, truncated to int */
        for (long n = 0; n < nfr; ++n) {
            const float *x = in->inData + 80 * n;
            float check = 0.0f;
            for (int i = 0; i < 80; ++i) check += x[i] * x[i];
            if ((int)check == 0) continue;
            memcpy(t->h_in + 80 * kept.size(), x, 80 * sizeof(float));
            kept.push_back(n);
        }
    } else { /* the 16 k-native kernel applies the gate itself and reports per frame what was written */
        memcpy(t->h_in, in->inData, (size_t)nfr * hop * sizeof(float));
        for (long n = 0; n < nfr; ++n) kept.push_back(n);
    }
    const int m = (int)kept.size();
    if (m == 0) return 0;
    HIP_TRY(hipMemcpyAsync(t->d_in, t->h_in, (size_t)m * hop * sizeof(float), hipMemcpyHostToDevice, t->stream));
    if (hop == 80) {
        if (sea_ns_streams_push_fd(t->d_in, t->d_out, t->d_prod, t->d_flags, t->d_cnt, t->d_state, 1, m, t->fresh, t->stream))
            return 1;
    } else {
        if (sea_ns16k_streams_push(t->d_in, t->d_out, t->d_prod, t->d_flags, t->d_cnt, t->d_wiener, t->d_state, 1, m, t->fresh,
                                   t->stream))
            return 1;
        HIP_TRY(hipMemcpyAsync(t->h_wiener, t->d_wiener, (size_t)m * 25 * sizeof(float), hipMemcpyDeviceToHost, t->stream));
    }
    HIP_TRY(hipMemcpyAsync(t->h_out, t->d_out, (size_t)m * hop * sizeof(float), hipMemcpyDeviceToHost, t->stream));
    HIP_TRY(hipMemcpyAsync(t->h_prod, t->d_prod, (size_t)m * sizeof(int), hipMemcpyDeviceToHost, t->stream));
    HIP_TRY(hipMemcpyAsync(t->h_cnt, t->d_cnt, (size_t)m * sizeof(int), hipMemcpyDeviceToHost, t->stream));
    HIP_TRY(hipMemcpyAsync(t->h_flags, t->d_flags, (size_t)m, hipMemcpyDeviceToHost, t->stream));
    HIP_TRY(hipStreamSynchronize(t->stream));
    t->fresh = 0;
    FILE *fp = (hop == 160) ? (FILE *)fp_Wiener : nullptr;
    for (int k = 0; k < m; ++k) {
        const long n = kept[k];
        if (t->h_prod[k] && out->outData) memcpy(out->outData + (size_t)hop * n, t->h_out + (size_t)hop * k, hop * sizeof(float));
        if (t->h_cnt[k] >= 1) { /* the first stage has run at this frame (it runs at every frame from its first run on) */
            const int f = t->h_flags[k];
            if (out->pSpeechFoundVar) out->pSpeechFoundVar[n] = f & 1;
            if (out->pSpeechFoundSpec) out->pSpeechFoundSpec[n] = (f >> 1) & 1;
            if (out->pSpeechFoundMel) out->pSpeechFoundMel[n] = (f >> 2) & 1;
            if (out->pSpeechFoundVADNS) out->pSpeechFoundVADNS[n] = (f >> 3) & 1;
            if (out->pFrameCounter) out->pFrameCounter[n] = t->h_cnt[k];
        }
        if (fp && t->h_prod[k]) { /* :1319-1328 */
            for (int i = 0; i < 25; ++i) fprintf(fp, "%f ", t->h_wiener[25 * k + i]);
            fprintf(fp, "\n");
        }
    }
    return 0;
}

int etsi_denoise_mapping_func(void *sm_glb_ins, void *sm_thd_ins, void *in_ins, void *out_ins)
{ /* NoiseSupExports.h:39: declared beside func_Wiener (its definition is commented out in the reference) */
    return etsi_denoise_mapping_func_Wiener(sm_glb_ins, sm_thd_ins, in_ins, out_ins, nullptr);
}

void etsi_denoise_mapping_thread_release(void **sm_thd_pins)
{ /* :1099-1119 */
    if (!sm_thd_pins || !*sm_thd_pins) return;
    MapThread *t = (MapThread *)*sm_thd_pins;
    {
        DeviceScope dev(t->device);
        t->release_io();
        if (t->d_state) (void)hipFree(t->d_state);
        if (t->stream) (void)hipStreamDestroy(t->stream);
    }
    delete t;
    *sm_thd_pins = nullptr;
}

void etsi_denoise_mapping_global_release(void **sm_glb_pins)
{ /* :1120-1125 */
    if (!sm_glb_pins || !*sm_glb_pins) return;
    delete (MapGlobal *)*sm_glb_pins;
    *sm_glb_pins = nullptr;
}

} // extern "C"
