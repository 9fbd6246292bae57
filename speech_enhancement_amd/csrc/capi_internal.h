/*
 * capi_internal.h -- what the translation units behind the C ABI share (capi.hip, hostpipe.hip): the
 * per-thread error string, the per-device context with the constant tables, small RAII helpers.
 */
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/sea_mi355x.h"
#include "sea_kernels.h"

namespace sea_capi {

int fail(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
const char *last_error();

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return ::sea_capi::fail("%s: %s", #expr, hipGetErrorString(e_));          \
    } while (0)

struct DeviceCtx {
    bool ready = false;
    sea_ns_tables *ns = nullptr;
    sea_cc_tables *cc = nullptr;
    sea_gt_tables *gt = nullptr;
    sea_ns16k_tables *ns16 = nullptr;
    int n_cu = 256;
};

/* per-device context for the CURRENT device; uploads the constant tables on first use */
int ctx(DeviceCtx **out);

template <typename T>
struct DevBuf {
    T *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n) { return hipMalloc(&p, n * sizeof(T) + 16); }
};

inline long long align8(long long v) { return (v + 7) & ~7LL; }

/* launch order of the utterance-per-workgroup kernels: longest first, every other row of n_cu reversed */
void launch_order(const long long *lens, int n, int n_cu, int *order);

/* NoiseSup kernel form for a batch of n_inflight utterances sharing the device (capi.hip::sea_ns_denoise_batch):
 * 3 six-wave, 2 four-wave, 4 four-wave / tables in LDS, 1 single wave; honours sea_ns_kernel_form / SEA_NS_KERNEL */
int ns_pick_form(int n_inflight, int n_cu);
/* one launch of that form over a.n_utt utterances; returns 0 or fail() */
int ns_launch(const sea::NsBatchArgs &a, int form, hipStream_t stream);

} // namespace sea_capi
