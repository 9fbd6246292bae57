// tools/fma_probe.hip -- dependent-chain cost of v_fma_f32 / v_add_f32 for one wave64, with the f32 denormal
// mode the product library is built with (-fno-gpu-flush-denormals-to-zero) and without.  Diagnostic only.
//   hipcc --offload-arch=gfx950 -O2 [-fno-gpu-flush-denormals-to-zero] tools/fma_probe.hip -o /tmp/fma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP256(x) REP16(REP16(x))
#define PROBE(name, body)                                                                  \
    __global__ void name(unsigned long long *out, float *sink, float a0)                   \
    {                                                                                        \
        float a = a0, b = a0 + 1, c = a0 + 2, e = 0.999f;                                    \
        unsigned long long msk = 1;                                                          \
        unsigned long long t0 = __builtin_readcyclecounter();                                \
        REP256(body)                                                                         \
        unsigned long long t1 = __builtin_readcyclecounter();                                \
        if (threadIdx.x == 0) out[0] = t1 - t0;                                              \
        sink[threadIdx.x] = a + b + c + (float)msk;                                          \
    }
PROBE(k_add_dep, asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(e));)
PROBE(k_fma_src0, asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(e), "v"(b));)
PROBE(k_fma_src1, asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(a) : "v"(e), "v"(b));)
PROBE(k_fma_src2, asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a) : "v"(e), "v"(b));)
PROBE(k_fma_capture, asm volatile("v_fma_f32 %0, %3, %0, %4\n s_lshl_b64 %2, %2, 1\n v_cndmask_b32_e64 %1, %1, %0, %2" : "+v"(a), "+v"(c), "+s"(msk) : "v"(e), "v"(b));)
PROBE(k_fma_indep_mov, asm volatile("v_fma_f32 %0, %2, %0, %3\n v_mov_b32 %1, %3" : "+v"(a), "+v"(c) : "v"(e), "v"(b));)
PROBE(k_fma_indep_cnd, asm volatile("v_fma_f32 %0, %3, %0, %4\n v_cndmask_b32_e64 %1, %1, %4, %2" : "+v"(a), "+v"(c), "+s"(msk) : "v"(e), "v"(b));)
PROBE(k_fma_prevcap, asm volatile("v_fma_f32 %1, %4, %0, %5\n v_cndmask_b32_e64 %2, %2, %0, %3\n s_lshl_b64 %3, %3, 1\n v_fma_f32 %0, %4, %1, %5\n v_cndmask_b32_e64 %2, %2, %1, %3\n s_lshl_b64 %3, %3, 1" : "+v"(a), "+v"(b), "+v"(c), "+s"(msk) : "v"(e), "v"(e));)
PROBE(k_mul_dep, asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a) : "v"(e));)
template <typename K> void run(const char *name, K k)
{
    unsigned long long *d; float *sink;
    hipMalloc(&d, 8); hipMalloc(&sink, 4096);
    k<<<1, 64>>>(d, sink, 1.0f);
    k<<<1, 64>>>(d, sink, 1.0f);
    unsigned long long h = 0; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("%-16s %6.2f clk per step\n", name, (double)h / 256.0);
    fflush(stdout);
    hipFree(d); hipFree(sink);
}
int main()
{
    run("k_add_dep", k_add_dep); run("k_mul_dep", k_mul_dep); run("k_fma_src0", k_fma_src0); run("k_fma_src1", k_fma_src1);
    run("k_fma_src2", k_fma_src2); run("k_fma_capture", k_fma_capture); run("k_fma_indep_mov", k_fma_indep_mov); run("k_fma_indep_cnd", k_fma_indep_cnd); run("k_fma_prevcap(2 steps)", k_fma_prevcap);
    return 0;
}
