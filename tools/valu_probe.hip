// tools/valu_probe.hip -- aggregate vector-instruction issue rate of ONE SIMD as a function of how many waves share it
// and of how dependent each wave's stream is.  Diagnostic only.
//   hipcc --offload-arch=gfx950 -O2 tools/valu_probe.hip -o /tmp/valu_probe && /tmp/valu_probe
// One workgroup of 64*4*W threads on one CU = W waves per SIMD.  Every wave runs N steps; a step is CH dependent
// chains advanced by one v_fma_f32 each (CH = 1: fully dependent stream; CH = 4: four independent chains).
// Output: shader clocks per vector instruction PER SIMD (elapsed / (W * N * CH)).
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)
template <int CH, int KIND>
__global__ void probe(unsigned long long *out, float *sink, float a0, int iters)
{
    float a = a0, b = a0 + 1, c = a0 + 2, d = a0 + 3, e = 0.999f, f = 0.5f;
    typedef float v2 __attribute__((ext_vector_type(2)));
    v2 pa = {a0, a0 + 1}, pb = {a0 + 2, a0 + 3}, pe = {0.999f, 0.999f};
    double da = a0, db = a0 + 1, de = 0.999;
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) { // v_fma_f32
            if (CH == 1) { REP64(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(e), "v"(f));) }
            if (CH == 2) { REP64(asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3" : "+v"(a), "+v"(b) : "v"(e), "v"(f));) }
            if (CH == 4) { REP64(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));) }
        } else if (KIND == 1) { // v_pk_fma_f32
            if (CH == 1) { REP64(asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(pa) : "v"(pe));) }
            if (CH == 2) { REP64(asm volatile("v_pk_fma_f32 %0, %0, %2, %2\n v_pk_fma_f32 %1, %1, %2, %2" : "+v"(pa), "+v"(pb) : "v"(pe));) }
        } else { // v_fma_f64
            if (CH == 1) { REP64(asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(da) : "v"(de));) }
            if (CH == 2) { REP64(asm volatile("v_fma_f64 %0, %0, %2, %2\n v_fma_f64 %1, %1, %2, %2" : "+v"(da), "+v"(db) : "v"(de));) }
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    __syncthreads();
    if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;
    sink[threadIdx.x] = a + b + c + d + pa.x + pa.y + pb.x + pb.y + (float)da + (float)db;
}
template <int CH, int KIND> void run(const char *name)
{
    unsigned long long *d; float *sink;
    hipMalloc(&d, 8 * 64); hipMalloc(&sink, 4096 * 4);
    const int iters = 64;
    printf("%-28s", name);
    for (int W = 1; W <= 4; ++W) { // waves per SIMD (1024 threads per workgroup = 16 waves = 4 per SIMD at most)
        probe<CH, KIND><<<1, 256 * W>>>(d, sink, 1.0f, iters);
        probe<CH, KIND><<<1, 256 * W>>>(d, sink, 1.0f, iters);
        unsigned long long h[16]; hipMemcpy(h, d, 8 * 4 * W, hipMemcpyDeviceToHost);
        unsigned long long mx = 0; for (int i = 0; i < 4 * W; ++i) mx = h[i] > mx ? h[i] : mx;
        printf("  W=%d: %5.2f", W, (double)mx / ((double)W * iters * 64 * CH));
    }
    printf("   clk per instruction per SIMD\n"); fflush(stdout);
    hipFree(d); hipFree(sink);
}
int main()
{
    run<1, 0>("v_fma_f32 dependent");
    run<2, 0>("v_fma_f32 2 chains");
    run<4, 0>("v_fma_f32 4 chains");
    run<1, 1>("v_pk_fma_f32 dependent");
    run<2, 1>("v_pk_fma_f32 2 chains");
    run<1, 2>("v_fma_f64 dependent");
    run<2, 2>("v_fma_f64 2 chains");
    return 0;
}
