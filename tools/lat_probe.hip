// tools/lat_probe.hip -- issue / dependency latencies of a LONE wave64 on gfx950, in shader clocks
// per instruction (s_memtime around 4 x 256 instructions).  Diagnostic only, not part of the product.
//   hipcc --offload-arch=gfx950 -O2 tools/lat_probe.hip -o /tmp/lat_probe && /tmp/lat_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP256(x) REP16(REP16(x))

#define PROBE(name, body, decl)                                                                     \
    __global__ void name(unsigned long long *out, float *sink, float a0)                            \
    {                                                                                                 \
        __shared__ float lds[1024];                                                                   \
        lds[threadIdx.x] = a0;                                                                        \
        float a = a0, b = a0 + 1, c = a0 + 2, d = a0 + 3, e = 1.0001f;                                \
        double da = a0, db = a0 + 1, de = 1.0001;                                                     \
        typedef float v2f __attribute__((ext_vector_type(2)));                                        \
        v2f pa = {a0, a0}, pb = {b, b}, pe = {e, e};                                                  \
        unsigned addr = threadIdx.x * 4;                                                              \
        decl;                                                                                         \
        __syncthreads();                                                                              \
        unsigned long long t0 = __builtin_readcyclecounter();                                         \
        for (int it = 0; it < 4; ++it) { REP256(body) }                                               \
        unsigned long long t1 = __builtin_readcyclecounter();                                         \
        if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;                                              \
        sink[threadIdx.x] = a + b + c + d + (float)da + (float)db + pa.x + pb.y + lds[threadIdx.x];  \
    }

PROBE(k_add_dep, asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(e));, )
PROBE(k_add_2chains, asm volatile("v_add_f32 %0, %0, %2\n v_add_f32 %1, %1, %2" : "+v"(a), "+v"(b) : "v"(e));, )
PROBE(k_add_4chains, asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));, )
PROBE(k_fma_dep, asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a) : "v"(e));, )
PROBE(k_pk_dep, asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(pa) : "v"(pe));, )
PROBE(k_pk_2chains, asm volatile("v_pk_add_f32 %0, %0, %2\n v_pk_add_f32 %1, %1, %2" : "+v"(pa), "+v"(pb) : "v"(pe));, )
PROBE(k_pkmul_dep, asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(pa) : "v"(pe));, )
PROBE(k_f64_dep, asm volatile("v_add_f64 %0, %0, %1" : "+v"(da) : "v"(de));, )
PROBE(k_f64_2chains, asm volatile("v_add_f64 %0, %0, %2\n v_add_f64 %1, %1, %2" : "+v"(da), "+v"(db) : "v"(de));, )
PROBE(k_fma64_dep, asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(da) : "v"(de));, )
PROBE(k_rcp_dep, asm volatile("v_rcp_f32 %0, %0" : "+v"(a));, )
PROBE(k_sqrt_dep, asm volatile("v_sqrt_f32 %0, %0" : "+v"(a));, )
PROBE(k_lds_rt, asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(a) : "v"(addr));, )
PROBE(k_lds_wr_rd, asm volatile("ds_write_b32 %1, %0\n ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "+v"(a) : "v"(addr));, )
PROBE(k_lds_b128_rt, asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(q) : "v"(addr4));, float4 q; unsigned addr4 = (threadIdx.x & 63) * 16)
PROBE(k_readfirstlane, asm volatile("v_readfirstlane_b32 %1, %0\n v_add_f32 %0, %1, %0" : "+v"(a), "=s"(si));, int si = 0)
PROBE(k_cndmask_dep, asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(e) : "vcc");, )
PROBE(k_dpp_shr, asm volatile("s_nop 1\n v_add_f32_dpp %0, %0, %1 wave_shr:1" : "+v"(a) : "v"(e));, )
PROBE(k_barrier, asm volatile("s_barrier");, )

template <typename K> void run(const char *name, K k, int blocks, int threads, int per_body)
{
    unsigned long long *d; float *sink;
    hipMalloc(&d, blocks * 8); hipMalloc(&sink, 4096);
    k<<<blocks, threads>>>(d, sink, 1.0f);
    k<<<blocks, threads>>>(d, sink, 1.0f);
    unsigned long long h[1]; hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    printf("%-18s blocks=%4d threads=%3d : %6.2f clk per instruction (%d per body)\n", name, blocks, threads,
           (double)h[0] / (1024.0 * per_body), per_body);
    hipFree(d); hipFree(sink);
}
#define RUN(k, n) run(#k, k, 1, 64, n); run(#k, k, 1, 256, n); run(#k, k, 1024, 256, n)
int main()
{
    RUN(k_add_dep, 1); RUN(k_add_2chains, 2); RUN(k_add_4chains, 4); RUN(k_fma_dep, 1);
    RUN(k_pk_dep, 1); RUN(k_pk_2chains, 2); RUN(k_pkmul_dep, 1);
    RUN(k_f64_dep, 1); RUN(k_f64_2chains, 2); RUN(k_fma64_dep, 1); RUN(k_rcp_dep, 1); RUN(k_sqrt_dep, 1);
    RUN(k_lds_rt, 1); RUN(k_lds_wr_rd, 1); RUN(k_lds_b128_rt, 1);
    RUN(k_readfirstlane, 2); RUN(k_cndmask_dep, 2); RUN(k_dpp_shr, 1);
    RUN(k_barrier, 1);
    return 0;
}
