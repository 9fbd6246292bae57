// tools/hwid_probe.hip -- where does the dispatcher put the waves of a 256-thread workgroup?
// Prints, for a launch shaped like ns_denoise_pipe_kernel (1024 blocks x 4 waves, 16 KB LDS), the
// (XCC, SE, CU, SIMD, slot) of every wave.  Diagnostic only, not part of the product.
//   hipcc --offload-arch=gfx950 -O2 tools/hwid_probe.hip -o /tmp/hwid_probe && /tmp/hwid_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
__global__ __launch_bounds__(256, 4) void probe(unsigned *out, int spin)
{
    __shared__ float lds[4096];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
    unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);
    float acc = lds[(threadIdx.x * 7) & 4095];
    for (int i = 0; i < spin; ++i) acc = acc * 1.0001f + 0.5f;   // keep every block resident for a while
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = hw;
        out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = xcc + (acc == 123.0f);
    }
}
int main()
{
    const int nb = 1024;
    unsigned *d;
    hipMalloc(&d, nb * 4 * 2 * sizeof(unsigned));
    probe<<<nb, 256>>>(d, 200000);
    std::vector<unsigned> h(nb * 8);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    std::map<unsigned, int> simdOfWave[4];
    std::map<std::vector<unsigned>, int> perSimd;
    int sameSlot = 0;
    for (int b = 0; b < nb; ++b) {
        bool ss = true;
        for (int w = 0; w < 4; ++w) {
            unsigned hw = h[(b * 4 + w) * 2], xcc = h[(b * 4 + w) * 2 + 1] & 15;
            unsigned slot = hw & 15, simd = (hw >> 4) & 3, cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            simdOfWave[w][simd]++;
            perSimd[{xcc, se, sh, cu, simd}]++;
            if (slot != (h[b * 8] & 15)) ss = false;
            if (b < 12 || (b % 256) < 2) printf("block %4d wave %d: xcc %u se %u sh %u cu %2u simd %u slot %u\n", b, w, xcc, se, sh, cu, simd, slot);
        }
        sameSlot += ss;
    }
    for (int w = 0; w < 4; ++w) {
        printf("wave %d -> simd histogram:", w);
        for (auto &kv : simdOfWave[w]) printf(" simd%u:%d", kv.first, kv.second);
        printf("\n");
    }
    std::map<int, int> occ;
    for (auto &kv : perSimd) occ[kv.second]++;
    printf("distinct SIMDs used: %zu; waves-per-SIMD histogram:", perSimd.size());
    for (auto &kv : occ) printf(" %d:%d", kv.first, kv.second);
    printf("\nblocks whose 4 waves share one slot id: %d of %d\n", sameSlot, nb);
    return 0;
}
