/*
 * ns_pipe_kernel.hip -- etsi_denoise over a packed batch, FOUR PIPELINED WAVEFRONTS per utterance.
 *
 * Frames of one utterance are serially dependent, so a batch of N utterances offers
 * only N independent chains: at BASELINE configs[1] (1024 utterances on 1024 SIMDs) a
 * one-wave-per-utterance kernel leaves every SIMD with a single latency-bound wave.  The frame
 * recursion however is a 4-deep software pipeline, and this kernel gives each depth its own wave
 * (workgroup = 256 threads = one utterance; the four waves land on the four SIMDs of a CU):
 *
 *   wave 0  frame f     load int16, zero-frame gate, push into stage-0 buffer, window+rfft+PSD  (FRONT 0)
 *   wave 1  frame f-1   VAD, FilterCalc, mel, gain, IDCT, 17-tap FIR of stage 0                 (BACK 0)
 *   wave 2  frame f-2   window+rfft+PSD of the stage-1 buffer                                   (FRONT 1)
 *   wave 3  frame f-3   stage-1 FilterCalc .. FIR, DC-offset filter, int16 cast, store          (BACK 1)
 *
 * with one s_barrier per frame.  FRONT halves depend only on the sample buffers; all recursive
 * state lives in the registers of waves 1 and 3.  The two 320-sample stage buffers of the
 * reference (NoiseSup.c:98-99) become 8-slot circular buffers of 80-sample frames in LDS that
 * several waves read while one writes the newest slot (slots 0..2 are mirrored behind the end so
 * that every 200- or 96-sample run is contiguous).  Small records (PSD, gain-factor energies,
 * frame validity / tick number) are handed down the pipeline through double-buffered LDS slots.
 *
 * Arithmetic is ns_core.h's, shared with the single-wave kernels: results are identical.
 */
#include "ns_core.h"

namespace sea {

namespace {

constexpr int kSlots = 8;
constexpr int kSlotLen = SEA_HOP;
constexpr int kCirc = kSlots * kSlotLen;   /* 640 */
constexpr int kMirror = 3 * kSlotLen;      /* slots 0..2 repeated behind the end */

struct __attribute__((aligned(16))) Rec01 { /* wave 0 -> wave 1 */
    float psd[68];
    int valid, tick, pad0, pad1;
};
struct __attribute__((aligned(16))) Rec12 { /* wave 1 -> wave 2 */
    float denEn[4];
    int valid, tick, pad0, pad1;
};
struct __attribute__((aligned(16))) Rec23 { /* wave 2 -> wave 3 */
    float psd[68];
    float denEn[4];
    int valid, tick, pad0, pad1;
};

struct __attribute__((aligned(16))) PipeLds {
    float circ[2][kCirc + kMirror]; /* stage-0 / stage-1 sample buffers */
    float work[2][256];             /* FFT workspaces of waves 0 and 2 */
    BackLds back[2];                /* scratch of waves 1 and 3 */
    float outb[80];                 /* wave 3: second-stage output / DC-filtered output */
    Rec01 r01[2];
    Rec12 r12[2];
    Rec23 r23[2];
};

/* start of the 320-sample window "buf[0..319]" of the reference at tick t: buf[240..319] is the
 * frame pushed at tick t (slot t&7), buf[0..79] the one pushed at tick t-3 */
__device__ __forceinline__ int window_base(int tick) { return ((tick - 3) & (kSlots - 1)) * kSlotLen; }

/* lanes 0..39 store samples 2l, 2l+1 of the frame of tick t into its slot (and the mirror) */
__device__ __forceinline__ void slot_store(float *circ, int tick, int lane, float a, float b)
{
    const int slot = tick & (kSlots - 1);
    float *p = circ + slot * kSlotLen + 2 * lane;
    *reinterpret_cast<float2 *>(p) = make_float2(a, b);
    if (slot < 3) *reinterpret_cast<float2 *>(p + kCirc) = make_float2(a, b);
}

/* workgroup barrier usable from role-specialised (wave-uniform) branches: every wave executes the
 * same NUMBER of barriers per frame, at different program counters */
__device__ __forceinline__ void block_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

/* constants a FRONT wave keeps in registers / a BACK wave keeps in registers (each wave loads only
 * its own: the role branches below have separate loops so that register allocation is per role) */
struct FrontConst {
    FftRegs fft;
    float win[4];
};

__device__ __forceinline__ void load_front_const(FrontConst &C, const sea_ns_tables *t, int lane)
{
    load_fft_regs(C.fft, &t->fft, lane);
#pragma unroll
    for (int k = 0; k < 4; ++k) C.win[k] = t->win[k][lane];
}

__device__ __forceinline__ void load_back_const(NsConst &C, const sea_ns_tables *t, int lane)
{
    C.melStart = t->melStart[lane];
    C.melLen = t->melLen[lane];
#pragma unroll
    for (int i = 0; i < SEA_MEL_TAPS; ++i) C.melW[i] = t->melW[i][lane];
#pragma unroll
    for (int f = 0; f < SEA_NMEL; ++f) C.idct[f] = t->idct[f][lane];
    C.irWin = t->irWin[lane];
    C.eps = t->eps;
}

} // namespace

__global__ __launch_bounds__(256, 4) void ns_denoise_pipe_kernel(NsBatchArgs a)
{
    __shared__ PipeLds L;
    const int lane = threadIdx.x & 63;
    const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int u = a.order ? a.order[blockIdx.x] : (int)blockIdx.x;
    const long long off = a.offsets[u];
    const long long nfr = a.lengths[u] / SEA_HOP;
    const long long niter = nfr + 3;

    for (int i = threadIdx.x; i < 2 * (kCirc + kMirror); i += 256) (&L.circ[0][0])[i] = 0.0f;
    if (threadIdx.x < 2) {
        L.r01[threadIdx.x].valid = 0;
        L.r12[threadIdx.x].valid = 0;
        L.r23[threadIdx.x].valid = 0;
    }
    block_sync();

    if (role == 0) {
        /* ---- wave 0: input, zero-frame gate (ParmInterface.c:244-251), FRONT of stage 0 ---- */
        FrontConst C;
        load_front_const(C, a.tables, lane);
        const uint32_t *in32 = reinterpret_cast<const uint32_t *>(a.in + off);
        uint32_t nextw = (lane < 40 && nfr > 0) ? in32[lane] : 0u;
        int tick = 0; /* frames seen since (and including) the first non-zero one */
        for (long long i = 0; i < niter; ++i) {
            const long long f = i;
            if (f < nfr) {
                Rec01 &r = L.r01[f & 1];
                const uint32_t w = nextw;
                if (f + 1 < nfr && lane < 40) nextw = in32[(f + 1) * 40 + lane];
                const bool any = __ballot(w != 0u) != 0ull;
                int valid = 0;
                if (any || tick > 0) {
                    valid = 1;
                    tick++;
                    const float x0 = (float)(short)(w & 0xFFFFu), x1 = (float)(short)(w >> 16);
                    if (lane < 40) slot_store(L.circ[0], tick, lane, x0, x1);
                    wave_sync();
                    if (tick >= 3) /* nbFramesInFirstStage - nbFramesInSecondStage > 2, NoiseSup.c:1152 */
                        ns_front(L.circ[0] + window_base(tick), L.work[0], r.psd, C.fft, C.win, lane);
                }
                if (lane == 0) {
                    r.valid = valid;
                    r.tick = tick;
                }
            }
            block_sync();
        }
    } else if (role == 1) {
        /* ---- wave 1: BACK of stage 0; its 80 outputs enter the stage-1 buffer ---- */
        NsConst C;
        load_back_const(C, a.tables, lane);
        NsRegs s;
        regs_init(s, C.eps);
        for (long long i = 0; i < niter; ++i) {
            const long long f = i - 1;
            if (f >= 0 && f < nfr) {
                const Rec01 &r = L.r01[f & 1];
                Rec12 &o = L.r12[f & 1];
                const int valid = r.valid, t = r.tick;
                if (valid && t >= 3) {
                    float *tmp = L.back[0].sq; /* FIR output staged here, then stored with its mirror */
                    ns_back<0>(r.psd, L.circ[0] + window_base(t), L.back[0], s, C, tmp, lane);
                    if (lane < 40) {
                        const float2 v = *reinterpret_cast<const float2 *>(tmp + 2 * lane);
                        slot_store(L.circ[1], t, lane, v.x, v.y);
                    }
                }
                if (lane == 0) {
                    o.valid = valid;
                    o.tick = t;
                    o.denEn[0] = s.denEn0;
                    o.denEn[1] = s.denEn1;
                    o.denEn[2] = s.denEn2;
                }
            }
            block_sync();
        }
    } else if (role == 2) {
        /* ---- wave 2: FRONT of stage 1 (nbFramesInSecondStage - nbFramesOut > 2 <=> tick >= 5) ---- */
        FrontConst C;
        load_front_const(C, a.tables, lane);
        for (long long i = 0; i < niter; ++i) {
            const long long f = i - 2;
            if (f >= 0 && f < nfr) {
                const Rec12 &r = L.r12[f & 1];
                Rec23 &o = L.r23[f & 1];
                const int valid = r.valid, t = r.tick;
                if (valid && t >= 5)
                    ns_front(L.circ[1] + window_base(t), L.work[1], o.psd, C.fft, C.win, lane);
                if (lane == 0) {
                    o.valid = valid;
                    o.tick = t;
                    o.denEn[0] = r.denEn[0];
                    o.denEn[1] = r.denEn[1];
                    o.denEn[2] = r.denEn[2];
                }
            }
            block_sync();
        }
    } else {
        /* ---- wave 3: BACK of stage 1, DC-offset filter, int16 cast, store ---- */
        NsConst C;
        load_back_const(C, a.tables, lane);
        NsRegs s;
        regs_init(s, C.eps);
        uint32_t *out32 = reinterpret_cast<uint32_t *>(a.out + off);
        float *outf = a.out_f32 ? a.out_f32 + off : nullptr;
        int firstOut = -1;
        for (long long i = 0; i < niter; ++i) {
            const long long f = i - 3;
            if (f >= 0 && f < nfr) {
                const Rec23 &r = L.r23[f & 1];
                const int valid = r.valid, t = r.tick;
                bool produced = false;
                if (valid && t >= 5) {
                    s.denEn0 = r.denEn[0];
                    s.denEn1 = r.denEn[1];
                    s.denEn2 = r.denEn[2];
                    ns_back<1>(r.psd, L.circ[1] + window_base(t), L.back[1], s, C, L.outb, lane);
                    /* DCOffsetFil (NoiseSup.c:182-198): differences in parallel, then the recurrence */
                    const float xm1 = (lane == 0) ? s.dcX : L.outb[lane - 1];
                    const float d0 = L.outb[lane] - xm1;
                    float d1 = 0.0f;
                    if (lane < 16) d1 = L.outb[64 + lane] - L.outb[63 + lane];
                    s.dcX = L.outb[79];
                    wave_sync();
                    L.back[1].sq[lane] = d0;
                    if (lane < 16) L.back[1].sq[64 + lane] = d1;
                    wave_sync();
                    dc_filter(L.back[1].sq, L.outb, s.dcY, lane);
                    produced = true;
                    if (firstOut < 0) firstOut = (int)f;
                }
                /* what etsi_denoise copies out for this frame (AdvFrontEnd.c:186-190): zeros until
                 * the first NoiseSup output; float -> int16 is the bare cast of ParmInterface.c:266 */
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
            block_sync();
        }
        if (a.first_out && lane == 0) a.first_out[u] = firstOut;
    }
}

} // namespace sea
