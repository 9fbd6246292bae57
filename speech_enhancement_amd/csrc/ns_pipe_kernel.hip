/*
 * ns_pipe_kernel.hip -- etsi_denoise over a packed batch, FOUR PIPELINED WAVEFRONTS per utterance.
 *
 * Frames of one utterance are serially dependent, so a batch of N utterances offers
 * only N independent chains: at BASELINE configs[1] (1024 utterances on 1024 SIMDs) a
 * one-wave-per-utterance kernel leaves every SIMD with a single latency-bound wave.  The frame
 * recursion however is a software pipeline, and this kernel gives each depth its own wave
 * (workgroup = 256 threads = one utterance, one wave per SIMD of a CU), one s_barrier per frame:
 *
 *   wave F   iteration i: load int16 frame i, zero-frame gate, push into the stage-0 buffer; then
 *            BOTH front halves side by side in one dual transform (lanes 0..31 / 32..63):
 *            window+rfft+PSD of stage 0 for frame i and of stage 1 for frame i-2
 *   wave B0  frame i-1   stage-0 FilterCalc, mel, IDCT, 17-tap FIR  -> stage-1 buffer
 *   wave B1  frame i-3   stage-1 FilterCalc, gain factorisation, mel, IDCT (its FIR runs in S)
 *   wave S   the lane-redundant scalar chains that need no lane parallelism and are either
 *            input-only or deferrable:  VAD frame log-energy of the frame pushed at i-1 (consumed
 *            by B0 two frames later), in-order sum of denSigSE1 of frame i-2 (consumed by B1),
 *            stage-1 FIR, DC-offset recurrence + int16 cast + store of frame i-4.
 *
 * Front halves depend only on the sample buffers; all recursive state lives in the registers of B0,
 * B1 and S.  The two 320-sample stage buffers of the reference (NoiseSup.c:98-99) become 8-slot
 * circular buffers of 80-sample frames in LDS that several waves read while one writes the newest
 * slot (slots 0..2 are mirrored behind the end so that every 200- or 96-sample run is contiguous).
 * Per-frame records travel down the pipeline through double-buffered LDS slots; values with a
 * longer life (frame energies, denSigSE1 sums) sit in 8-entry rings indexed by tick number.
 *
 * Arithmetic is ns_core.h's, shared with the single-wave kernels: results are identical.
 */
#include "ns_core.h"

namespace sea {

namespace p4 { /* everything of the four-wave form */

constexpr int kSlots = 8;
constexpr int kSlotLen = SEA_HOP;
constexpr int kCirc = kSlots * kSlotLen;   /* 640 */
constexpr int kMirror = 3 * kSlotLen;      /* slots 0..2 repeated behind the end */
constexpr int kPipeWaves = 4;

/* timing-only diagnostic: bit set = that piece of work is done (default all) */
/* 1: the second-stage FIR runs in the helper wave S instead of B1 (shorter lone-workgroup frame period,
 * slightly more instructions) */
#ifndef SEA_FIR_IN_S
#define SEA_FIR_IN_S 1
#endif
/* 1 (experiment, off): the second-stage mel IDCT (DoMelIDCT: nine in-order sums of 25 terms) SPLIT between B1, the
 * longest role, and the transform wave F, which has ~1000 clk of slack per frame: B1 adds the first SEA_IDCT_SPLIT
 * terms, F the rest one beat later, then windows and mirrors the taps (one more beat of pipeline depth).  Bit-identical
 * and measured: the role timers of the top-priority workgroup balance (whole IDCT in B1: F 3140 / B1 4100 clk per frame;
 * 17 terms in B1: F 3650 / B1 3825, frame period 4310 -> 4060), but the bench step gets SLOWER for every split
 * (2.27-2.35 ms against 2.19-2.23): the extra beat and LDS traffic cost the lower launch rows more than the top row
 * gains (profiles/r03_ns_idct_split_experiment.txt). */
#ifndef SEA_IDCT_IN_F
#define SEA_IDCT_IN_F 0
#endif
#ifndef SEA_IDCT_SPLIT
#define SEA_IDCT_SPLIT 13
#endif
/* > 0: the same split with the TAIL in the helper wave S, which applies the taps anyway one beat after B1 (no extra
 * beat, no extra record): B1 adds the first 25 - SEA_IDCT_TAIL_S terms, S the last SEA_IDCT_TAIL_S, windows and mirrors */
#ifndef SEA_IDCT_TAIL_S
#define SEA_IDCT_TAIL_S 8 /* round 4 (LDS-free taps, LRPT priorities): 4: 2.14 ms, 8: 2.11, 12: 2.11, 16: 2.13; round 3, measured on configs[1], alternating A/B (tools/ns_ab.sh): 0: 2.19-2.24 ms, 4: 2.14-2.19, 7: 2.16-2.19, 10: 2.17-2.18 */
#endif
/* 1 (experiment, off): the int16 cast and the output store (ParmInterface.c:266) of a frame run in the transform wave F one
 * beat after the helper wave finished it (F has ~900 clk of slack per frame, S none): double-buffered output frame in
 * LDS.  Bit-identical; measured on configs[1], alternating A/B: 2.29-2.31 ms against 2.20-2.22 (also with 6 or 8 IDCT
 * terms moved to S on top): the extra beat costs the lower launch rows more than the top row gains, as with the
 * IDCT split in F. */
#ifndef SEA_STORE_IN_F
#define SEA_STORE_IN_F 0
#endif
constexpr bool kIdctSplit = (SEA_IDCT_IN_F || SEA_IDCT_TAIL_S > 0) && SEA_FIR_IN_S;
constexpr int kIdctHead = SEA_IDCT_IN_F ? SEA_IDCT_SPLIT : 25 - SEA_IDCT_TAIL_S;
constexpr int kLagS = (SEA_IDCT_IN_F && SEA_FIR_IN_S) ? 5 : 4; /* beats between a frame's intake and its output store */
constexpr int kRec34 = (SEA_IDCT_IN_F && SEA_FIR_IN_S) ? 4 : 2;    /* with the split: B1 writes at beat i, F at i + 1, S reads at i + 2 */
/* 1 (four-wave forms with the address tables in VGPRs): the filter taps reach the 17-tap filters as scalar operands through
 * v_readlane and the filters' outputs stay in registers (B0: straight into the stage-1 buffer; S: straight into the DC filter's
 * differences), the helper wave fetches the operands of its exactness check and of the output store in one batch -- two LDS
 * round trips (~200 clk each) less on B0's chain, two less on S's */
#ifndef SEA_TAPS_RL
#define SEA_TAPS_RL 1
#endif
/* 1 (four-wave forms with the address tables in VGPRs): the two BACK waves keep their column of the 9 x 25 IDCT basis in
 * registers (the kernel's register budget is set by the transform wave's address tables, these roles have room) instead
 * of 25 + 21 LDS reads of constants per frame */
#ifndef SEA_BASIS_REGS
#define SEA_BASIS_REGS 1
#endif
/* 1: issue priority by REMAINING frames instead of by launch row (longest-remaining-processing-time first, the makespan
 * rule).  Every SEA_PRIO_STEP frames each wave of a workgroup sets s_setprio from
 *     L = SEA_PRIO_LEVELS * (frames its utterance has left) / (frames of the batch's longest utterance) + SEA_PRIO_ROWBIAS * launch row
 * dithered over SEA_PRIO_LEVELS / 4 consecutive evaluations into the four hardware levels: level (L + d) / (LEVELS / 4) with
 * d = 0 .. LEVELS / 4 - 1 in turn.  An utterance keeps the top level only while more of it is left than of its neighbours, so the
 * utterances of a CU converge on a common finishing time whatever their lengths.  (The static rows gave the longest utterance of
 * a CU its lone frame period from start to end and starved the third row: rows 0 / 1 finished at 2.09 / 2.15 ms, row 2 at 2.39,
 * tools/ns_finish_order.py; 2.24 -> 2.15 ms per configs[1] step, profiles/r04_ns_priority_and_lone_period.txt.)  The launch row
 * enters because among equal levels the hardware prefers the oldest wave.  Whole utterances in the four-wave form only. */
#ifndef SEA_PRIO_LRPT
#define SEA_PRIO_LRPT 1
#endif
#ifndef SEA_PRIO_STEP
#define SEA_PRIO_STEP 16
#endif
#ifndef SEA_PRIO_LEVELS
#define SEA_PRIO_LEVELS 32 /* 4 (no dither to speak of): 2.21 ms; 16: 2.17; 32: 2.15; 64 with step 8: 2.16 */
#endif
#ifndef SEA_PRIO_ROWBIAS
#define SEA_PRIO_ROWBIAS 1
#endif
#ifndef SEA_PRIO_DITHER
#define SEA_PRIO_DITHER 1
#endif
/* waves per SIMD the register allocation must leave room for (= workgroups per CU of this 4-wave kernel) */
#ifndef SEA_NS_MIN_WAVES
#define SEA_NS_MIN_WAVES 4
#endif
#ifndef SEA_ROLE_MASK
#define SEA_ROLE_MASK 127
#endif
/* timing-only ablations of the helper wave (results wrong by construction): 1 no second-stage FIR, 2 no VAD log,
 * 4 no output store, 8 no chains */
#ifndef SEA_ABL_S
#define SEA_ABL_S 0
#endif
/* timing-only ablation (results wrong by construction): the frame barrier on every second beat only -- no gain on configs[1]
 * or on the configs[4] shard: synchronising less often than once per frame is not what this pipeline lacks */
#ifndef SEA_ABL_HALFSYNC
#define SEA_ABL_HALFSYNC 0
#endif


/* timing-only diagnostic (-DSEA_NS_TIMING): shader-clock cycles each role of workgroup 0 spends
 * working / waiting at the frame barrier -> g_ns_timing[role*2 + {0,1}] */
#ifdef SEA_NS_TIMING
__device__ unsigned long long g_ns_timing[24]; /* [8..15]: checkpoints inside S */
__device__ unsigned g_ns_hw[4096 * 4];           /* per workgroup and role: HW_ID of the wave (SIMD placement, tools/ns_placement.py) */
__device__ unsigned g_ns_wg[4096 * 4];           /* per workgroup: frame-loop span in 10 ns ticks, HW_ID, XCC_ID, start tick (low 32 bits) */
struct RoleTimer {
    unsigned long long work = 0, wait = 0, t0 = 0, t1 = 0;
    __device__ __forceinline__ void begin() { t0 = clock64(); }
    __device__ __forceinline__ void mid() { t1 = clock64(); work += t1 - t0; }
    __device__ __forceinline__ void end() { wait += clock64() - t1; }
    __device__ __forceinline__ void flush(int slot)
    {
        if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) {
            g_ns_timing[slot] = work;
            g_ns_timing[slot + 1] = wait;
        }
    }
};
#define NS_T_DECL RoleTimer rt_
#define NS_T_BEGIN rt_.begin()
#define NS_T_MID rt_.mid()
#define NS_T_END rt_.end()
#define NS_T_FLUSH(slot) rt_.flush(slot)
#ifdef SEA_NS_TIMING_NOCK /* role totals only: the checkpoints inside a role cost ~100 clk each and move the schedule */
#define NS_T_CK(k)
#define NS_T_CK_DECL unsigned long long ck_[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define NS_T_CK_START
#else
#define NS_T_CK(k) do { const unsigned long long c_ = clock64(); ck_[k] += c_ - ckt_; ckt_ = c_; } while (0)
#define NS_T_CK_DECL unsigned long long ck_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ckt_ = 0
#define NS_T_CK_START ckt_ = clock64()
#endif
#define NS_T_CK_FLUSH do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) for (int q_ = 0; q_ < 8; ++q_) g_ns_timing[8 + q_] = ck_[q_]; } while (0)
#else
#define NS_T_CK(k)
#define NS_T_CK_DECL
#define NS_T_CK_START
#define NS_T_CK_FLUSH
#define NS_T_DECL
#define NS_T_BEGIN
#define NS_T_MID
#define NS_T_END
#define NS_T_FLUSH(slot)
#endif

/* F -> B0, S (r01) and F -> B1 (r23) */
struct __attribute__((aligned(16))) RecPsd {
    float psd[68];
    int valid, tick, pad0, pad1;
};
struct __attribute__((aligned(16))) Rec12 { /* B0 -> F, S */
    float den[68]; /* denSigSE1 of this tick, summed in order by S */
    int valid, tick, pad0, pad1;
};
struct __attribute__((aligned(16))) RecFd { /* S -> F, same variant: the measures' three sums of the frame S summed at this beat */
    float mean, var, tempEn, m1, m2, m3;
    int nb16, vadns, tick, pad0, pad1, pad2; /* tick 0: the first stage did not run */
};
struct __attribute__((aligned(16))) Rec34 { /* B1 -> F -> S */
    float mel[kIdctSplit ? 40 : 4]; /* split IDCT: [k..24] second-stage mel gains (gain factor applied) whose IDCT terms F still has to
                    * add, [28..36] B1's partial sums of rows 0..8 (ns_idct_head / ns_idct_tail) */
    float fir[20]; /* SEA_FIR_IN_S: the 17 taps of the second-stage filter, S applies them */
    float out[80]; /* otherwise: second-stage filter output before the DC-offset filter */
    int produced, tick, pad1, pad2;
};

template <bool ADDR_LDS, bool FD = false>
struct __attribute__((aligned(16))) PipeLds {
    float circ[2][kCirc + kMirror]; /* stage-0 / stage-1 sample buffers */
    float work[512];                /* the two FFT frames of F */
    uint4 fftAddr[ADDR_LDS ? SEA_FFT_LSTAGES * 64 : 1]; /* F's butterfly operand addresses (Fft2Regs) */
    BackLds back[2];                /* scratch of B0 and B1 */
    float ssq[80], sdif[80], sout[SEA_STORE_IN_F ? 2 : 1][80]; /* scratch of S (sout: the DC-filtered output frame) */
    int outProd[2];                 /* SEA_STORE_IN_F: frame fo & 1 holds an output (1) / is a latency frame to be zero-filled (0) */
    float szero[4];                 /* zeros: what the shorter chain reads past its end */
    float sfir[80];                 /* second-stage filter output before the DC-offset filter */
    float frameEn[kSlots];          /* 64 + in-order sum of squares of the VAD's frame for tick t at [t & 7] */
    float denSum[kSlots];           /* sum of denSigSE1 of tick t at [t & 7] */
    int fdFlags[kSlots];            /* speech flags of tick t at [t & 7] (frame-dropping VAD variant) */
    float idctT[SEA_NMEL * 16];     /* mel-IDCT basis rows 0..8: [f][16], shared by B0 and B1 */
    RecPsd r01[2];
    Rec12 r12[2];
    RecPsd r23[2];
    Rec34 r34[kRec34];
    /* frame-dropping VAD variant only, BEHIND everything else (the other forms keep their layout): what B0 leaves of frame f
     * (by parity) for the speech measures (ns_core.h, kFdRecFloats), and their sums on the way from S to F */
    float fdRec[2][FD ? kFdRecFloats : 4];
    RecFd rfd[FD ? 2 : 1];
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
 * same NUMBER of barriers per frame, at different program counters.  The fences cover LDS only
 * (everything the waves exchange lives there), so global prefetches and stores stay in flight
 * across the barrier instead of being drained (vmcnt(0)) once per frame. */
__device__ __forceinline__ void block_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

/* constants a BACK wave keeps in registers (each wave loads only its own: the role branches below
 * have separate loops so that register allocation is per role) */
template <bool BASIS = false>
__device__ __forceinline__ void load_back_const(NsConst &C, const sea_ns_tables *t, int lane)
{
    if (BASIS) {
#pragma unroll
        for (int f = 0; f < SEA_NMEL; ++f) C.idct[f] = t->idct[f][lane <= 8 ? lane : 8];
    }
    C.melStart = t->melStart[lane];
    C.melLen = t->melLen[lane];
#pragma unroll
    for (int i = 0; i < SEA_MEL_TAPS; ++i) C.melW[i] = t->melW[i][lane];
    C.irWin = t->irWin[lane];
    C.eps = t->eps;
}

template <bool FD, bool ADDR_LDS, bool SLICES = false>
__device__ __forceinline__ void ns_pipe_body(const NsBatchArgs &a, PipeLds<ADDR_LDS, FD> &L)
{
    const int lane = threadIdx.x & 63;
    const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int u = a.order ? a.order[blockIdx.x] : (int)blockIdx.x;
    /* Issue priority.  Whole utterances in the four-wave form: by remaining frames (SEA_PRIO_LRPT above, prio_by_remaining below).
     * Otherwise (time slices, which hold the same frame range of every utterance) by launch row: the launch order puts the longest
     * utterances first, in rows of one workgroup per CU; row 0 gets s_setprio 3, row 1 -> 2, row 2 -> 1.  prio_row = 0: off. */
    constexpr bool kLrptForm = SEA_PRIO_LRPT && !ADDR_LDS && !SLICES; /* whole utterances in the four-wave form */
    const bool lrpt = kLrptForm && a.prio_row > 0 && a.order && !a.state;
    if (a.prio_row > 0 && !lrpt) {
        const int row = a.prio_base + (int)blockIdx.x / a.prio_row;
        if (row == 0) __builtin_amdgcn_s_setprio(3);
        else if (row == 1) __builtin_amdgcn_s_setprio(2);
        else if (row == 2) __builtin_amdgcn_s_setprio(1);
    }
    const long long off = a.offsets[u];
    const long long nfr = a.lengths[u] / SEA_HOP;
    /* the batch's longest utterance is block 0's (the launch order is longest first; any other order only makes the rule less sharp) */
    const long long longestFr = lrpt ? a.lengths[a.order[0]] / SEA_HOP : 0;
    const float lrptScale = (float)SEA_PRIO_LEVELS / (float)(longestFr > 0 ? longestFr : 1);
    const int lrptBias = lrpt ? SEA_PRIO_ROWBIAS * ((int)blockIdx.x / a.prio_row) : 0; /* equal levels: the hardware prefers the oldest wave */
    auto prio_by_remaining = [&](long long i) {
        if (kLrptForm && lrpt && (i & (SEA_PRIO_STEP - 1)) == 0) {
            const int L = __builtin_amdgcn_readfirstlane((int)((float)(nfr - i) * lrptScale)) + lrptBias;
            const int d = SEA_PRIO_DITHER ? (int)((i / SEA_PRIO_STEP) & (SEA_PRIO_LEVELS / 4 - 1)) : 0;
            const int pr = (L + d) / (SEA_PRIO_LEVELS / 4);
            if (pr >= 3) __builtin_amdgcn_s_setprio(3);
            else if (pr == 2) __builtin_amdgcn_s_setprio(2);
            else if (pr == 1) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
        }
    };
    const long long niter = nfr + kLagS + (SEA_STORE_IN_F ? 1 : 0);

    /* time slices (NsBatchArgs::state): the recursion of utterance u between two launches */
    float *const blob = (SLICES && !FD && a.state) ? a.state + (size_t)u * kNsPipeStateFloats : nullptr;
    const bool resume = blob && a.resume;
    constexpr int kBlobLane = 2 * kCirc, kBlobRing = kBlobLane + 12 * 64, kBlobScal = kBlobRing + 3 * kSlots;
    if (resume) { /* the two stage buffers with their mirrors, the tick-indexed rings */
        for (int i = threadIdx.x; i < 2 * (kCirc + kMirror); i += 64 * kPipeWaves) {
            const int st = i / (kCirc + kMirror), x = i - st * (kCirc + kMirror);
            L.circ[st][x] = blob[st * kCirc + (x < kCirc ? x : x - kCirc)];
        }
    } else {
        for (int i = threadIdx.x; i < 2 * (kCirc + kMirror); i += 64 * kPipeWaves) (&L.circ[0][0])[i] = 0.0f;
    }
    for (int i = threadIdx.x; i < SEA_NMEL * 16; i += 64 * kPipeWaves) L.idctT[i] = a.tables->idct[i >> 4][i & 15];
    if (threadIdx.x < 4) L.szero[threadIdx.x] = 0.0f;
    if (threadIdx.x < kSlots) {
        L.frameEn[threadIdx.x] = resume ? blob[kBlobRing + threadIdx.x] : 0.0f;
        L.denSum[threadIdx.x] = resume ? blob[kBlobRing + kSlots + threadIdx.x] : 0.0f;
    }
    if (threadIdx.x < 2) {
        L.r01[threadIdx.x].valid = 0;
        L.r12[threadIdx.x].valid = 0;
        L.r12[threadIdx.x].den[65] = L.r12[threadIdx.x].den[66] = L.r12[threadIdx.x].den[67] = 0.0f; /* read as zeros by S */
        L.r23[threadIdx.x].valid = 0;
    }
    if (threadIdx.x < kRec34) L.r34[threadIdx.x].produced = 0;
    if (FD && threadIdx.x < 2) {
        L.rfd[threadIdx.x].tick = 0;
        L.fdRec[threadIdx.x][153] = L.fdRec[threadIdx.x][154] = L.fdRec[threadIdx.x][155] = 0.0f; /* the mel chain's zeros */
    }
    block_sync();

    NS_T_DECL;
#ifdef SEA_NS_TIMING
    if (blockIdx.x < 4096 && lane == 0)
        g_ns_hw[blockIdx.x * 4 + role] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4) | ((__builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20) & 0xFu) << 28);
#endif
    if (role == 0) {
        /* ---- F: input + zero-frame gate (ParmInterface.c:244-251); front halves of both stages ---- */
        Fft2Regs fft;
        load_fft2_regs<ADDR_LDS>(fft, &a.tables->fft, lane, L.fftAddr);
        wave_sync();
        float win8[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) win8[k] = a.tables->win8[k][lane];
        const float irWin = a.tables->irWin[lane];
        const uint32_t *in32 = reinterpret_cast<const uint32_t *>(a.in + off);
        uint32_t nextw = (lane < 40 && nfr > 0) ? in32[lane] : 0u;
        int tick = resume ? __float_as_int(blob[kBlobScal + 0]) : 0; /* frames seen since (and including) the first non-zero one */
        int onset = (int)nfr;
        /* The intake of a frame (zero-frame gate, int16 -> float, store into its slot of the stage-0 buffer)
         * runs at the BOTTOM of the previous iteration, when its words (requested one iteration earlier still)
         * have long arrived: the new slot is outside every window read during that iteration, and the
         * transform's window loads at the top of the next one no longer wait for these LDS stores.
         * (valid, tick) of frames i, i-1, i-2 stay in registers: B0 only copies them from Rec01 to Rec12. */
        int vCur = 0, tCur = 0, v1 = 0, t1 = 0, v2 = 0, t2 = 0;
        NsFd fdF; /* FD: the frame-dropping VAD's measures (SpeechQVar / Spec / Mel) live here */
        fd_init(fdF);
        auto intake = [&](long long f) {
            /* the 80-VGPR form has no register left for the lane id across the transform: the allocator would park it
             * (and 8 * lane) in scratch and reload both every frame; two v_mbcnt recompute it instead */
            int ln = lane;
            if (ADDR_LDS) asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
            const uint32_t w = nextw;
            if (f + 1 < nfr && ln < 40) nextw = in32[(f + 1) * 40 + ln];
            const bool any = __ballot(w != 0u) != 0ull;
            vCur = 0;
            if (any || tick > 0) {
                vCur = 1;
                if (FD && tick == 0) onset = (int)f;
                tick++;
                const float x0 = (float)(short)(w & 0xFFFFu), x1 = (float)(short)(w >> 16);
                if (ln < 40) slot_store(L.circ[0], tick, ln, x0, x1);
            }
            tCur = tick;
        };
        if (nfr > 0) intake(0);
        NS_T_CK_DECL;
#ifdef SEA_NS_TIMING
        const unsigned long long clk0_ = clock64(), wall0_ = wall_clock64();
#endif
        uint32_t *out32F = reinterpret_cast<uint32_t *>(a.out + off);
        float *outfF = a.out_f32 ? a.out_f32 + off : nullptr;
        for (long long i = 0; i < niter; ++i) {
            NS_T_BEGIN;
            prio_by_remaining(i);
            NS_T_CK_START;
            if (SEA_STORE_IN_F) { /* cast + store of the frame the helper wave finished one beat ago */
                const long long fs = i - kLagS - 1;
                if (fs >= 0 && fs < nfr) {
                    int ln = lane;
                    if (ADDR_LDS) asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
                    const int prod = L.outProd[fs & 1];
                    if (ln < 40) {
                        uint32_t packed = 0u;
                        if (prod) {
                            const float2 v = *reinterpret_cast<const float2 *>(&L.sout[fs & 1][2 * ln]);
                            packed = (uint32_t)cast_i16(v.x) | ((uint32_t)cast_i16(v.y) << 16);
                            if (outfF) *reinterpret_cast<float2 *>(outfF + fs * SEA_HOP + 2 * ln) = v;
                        }
                        out32F[fs * 40 + ln] = packed;
                    }
                }
            }
            /* stage 0, frame i */
            bool actA = false;
            auto &rA = L.r01[i & 1];
            if (i < nfr) {
                /* nbFramesInFirstStage - nbFramesInSecondStage > 2 (NoiseSup.c:1152) <=> tick >= 3 */
                actA = (SEA_ROLE_MASK & 1) && vCur && tCur >= 3;
                if (lane == 0) {
                    rA.valid = vCur;
                    rA.tick = tCur;
                }
            }
            /* stage 1, frame i-2: nbFramesInSecondStage - nbFramesOut > 2 (NoiseSup.c:1178) <=> tick >= 5 */
            const long long fB = i - 2;
            bool actB = false;
            const int tA = tCur, tB = t2;
            auto &rB = L.r23[fB & 1];
            if (fB >= 0 && fB < nfr) {
                actB = (SEA_ROLE_MASK & 1) && v2 && tB >= 5;
                if (lane == 0) {
                    rB.valid = v2;
                    rB.tick = tB;
                }
            }
            NS_T_CK(5);
            if (actA || actB) {
                wave_sync();
                ns_front_dual<ADDR_LDS>(L.circ[0] + window_base(tA), actA, rA.psd, L.circ[1] + window_base(tB), actB, rB.psd,
                              L.work, fft, win8, lane);
            }
            NS_T_CK(6);
            if (SEA_IDCT_IN_F && SEA_FIR_IN_S) { /* the taps of the frame B1 finished one beat ago */
                const long long fg = i - 4;
                if (fg >= 0 && fg < nfr) {
                    Rec34 &g = L.r34[fg & (kRec34 - 1)];
                    if (g.produced) ns_idct_tail<SEA_IDCT_SPLIT>(g.mel, L.idctT, irWin, g.fir, lane);
                }
            }
            if (FD) { /* the scalar logic of the speech measures of the frame whose sums S finished one beat ago */
                const long long ff = i - 3;
                if (ff >= 0 && ff < nfr) {
                    const RecFd &q = L.rfd[ff & 1];
                    const int tf = q.tick;
                    if (tf > 0) {
                        int bits = fd_var_sums<64>(fdF, uniform_f(q.mean), uniform_f(q.var), q.nb16);
                        bits |= fd_spec_mel_sums(fdF, uniform_f(q.tempEn), q.m1, q.m2, q.m3, q.nb16) << 1;
                        bits |= q.vadns ? 8 : 0;
                        if (lane == 0) L.fdFlags[tf & (kSlots - 1)] = bits; /* bit 0 Var, 1 Spec, 2 Mel, 3 VADNS */
                    }
                }
            }
            v2 = v1, t2 = t1, v1 = vCur, t1 = tCur;
            vCur = 0;
            if (i + 1 < nfr) intake(i + 1);
            NS_T_MID;
            if (!SEA_ABL_HALFSYNC || (i & 1)) block_sync();
            NS_T_END;
        }
        if (FD && a.onset_out && lane == 0) a.onset_out[u] = onset;
        if (blob && lane == 0) blob[kBlobScal + 0] = __int_as_float(tick);
        NS_T_FLUSH(0);
#ifdef SEA_NS_TIMING
        if (blockIdx.x < 4096 && lane == 0) {
            const unsigned long long w1_ = wall_clock64();
            g_ns_wg[blockIdx.x * 4 + 0] = (unsigned)(w1_ - wall0_);
            g_ns_wg[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
            g_ns_wg[blockIdx.x * 4 + 2] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);
            g_ns_wg[blockIdx.x * 4 + 3] = (unsigned)wall0_;
        }
        if (blockIdx.x == 0 && lane == 0) {
            g_ns_timing[16] = ck_[5];
            g_ns_timing[17] = ck_[6];
            g_ns_timing[18] = clock64() - clk0_;        /* shader clocks over the whole frame loop ... */
            g_ns_timing[19] = wall_clock64() - wall0_;  /* ... and the same span on the constant 100 MHz counter */
        }
#endif
    } else if (role == 1) {
        /* ---- B0: BACK of stage 0; its 80 outputs enter the stage-1 buffer ---- */
        NsConst C;
        constexpr bool kBregs = SEA_BASIS_REGS && !ADDR_LDS && SEA_TAPS_RL;
        load_back_const<kBregs>(C, a.tables, lane);
        NsRegs s;
        regs_init(s, C.eps);
        NsFd fd;
        fd_init(fd);
        if (resume) {
            const float *p = blob + kBlobLane + lane, *q = blob + kBlobScal;
            s.noiseLo[0] = p[0 * 64]; s.noiseHi[0] = p[1 * 64]; s.denLo[0] = p[2 * 64];
            s.denHi[0] = p[3 * 64]; s.prevLo[0] = p[4 * 64]; s.prevHi[0] = p[5 * 64];
            s.nbFrame[0] = __float_as_int(q[1]); s.meanEn = q[2]; s.flagVAD = __float_as_int(q[3]);
            s.hangOver = __float_as_int(q[4]); s.nbSpeech = __float_as_int(q[5]); s.psdOk[0] = __float_as_int(q[6]);
        }
        for (long long i = 0; i < niter; ++i) {
            NS_T_BEGIN;
            prio_by_remaining(i);
            const long long f = i - 1;
            if (f >= 0 && f < nfr) {
                const auto &r = L.r01[f & 1];
                Rec12 &o = L.r12[f & 1];
                const int valid = r.valid, t = r.tick;
                if ((SEA_ROLE_MASK & 2) && valid && t >= 3) {
                    float *tmp = L.back[0].sq; /* FIR output staged here, then stored with its mirror */
                    int bits = 0;
                    /* the helper wave leaves the frame's in-order sum of squares; the log-energy (NoiseSup.c:391) is taken
                     * here, by its consumer: this wave has ~1000 clk of slack per frame, the helper wave none */
                    /* FD: the speech measures' inputs go into o.fd; S sums them in free lanes of its chain one beat later, F
                     * runs their scalar logic the beat after (this wave has no slack left for ~1300 clk of them) */
                    constexpr bool kRegs = SEA_TAPS_RL && !ADDR_LDS; /* (the table-in-LDS form gains nothing from any of it: 464 M frames/s on the configs[4] shard either way) */
                    float y01[2] = {0.0f, 0.0f};
                    ns_back<0, true, FD, false, !ADDR_LDS>(r.psd, L.circ[0] + window_base(t), L.back[0], s, C, tmp, lane,
                                         vad_frame_energy(L.frameEn[t & (kSlots - 1)]), o.den, L.idctT, &fd, &bits,
                                         FD ? L.fdRec[f & 1] : nullptr, kRegs ? y01 : nullptr, kBregs ? C.idct : nullptr);
                    if (lane < 40) {
                        if (kRegs) {
                            slot_store(L.circ[1], t, lane, y01[0], y01[1]);
                        } else {
                            const float2 v = *reinterpret_cast<const float2 *>(tmp + 2 * lane);
                            slot_store(L.circ[1], t, lane, v.x, v.y);
                        }
                    }
                }
                if (lane == 0) {
                    o.valid = valid;
                    o.tick = t;
                }
            }
            NS_T_MID;
            if (!SEA_ABL_HALFSYNC || (i & 1)) block_sync();
            NS_T_END;
        }
        if (blob) {
            float *p = blob + kBlobLane + lane, *q = blob + kBlobScal;
            p[0 * 64] = s.noiseLo[0]; p[1 * 64] = s.noiseHi[0]; p[2 * 64] = s.denLo[0];
            p[3 * 64] = s.denHi[0]; p[4 * 64] = s.prevLo[0]; p[5 * 64] = s.prevHi[0];
            if (lane == 0) {
                q[1] = __int_as_float(s.nbFrame[0]); q[2] = s.meanEn; q[3] = __int_as_float(s.flagVAD);
                q[4] = __int_as_float(s.hangOver); q[5] = __int_as_float(s.nbSpeech); q[6] = __int_as_float(s.psdOk[0]);
            }
        }
        NS_T_FLUSH(2);
    } else if (role == 2) {
        /* ---- B1: BACK of stage 1 ---- */
        NsConst C;
        constexpr bool kBregs = SEA_BASIS_REGS && !ADDR_LDS && kIdctSplit;
        load_back_const<kBregs>(C, a.tables, lane);
        NsRegs s;
        regs_init(s, C.eps);
        if (resume) {
            const float *p = blob + kBlobLane + lane, *q = blob + kBlobScal;
            s.noiseLo[1] = p[6 * 64]; s.noiseHi[1] = p[7 * 64]; s.denLo[1] = p[8 * 64];
            s.denHi[1] = p[9 * 64]; s.prevLo[1] = p[10 * 64]; s.prevHi[1] = p[11 * 64];
            s.nbFrame[1] = __float_as_int(q[7]); s.lowSNRtrack = q[8]; s.alfaGF = q[9]; s.psdOk[1] = __float_as_int(q[10]);
        }
        for (long long i = 0; i < niter; ++i) {
            NS_T_BEGIN;
            prio_by_remaining(i);
            const long long f = i - 3;
            if (f >= 0 && f < nfr) {
                const auto &r = L.r23[f & 1];
                Rec34 &o = L.r34[f & (kRec34 - 1)];
                const int valid = r.valid, t = r.tick;
                int produced = 0;
                if ((SEA_ROLE_MASK & 8) && valid && t >= 5) {
                    /* denEn1[0..2] (NoiseSup.c:595-598) = sums of denSigSE1 of ticks t-2, t-1, t */
                    s.denEn0 = L.denSum[(t - 2) & (kSlots - 1)];
                    s.denEn1 = L.denSum[(t - 1) & (kSlots - 1)];
                    s.denEn2 = L.denSum[t & (kSlots - 1)];
                    ns_back<1, true, false, SEA_FIR_IN_S != 0, !ADDR_LDS, kIdctSplit ? kIdctHead : -1>(
                        r.psd, L.circ[1] + window_base(t), L.back[1], s, C,
                        SEA_FIR_IN_S ? (kIdctSplit ? o.mel : o.fir) : o.out, lane, 0.0f, nullptr, L.idctT, nullptr, nullptr, nullptr,
                        nullptr, kBregs ? C.idct : nullptr);
                    produced = 1;
                }
                if (lane == 0) {
                    o.produced = produced;
                    o.tick = t;
                }
            }
            NS_T_MID;
            if (!SEA_ABL_HALFSYNC || (i & 1)) block_sync();
            NS_T_END;
        }
        if (blob) {
            float *p = blob + kBlobLane + lane, *q = blob + kBlobScal;
            p[6 * 64] = s.noiseLo[1]; p[7 * 64] = s.noiseHi[1]; p[8 * 64] = s.denLo[1];
            p[9 * 64] = s.denHi[1]; p[10 * 64] = s.prevLo[1]; p[11 * 64] = s.prevHi[1];
            if (lane == 0) {
                q[7] = __int_as_float(s.nbFrame[1]); q[8] = s.lowSNRtrack; q[9] = s.alfaGF; q[10] = __int_as_float(s.psdOk[1]);
            }
        }
        NS_T_FLUSH(4);
    } else {
        /* ---- S: scalar chains ---- */
        uint32_t *out32 = reinterpret_cast<uint32_t *>(a.out + off);
        float *outf = a.out_f32 ? a.out_f32 + off : nullptr;
        float dcX = resume ? blob[kBlobScal + 11] : 0.0f, dcY = resume ? blob[kBlobScal + 12] : 0.0f; /* prevSamples, NoiseSup.c:908-909 */
        int firstOut = resume ? __float_as_int(blob[kBlobScal + 13]) : -1;
        const float irWinS = (SEA_IDCT_TAIL_S > 0) ? a.tables->irWin[lane] : 0.0f;
        NS_T_CK_DECL;
        for (long long i = 0; i < niter; ++i) {
            NS_T_BEGIN;
            prio_by_remaining(i);
            NS_T_CK_START;
            /* (1) VAD log-energy (NoiseSup.c:386-391) of the frame pushed at i-1 = tick tp; it is
             *     the "current frame" buf[80..159] of tick tp+2
             * (2) in-order sum of denSigSE1 (NoiseSup.c:597-598) of the tick B0 finished at i-1
             * (3) DC-offset filter (NoiseSup.c:182-198), int16 cast (ParmInterface.c:266), store of
             *     the frame B1 finished at i-1.  etsi_denoise copies zeros until the first NoiseSup
             *     output (AdvFrontEnd.c:186-190).
             * The three serial chains are independent of each other and run interleaved. */
            const long long fp = i - 1, fd = i - 2, fo = i - kLagS;
            bool doVad = false, doDen = false, produced = false;
            int tp = 0, td = 0;
            const float *denSrc = L.r12[0].den;
            const float *fdSrc = L.fdRec[0];
            float fdSums[3] = {0.0f, 0.0f, 0.0f};
            if (fp >= 0 && fp < nfr) {
                const auto &r = L.r01[fp & 1];
                doVad = (SEA_ROLE_MASK & 16) && r.valid;
                tp = r.tick;
            }
            if (fd >= 0 && fd < nfr) {
                const Rec12 &r = L.r12[fd & 1];
                doDen = (SEA_ROLE_MASK & 32) && r.valid && r.tick >= 3;
                td = r.tick;
                denSrc = r.den;
                fdSrc = L.fdRec[fd & 1];
            }
            const bool haveOut = fo >= 0 && fo < nfr;
            float *soutS = L.sout[SEA_STORE_IN_F ? (fo & 1) : 0];
            constexpr bool kTake = SEA_TAPS_RL && !ADDR_LDS && !SEA_STORE_IN_F; /* dc_verify_take: check + output in one batch */
            float2 vOut = make_float2(0.0f, 0.0f);
            if (haveOut) produced = (SEA_ROLE_MASK & 64) && L.r34[fo & (kRec34 - 1)].produced != 0;
            /* everything the chains need goes into LDS in one batch: the squares of the VAD frame and the DC filter's
             * input differences, straight from the second-stage FIR's registers (stage-1 17-tap FIR, NoiseSup.c:324-340) */
            {
                const float *frame = L.circ[0] + (tp & (kSlots - 1)) * kSlotLen;
                float x = 0.0f, yv = 0.0f, d0 = 0.0f, d1 = 0.0f;
                if (doVad) {
                    x = frame[lane];
                    if (lane < 16) yv = frame[64 + lane];
                }
                if (produced) {
                    Rec34 &r = L.r34[fo & (kRec34 - 1)];
                    constexpr bool kTapsRl = SEA_TAPS_RL && !ADDR_LDS && SEA_IDCT_TAIL_S > 0 && !SEA_IDCT_IN_F && SEA_FIR_IN_S;
                    if (!kTapsRl && SEA_IDCT_TAIL_S > 0 && !SEA_IDCT_IN_F && SEA_FIR_IN_S) /* finish the taps B1 started */
                        ns_idct_tail<kIdctHead>(r.mel, L.idctT, irWinS, r.fir, lane);
                    if (SEA_ABL_S & 1) {
                        d0 = d1 = dcX;
                    } else if (kTapsRl) { /* the taps B1 started, finished in lanes 0..8 and read as scalars; no trip through LDS */
                        dcX = ns_fir_dif_rl(fir_taps_rl(ns_idct_tail_rl<kIdctHead>(r.mel, L.idctT, irWinS, lane)),
                                            L.circ[1] + window_base(r.tick), lane, dcX, d0, d1);
                    } else if (SEA_FIR_IN_S) {
                        dcX = ns_fir_dif(r.fir, L.circ[1] + window_base(r.tick), lane, dcX, d0, d1);
                    } else {
                        const float *y2 = r.out;
                        const float xm1 = (lane == 0) ? dcX : y2[lane - 1];
                        L.sdif[lane] = y2[lane] - xm1;
                        if (lane < 16) L.sdif[64 + lane] = y2[64 + lane] - y2[63 + lane];
                        dcX = y2[79];
                    }
                }
                if (doVad) {
                    L.ssq[lane] = x * x;
                    if (lane < 16) L.ssq[64 + lane] = yv * yv;
                }
                if (produced && SEA_FIR_IN_S && lane < 40) *reinterpret_cast<float2 *>(&L.sdif[2 * lane]) = make_float2(d0, d1);
            }
            NS_T_CK(0);
            if (FD && fd >= 0 && fd < nfr && !doDen && lane == 0) L.rfd[fd & 1].tick = 0;
            if (doVad || doDen || produced) {
                wave_sync();
                float vadSum, denTotal, y = dcY;
                if (SEA_ABL_S & 8) {
                    vadSum = L.ssq[3] + 64.0f, denTotal = denSrc[5], y = L.sdif[7];
                } else
                    helper_chains<ADDR_LDS ? 10 : 4, FD>(L.ssq, denSrc, L.sdif, soutS, L.szero, vadSum, denTotal, y, lane, nullptr,
                                                         FD ? fdSrc : nullptr, FD ? fdSums : nullptr);
                NS_T_CK(1);
                if (doVad && lane == 0) L.frameEn[(tp + 2) & (kSlots - 1)] = vadSum; /* 64 + sum of squares; B0 takes the log */
                NS_T_CK(2);
                if (doDen && lane == 0) L.denSum[td & (kSlots - 1)] = denTotal;
                if (FD && doDen && lane == 0) { /* hand the measures' sums of that frame to F */
                    RecFd &q = L.rfd[fd & 1];
                    q.mean = fdSums[0];
                    q.var = fdSums[1];
                    q.tempEn = fdSums[2];
                    q.m1 = fdSrc[129];
                    q.m2 = fdSrc[130];
                    q.m3 = fdSrc[131];
                    q.nb16 = reinterpret_cast<const int *>(fdSrc)[156];
                    q.vadns = reinterpret_cast<const int *>(fdSrc)[157];
                    q.tick = td;
                }
                if (produced) {
                    /* (checking the recurrence's exactness condition on the sixteen recomputing lanes' registers instead
                     * was measured slower: five checks in a row per lane against two per lane here) */
                    if (kTake) vOut = dc_verify_take(L.sdif, soutS, dcY, y, lane);
                    else dc_verify(L.sdif, soutS, dcY, y, lane);
                    dcY = y;
                    if (firstOut < 0) firstOut = (int)fo + (blob ? a.frame_base : 0);
                }
                NS_T_CK(3);
            }
            if (haveOut && SEA_STORE_IN_F) {
                if (lane == 0) L.outProd[fo & 1] = produced ? 1 : 0;
                if (FD && produced && lane == 0 && a.flags_out)
                    a.flags_out[off / 8 + 10 * fo] = (unsigned char)L.fdFlags[L.r34[fo & (kRec34 - 1)].tick & (kSlots - 1)];
            } else if (haveOut) {
                int ln = lane; /* 80-VGPR form: recomputed, or the per-lane store address lives in scratch (see F's intake) */
                if (ADDR_LDS) asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
                if (ln < 40 && !(SEA_ABL_S & 4)) {
                    uint32_t packed = 0u;
                    if (produced) {
                        const float2 v = kTake ? vOut : *reinterpret_cast<const float2 *>(&soutS[2 * ln]);
                        packed = (uint32_t)cast_i16(v.x) | ((uint32_t)cast_i16(v.y) << 16);
                        if (outf) *reinterpret_cast<float2 *>(outf + fo * SEA_HOP + 2 * ln) = v;
                    }
                    out32[fo * 40 + ln] = packed;
                }
                if (FD && produced && lane == 0 && a.flags_out)
                    a.flags_out[off / 8 + 10 * fo] = (unsigned char)L.fdFlags[L.r34[fo & (kRec34 - 1)].tick & (kSlots - 1)];
                wave_sync();
            }
            NS_T_CK(4);
            NS_T_MID;
            if (!SEA_ABL_HALFSYNC || (i & 1)) block_sync();
            NS_T_END;
        }
        if (a.first_out && lane == 0) a.first_out[u] = firstOut;
        if (blob && lane == 0) {
            blob[kBlobScal + 11] = dcX;
            blob[kBlobScal + 12] = dcY;
            blob[kBlobScal + 13] = __int_as_float(firstOut);
        }
        NS_T_FLUSH(6);
        NS_T_CK_FLUSH;
    }
    if (blob) { /* every wave has passed the last frame barrier: nothing writes LDS any more */
        for (int i = threadIdx.x; i < 2 * kCirc; i += 64 * kPipeWaves) blob[i] = L.circ[i / kCirc][i % kCirc];
        if (threadIdx.x < kSlots) {
            blob[kBlobRing + threadIdx.x] = L.frameEn[threadIdx.x];
            blob[kBlobRing + kSlots + threadIdx.x] = L.denSum[threadIdx.x];
        }
    }
}

} // namespace p4

#ifndef SEA_NS_BODY_ONLY
__global__ __launch_bounds__(256, SEA_NS_MIN_WAVES) void ns_denoise_pipe_kernel(NsBatchArgs a)
{
    __shared__ p4::PipeLds<false> L;
    p4::ns_pipe_body<false, false>(a, L);
}

/* the same arithmetic with the transform's address tables in LDS instead of VGPRs: 80 instead of 110
 * VGPRs, six workgroups per CU instead of four -- the form to launch when the batch has more than four
 * utterances per CU (373 vs 331 M frames/s at 4096 utterances; 267 vs 298 M at 1024) */
#ifndef SEA_NS_BIG_WAVES
#define SEA_NS_BIG_WAVES 6
#endif
__global__ __launch_bounds__(256, SEA_NS_BIG_WAVES) void ns_denoise_pipe_big_kernel(NsBatchArgs a)
{
    __shared__ p4::PipeLds<true> L;
    p4::ns_pipe_body<false, true>(a, L);
}

/* both forms for utterances processed in time slices (NsBatchArgs::state): the recursion is loaded at the start and
 * stored at the end of the launch; kernels of their own so that the whole-utterance forms keep their register budgets */
__global__ __launch_bounds__(256, SEA_NS_MIN_WAVES) void ns_denoise_pipe_slice_kernel(NsBatchArgs a)
{
    __shared__ p4::PipeLds<false> L;
    p4::ns_pipe_body<false, false, true>(a, L);
}
__global__ __launch_bounds__(256, SEA_NS_BIG_WAVES) void ns_denoise_pipe_big_slice_kernel(NsBatchArgs a)
{
    __shared__ p4::PipeLds<true> L;
    p4::ns_pipe_body<false, true, true>(a, L);
}

/* the same pipeline with the first stage's speech measures (SpeechQVar/Spec/Mel, VADNS) evaluated in
 * B0 and their four bits stored per output frame: input of the frame-dropping VAD */
__global__ __launch_bounds__(256, SEA_NS_MIN_WAVES) void ns_denoise_pipe_fd_kernel(NsBatchArgs a)
{
    __shared__ p4::PipeLds<false, true> L;
    p4::ns_pipe_body<true, false>(a, L);
}
#endif

} // namespace sea

#if defined(SEA_NS_TIMING) && !defined(SEA_NS_BODY_ONLY)
extern "C" int sea_debug_ns_back_ck(unsigned long long *out16, int reset)
{
    if (reset) {
        unsigned long long z[16] = {0};
        return (int)hipMemcpyToSymbol(HIP_SYMBOL(sea::g_back_ck), z, sizeof z);
    }
    return (int)hipMemcpyFromSymbol(out16, HIP_SYMBOL(sea::g_back_ck), 16 * sizeof(unsigned long long));
}
extern "C" int sea_debug_ns_wg(unsigned *out, int n_wg)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(sea::p4::g_ns_wg), (size_t)n_wg * 4 * sizeof(unsigned));
}
extern "C" int sea_debug_ns_hw(unsigned *out, int n_wg)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(sea::p4::g_ns_hw), (size_t)n_wg * 4 * sizeof(unsigned));
}
extern "C" int sea_debug_ns_timing(unsigned long long *out8)
{
    return (int)hipMemcpyFromSymbol(out8, HIP_SYMBOL(sea::p4::g_ns_timing), 24 * sizeof(unsigned long long));
}
#endif
