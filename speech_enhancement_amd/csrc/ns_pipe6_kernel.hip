/*
 * ns_pipe6_kernel.hip -- etsi_denoise over a packed batch, SIX pipelined wavefronts per utterance.
 *
 * Same arithmetic as ns_pipe_kernel.hip (four waves), cut finer.  With up to four utterances per CU the
 * run time of a launch is the longest utterance's chain of frames, i.e. (number of frames) x (frame
 * period of one workgroup), and the frame period is the longest role.  The role timers of the
 * four-wave kernel (tools/ns_timing.py) read F 4670, B0 3590, B1 4690, S 4310 clk; here the dual
 * transform is cut at its middle level and the second-stage back half into noise tracking | gains:
 *
 *   wave FA  iteration i: load int16 frame i, zero-frame gate, push; window + register-resident start
 *            + levels n2 = 8, 16, 32 of BOTH transforms (stage 0 of frame i, stage 1 of frame i-3)
 *   wave FB  i+1 / i+4:   levels n2 = 64, 128, 256 and the two 65-bin PSDs
 *   wave B0  frame i-2:   stage-0 FilterCalc, VAD, mel, IDCT, FIR -> stage-1 buffer
 *   wave N1  frame i-5:   stage-1 PSD mean, noise tracking, in-order noise sum, gain-factor scalars
 *   wave G1  frame i-6:   stage-1 Wiener gains, mel, gain factorisation, IDCT, FIR
 *   wave S   the lane-grouped scalar chains: VAD log-energy of the frame pushed at i-1, in-order sum of
 *            denSigSE1 of frame i-3, DC-offset recurrence + cast + store of frame i-7
 *
 * All records between neighbouring stages are double-buffered by frame parity (written at one
 * iteration, read at the next); the two transform work areas alternate between FA and FB.
 *
 * Round 4: this is the form for up to FOUR utterances per CU, configs[1] included (capi.hip::ns_pick_form).  Up to three per CU
 * ns_denoise_pipe6_kernel (80 VGPRs, the lighter helper wave: SEA_P6_LIGHT_S); for the fourth ns_denoise_pipe6_dense_kernel, the same
 * body compiled for seven waves per SIMD -- with six, the dispatcher never found room for the fourth six-wave workgroup of a CU, which is
 * what rounds 1-3 measured as "the six-wave form loses at four per CU" (see the comment at the kernels below).  With more than one
 * utterance per CU the waves set their issue priority by the frames their utterance has left (prio_by_remaining, the rule of
 * ns_pipe_kernel.hip), and the wave -> role map (SEA_NS6_PERM) puts B0 and S on the two oldest waves: among equal priorities a SIMD
 * issues its oldest wave first.  configs[1]: 1.91-1.95 ms = 420-427 M frames/s (four-wave form 2.08-2.13);
 * profiles/r04_ns_six_wave_dense.txt.
 */
#include "ns_core.h"

namespace sea {

namespace p6 { /* everything of the six-wave form */

/* timing-only diagnostic (-DSEA_NS6_TIMING): shader clocks workgroup 0's six roles spend working / waiting at the frame barrier
 * -> g_ns6_timing[role * 2 + {0, 1}], [12] = frames; read through sea_debug_ns6_timing (tools/ns6_roles.py) */
#ifdef SEA_NS6_TIMING
__device__ unsigned long long g_ns6_timing[16];
struct RoleTimer6 {
    unsigned long long work = 0, wait = 0, t0 = 0, t1 = 0;
    __device__ __forceinline__ void begin() { t0 = clock64(); }
    __device__ __forceinline__ void mid() { t1 = clock64(); work += t1 - t0; }
    __device__ __forceinline__ void end() { wait += clock64() - t1; }
};
#define NS6_T_DECL RoleTimer6 rt_
#define NS6_T_BEGIN rt_.begin()
#define NS6_T_MID rt_.mid()
#define NS6_T_END rt_.end()
#define NS6_T_FLUSH(r, n) do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) { g_ns6_timing[2 * (r)] = rt_.work; g_ns6_timing[2 * (r) + 1] = rt_.wait; g_ns6_timing[12] = (unsigned long long)(n); } } while (0)
#else
#define NS6_T_DECL
#define NS6_T_BEGIN
#define NS6_T_MID
#define NS6_T_END
#define NS6_T_FLUSH(r, n)
#endif

constexpr int kSlots = 8;
constexpr int kSlotLen = SEA_HOP;
constexpr int kCirc = kSlots * kSlotLen;
constexpr int kMirror = 3 * kSlotLen;
constexpr int kWaves = 6;
constexpr int kDepth = 7; /* S stores frame i - kDepth */

struct __attribute__((aligned(16))) RecA { /* FA -> FB, S: what was pushed at this iteration */
    int valid, tick;      /* stage 0, frame i */
    int valid1, tick1;    /* stage 1, frame i-3 */
};
struct __attribute__((aligned(16))) RecPsd { /* FB -> B0 (stage 0) / N1 (stage 1) */
    float psd[68];
    int valid, tick, pad0, pad1;
};
struct __attribute__((aligned(16))) RecDen { /* B0 -> FA (valid / tick of the stage-1 frame), S (den) */
    float den[68];
    int valid, tick, pad0, pad1;
};
struct __attribute__((aligned(16))) RecN { /* N1 -> G1 */
    float psd[68], P[68], noise[68];
    float alfa;
    int produced, tick, pad0;
};
struct __attribute__((aligned(16))) RecOut { /* G1 -> S */
    float out[80]; /* second-stage filter output before the DC-offset filter */
    int produced, tick, pad0, pad1;
};

struct __attribute__((aligned(16))) Pipe6Lds {
    float circ[2][kCirc + kMirror];
    float work[2][512];             /* transform work areas: FA fills [i & 1], FB finishes [(i-1) & 1] */
    BackLds back[2];                /* scratch of B0 and G1 */
    float ssq[80], sdif[80], sout[80], szero[4]; /* scratch of S */
    float frameEn[kSlots], denSum[kSlots];
    int vadTodo[2];                 /* SEA_P6_LOG_IN = 3: the ring entry whose log N1 takes at beat i + 1, by i & 1 (-1: none) */
    float frameEnLog[kSlots];       /* SEA_P6_LIGHT_S: frameEn = 64 + sum of squares (S), frameEnLog = its log-energy (FA, one beat later) */
    int fdFlags[kSlots];
    float idctT[SEA_NMEL * 16];
    RecA ra[2];
    RecPsd p0[2], p1[2];
    RecDen rd[2];
    RecN rn[2];
    RecOut ro[2];
};

__device__ __forceinline__ int window_base(int tick) { return ((tick - 3) & (kSlots - 1)) * kSlotLen; }

__device__ __forceinline__ void slot_store(float *circ, int tick, int lane, float a, float b)
{
    const int slot = tick & (kSlots - 1);
    float *p = circ + slot * kSlotLen + 2 * lane;
    *reinterpret_cast<float2 *>(p) = make_float2(a, b);
    if (slot < 3) *reinterpret_cast<float2 *>(p + kCirc) = make_float2(a, b);
}

__device__ __forceinline__ void block_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

__device__ __forceinline__ void load_back_const(NsConst &C, const sea_ns_tables *t, int lane)
{
    C.melStart = t->melStart[lane];
    C.melLen = t->melLen[lane];
#pragma unroll
    for (int i = 0; i < SEA_MEL_TAPS; ++i) C.melW[i] = t->melW[i][lane];
    C.irWin = t->irWin[lane];
    C.eps = t->eps;
}

template <bool FD, bool LIGHT /* the VAD log leaves S */, bool DIFG1 = LIGHT /* G1 hands over the DC differences */>
__device__ __forceinline__ void ns_pipe6_body(const NsBatchArgs &a, Pipe6Lds &L)
{
    const int lane = threadIdx.x & 63;
    /* wave -> role (octal digits, wave 0 rightmost; roles 0 FA, 1 FB, 2 B0, 3 N1, 4 G1, 5 S).  This form is for up to two
     * utterances per CU.  At four per CU (configs[1], where the four-wave form runs at 2.20 ms) the placement of the roles on
     * the SIMDs decides: identity 3.19 ms, 0104352 (B0, S, N1, G1, FA, FB) 2.78 ms, the ten even / odd splits in wave order
     * 2.97-3.26 ms (round 3, tools/build_variant.sh -DSEA_NS6_PERM=...): none reaches the four-wave form. */
#ifndef SEA_P6_TAPS_RL
#define SEA_P6_TAPS_RL 1
#endif
/* 1 (round 4; the forms for up to two utterances per CU, where the run time is one utterance's chain of frames): the helper
 * wave S was this form's longest role (3397 clk per frame alone, B0 3089, G1 2985, N1 2623, FB 2608, FA 2268): (a) the VAD's
 * log-energy (NoiseSup.c:391) is taken by the first transform wave FA one beat after S left the frame's sum of squares -- FA has
 * ~1000 clk of slack, the value is consumed by B0 two beats later still (SEA_P6_LOG_IN = 3: by N1 instead); (b) G1 hands over the
 * DC filter's input differences instead of the filtered frame (ns_gain1_dif), S's own pass over the frame goes.  S 3397 -> 2836,
 * G1 2985 -> 3134, FA 2268 -> 2694: 256 utterances 1.777 -> 1.664 ms.  NOT in the dense form: at four workgroups per CU the step is
 * a throughput limit and the same change costs 2 % there (1.99 against 1.95 ms). */
#ifndef SEA_P6_LIGHT_S
#define SEA_P6_LIGHT_S 1
#endif
#ifndef SEA_P6D_LOGMOVE /* the same two changes in the dense form, separately */
#define SEA_P6D_LOGMOVE 0
#endif
#ifndef SEA_P6D_DIFG1
#define SEA_P6D_DIFG1 0
#endif
#ifndef SEA_P6_LOG_IN /* which wave takes the VAD log with LIGHT: 0 FA itself, 3 N1 (FA only leaves the ring index) */
#define SEA_P6_LOG_IN 0
#endif
#ifndef SEA_P6_S_CHUNKS
#define SEA_P6_S_CHUNKS 10
#endif
#ifndef SEA_P6_LRPT
#define SEA_P6_LRPT 1
#endif
#ifndef SEA_P6_PRIO_LEVELS
#define SEA_P6_PRIO_LEVELS 32
#endif
#ifndef SEA_P6_PRIO_STEP
#define SEA_P6_PRIO_STEP 16
#endif
#ifndef SEA_P6_PRIO_ROWBIAS
#define SEA_P6_PRIO_ROWBIAS 1
#endif
#ifndef SEA_NS6_PERM
#define SEA_NS6_PERM 0014352
#endif
    const int role = ((a.perm6 ? a.perm6 : SEA_NS6_PERM) >> (3 * __builtin_amdgcn_readfirstlane(threadIdx.x >> 6))) & 7;
    const int u = a.order ? a.order[blockIdx.x] : (int)blockIdx.x;
    const long long off = a.offsets[u];
    const long long nfr = a.lengths[u] / SEA_HOP;
    const long long niter = nfr + kDepth;
    /* issue priority by remaining frames, the rule of the four-wave form (ns_pipe_kernel.hip, SEA_PRIO_LRPT): on whenever the
     * launch has more utterances than CUs to put them on and an order whose first entry is the longest */
    const bool lrpt = SEA_P6_LRPT && a.prio_row > 0 && a.order;
    const long long longestFr = lrpt ? a.lengths[a.order[0]] / SEA_HOP : 0;
    const float lrptScale = (float)SEA_P6_PRIO_LEVELS / (float)(longestFr > 0 ? longestFr : 1);
    const int lrptBias = lrpt ? SEA_P6_PRIO_ROWBIAS * ((int)blockIdx.x / a.prio_row) : 0;
    auto prio_by_remaining = [&](long long i) {
        if (SEA_P6_LRPT && lrpt && (i & (SEA_P6_PRIO_STEP - 1)) == 0) {
            const int L = __builtin_amdgcn_readfirstlane((int)((float)(nfr - i) * lrptScale)) + lrptBias;
            const int pr = (L + (int)((i / SEA_P6_PRIO_STEP) & (SEA_P6_PRIO_LEVELS / 4 - 1))) / (SEA_P6_PRIO_LEVELS / 4);
            if (pr >= 3) __builtin_amdgcn_s_setprio(3);
            else if (pr == 2) __builtin_amdgcn_s_setprio(2);
            else if (pr == 1) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
        }
    };

    for (int i = threadIdx.x; i < 2 * (kCirc + kMirror); i += 64 * kWaves) (&L.circ[0][0])[i] = 0.0f;
    for (int i = threadIdx.x; i < SEA_NMEL * 16; i += 64 * kWaves) L.idctT[i] = a.tables->idct[i >> 4][i & 15];
    if (threadIdx.x < 4) L.szero[threadIdx.x] = 0.0f;
    if (threadIdx.x < kSlots) {
        L.frameEn[threadIdx.x] = 0.0f;
        L.frameEnLog[threadIdx.x] = 0.0f;
        L.denSum[threadIdx.x] = 0.0f;
    }
    if (threadIdx.x < 2) {
        const int k = threadIdx.x;
        L.vadTodo[k] = -1;
        L.ra[k].valid = L.ra[k].valid1 = 0;
        L.p0[k].valid = L.p1[k].valid = 0;
        L.rd[k].valid = 0;
        L.rd[k].den[65] = L.rd[k].den[66] = L.rd[k].den[67] = 0.0f; /* read as zeros by S */
        L.rn[k].produced = 0;
        L.ro[k].produced = 0;
    }
    block_sync();
    NS6_T_DECL;

    if (role == 0) {
        /* ---- FA: input + zero-frame gate (ParmInterface.c:244-251); first half of both transforms ---- */
        Fft2Regs fft;
        load_fft2_regs<false>(fft, &a.tables->fft, lane, nullptr);
        float win8[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) win8[k] = a.tables->win8[k][lane];
        const uint32_t *in32 = reinterpret_cast<const uint32_t *>(a.in + off);
        uint32_t nextw = (lane < 40 && nfr > 0) ? in32[lane] : 0u;
        int tick = 0; /* frames seen since (and including) the first non-zero one */
        int onset = (int)nfr;
        int vH1 = 0, tH1 = 0, vH2 = 0, tH2 = 0; /* (valid, tick) of frames i-1 and i-2 */
        for (long long i = 0; i < niter; ++i) {
            NS6_T_BEGIN;
            prio_by_remaining(i);
            if (LIGHT && SEA_P6_LOG_IN == 0 && vH2) { /* the log-energy of the sum S left one beat ago (the frame pushed at i-2 is tick tH2 + 2's "current frame") */
                const int e = (tH2 + 2) & (kSlots - 1);
                const float en = vad_frame_energy(L.frameEn[e]);
                if (lane == 0) L.frameEnLog[e] = en;
            }
            if (LIGHT && SEA_P6_LOG_IN == 3 && lane == 0) L.vadTodo[i & 1] = vH2 ? ((tH2 + 2) & (kSlots - 1)) : -1;
            RecA &r = L.ra[i & 1];
            int valid = 0;
            bool actA = false;
            if (i < nfr) {
                const uint32_t w = nextw;
                if (i + 1 < nfr && lane < 40) nextw = in32[(i + 1) * 40 + lane];
                const bool any = __ballot(w != 0u) != 0ull;
                if (any || tick > 0) {
                    valid = 1;
                    if (FD && tick == 0) onset = (int)i;
                    tick++;
                    const float x0 = (float)(short)(w & 0xFFFFu), x1 = (float)(short)(w >> 16);
                    if (lane < 40) slot_store(L.circ[0], tick, lane, x0, x1);
                    actA = tick >= 3; /* NoiseSup.c:1152 */
                }
            }
            /* stage 1, frame i-3 (B0 finished it at the previous iteration): NoiseSup.c:1178 <=> tick >= 5 */
            const long long f1 = i - 3;
            int valid1 = 0, t1 = 0;
            bool actB = false;
            if (f1 >= 0 && f1 < nfr) {
                const RecDen &d = L.rd[f1 & 1];
                valid1 = d.valid;
                t1 = d.tick;
                actB = valid1 && t1 >= 5;
            }
            if (lane == 0) {
                r.valid = valid;
                r.tick = tick;
                r.valid1 = valid1;
                r.tick1 = t1;
            }
            vH2 = vH1, tH2 = tH1, vH1 = valid, tH1 = tick;
            if (actA || actB) {
                wave_sync();
                float e[8];
                ns_window8(L.circ[0] + window_base(tick), actA, L.circ[1] + window_base(t1), actB, win8, lane, e);
                rfft256_dual_lo<false>(e, L.work[i & 1], fft);
            }
            NS6_T_MID;
            block_sync();
            NS6_T_END;
        }
        NS6_T_FLUSH(role, niter);
        if (FD && a.onset_out && lane == 0) a.onset_out[u] = onset;
    } else if (role == 1) {
        /* ---- FB: second half of both transforms, FFTtoPSD ---- */
        Fft2Regs fft;
        load_fft2_regs<false>(fft, &a.tables->fft, lane, nullptr);
        for (long long i = 0; i < niter; ++i) {
            NS6_T_BEGIN;
            prio_by_remaining(i);
            const long long g = i - 1; /* FA's iteration */
            if (g >= 0) {
                const RecA &r = L.ra[g & 1];
                const long long f0 = g, f1 = g - 3;
                const int valid = r.valid, t0 = r.tick, valid1 = r.valid1, t1 = r.tick1;
                const bool actA = (f0 < nfr) && valid && t0 >= 3;
                const bool actB = (f1 >= 0 && f1 < nfr) && valid1 && t1 >= 5;
                float *work = L.work[g & 1];
                if (actA || actB) {
                    if (SEA_PSD_REGS) { /* the last level stays in registers and feeds both PSDs (ns_core.h, psd_from_last_level) */
                        float o[8];
                        rfft256_dual_hi_keep_last<false>(work, fft, o);
                        psd_from_last_level(o, fft, L.p0[f0 & 1].psd, actA, L.p1[(f1 < 0 ? 0 : f1) & 1].psd, actB, lane);
                    } else {
                        rfft256_dual_hi<false>(work, fft);
                        if (actA) psd_from_fft2(work, L.p0[f0 & 1].psd, fft, lane);
                        if (actB) psd_from_fft2(work + 256, L.p1[f1 & 1].psd, fft, lane);
                    }
                    wave_sync();
                }
                if (lane == 0) {
                    if (f0 < nfr) {
                        L.p0[f0 & 1].valid = valid;
                        L.p0[f0 & 1].tick = t0;
                    }
                    if (f1 >= 0 && f1 < nfr) {
                        L.p1[f1 & 1].valid = valid1;
                        L.p1[f1 & 1].tick = t1;
                    }
                }
            }
            NS6_T_MID;
            block_sync();
            NS6_T_END;
        }
        NS6_T_FLUSH(role, niter);
    } else if (role == 2) {
        /* ---- B0: BACK of stage 0; its 80 outputs enter the stage-1 buffer ---- */
        NsConst C;
        load_back_const(C, a.tables, lane);
        NsRegs s;
        regs_init(s, C.eps);
        NsFd fd;
        fd_init(fd);
        for (long long i = 0; i < niter; ++i) {
            NS6_T_BEGIN;
            prio_by_remaining(i);
            const long long f = i - 2;
            if (f >= 0 && f < nfr) {
                const RecPsd &r = L.p0[f & 1];
                RecDen &o = L.rd[f & 1];
                const int valid = r.valid, t = r.tick;
                if (valid && t >= 3) {
                    float *tmp = L.back[0].sq;
                    int bits = 0;
                    /* SEA_P6_TAPS_RL: the filter taps as scalar operands (v_readlane), the filter's outputs in registers straight
                     * into the stage-1 buffer -- two LDS round trips less on this role's chain (ns_core.h, fir_taps_rl) */
                    float y01[2] = {0.0f, 0.0f};
                    ns_back<0, true, FD, false, true>(r.psd, L.circ[0] + window_base(t), L.back[0], s, C, tmp, lane,
                                         (LIGHT ? L.frameEnLog : L.frameEn)[t & (kSlots - 1)], o.den, L.idctT, &fd, &bits, nullptr,
                                         SEA_P6_TAPS_RL ? y01 : nullptr);
                    if (FD && lane == 0) L.fdFlags[t & (kSlots - 1)] = bits;
                    if (lane < 40) {
                        if (SEA_P6_TAPS_RL) {
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
            NS6_T_MID;
            block_sync();
            NS6_T_END;
        }
        NS6_T_FLUSH(role, niter);
    } else if (role == 3) {
        /* ---- N1: stage-1 noise tracking, noise-spectrum sum, gain-factor scalars ---- */
        const float eps = a.tables->eps;
        NsRegs s;
        regs_init(s, eps);
        for (long long i = 0; i < niter; ++i) {
            NS6_T_BEGIN;
            prio_by_remaining(i);
            if (LIGHT && SEA_P6_LOG_IN == 3 && i > 0) { /* the VAD log of the entry FA named one beat ago; B0 reads it one beat from now */
                const int e = L.vadTodo[(i - 1) & 1];
                if (e >= 0) {
                    const float en = vad_frame_energy(L.frameEn[e]);
                    if (lane == 0) L.frameEnLog[e] = en;
                }
            }
            const long long f = i - 5;
            if (f >= 0 && f < nfr) {
                const RecPsd &r = L.p1[f & 1];
                RecN &o = L.rn[f & 1];
                const int valid = r.valid, t = r.tick;
                int produced = 0;
                if (valid && t >= 5) {
                    /* denEn1[0..2] (NoiseSup.c:595-598) = sums of denSigSE1 of ticks t-2, t-1, t */
                    s.denEn0 = L.denSum[(t - 2) & (kSlots - 1)];
                    s.denEn1 = L.denSum[(t - 1) & (kSlots - 1)];
                    s.denEn2 = L.denSum[t & (kSlots - 1)];
                    o.psd[lane] = r.psd[lane];
                    if (lane == 0) o.psd[64] = r.psd[64];
                    const float alfa = ns_noise1(r.psd, o.P, o.noise, s, eps, lane);
                    if (lane == 0) o.alfa = alfa;
                    produced = 1;
                }
                if (lane == 0) {
                    o.produced = produced;
                    o.tick = t;
                }
            }
            NS6_T_MID;
            block_sync();
            NS6_T_END;
        }
        NS6_T_FLUSH(role, niter);
    } else if (role == 4) {
        /* ---- G1: stage-1 Wiener gains, mel, gain factorisation, IDCT, FIR ---- */
        NsConst C;
        load_back_const(C, a.tables, lane);
        NsRegs s;
        regs_init(s, C.eps);
        float lastIn = 0.0f; /* LIGHT: y[79] of the previous filtered frame (prevSamples, NoiseSup.c:908) */
        for (long long i = 0; i < niter; ++i) {
            NS6_T_BEGIN;
            prio_by_remaining(i);
            const long long f = i - 6;
            if (f >= 0 && f < nfr) {
                const RecN &r = L.rn[f & 1];
                RecOut &o = L.ro[f & 1];
                const int produced = r.produced, t = r.tick;
                if (produced && DIFG1)
                    ns_gain1_dif(r.psd, r.P, r.noise, r.alfa, L.circ[1] + window_base(t), L.back[1], s, C, o.out, lane, L.idctT, lastIn);
                else if (produced)
                    ns_gain1(r.psd, r.P, r.noise, r.alfa, L.circ[1] + window_base(t), L.back[1], s, C, o.out, lane, L.idctT);
                if (lane == 0) {
                    o.produced = produced;
                    o.tick = t;
                }
            }
            NS6_T_MID;
            block_sync();
            NS6_T_END;
        }
        NS6_T_FLUSH(role, niter);
    } else {
        /* ---- S: the lane-grouped scalar chains (helper_chains) ---- */
        uint32_t *out32 = reinterpret_cast<uint32_t *>(a.out + off);
        float *outf = a.out_f32 ? a.out_f32 + off : nullptr;
        float dcX = 0.0f, dcY = 0.0f; /* prevSamples, NoiseSup.c:908-909 */
        int firstOut = -1;
        for (long long i = 0; i < niter; ++i) {
            NS6_T_BEGIN;
            prio_by_remaining(i);
            /* (1) VAD log-energy (NoiseSup.c:386-391) of the frame pushed at i-1 = tick tp ("current frame" of
             *     tick tp+2); (2) in-order sum of denSigSE1 of the frame B0 finished at i-1; (3) DC-offset
             *     filter, int16 cast, store of the frame G1 finished at i-1 */
            const long long fp = i - 1, fd = i - 3, fo = i - kDepth;
            bool doVad = false, doDen = false, produced = false;
            int tp = 0, td = 0;
            const float *denSrc = L.rd[0].den;
            if (fp >= 0 && fp < nfr) {
                const RecA &r = L.ra[fp & 1];
                doVad = r.valid != 0;
                tp = r.tick;
            }
            if (fd >= 0 && fd < nfr) {
                const RecDen &r = L.rd[fd & 1];
                doDen = r.valid && r.tick >= 3;
                td = r.tick;
                denSrc = r.den;
            }
            const bool haveOut = fo >= 0 && fo < nfr;
            float2 vOut = make_float2(0.0f, 0.0f);
            if (haveOut) produced = L.ro[fo & 1].produced != 0;
            if (doVad) {
                const float *frame = L.circ[0] + (tp & (kSlots - 1)) * kSlotLen;
                const float x = frame[lane];
                L.ssq[lane] = x * x;
                if (lane < 16) {
                    const float yv = frame[64 + lane];
                    L.ssq[64 + lane] = yv * yv;
                }
            }
            const float *difS = DIFG1 ? L.ro[fo & 1].out : L.sdif; /* DIFG1: G1 left the differences themselves */
            if (produced && !DIFG1) {
                const float *y2 = L.ro[fo & 1].out;
                const float xm1 = (lane == 0) ? dcX : y2[lane - 1];
                L.sdif[lane] = y2[lane] - xm1;
                if (lane < 16) L.sdif[64 + lane] = y2[64 + lane] - y2[63 + lane];
                dcX = y2[79];
            }
            if (doVad || doDen || produced) {
                wave_sync();
                float vadSum, denTotal, y = dcY;
                helper_chains<SEA_P6_S_CHUNKS>(L.ssq, denSrc, difS, L.sout, L.szero, vadSum, denTotal, y, lane);
                if (doVad) {
                    const float en = LIGHT ? vadSum : vad_frame_energy(vadSum); /* LIGHT_S: FA takes the log one beat later */
                    if (lane == 0) L.frameEn[(tp + 2) & (kSlots - 1)] = en;
                }
                if (doDen && lane == 0) L.denSum[td & (kSlots - 1)] = denTotal;
                if (produced) {
                    if (SEA_P6_TAPS_RL) vOut = dc_verify_take(difS, L.sout, dcY, y, lane); /* check + output in one batch of reads */
                    else dc_verify(difS, L.sout, dcY, y, lane);
                    dcY = y;
                    if (firstOut < 0) firstOut = (int)fo;
                }
            }
            if (haveOut) {
                if (lane < 40) {
                    uint32_t packed = 0u;
                    if (produced) {
                        const float2 v = SEA_P6_TAPS_RL ? vOut : *reinterpret_cast<const float2 *>(&L.sout[2 * lane]);
                        packed = (uint32_t)cast_i16(v.x) | ((uint32_t)cast_i16(v.y) << 16);
                        if (outf) *reinterpret_cast<float2 *>(outf + fo * SEA_HOP + 2 * lane) = v;
                    }
                    out32[fo * 40 + lane] = packed;
                }
                if (FD && produced && lane == 0 && a.flags_out)
                    a.flags_out[off / 8 + 10 * fo] = (unsigned char)L.fdFlags[L.ro[fo & 1].tick & (kSlots - 1)];
                wave_sync();
            }
            NS6_T_MID;
            block_sync();
            NS6_T_END;
        }
        NS6_T_FLUSH(role, niter);
        if (a.first_out && lane == 0) a.first_out[u] = firstOut;
    }
}

} // namespace p6

#ifndef SEA_NS_BODY_ONLY
/* launched for at most two utterances per CU (capi.hip::ns_pick_form): twelve waves per CU, three per SIMD, so the
 * register allocation could use up to 168 VGPRs.  The plain form stays compiled for 80 (measured: 1644 against 1679 ns
 * per frame with the looser bound, which only changes the schedule); the _fd form takes the looser bound, which
 * removes its 7 spilled registers (91 VGPRs). */
#ifndef SEA_NS6_BLOCKS
#define SEA_NS6_BLOCKS 6
#endif
__global__ __launch_bounds__(384, SEA_NS6_BLOCKS) void ns_denoise_pipe6_kernel(NsBatchArgs a)
{
    __shared__ p6::Pipe6Lds L;
    p6::ns_pipe6_body<false, SEA_P6_LIGHT_S != 0>(a, L);
}
/* The same body compiled for SEVEN waves per SIMD (72 VGPRs, 8 spilled): the form for three or four utterances per CU
 * (round 4).  With 80 VGPRs a SIMD holds six waves, four six-wave workgroups are exactly the 24 a CU then holds -- and the
 * dispatcher, which deals the six waves of a workgroup 2 / 2 / 1 / 1 over the SIMDs, does not find room for the fourth: it
 * waited for one of the first three to end (configs[1]: 3.20 ms, which rounds 1-3 read as "the six-wave form loses at four
 * per CU").  One wave slot of slack per SIMD lets all four co-reside: 2.14 ms without priorities, **2.00-2.05 ms** with the
 * issue priority by remaining frames and the wave -> role map below, against 2.08-2.13 for the four-wave form
 * (profiles/r04_ns_six_wave_dense.txt). */
#ifdef SEA_NS6_TIMING
__device__ unsigned g_ns6_wg[16384 * 4]; /* dense form, per workgroup: start, end (constant 100 MHz counter), HW_ID, XCC_ID (tools/ns6_residency.py) */
#endif
__global__ __launch_bounds__(384, 7) void ns_denoise_pipe6_dense_kernel(NsBatchArgs a)
{
    __shared__ p6::Pipe6Lds L;
#ifdef SEA_NS6_TIMING
    if (threadIdx.x == 0 && blockIdx.x < 16384) {
        g_ns6_wg[4 * blockIdx.x] = (unsigned)wall_clock64();
        g_ns6_wg[4 * blockIdx.x + 2] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
        g_ns6_wg[4 * blockIdx.x + 3] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);
    }
#endif
    p6::ns_pipe6_body<false, SEA_P6D_LOGMOVE != 0, SEA_P6D_DIFG1 != 0>(a, L);
#ifdef SEA_NS6_TIMING
    if (threadIdx.x == 0 && blockIdx.x < 16384) g_ns6_wg[4 * blockIdx.x + 1] = (unsigned)wall_clock64();
#endif
}
__global__ __launch_bounds__(384, 2) void ns_denoise_pipe6_fd_kernel(NsBatchArgs a)
{
    __shared__ p6::Pipe6Lds L;
    p6::ns_pipe6_body<true, SEA_P6_LIGHT_S != 0>(a, L);
}
#endif

} // namespace sea

#if defined(SEA_NS6_TIMING) && !defined(SEA_NS_BODY_ONLY)
extern "C" int sea_debug_ns6_timing(unsigned long long *out16)
{
    return (int)hipMemcpyFromSymbol(out16, HIP_SYMBOL(sea::p6::g_ns6_timing), 16 * sizeof(unsigned long long));
}
extern "C" int sea_debug_ns6_wg(unsigned *out, int n_wg)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(sea::g_ns6_wg), (size_t)n_wg * 4 * sizeof(unsigned));
}
#endif
