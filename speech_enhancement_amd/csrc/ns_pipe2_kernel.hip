/*
 * ns_pipe2_kernel.hip -- etsi_denoise over a packed batch, TWO UTTERANCES PER WORKGROUP: the large-batch form.
 *
 * The four-wave pipeline of ns_pipe_kernel.hip spends most of its vector instructions on phases that use a fraction of
 * the 64 lanes: bin 64 of the 65-bin spectra (one lane's worth of work at the price of a whole FilterCalc pass, in both
 * stages), the mel filter bank (25 lanes), the IDCT (9 lanes), the 17-tap filters (40 lanes), the in-order sums and the
 * scalar recursions (one lane).  With one utterance per wave that is the price of exactness (every sum in the reference's
 * order); with MANY utterances per device -- BASELINE configs[4]: 12 500 per GPU, the kernel is vector-issue-bound -- the
 * idle lanes can carry a SECOND utterance.  A workgroup is five waves around two utterances a, b:
 *
 *   F_a, F_b   the transform wave of ns_pipe_kernel.hip, one per utterance (its dual transform already fills the wave:
 *              stage 0 of frame i beside stage 1 of frame i - 2)
 *   B0         stage 0 of BOTH utterances for frame i - 1: FilterCalc of bins 0..63 one utterance after the other
 *              (lane = bin), then ONE pass each for what is lane-sparse, half-wave = utterance: bin 64 of both, the mel
 *              filter banks (lanes 0..24 | 32..56), the IDCTs (0..8 | 32..40), the 17-tap filters (27 lanes x 3 outputs each)
 *   B1         stage 1 of both for frame i - 3, likewise, with the in-order noise sums and DoGainFact's scalars
 *              (NoiseSup.c:600-637) evaluated per lane instead of wave-uniformly
 *   S          the helper wave of both: SIX in-order chains in its one 80-step dependent stream (VAD sum, denSigSE1 sum
 *              and DC recurrence of a and of b), the second-stage filters, casts and stores
 *
 * Measured (round 4, profiles/r04_ns_pair_form_experiment.txt): 1160 instead of 1491 vector instructions per frame on the
 * configs[4] shard (-22 %: SQ_INSTS_VALU 11.59 G against 14.89 G per launch), bit-identical -- and SLOWER than the four-wave
 * large-batch form: 434 M frames/s (one transform wave for both utterances, kFWaves = 1: four-wave workgroups, four resident
 * per CU) / 401 M (a transform wave per utterance: five-wave workgroups, of which the CU's dispatcher keeps only three
 * resident -- it does not fill the four SIMDs evenly with workgroups whose wave count is not a multiple of four -- ) / 294 M
 * (two pairs per ten-wave workgroup: ONE resident per CU) against 465 M.  What the packing saves in instructions it loses in
 * overlap: each of its waves is a longer dependent chain (lone beat 6.4-7.9 k clk for two frames against ~4.3 k for one), the
 * LDS footprint of two utterances allows 16-20 waves per CU against the four-wave form's 24, and a SIMD that holds four or
 * five latency-bound waves issues 40 % of the time against 59 %.  So this form is NOT chosen by ns_pick_form; it stays as
 * SEA_NS_KERNEL=pair / sea_ns_kernel_form(5) -- the measured answer to VERDICT r03 #1(a) -- and under the same parity tests
 * as every other form (tests/test_gpu_parity.py::test_ns_all_kernel_forms_agree).
 */
#include "ns_core.h"

namespace sea {

namespace p2 {

constexpr int kSlots = 8, kSlotLen = SEA_HOP, kCirc = kSlots * kSlotLen, kMirror = 3 * kSlotLen;
constexpr int kLagS = 4;   /* beats between a frame's intake and its output store */
#ifndef SEA_P2_FWAVES
#define SEA_P2_FWAVES 1
#endif
constexpr int kFWaves = SEA_P2_FWAVES; /* 2: a transform wave per utterance (five-wave workgroups: three resident per CU, see kPairs);
                                        * 1: ONE transform wave runs both utterances, one after the other (four-wave workgroups) */
constexpr int kWaves = 3 + kFWaves;   /* per pair of utterances */
constexpr int kPairs = 1;   /* pairs per workgroup.  Measured (profiles/r04_ns_pair_form_experiment.txt): the CU's dispatcher does not fill
                             * the four SIMDs evenly with workgroups whose wave count is not a multiple of four -- five-wave workgroups:
                             * three resident per CU although registers and LDS allow four (1536 utterances 375 M frames/s, 2048: 340 M);
                             * ten-wave workgroups (kPairs = 2): ONE resident per CU (294 M frames/s on the configs[4] shard) */

/* timing-only diagnostic (-DSEA_P2_TIMING, tools/ns_pair_roles.py): shader clocks the five waves of workgroup 0 spend working /
 * waiting at the beat barrier */
#ifdef SEA_P2_TIMING
__device__ unsigned long long g_p2_timing[16];
#define P2T_DECL unsigned long long tw_ = 0, tb_ = 0, t0_ = 0, t1_ = 0
#define P2T_BEGIN t0_ = clock64()
#define P2T_MID do { t1_ = clock64(); tw_ += t1_ - t0_; } while (0)
#define P2T_END tb_ += clock64() - t1_
#define P2T_FLUSH(w) do { if (blockIdx.x == 0 && threadIdx.x < 64 * kWaves && (threadIdx.x & 63) == 0) { g_p2_timing[2 * (w)] = tw_; g_p2_timing[2 * (w) + 1] = tb_; } } while (0)
/* checkpoints inside one role (-DSEA_P2_CK=<wave>): clocks between successive P2CK(k) of that wave -> g_p2_ck[k] */
__device__ unsigned long long g_p2_ck[8];
#ifdef SEA_P2_CK
#define P2CK_START unsigned long long ckt_ = clock64()
#define P2CK(k) do { const unsigned long long c_ = clock64(); if (blockIdx.x == 0 && threadIdx.x == 64 * (SEA_P2_CK - 2 + kFWaves)) g_p2_ck[k] += c_ - ckt_; ckt_ = c_; } while (0)
#endif
#else
#define P2T_DECL
#define P2T_BEGIN
#define P2T_MID
#define P2T_END
#define P2T_FLUSH(w)
#endif
#ifndef P2CK
#define P2CK_START
#define P2CK(k)
#endif

struct __attribute__((aligned(16))) Rec01 { /* F -> B0, S */
    float psd[68];
    int valid, tick, pad0, pad1;
};
struct __attribute__((aligned(16))) Rec12 { /* B0 -> S */
    float den[68]; /* denSigSE1 of this tick ([65..67] stay zero: S reads them as the chain's tail) */
    int valid, tick, pad0, pad1;
};
struct __attribute__((aligned(16))) Rec34 { /* B1 -> S */
    float fir[20]; /* the 17 taps of the second-stage filter */
    int produced, tick, pad0, pad1;
};
struct __attribute__((aligned(16))) BackScratch {
    float wbuf[68]; /* Wiener gains W[65] */
    float sbuf[68]; /* noiseSE2, summed in order */
    float mel[28];  /* 25 mel gains */
    float fir[20];  /* 17 taps */
};
struct __attribute__((aligned(16))) UttLds {
    float circ[2][kCirc + kMirror]; /* stage-0 / stage-1 sample rings (slots 0..2 mirrored behind the end) */
    float work[kFWaves == 2 ? 512 : 4]; /* the two FFT frames of this utterance's F wave (kFWaves == 1: one work area per pair) */
    BackScratch back[2];
    float ssq[80], sdif[80], sout[80];
    float frameEn[kSlots], denSum[kSlots];
    Rec01 r01[2];
    Rec12 r12[2];
    Rec01 r23[2];
    Rec34 r34[2];
};
struct __attribute__((aligned(16))) PairLds {
    UttLds u[2 * kPairs];
    float work1[kFWaves == 1 ? 512 * kPairs : 4]; /* kFWaves == 1: the transform wave's work area (it runs a, then b) */
    uint4 fftAddr[SEA_FFT_LSTAGES * 64]; /* the transform's operand addresses: identical for both F waves */
    float idctT[SEA_NMEL * 16];          /* mel-IDCT basis rows 0..8: [f][16] */
    float szero[4];
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

/* value of the first lane of each half, as a wave-uniform pair */
__device__ __forceinline__ float half_lane(float v, int h)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), h ? 32 : 0));
}

/* per-lane constants of the packed back phases: the tables' columns of lane & 31 in both halves */
struct PackConst {
    int melStart;
    float melW[SEA_MEL_TAPS];
    float irWin;
    float eps;
};
__device__ __forceinline__ void load_pack_const(PackConst &C, const sea_ns_tables *t, int lane)
{
    const int l5 = lane & 31;
    C.melStart = t->melStart[l5];
#pragma unroll
    for (int i = 0; i < SEA_MEL_TAPS; ++i) C.melW[i] = t->melW[i][l5];
    C.irWin = t->irWin[l5];
    C.eps = t->eps;
}

/* DoMelFB (MelProc.c:82-104) of both utterances: band lane & 31 (< 25) of the half's utterance over its gains wbuf */
__device__ __forceinline__ float mel_fb2(const float *wbuf, const PackConst &C, int lane)
{
    float melOut = 0.0f;
#pragma unroll
    for (int i = 0; i < SEA_MEL_TAPS; ++i) {
        const int idx = C.melStart + i;
        melOut = melOut + wbuf[idx < 65 ? idx : 64] * C.melW[i]; /* taps past the band's length carry weight 0 */
    }
    return ((lane & 31) < SEA_NMEL) ? melOut : 0.0f;
}

/* DoMelIDCT rows 0..8 (MelProc.c:357-378) + mirror + Hanning(17) (NoiseSup.c:660-669) of both utterances: mel / fir are the
 * half's own scratch; ends with wave_sync() */
__device__ __forceinline__ void idct_taps2(float melOut, float *mel, float *fir, const float *idctLds, float irWin, int lane)
{
    const int l5 = lane & 31;
    if (l5 < SEA_NMEL) mel[l5] = melOut;
    wave_sync();
    const int l = (l5 <= 8) ? l5 : 8; /* every lane computes (no divergence), rows 0..8 store */
    float h = 0.0f;
#pragma unroll
    for (int f4 = 0; f4 < 24; f4 += 4) {
        const float4 m = *reinterpret_cast<const float4 *>(&mel[f4]);
        h += m.x * idctLds[(f4 + 0) * 16 + l];
        h += m.y * idctLds[(f4 + 1) * 16 + l];
        h += m.z * idctLds[(f4 + 2) * 16 + l];
        h += m.w * idctLds[(f4 + 3) * 16 + l];
    }
    h += mel[24] * idctLds[24 * 16 + l];
    const float tap = h * irWin;
    if (l5 <= 8) {
        fir[8 + l5] = tap;
        fir[8 - l5] = tap;
    }
    wave_sync();
}

/* ApplyWF (NoiseSup.c:324-340) of both utterances: lane & 31 = l < 27 produces outputs 3l, 3l + 1, 3l + 2 of its half's
 * utterance: y[n] = sum_{j=-8..8} fir[j + 8] * buf[80 + n - j] in that order (buf = the 320-sample stage window) */
__device__ __forceinline__ void fir3(const float *fir, const float *buf, int lane, float &y0, float &y1, float &y2)
{
    float c[SEA_NTAP];
#pragma unroll
    for (int k4 = 0; k4 < 16; k4 += 4) {
        const float4 v = *reinterpret_cast<const float4 *>(&fir[k4]);
        c[k4] = v.x, c[k4 + 1] = v.y, c[k4 + 2] = v.z, c[k4 + 3] = v.w;
    }
    c[16] = fir[16];
    const int l5 = lane & 31, l = (l5 < 27) ? l5 : 26;
    const float *src = buf + 72 + 3 * l; /* x[m] = buf[72 + 3l + m] */
    float x[19];
#pragma unroll
    for (int m = 0; m < 19; ++m) x[m] = src[m];
    y0 = y1 = y2 = 0.0f;
#pragma unroll
    for (int k = 0; k < SEA_NTAP; ++k) {
        y0 += c[k] * x[16 - k];
        y1 += c[k] * x[17 - k];
        y2 += c[k] * x[18 - k];
    }
}

/* DoGainFact's second-stage scalars (NoiseSup.c:600-637), per lane: ns_core.h's gain_fact_update without the
 * wave-uniform shortcuts (every lane of a half holds its utterance's values) */
__device__ __forceinline__ void gain_fact_v(float denEn0, float denEn1, float denEn2, float noiseEn, int nb, float &lowSNRtrack,
                                            float &alfaGF)
{
    float averSNR = (denEn0 * denEn1 * denEn2) / (noiseEn * noiseEn * noiseEn);
    const bool pos = (double)averSNR > 0.00001;
    const float lg = ns_aversnr_expr<false>(pos ? averSNR : 1.0f);
    averSNR = pos ? lg : (float)(-100.0 / 3.0);
    if (((double)(averSNR - lowSNRtrack) < 10.0) || (nb < 10)) {
        float lambdaSNR;
        if (nb < 10)
            lambdaSNR = (float)(1.0 - 1.0 / (double)(float)nb);
        else
            lambdaSNR = (averSNR < lowSNRtrack) ? (float)0.95 : (float)0.99;
        lowSNRtrack = (float)((double)lowSNRtrack + (1.0 - (double)lambdaSNR) * (double)(averSNR - lowSNRtrack));
    }
    if (denEn2 > 100.0f) {
        if ((double)averSNR < ((double)lowSNRtrack + 3.5)) {
            alfaGF = (float)((double)alfaGF + 0.15);
            if ((double)alfaGF > 0.8) alfaGF = (float)0.8;
        } else {
            alfaGF = (float)((double)alfaGF - 0.3);
            if ((double)alfaGF < 0.1) alfaGF = (float)0.1;
        }
    }
}

/* the FilterCalc pass of bins 0..63 of ONE utterance (lane = bin) and what surrounds it: PSDMean (NoiseSup.c:289-303), the
 * frame counter, stage 0's VAD (:359-430), the fast-division domain of ns_core.h::ns_back.  Returns the gain; fast / nb16
 * are handed to the packed bin-64 pass. */
struct LoState {
    float noise, den, prev; /* lane = bin */
    int nbFrame, psdOk;
    /* stage 0 only */
    float meanEn;
    int flagVAD, hangOver, nbSpeech;
};
template <int ST>
__device__ __forceinline__ float lo_pass(const float *psd, LoState &q, float hiNoise, float eps, float frameSum, int lane, bool &fast,
                                         int &nb16, float &nSigHi)
{
    const float nSigLo = psd[lane];
    nSigHi = psd[64];
    const float PLo = (q.prev + nSigLo) * 0.5f;
    q.prev = nSigLo;
    {
        int nb = q.nbFrame;
        if (nb < 2147483647) nb++;
        q.nbFrame = nb;
    }
    if (ST == 0) {
        NsRegs s;
        s.nbFrame[0] = q.nbFrame, s.meanEn = q.meanEn, s.flagVAD = q.flagVAD, s.hangOver = q.hangOver, s.nbSpeech = q.nbSpeech;
        vad_update(s, vad_frame_energy(frameSum));
        q.meanEn = s.meanEn, q.flagVAD = s.flagVAD, q.hangOver = s.hangOver, q.nbSpeech = s.nbSpeech;
    }
    nb16 = (int)(short)q.nbFrame;
    const bool psdOk = (__ballot(!ns_psd_in_domain(nSigLo)) == 0ull) && ns_psd_in_domain(nSigHi);
    const bool noiseOk = (__ballot(!(q.noise <= 0x1p28f && q.noise >= 0x1p-15f)) == 0ull) && hiNoise <= 0x1p28f && hiNoise >= 0x1p-15f;
    fast = SEA_NS_FAST_DIV && psdOk && noiseOk && (q.psdOk != 0);
    q.psdOk = psdOk ? 1 : 0;
    return fast ? filter_bin<ST, true>(PLo, nSigLo, q.noise, q.den, nb16, q.flagVAD, eps)
                : filter_bin<ST, false>(PLo, nSigLo, q.noise, q.den, nb16, q.flagVAD, eps);
}

/* The same cut in two, so that the FilterCalc chains of utterance a, of utterance b and of their bins 64 can sit in ONE basic
 * block and the scheduler can interleave them (a wave issues a dependent instruction every ~8 clk, an independent one every
 * ~2: three chains side by side cost little more than one):  lo_pre = everything before FilterCalc (with its scalar
 * branches), filter_steady = FilterCalc without a branch -- inside the fast-division domain, and for the second stage from
 * frame 11 on (noise_track1's start-up branch) -- the reference's `if (flagVAD == 0)` as a select of the same values. */
struct LoPre {
    float P, nSig, nSigHi;
    bool fast;
    int nb16;
};
template <int ST>
__device__ __forceinline__ LoPre lo_pre(const float *psd, LoState &q, float hiNoise, float frameSum, int lane)
{
    LoPre r;
    r.nSig = psd[lane];
    r.nSigHi = psd[64];
    r.P = (q.prev + r.nSig) * 0.5f;
    q.prev = r.nSig;
    {
        int nb = q.nbFrame;
        if (nb < 2147483647) nb++;
        q.nbFrame = nb;
    }
    if (ST == 0) {
        NsRegs s;
        s.nbFrame[0] = q.nbFrame, s.meanEn = q.meanEn, s.flagVAD = q.flagVAD, s.hangOver = q.hangOver, s.nbSpeech = q.nbSpeech;
        vad_update(s, vad_frame_energy(frameSum));
        q.meanEn = s.meanEn, q.flagVAD = s.flagVAD, q.hangOver = s.hangOver, q.nbSpeech = s.nbSpeech;
    }
    r.nb16 = (int)(short)q.nbFrame;
    const bool psdOk = (__ballot(!ns_psd_in_domain(r.nSig)) == 0ull) && ns_psd_in_domain(r.nSigHi);
    const bool noiseOk = (__ballot(!(q.noise <= 0x1p28f && q.noise >= 0x1p-15f)) == 0ull) && hiNoise <= 0x1p28f && hiNoise >= 0x1p-15f;
    r.fast = SEA_NS_FAST_DIV && psdOk && noiseOk && (q.psdOk != 0);
    q.psdOk = psdOk ? 1 : 0;
    return r;
}
/* S's one dependent instruction stream for BOTH utterances: acc = fma(m, acc, x[n]), n = 0..79, six chains in six lane
 * groups (ns_core.h::helper_chains is the one-utterance form):
 *   lanes 0..7 / 8..15    64 + sum sq[n]          of a / b   (VAD frame sum, NoiseSup.c:386-391)
 *   lanes 16..23 / 24..31 sum den[n], 65 terms    of a / b   (denSigSE1, :597-598; zeros from term 68 on)
 *   lanes 32..47 / 48..63 y = 1023/1024 y + dif[n] of a / b  (DC-offset filter, :182-198, float-FMA form)
 * Lane 32 + j / 48 + j captures y[5j - 1]; those lanes then recompute five outputs each and check the FMA form's exactness
 * condition on their registers (bad[h]).  All six are always computed; the caller discards what it does not need.  Ends
 * with wave_sync(). */
struct Chains2 {
    float vad[2], den[2], y[2];
    bool bad[2];
};
__device__ __forceinline__ void pair_chains(UttLds &A, UttLds &B, const float *denA, const float *denB, const float *zero4, float yA, float yB,
                                            int lane, Chains2 &r)
{
    const int g = lane >> 3; /* 0 VAD a, 1 VAD b, 2 den a, 3 den b, 4-5 DC a, 6-7 DC b */
    const bool isB = (lane < 32) ? (g & 1) != 0 : lane >= 48;
    const bool isDen = g == 2 || g == 3, isDc = lane >= 32;
    const float *src = isDc ? (isB ? B.sdif : A.sdif) : (isDen ? (isB ? denB : denA) : (isB ? B.ssq : A.ssq));
    const float m = isDc ? 0.9990234375f : 1.0f;
    float acc = isDc ? (isB ? yB : yA) : (isDen ? 0.0f : 64.0f);
    constexpr int kChunks = 5, kQ = SEA_HOP / 4 / kChunks, kSeg = 5;
    float4 x[2][kQ];
    auto request = [&](int c, float4(&dstq)[kQ]) {
#pragma unroll
        for (int k = 0; k < kQ; ++k) {
            const int n = 4 * (c * kQ + k);
            const float *p = (n >= 68 && isDen) ? zero4 : src + n;
            dstq[k] = *reinterpret_cast<const float4 *>(p);
        }
    };
    const int seg = lane & 15;
    const float *dif = isB ? B.sdif : A.sdif;
    float *out = isB ? B.sout : A.sout;
    float d5[kSeg];
#pragma unroll
    for (int k = 0; k < kSeg; ++k) d5[k] = dif[kSeg * seg + k];
    float cap = acc; /* segment 0 starts from the incoming state */
    auto step = [&](float xv, int n) {
        float next;
        asm volatile("v_fma_f32 %0, %2, %1, %3" : "=&v"(next) : "v"(m), "v"(acc), "v"(xv));
        if (n > 0 && n % kSeg == 0) {
            const unsigned long long bit = (1ull << (32 + n / kSeg)) | (1ull << (48 + n / kSeg));
            asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(cap) : "v"(acc), "s"(bit));
        }
        acc = next;
    };
    request(0, x[0]);
#pragma unroll
    for (int c = 0; c < kChunks; ++c) {
        if (c + 1 < kChunks) request(c + 1, x[(c + 1) & 1]);
#pragma unroll
        for (int k = 0; k < kQ; ++k) {
            const float4 v = x[c & 1][k];
            const int n = 4 * (c * kQ + k);
            step(v.x, n);
            step(v.y, n + 1);
            step(v.z, n + 2);
            step(v.w, n + 3);
        }
        if (c + 1 < kChunks) __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("" : "+v"(d5[0]), "+v"(d5[1]), "+v"(d5[2]), "+v"(d5[3]), "+v"(d5[4]), "+v"(cap));
    {
        float v = cap;
        bool bad = false;
#pragma unroll
        for (int k = 0; k < kSeg; ++k) {
            bad |= !dc_step_ok(d5[k], v);
            v = __fmaf_rn(0.9990234375f, v, d5[k]);
            if (isDc) out[kSeg * seg + k] = v;
        }
        const unsigned long long mask = __ballot(bad && isDc);
        r.bad[0] = (mask & 0x0000FFFF00000000ull) != 0ull;
        r.bad[1] = (mask & 0xFFFF000000000000ull) != 0ull;
    }
    r.vad[0] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc), 0));
    r.vad[1] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc), 8));
    r.den[0] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc), 16));
    r.den[1] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc), 24));
    r.y[0] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc), 32));
    r.y[1] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc), 48));
    wave_sync();
}

} // namespace p2

__global__ __launch_bounds__(64 * p2::kWaves * p2::kPairs, p2::kFWaves == 2 ? 5 : 4) void ns_denoise_pipe_pair_kernel(NsBatchArgs a)
{
    using namespace p2;
    __shared__ PairLds L;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pair = wave / kWaves;
    const int role = (kFWaves == 2) ? wave - pair * kWaves : ((wave - pair * kWaves) == 0 ? 0 : wave - pair * kWaves + 1); /* 0, 1: F; 2 B0; 3 B1; 4 S */
    UttLds *const U2 = &L.u[2 * pair]; /* this pair's two utterances */
    const int h = lane >> 5, l5 = lane & 31;
    /* the two utterances of this workgroup: neighbours of the launch order (sorted by length) */
    int uu[2];
    long long off[2], nfr[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int pos = 2 * (kPairs * (int)blockIdx.x + pair) + k;
        const bool have = pos < a.n_utt;
        uu[k] = have ? (a.order ? a.order[pos] : pos) : -1;
        off[k] = have ? a.offsets[uu[k]] : 0;
        nfr[k] = have ? a.lengths[uu[k]] / SEA_HOP : 0;
    }
    /* both pairs of the workgroup run the same number of beats: the longest of its four utterances (the launch order is sorted by
     * length: the first position of the workgroup) */
    long long longest = 0;
#pragma unroll
    for (int k = 0; k < 2 * kPairs; ++k) {
        const int pos = 2 * kPairs * (int)blockIdx.x + k;
        if (pos < a.n_utt) {
            const long long n = a.lengths[a.order ? a.order[pos] : pos] / SEA_HOP;
            longest = n > longest ? n : longest;
        }
    }
    const long long niter = longest + kLagS;

    for (int i = threadIdx.x; i < (int)(sizeof(UttLds) / 4) * 2 * kPairs; i += 64 * kWaves * kPairs) reinterpret_cast<float *>(&L.u[0])[i] = 0.0f;
    for (int i = threadIdx.x; i < SEA_NMEL * 16; i += 64 * kWaves * kPairs) L.idctT[i] = a.tables->idct[i >> 4][i & 15];
    if (threadIdx.x < 4) L.szero[threadIdx.x] = 0.0f;
    block_sync();

    if (role < 2) {
        /* ---- F_a / F_b: input + zero-frame gate (ParmInterface.c:244-251); front halves of both stages (ns_pipe_kernel.hip).
         *      kFWaves == 1: ONE transform wave runs utterance a, then utterance b (four-wave workgroup) ---- */
        const int u0 = (kFWaves == 2) ? role : 0, u1 = (kFWaves == 2) ? role + 1 : 2;
        Fft2Regs fft;
        load_fft2_regs<true>(fft, &a.tables->fft, lane, L.fftAddr); /* both F waves write the same words */
        wave_sync();
        float win8[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) win8[k] = a.tables->win8[k][lane];
        struct FState {
            const uint32_t *in32;
            uint32_t nextw;
            int tick, vCur, tCur, v1, t1, v2, t2;
        } st[2];
        auto intake = [&](FState &q, UttLds &U, long long myN, long long f) {
            int ln = lane;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
            const uint32_t w = q.nextw;
            if (f + 1 < myN && ln < 40) q.nextw = q.in32[(f + 1) * 40 + ln];
            const bool any = __ballot(w != 0u) != 0ull;
            q.vCur = 0;
            if (any || q.tick > 0) {
                q.vCur = 1;
                q.tick++;
                const float x0 = (float)(short)(w & 0xFFFFu), x1 = (float)(short)(w >> 16);
                if (ln < 40) slot_store(U.circ[0], q.tick, ln, x0, x1);
            }
            q.tCur = q.tick;
        };
#pragma unroll
        for (int u = 0; u < 2; ++u)
            if (u >= u0 && u < u1) {
                st[u].in32 = reinterpret_cast<const uint32_t *>(a.in + off[u]);
                st[u].nextw = (lane < 40 && nfr[u] > 0) ? st[u].in32[lane] : 0u;
                st[u].tick = st[u].vCur = st[u].tCur = st[u].v1 = st[u].t1 = st[u].v2 = st[u].t2 = 0;
                if (nfr[u] > 0) intake(st[u], U2[u], nfr[u], 0);
            }
        block_sync(); /* the address table is complete before either F wave reads the other's half of it */
        P2T_DECL;
        for (long long i = 0; i < niter; ++i) {
            P2T_BEGIN;
#pragma unroll
            for (int u = 0; u < 2; ++u)
                if (u >= u0 && u < u1) {
                    FState &q = st[u];
                    UttLds &U = U2[u];
                    const long long myN = nfr[u];
                    bool actA = false;
                    Rec01 &rA = U.r01[i & 1];
                    if (i < myN) {
                        actA = q.vCur && q.tCur >= 3; /* nbFramesInFirstStage - nbFramesInSecondStage > 2 (NoiseSup.c:1152) */
                        if (lane == 0) {
                            rA.valid = q.vCur;
                            rA.tick = q.tCur;
                        }
                    }
                    const long long fB = i - 2;
                    bool actB = false;
                    const int tA = q.tCur, tB = q.t2;
                    Rec01 &rB = U.r23[fB & 1];
                    if (fB >= 0 && fB < myN) {
                        actB = q.v2 && tB >= 5; /* nbFramesInSecondStage - nbFramesOut > 2 (NoiseSup.c:1178) */
                        if (lane == 0) {
                            rB.valid = q.v2;
                            rB.tick = tB;
                        }
                    }
                    if (actA || actB) {
                        wave_sync();
                        ns_front_dual<true>(U.circ[0] + window_base(tA), actA, rA.psd, U.circ[1] + window_base(tB), actB, rB.psd,
                                            kFWaves == 2 ? U.work : L.work1 + 512 * pair, fft, win8, lane);
                    }
                    q.v2 = q.v1, q.t2 = q.t1, q.v1 = q.vCur, q.t1 = q.tCur;
                    q.vCur = 0;
                    if (i + 1 < myN) intake(q, U, myN, i + 1);
                }
            P2T_MID;
            block_sync();
            P2T_END;
        }
        P2T_FLUSH(role);
    } else if (role == 2) {
        /* ---- B0: stage 0 of both utterances ---- */
        PackConst C;
        load_pack_const(C, a.tables, lane);
        LoState q[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            q[k].noise = C.eps, q[k].den = 0.0f, q[k].prev = 0.0f, q[k].nbFrame = 0, q[k].psdOk = 1;
            q[k].meanEn = 0.0f, q[k].flagVAD = q[k].hangOver = q[k].nbSpeech = 0;
        }
        float hiNoise = C.eps, hiDen = 0.0f, hiPrev = 0.0f; /* bin 64: lanes 0..31 hold a's, 32..63 b's */
        block_sync();
        P2T_DECL;
        for (long long i = 0; i < niter; ++i) {
            P2T_BEGIN;
            const long long f = i - 1;
            bool act[2] = {false, false};
            int tk[2] = {0, 0};
#pragma unroll
            for (int k = 0; k < 2; ++k)
                if (f >= 0 && f < nfr[k]) {
                    const Rec01 &r = U2[k].r01[f & 1];
                    const int valid = r.valid;
                    tk[k] = r.tick;
                    act[k] = valid && tk[k] >= 3;
                    if (lane == 0) {
                        U2[k].r12[f & 1].valid = valid;
                        U2[k].r12[f & 1].tick = tk[k];
                    }
                }
            if (act[0] || act[1]) {
#if defined(SEA_P2_CK) && SEA_P2_CK == 2
                P2CK_START;
#define CK2(k) P2CK(k)
#else
#define CK2(k)
#endif
                LoPre p[2];
                p[0].P = p[0].nSig = p[0].nSigHi = p[1].P = p[1].nSig = p[1].nSigHi = 0.0f;
                p[0].fast = p[1].fast = true;
                p[0].nb16 = p[1].nb16 = 1;
#pragma unroll
                for (int k = 0; k < 2; ++k)
                    if (act[k]) p[k] = lo_pre<0>(U2[k].r01[f & 1].psd, q[k], half_lane(hiNoise, k), U2[k].frameEn[tk[k] & (kSlots - 1)], lane);
                CK2(0);
                const bool actH = h ? act[1] : act[0];
                const float nSigH = h ? p[1].nSigHi : p[0].nSigHi;
                const float PH = (hiPrev + nSigH) * 0.5f;
                const int nbH = h ? p[1].nb16 : p[0].nb16, vadH = h ? q[1].flagVAD : q[0].flagVAD;
                float nz = hiNoise, dn = hiDen, W0 = 0.0f, W1 = 0.0f, WH;
                if (act[0] && act[1] && p[0].fast && p[1].fast) {
                    /* steady state: bins 0..63 of a, of b and the two bins 64 as three independent chains in one block */
                    W0 = filter_steady<0>(p[0].P, p[0].nSig, q[0].noise, q[0].den, p[0].nb16, q[0].flagVAD, C.eps);
                    W1 = filter_steady<0>(p[1].P, p[1].nSig, q[1].noise, q[1].den, p[1].nb16, q[1].flagVAD, C.eps);
                    WH = filter_steady<0>(PH, nSigH, nz, dn, nbH, vadH, C.eps);
                } else {
                    if (act[0])
                        W0 = p[0].fast ? filter_bin<0, true>(p[0].P, p[0].nSig, q[0].noise, q[0].den, p[0].nb16, q[0].flagVAD, C.eps)
                                       : filter_bin<0, false>(p[0].P, p[0].nSig, q[0].noise, q[0].den, p[0].nb16, q[0].flagVAD, C.eps);
                    if (act[1])
                        W1 = p[1].fast ? filter_bin<0, true>(p[1].P, p[1].nSig, q[1].noise, q[1].den, p[1].nb16, q[1].flagVAD, C.eps)
                                       : filter_bin<0, false>(p[1].P, p[1].nSig, q[1].noise, q[1].den, p[1].nb16, q[1].flagVAD, C.eps);
                    WH = (p[0].fast && p[1].fast) ? filter_bin<0, true>(PH, nSigH, nz, dn, nbH, vadH, C.eps)
                                                  : filter_bin<0, false>(PH, nSigH, nz, dn, nbH, vadH, C.eps);
                }
                hiPrev = actH ? nSigH : hiPrev;
                hiNoise = actH ? nz : hiNoise;
                hiDen = actH ? dn : hiDen;
                if (act[0]) {
                    U2[0].back[0].wbuf[lane] = W0;
                    U2[0].r12[f & 1].den[lane] = q[0].den;
                }
                if (act[1]) {
                    U2[1].back[0].wbuf[lane] = W1;
                    U2[1].r12[f & 1].den[lane] = q[1].den;
                }
                if (actH && l5 == 0) {
                    U2[h].back[0].wbuf[64] = WH;
                    U2[h].r12[f & 1].den[64] = dn;
                }
                wave_sync();
                CK2(1);
                UttLds &U = U2[h];
                float melOut = mel_fb2(U.back[0].wbuf, C, lane);
                CK2(2);
                idct_taps2(melOut, U.back[0].mel, U.back[0].fir, L.idctT, C.irWin, lane);
                CK2(3);
                const int tkH = h ? tk[1] : tk[0];
                float y0, y1, y2;
                fir3(U.back[0].fir, U.circ[0] + window_base(tkH), lane, y0, y1, y2);
                CK2(4);
                if (actH && l5 < 27) {
                    const int slot = tkH & (kSlots - 1);
                    float *p = U.circ[1] + slot * kSlotLen + 3 * l5;
                    p[0] = y0;
                    p[1] = y1;
                    if (l5 < 26) p[2] = y2;
                    if (slot < 3) {
                        p[kCirc] = y0;
                        p[kCirc + 1] = y1;
                        if (l5 < 26) p[kCirc + 2] = y2;
                    }
                }
            }
            P2T_MID;
            block_sync();
            P2T_END;
        }
        P2T_FLUSH(2);
    } else if (role == 3) {
        /* ---- B1: stage 1 of both utterances ---- */
        PackConst C;
        load_pack_const(C, a.tables, lane);
        LoState q[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            q[k].noise = C.eps, q[k].den = 0.0f, q[k].prev = 0.0f, q[k].nbFrame = 0, q[k].psdOk = 1;
            q[k].meanEn = 0.0f, q[k].flagVAD = q[k].hangOver = q[k].nbSpeech = 0;
        }
        float hiNoise = C.eps, hiDen = 0.0f, hiPrev = 0.0f;
        float lowSNRtrack = 0.0f, alfaGF = (float)0.8; /* per half */
        block_sync();
        P2T_DECL;
        for (long long i = 0; i < niter; ++i) {
            P2T_BEGIN;
            const long long f = i - 3;
            bool act[2] = {false, false};
            int tk[2] = {0, 0};
#pragma unroll
            for (int k = 0; k < 2; ++k)
                if (f >= 0 && f < nfr[k]) {
                    const Rec01 &r = U2[k].r23[f & 1];
                    tk[k] = r.tick;
                    act[k] = r.valid && tk[k] >= 5;
                    if (lane == 0) {
                        U2[k].r34[f & 1].produced = act[k] ? 1 : 0;
                        U2[k].r34[f & 1].tick = tk[k];
                    }
                }
            if (act[0] || act[1]) {
#if defined(SEA_P2_CK) && SEA_P2_CK == 3
                P2CK_START;
#define CK3(k) P2CK(k)
#else
#define CK3(k)
#endif
                LoPre p[2];
                p[0].P = p[0].nSig = p[0].nSigHi = p[1].P = p[1].nSig = p[1].nSigHi = 0.0f;
                p[0].fast = p[1].fast = true;
                p[0].nb16 = p[1].nb16 = 1;
#pragma unroll
                for (int k = 0; k < 2; ++k)
                    if (act[k]) p[k] = lo_pre<1>(U2[k].r23[f & 1].psd, q[k], half_lane(hiNoise, k), 0.0f, lane);
                CK3(0);
                const bool actH = h ? act[1] : act[0];
                const float nSigH = h ? p[1].nSigHi : p[0].nSigHi;
                const float PH = (hiPrev + nSigH) * 0.5f;
                const int nbH = h ? p[1].nb16 : p[0].nb16;
                float nz = hiNoise, dn = hiDen, W0 = 0.0f, W1 = 0.0f, WH;
                if (act[0] && act[1] && p[0].fast && p[1].fast && p[0].nb16 >= 11 && p[1].nb16 >= 11) {
                    W0 = filter_steady<1>(p[0].P, p[0].nSig, q[0].noise, q[0].den, p[0].nb16, 0, C.eps);
                    W1 = filter_steady<1>(p[1].P, p[1].nSig, q[1].noise, q[1].den, p[1].nb16, 0, C.eps);
                    WH = filter_steady<1>(PH, nSigH, nz, dn, nbH, 0, C.eps);
                } else {
                    if (act[0])
                        W0 = p[0].fast ? filter_bin<1, true>(p[0].P, p[0].nSig, q[0].noise, q[0].den, p[0].nb16, 0, C.eps)
                                       : filter_bin<1, false>(p[0].P, p[0].nSig, q[0].noise, q[0].den, p[0].nb16, 0, C.eps);
                    if (act[1])
                        W1 = p[1].fast ? filter_bin<1, true>(p[1].P, p[1].nSig, q[1].noise, q[1].den, p[1].nb16, 0, C.eps)
                                       : filter_bin<1, false>(p[1].P, p[1].nSig, q[1].noise, q[1].den, p[1].nb16, 0, C.eps);
                    WH = (p[0].fast && p[1].fast) ? filter_bin<1, true>(PH, nSigH, nz, dn, nbH, 0, C.eps)
                                                  : filter_bin<1, false>(PH, nSigH, nz, dn, nbH, 0, C.eps);
                }
                hiPrev = actH ? nSigH : hiPrev;
                hiNoise = actH ? nz : hiNoise;
                hiDen = actH ? dn : hiDen;
                if (act[0]) {
                    U2[0].back[1].wbuf[lane] = W0;
                    U2[0].back[1].sbuf[lane] = q[0].noise;
                }
                if (act[1]) {
                    U2[1].back[1].wbuf[lane] = W1;
                    U2[1].back[1].sbuf[lane] = q[1].noise;
                }
                if (actH && l5 == 0) {
                    U2[h].back[1].wbuf[64] = WH;
                    U2[h].back[1].sbuf[64] = nz;
                }
                wave_sync();
                CK3(1);
                UttLds &U = U2[h];
                const int tkH = h ? tk[1] : tk[0];
                /* in-order sum of the 65 noise magnitudes (NoiseSup.c:600-601) and the gain-factor scalars, per half */
                const float noiseEn = serial_sum<65>(U.back[1].sbuf, 0.0f);
                CK3(2);
                const float d0 = U.denSum[(tkH - 2) & (kSlots - 1)], d1 = U.denSum[(tkH - 1) & (kSlots - 1)], d2 = U.denSum[tkH & (kSlots - 1)];
                {
                    float lt = lowSNRtrack, al = alfaGF;
                    gain_fact_v(d0, d1, d2, actH ? noiseEn : 1.0f, h ? q[1].nbFrame : q[0].nbFrame, lt, al);
                    lowSNRtrack = actH ? lt : lowSNRtrack;
                    alfaGF = actH ? al : alfaGF;
                }
                CK3(3);
                float melOut = mel_fb2(U.back[1].wbuf, C, lane);
                melOut = (float)((double)(alfaGF * melOut) + (1.0 - (double)alfaGF) * 1.0); /* :639-640 */
                /* the 17 taps straight into the record S reads one beat later (an inactive half writes into its own scratch) */
                CK3(4);
                idct_taps2(melOut, U.back[1].mel, actH ? U.r34[f & 1].fir : U.back[1].fir, L.idctT, C.irWin, lane);
                CK3(5);
            }
            P2T_MID;
            block_sync();
            P2T_END;
        }
        P2T_FLUSH(3);
    } else {
        /* ---- S: the scalar chains, second-stage filters, casts and stores of both utterances ---- */
        float dcX = 0.0f, dcY = 0.0f; /* per half: prevSamples (NoiseSup.c:908-909) */
        int firstOut[2] = {-1, -1};
        block_sync();
        P2T_DECL;
        for (long long i = 0; i < niter; ++i) {
            P2T_BEGIN;
#if defined(SEA_P2_CK) && SEA_P2_CK == 4
            P2CK_START;
#define CK4(k) P2CK(k)
#else
#define CK4(k)
#endif
            const long long fp = i - 1, fd = i - 2, fo = i - kLagS;
            bool doVad[2] = {false, false}, doDen[2] = {false, false}, produced[2] = {false, false}, haveOut[2] = {false, false};
            int tp[2] = {0, 0}, td[2] = {0, 0}, to[2] = {0, 0};
            const float *denSrc[2] = {U2[0].r12[0].den, U2[1].r12[0].den};
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                UttLds &U = U2[k];
                if (fp >= 0 && fp < nfr[k]) {
                    doVad[k] = U.r01[fp & 1].valid != 0;
                    tp[k] = U.r01[fp & 1].tick;
                }
                if (fd >= 0 && fd < nfr[k]) {
                    const Rec12 &r = U.r12[fd & 1];
                    doDen[k] = r.valid && r.tick >= 3;
                    td[k] = r.tick;
                    denSrc[k] = r.den;
                }
                haveOut[k] = fo >= 0 && fo < nfr[k];
                if (haveOut[k]) {
                    produced[k] = U.r34[fo & 1].produced != 0;
                    to[k] = U.r34[fo & 1].tick;
                }
                if (doVad[k]) { /* squares of the frame pushed one beat ago (the VAD's "current frame" two ticks later) */
                    const float *frame = U.circ[0] + (tp[k] & (kSlots - 1)) * kSlotLen;
                    const float x = frame[lane];
                    U.ssq[lane] = x * x;
                    if (lane < 16) {
                        const float yv = frame[64 + lane];
                        U.ssq[64 + lane] = yv * yv;
                    }
                }
            }
            CK4(0);
            const bool prodH = h ? produced[1] : produced[0];
            if (produced[0] || produced[1]) {
                /* second-stage ApplyWF of both utterances (3 outputs per lane), then the DC filter's input differences
                 * d[n] = y[n] - y[n-1] (NoiseSup.c:190-194) across lanes with a one-lane DPP shift */
                UttLds &U = U2[h];
                const int toH = h ? to[1] : to[0];
                float y0, y1, y2;
                fir3(U.r34[fo & 1].fir, U.circ[1] + window_base(toH), lane, y0, y1, y2);
                float below = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(dcX), __float_as_int(y2), 0x138 /* wave_shr:1 */, 0xf, 0xf, false));
                below = (l5 == 0) ? dcX : below;
                const float e0 = y0 - below, e1 = y1 - y0, e2 = y2 - y1;
                const float last = h ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(y1), 58))
                                     : __int_as_float(__builtin_amdgcn_readlane(__float_as_int(y1), 26)); /* y[79] */
                dcX = prodH ? last : dcX;
                if (prodH && l5 < 27) {
                    U.sdif[3 * l5] = e0;
                    U.sdif[3 * l5 + 1] = e1;
                    if (l5 < 26) U.sdif[3 * l5 + 2] = e2;
                }
            }
            CK4(1);
            if (doVad[0] || doVad[1] || doDen[0] || doDen[1] || produced[0] || produced[1]) {
                wave_sync();
                Chains2 r;
                pair_chains(U2[0], U2[1], denSrc[0], denSrc[1], L.szero, half_lane(dcY, 0), half_lane(dcY, 1), lane, r);
                CK4(2);
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    UttLds &U = U2[k];
                    if (doVad[k] && lane == 0) U.frameEn[(tp[k] + 2) & (kSlots - 1)] = r.vad[k]; /* 64 + sum of squares; B0 takes the log */
                    if (doDen[k] && lane == 0) U.denSum[td[k] & (kSlots - 1)] = r.den[k];
                    if (produced[k]) {
                        float y = r.y[k];
                        if (r.bad[k]) dc_redo_exact(U.sdif, U.sout, half_lane(dcY, k), y); /* never yet observed */
                        if ((h == k)) dcY = y;
                        if (firstOut[k] < 0) firstOut[k] = (int)fo;
                    }
                }
            }
            CK4(3);
            /* int16 cast (ParmInterface.c:266) and store: lane & 31 < 20 stores four samples of its half's utterance;
             * etsi_denoise copies zeros until the first NoiseSup output (AdvFrontEnd.c:186-190) */
            {
                const bool haveH = h ? haveOut[1] : haveOut[0];
                if (haveH && l5 < 20) {
                    UttLds &U = U2[h];
                    const long long o = (h ? off[1] : off[0]) + fo * SEA_HOP + 4 * l5;
                    uint2 packed = make_uint2(0u, 0u);
                    if (prodH) {
                        const float4 v = *reinterpret_cast<const float4 *>(&U.sout[4 * l5]);
                        packed.x = (uint32_t)cast_i16(v.x) | ((uint32_t)cast_i16(v.y) << 16);
                        packed.y = (uint32_t)cast_i16(v.z) | ((uint32_t)cast_i16(v.w) << 16);
                        if (a.out_f32) *reinterpret_cast<float4 *>(a.out_f32 + o) = v;
                    }
                    *reinterpret_cast<uint2 *>(a.out + o) = packed;
                }
            }
            wave_sync();
            CK4(4);
            P2T_MID;
            block_sync();
            P2T_END;
        }
        P2T_FLUSH(4);
        if (a.first_out && lane == 0) {
            if (uu[0] >= 0) a.first_out[uu[0]] = firstOut[0];
            if (uu[1] >= 0) a.first_out[uu[1]] = firstOut[1];
        }
    }
}

} // namespace sea

namespace sea {
int ns_pair_threads() { return 64 * p2::kWaves * p2::kPairs; }
} // namespace sea

#ifdef SEA_P2_TIMING
extern "C" int sea_debug_p2_timing(unsigned long long *out16)
{
    return (int)hipMemcpyFromSymbol(out16, HIP_SYMBOL(sea::p2::g_p2_timing), 16 * sizeof(unsigned long long));
}
extern "C" int sea_debug_p2_ck(unsigned long long *out8, int reset)
{
    if (reset) {
        unsigned long long z[8] = {0};
        return (int)hipMemcpyToSymbol(HIP_SYMBOL(sea::p2::g_p2_ck), z, sizeof z);
    }
    return (int)hipMemcpyFromSymbol(out8, HIP_SYMBOL(sea::p2::g_p2_ck), 8 * sizeof(unsigned long long));
}
#endif
