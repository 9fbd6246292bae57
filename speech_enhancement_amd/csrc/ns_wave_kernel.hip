/*
 * ns_wave_kernel.hip -- etsi_denoise over a packed batch, ONE WAVEFRONT PER UTTERANCE with the roles of the pipelined forms run one
 * after the other: the form for LARGE batches (BASELINE configs[4]: 12 500 utterances per GPU).
 *
 * What round 4 measured about that regime (profiles/r04_ns_six_wave_dense.txt, section 5): it is bound by vector issue, and it wants
 * UTTERANCES per CU, not waves per utterance -- six four-wave workgroups per CU (table-in-LDS form) 465 M frames/s, four of them
 * (VGPR-table form) 409, five six-wave workgroups 400.  The limit of that direction is a wave that owns its utterance: no workgroup
 * barrier, no double-buffered records, no pipeline fill and drain, and as many independent dependent-instruction streams per SIMD as
 * the register file and the LDS hold.  The arithmetic is ns_core.h's, role by role, in an order that needs ONE frame of lag:
 *
 *   beat i:  intake of frame i (zero-frame gate, push into the stage-0 ring)
 *            dual transform: stage 0 of frame i beside stage 1 of frame i - 1          (ns_front_dual, as the F wave)
 *            BACK of stage 1, frame i - 1: noise tracking, gains, mel, IDCT, 17-tap filter -> DC-filter differences   (as B1 + S)
 *            BACK of stage 0, frame i: VAD, gains, mel, IDCT, 17-tap filter -> stage-1 ring                            (as B0)
 *            the three in-order chains in one stream (helper_chains): VAD sum of frame i (used two ticks later), denSigSE1 sum of
 *            frame i (used by stage 1 next beat), DC-offset recurrence of frame i - 1; cast + store of output frame i - 1
 *
 * (stage 1 runs before stage 0 inside a beat: with four-slot rings stage 0's new frame overwrites the slot whose last eight samples
 * stage 1's filter still reads).  Per utterance 9.1 KB of LDS (two four-slot sample rings with three mirrored slots, the transform
 * work area, one back-half scratch, the chains' scratch); the transform's operand-address table and the IDCT basis are shared by
 * the waves of a workgroup.
 *
 * MEASURED (round 4, end): bit-identical to every other form at the first run (test_ns_all_kernel_forms_agree, form 7) -- and
 * 363 M frames/s on the configs[4] shard against the table-in-LDS form's 465.  A wave that runs every role holds every role's
 * constants and state: 167 VGPRs -> three waves per SIMD, twelve utterances per CU, vector issue ~48 % busy; compiled for four waves
 * per SIMD (128 VGPRs, 25 spilled registers, eight utterances per workgroup) 303 M.  The four-wave forms give an utterance 4 x 80
 * registers in four waves and still fit six utterances per CU; this form would need its constants (window, twiddles, mel weights:
 * ~40 VGPRs) in LDS to reach sixteen waves per CU without spills, for an extrapolated ~480 M.  NOT chosen by ns_pick_form; kept as
 * SEA_NS_KERNEL=wave / sea_ns_kernel_form(7), the measured end point of "utterances per CU instead of waves per utterance".
 */
#include "ns_core.h"

namespace sea {

namespace w1 {

#ifndef SEA_W1_WAVES
#define SEA_W1_WAVES 4 /* utterances (= waves) per workgroup.  4: 43 KB, three workgroups = 12 waves per CU at 167 VGPRs, 363 M frames/s on
                        * the configs[4] shard; 8 (77.8 KB, two workgroups = 16 waves per CU, which needs 128 VGPRs: 25 spilled): 303 M */
#endif
#ifndef SEA_W1_MINWAVES
#define SEA_W1_MINWAVES 4 /* waves per SIMD the register allocation leaves room for */
#endif
#ifndef SEA_W1_RL
#define SEA_W1_RL 0 /* 1: the lane-read forms of the in-order sums and the IDCT (fewer LDS trips, more vector instructions) */
#endif
constexpr int kWaves = SEA_W1_WAVES;
constexpr int kSlots = 4, kSlotLen = SEA_HOP, kCirc = kSlots * kSlotLen, kMirror = 3 * kSlotLen;
constexpr bool kRL = SEA_W1_RL != 0;

struct __attribute__((aligned(16))) WaveLds {
    float circ[2][kCirc + kMirror]; /* stage-0 / stage-1 sample rings: four 80-sample slots, slots 0..2 repeated behind the end */
    float work[512];                /* the two frames of the dual transform */
    float psd0[68], psd1[68];
    BackLds back;                   /* scratch of the BACK halves (one after the other) */
    float den[68];                  /* denSigSE1 of this beat's stage-0 frame; [65..67] stay zero (helper_chains) */
    float ssq[80], sdif[80];        /* the chains' inputs; their output frame goes to back.sq (free once stage 0's filter output has left it) */
    float frameEn[8], denSum[8];    /* by tick & 7 */
    float szero[4];
};
struct __attribute__((aligned(16))) GroupLds {
    uint4 fftAddr[SEA_FFT_LSTAGES * 64]; /* the transform's operand addresses: identical for every wave */
    float idctT[SEA_NMEL * 16];
    WaveLds w[kWaves];
};

__device__ __forceinline__ int window_base(int tick) { return ((tick - 3) & (kSlots - 1)) * kSlotLen; }
__device__ __forceinline__ void slot_store(float *circ, int tick, int lane, float a, float b)
{
    const int slot = tick & (kSlots - 1);
    float *p = circ + slot * kSlotLen + 2 * lane;
    *reinterpret_cast<float2 *>(p) = make_float2(a, b);
    if (slot < 3) *reinterpret_cast<float2 *>(p + kCirc) = make_float2(a, b);
}

} // namespace w1

__global__ __launch_bounds__(64 * w1::kWaves, SEA_W1_MINWAVES) void ns_denoise_wave_kernel(NsBatchArgs a)
{
    using namespace w1;
    __shared__ GroupLds G;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    WaveLds &L = G.w[wave];

    /* shared tables (every wave writes the same words of fftAddr) and this wave's own area */
    Fft2Regs fft;
    load_fft2_regs<true>(fft, &a.tables->fft, lane, G.fftAddr);
    for (int i = threadIdx.x; i < SEA_NMEL * 16; i += 64 * kWaves) G.idctT[i] = a.tables->idct[i >> 4][i & 15];
    for (int i = lane; i < (int)(sizeof(WaveLds) / sizeof(float)); i += 64) reinterpret_cast<float *>(&L)[i] = 0.0f;
    __syncthreads();

    const long long idx = (long long)blockIdx.x * kWaves + wave;
    if (idx >= a.n_utt) return; /* no barrier below this line */
    const int u = a.order ? a.order[idx] : (int)idx;
    const long long off = a.offsets[u];
    const long long nfr = a.lengths[u] / SEA_HOP;

    float win8[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) win8[k] = a.tables->win8[k][lane];
    NsConst C;
    C.melStart = a.tables->melStart[lane];
    C.melLen = a.tables->melLen[lane];
#pragma unroll
    for (int i = 0; i < SEA_MEL_TAPS; ++i) C.melW[i] = a.tables->melW[i][lane];
    C.irWin = a.tables->irWin[lane];
    C.eps = a.tables->eps;
    NsRegs s;
    regs_init(s, C.eps);

    const uint32_t *in32 = reinterpret_cast<const uint32_t *>(a.in + off);
    uint32_t *out32 = reinterpret_cast<uint32_t *>(a.out + off);
    float *outf = a.out_f32 ? a.out_f32 + off : nullptr;
    uint32_t nextw = (lane < 40 && nfr > 0) ? in32[lane] : 0u;
    int tick = 0, v1 = 0, t1 = 0; /* frames seen since the first non-zero one; (valid, tick) of frame i - 1 */
    float dcX = 0.0f, dcY = 0.0f; /* prevSamples, NoiseSup.c:908-909 */
    int firstOut = -1;

    for (long long i = 0; i <= nfr; ++i) {
        /* ---- intake of frame i: zero-frame gate (ParmInterface.c:244-251), int16 -> float, push ---- */
        int vCur = 0;
        if (i < nfr) {
            const uint32_t w = nextw;
            if (i + 1 < nfr && lane < 40) nextw = in32[(i + 1) * 40 + lane];
            const bool any = __ballot(w != 0u) != 0ull;
            if (any || tick > 0) {
                vCur = 1;
                tick++;
                const float x0 = (float)(short)(w & 0xFFFFu), x1 = (float)(short)(w >> 16);
                if (lane < 40) slot_store(L.circ[0], tick, lane, x0, x1);
            }
        }
        const int tCur = tick;
        /* nbFramesInFirstStage - nbFramesInSecondStage > 2 (NoiseSup.c:1152) <=> tick >= 3; second stage (:1178) <=> tick >= 5 */
        const bool actA = vCur && tCur >= 3;
        const bool actB = v1 && t1 >= 5;

        /* ---- both front halves side by side: stage 0 of frame i, stage 1 of frame i - 1 ---- */
        if (actA || actB) {
            wave_sync();
            ns_front_dual<true>(L.circ[0] + window_base(tCur), actA, L.psd0, L.circ[1] + window_base(t1), actB, L.psd1, L.work, fft,
                                win8, lane);
        }

        /* ---- BACK of stage 1, frame i - 1; its filter output as the DC filter's input differences ---- */
        bool produced = false;
        if (actB) {
            s.denEn0 = L.denSum[(t1 - 2) & 7]; /* denEn1[0..2] (NoiseSup.c:595-598) = sums of denSigSE1 of ticks t-2, t-1, t */
            s.denEn1 = L.denSum[(t1 - 1) & 7];
            s.denEn2 = L.denSum[t1 & 7];
            ns_back<1, true, false, true, kRL>(L.psd1, L.circ[1] + window_base(t1), L.back, s, C, L.back.fir, lane, 0.0f, nullptr, G.idctT);
            float d0, d1;
            dcX = ns_fir_dif(L.back.fir, L.circ[1] + window_base(t1), lane, dcX, d0, d1);
            if (lane < 40) *reinterpret_cast<float2 *>(&L.sdif[2 * lane]) = make_float2(d0, d1);
            produced = true;
        }

        /* ---- BACK of stage 0, frame i; its 80 outputs enter the stage-1 ring ---- */
        if (actA) {
            float *tmp = L.back.sq;
            ns_back<0, true, false, false, kRL>(L.psd0, L.circ[0] + window_base(tCur), L.back, s, C, tmp, lane,
                                                vad_frame_energy(L.frameEn[tCur & 7]), L.den, G.idctT);
            if (lane < 40) {
                const float2 v = *reinterpret_cast<const float2 *>(tmp + 2 * lane);
                slot_store(L.circ[1], tCur, lane, v.x, v.y);
            }
        }

        /* ---- the three in-order chains in one stream: VAD sum (64 + sum of squares) of frame i, denSigSE1 sum of frame i,
         *      DC-offset recurrence (NoiseSup.c:182-198) of frame i - 1 ---- */
        float2 vOut = make_float2(0.0f, 0.0f);
        if (vCur || produced) {
            if (vCur) {
                const float *frame = L.circ[0] + (tCur & (kSlots - 1)) * kSlotLen;
                const float x = frame[lane];
                L.ssq[lane] = x * x;
                if (lane < 16) {
                    const float yv = frame[64 + lane];
                    L.ssq[64 + lane] = yv * yv;
                }
            }
            wave_sync();
            float vadSum, denTotal, y = dcY;
            helper_chains<10>(L.ssq, L.den, L.sdif, L.back.sq, L.szero, vadSum, denTotal, y, lane);
            if (vCur && lane == 0) L.frameEn[(tCur + 2) & 7] = vadSum; /* the "current frame" of tick t + 2; its consumer takes the log */
            if (actA && lane == 0) L.denSum[tCur & 7] = denTotal;
            if (produced) {
                vOut = dc_verify_take(L.sdif, L.back.sq, dcY, y, lane);
                dcY = y;
            }
        }

        /* ---- what etsi_denoise copies for frame i - 1 (AdvFrontEnd.c:186-190): zeros until the first NoiseSup output ---- */
        const long long fo = i - 1;
        if (fo >= 0) {
            if (produced && firstOut < 0) firstOut = (int)fo;
            if (lane < 40) {
                uint32_t packed = 0u;
                if (produced) {
                    packed = (uint32_t)cast_i16(vOut.x) | ((uint32_t)cast_i16(vOut.y) << 16);
                    if (outf) *reinterpret_cast<float2 *>(outf + fo * SEA_HOP + 2 * lane) = vOut;
                }
                out32[fo * 40 + lane] = packed;
            }
        }
        wave_sync();
        v1 = vCur;
        t1 = tCur;
    }
    if (a.first_out && lane == 0) a.first_out[u] = firstOut;
}

int ns_wave_utts_per_block() { return w1::kWaves; }

} // namespace sea
