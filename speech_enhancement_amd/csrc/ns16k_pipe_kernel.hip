/*
 * ns16k_pipe_kernel.hip -- the 16 k-native NoiseSup variant as FOUR PIPELINED WAVEFRONTS per stream.
 *
 *     function/20141106_speech_enhancement/aurora_etsi/NoiseSup.cpp:1140-1407   etsi_denoise_mapping_func_Wiener
 *     function/20141106_speech_enhancement/aurora_etsi/NoiseSup.h:36-53         hop 160, window 480 at offset 80 of a 640-sample
 *                                                                               stage buffer, 129 spectral values, rfft (x, 512, 8)
 *     function/20141106_speech_enhancement/aurora_etsi/MelProc.cpp:119-135,556-576   DoGamma, DoGammaIDCT
 *
 * Round 3's form (ns16k_kernel.hip, kept: SEA_NS16K_KERNEL=single) runs a stream on ONE wavefront: 1024 streams are one
 * latency-bound wave per SIMD, 45 k clk per frame, 59 % of its LDS cycles bank conflicts of the un-swizzled transform.  The
 * frame recursion is the same software pipeline as etsi/'s (ns_pipe_kernel.hip), so a stream gets four role waves, one
 * s_barrier per frame (beat), frame f at beat:
 *
 *   f      S  intake: the frame gate's in-order sum of squares (:1160-1171) and the VAD's frame sum (:373-376) ride in two
 *             lanes of S's ONE 160-step dependent stream (the DC recurrence of the frame leaving the pipeline rides in the
 *             other half); a frame that passes the gate gets the next TICK and its slot of the stage-0 ring
 *   f + 1  F  window + rfft (x, 512, 8) + FFTtoPSD of stage 0 -- and, side by side in the same instruction stream, of stage 1
 *             for frame f - 2: register-resident start (length-2, n2 = 4, n2 = 8 on the lane's eight places), levels
 *             n2 = 16 .. 256 one work item per lane on an XOR-swizzled work area (sea_tables.c::build_ns16k_pipe)
 *   f + 2  B0 stage 0: PSDMean, VAD, FilterCalc of 129 values, SpeechQ measures, DoGamma (25 in-order chains of 128 terms
 *             + 3 riding sums), IDCT, 17-tap filter -> stage-1 ring
 *   f + 3  F  (stage 1 of frame f)
 *   f + 4  B1 stage 1: ..., gain factorisation, IDCT -> the 17 taps
 *   f + 5  S  second-stage filter, DC-offset filter, store
 *
 * Two streams (eight waves) share a workgroup and with it the 13.7 KB of DoGamma coefficients in LDS; every stream of a
 * push has the same number of frames, so the lock step costs nothing.  All recursive state lives in the role waves'
 * registers and per-stream LDS; between pushes in the same kNs16StateFloats-float blob the one-wave form uses (linear
 * stage buffers: a stream may change kernels between two pushes).  Arithmetic: ns_core.h's and ns16k_kernel.hip's,
 * operation by operation; bit-identical to the one-wave form and to oracle/ns16k_oracle.c (tests/test_gpu_ns16k.py).
 */
#include "ns_core.h"

#ifndef SEA16P_LOG_IN_S
#define SEA16P_LOG_IN_S 1
#endif

namespace sea {

namespace p16 {

constexpr int kHop = SEA16_HOP, kSpec = SEA16_NSPEC, kSpecPad = 132;
constexpr int kSlots = 8, kCirc = kSlots * kHop, kMirror = 3 * kHop; /* slots 0..2 repeated behind the end: every 640-run is contiguous */
constexpr int kStreams = 2;                                           /* per workgroup */
constexpr int kLag = 5;                                               /* beats between a frame's intake and its store */

/* timing-only diagnostic (-DSEA16P_TIMING, tools/ns16k_roles.py): shader clocks the roles of stream 0 spend working /
 * waiting at the beat barrier */
#ifdef SEA16P_TIMING
__device__ unsigned long long g_ns16p_timing[16];
#define T16_DECL unsigned long long tw_ = 0, tb_ = 0, t0_ = 0, t1_ = 0
#define T16_BEGIN t0_ = clock64()
#define T16_MID do { t1_ = clock64(); tw_ += t1_ - t0_; } while (0)
#define T16_END tb_ += clock64() - t1_
#define T16_FLUSH(role) do { if (blockIdx.x == 0 && threadIdx.x == 64 * (role)) { g_ns16p_timing[2 * (role)] = tw_; g_ns16p_timing[2 * (role) + 1] = tb_; } } while (0)
#else
#define T16_DECL
#define T16_BEGIN
#define T16_MID
#define T16_END
#define T16_FLUSH(role)
#endif

struct __attribute__((aligned(16))) RecPsd { /* F -> B0 / B1 */
    float psd[kSpecPad];
    int valid, tick, pad0, pad1;
};
struct __attribute__((aligned(16))) RecIn { /* S -> F */
    int valid, tick, pad0, pad1;
};
struct __attribute__((aligned(16))) RecTaps { /* B1 -> S */
    float fir[20];
    int produced, tick, pad0, pad1;
};

struct __attribute__((aligned(16))) StreamLds {
    float circ[2][kCirc + kMirror]; /* stage-0 / stage-1 sample rings, slot = tick & 7 */
    float work[2][SEA16_NFFT];      /* F: the two transforms' work areas (swizzled) */
    float bins[6][kSpecPad];        /* noiseSE1/2, denSigSE1/2, previous PSD 1/2 */
    float W[2][kSpecPad];           /* Wiener gains of B0 / B1 */
    float gam[2][32], fir[2][20];   /* their window outputs and taps */
    float ssq[kHop], sdif[kHop], sout[kHop]; /* S: squares of the frame taken in, the DC filter's differences, its output */
    float frameEn[kSlots];          /* 64 + in-order sum of squares of the frame of tick t at [t & 7] */
    float denSum[kSlots];           /* sum of denSigSE1 of tick t at [t & 7] */
    RecIn rin[2];
    RecPsd r01[2], r23[2];
    RecTaps r34[2];
    int tickEnd, psdOkEnd[2], padEnd; /* what the role waves leave for the state blob */
};
struct __attribute__((aligned(16))) Tab { /* shared by the streams of a workgroup */
    float ones[kSpecPad];                 /* 1.0f: the "window" of the plain in-order sums that ride along with DoGamma */
    float gammaC[SEA16_NGAM][kSpecPad];   /* [c][i], rows 132 floats apart: the 25 lanes' float4 reads fall on different banks */
};

__device__ __forceinline__ int window_base(int tick) { return ((tick - 3) & (kSlots - 1)) * kHop; }

/* lane i + 64 k (< 160) of the frame of tick t into its slot (and the mirror) */
__device__ __forceinline__ void slot_store3(float *circ, int tick, int lane, const float (&v)[3])
{
    const int slot = tick & (kSlots - 1);
    float *p = circ + slot * kHop;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int i = lane + kLanes * k;
        if (i < kHop) {
            p[i] = v[k];
            if (slot < 3) p[kCirc + i] = v[k];
        }
    }
}

__device__ __forceinline__ void block_sync()
{ /* every wave executes the same NUMBER of barriers per beat; LDS-only fences: global traffic stays in flight */
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

/* ---------------------------------------------------------------------------------------------------------------
 * F: two transforms side by side in one instruction stream (A in work[0], B in work[1]: the same per-lane tables, the
 * second work area 2 KB behind the first)
 * ------------------------------------------------------------------------------------------------------------- */
struct FftRegs16 {
    unsigned headFlags, head[4];
    unsigned kind[SEA16_PIPE_LEVELS], addr[SEA16_PIPE_LEVELS][4];
    float tw[SEA16_PIPE_LEVELS][4];
    unsigned psd[2][2], nyq;
    float win8[8];
    unsigned srcOff;   /* byte offset of element bitrev6(lane) of the window: 4 * (80 + bitrev6(lane)) */
    bool tailZero;     /* element bitrev6(lane) + 448 lies beyond the 480-sample window: a literal zero */
};
__device__ __forceinline__ void load_fft16(FftRegs16 &R, const sea_ns16k_pipe_tables *p, int lane)
{
    R.headFlags = p->head8Flags[lane];
#pragma unroll
    for (int q = 0; q < 4; ++q) R.head[q] = p->head8Addr[q][lane];
#pragma unroll
    for (int s = 0; s < SEA16_PIPE_LEVELS; ++s) {
        R.kind[s] = p->kind[s][lane];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            R.addr[s][q] = p->addr[s][q][lane];
            R.tw[s][q] = p->tw[s][q][lane];
        }
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        R.psd[q][0] = p->psd[q][0][lane];
        R.psd[q][1] = p->psd[q][1][lane];
    }
    R.nyq = p->nyq;
#pragma unroll
    for (int j = 0; j < 8; ++j) R.win8[j] = p->win8[j][lane];
    const unsigned rl = __brev((unsigned)lane) >> 26;
    R.srcOff = 4u * ((unsigned)SEA16_AWIN + rl);
    R.tailZero = rl + 448u >= (unsigned)SEA16_WIN;
}

/* the lane's eight windowed elements bitrev6(lane) + 64 bitrev3(j) of the 480-sample window at buf[80..559] (DoSigWindowing,
 * NoiseSup.cpp:209-222; zero padding to 512: literal zeros).  An inactive side is fed zeros. */
__device__ __forceinline__ void window8(const float *buf, bool act, const FftRegs16 &R, float (&e)[8])
{
    constexpr int kRev3[8] = {0, 4, 2, 6, 1, 5, 3, 7};
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = fft_at(buf, R.srcOff + 4u * 64u * (unsigned)kRev3[j]); /* <= buf[80 + 511]: inside the 640-run */
    asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float p = v[j] * R.win8[j];
        const bool zero = !act || (kRev3[j] == 7 && R.tailZero);
        e[j] = zero ? 0.0f : p;
    }
}

/* length-2 butterflies (rfft.cpp:83-97), the n2 = 4 level (plain butterflies only) and the n2 = 8 level (plain + pi/4) on the
 * lane's eight places, gated per block by the is/id schedule; results to their swizzled places */
__device__ __forceinline__ void head8(float (&e)[8], float *work, const FftRegs16 &R)
{
    const unsigned fl = R.headFlags;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const float a = e[2 * p], b = e[2 * p + 1], sum = a + b, dif = a - b;
        const bool f = (fl & (1u << p)) != 0;
        e[2 * p] = f ? sum : a;
        e[2 * p + 1] = f ? dif : b;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const float g0 = e[4 * h], g2 = e[4 * h + 2], g3 = e[4 * h + 3];
        const float t1 = g3 + g2;
        const float n3 = g3 - g2, n2 = g0 - t1, n0 = g0 + t1;
        const bool f = (fl & (16u << h)) != 0;
        e[4 * h + 3] = f ? n3 : g3;
        e[4 * h + 2] = f ? n2 : g2;
        e[4 * h] = f ? n0 : g0;
    }
    {
        const float x1 = e[0], x2 = e[2], x3 = e[4], x4 = e[6], x5 = e[1], x6 = e[3], x7 = e[5], x8 = e[7];
        const float t1 = x4 + x3;
        const float o4 = x4 - x3, o3 = x1 - t1, o1 = x1 + t1;
        const float u1 = (float)((double)(x7 + x8) * 0.70710678118654752440); /* == / M_SQRT2 for every float (sea_selftest_pi4) */
        const float u2 = (float)((double)(x7 - x8) * 0.70710678118654752440);
        const float o8 = x6 - u1, o7 = -x6 - u1, o6 = x5 - u2, o5 = x5 + u2;
        const bool f = (fl & 64u) != 0;
        e[0] = f ? o1 : x1;
        e[2] = x2;
        e[4] = f ? o3 : x3;
        e[6] = f ? o4 : x4;
        e[1] = f ? o5 : x5;
        e[3] = f ? o6 : x6;
        e[5] = f ? o7 : x7;
        e[7] = f ? o8 : x8;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        fft_at(work, R.head[q] & 0xffffu) = e[2 * q];
        fft_at(work, R.head[q] >> 16) = e[2 * q + 1];
    }
}

/* one work item: PAIR = plain butterfly on x[0..3] (rfft.cpp:113-120) + pi/4 butterfly on x[4..7] (:122-133), or the
 * twiddled butterfly (:139-176); both evaluated, the lane's kind selects (a wave executes both sides of a divergent branch
 * anyway, and without the branch the chains of the two transforms interleave) */
__device__ __forceinline__ void butterfly16(unsigned kind, const float (&tw)[4], const float (&x)[8], float (&o)[8])
{
    const bool isTw = kind == SEA_BF_TWIDDLE;
    float q[8], p[8];
    {
        const float cc1 = tw[0], ss1 = tw[1], cc3 = tw[2], ss3 = tw[3];
        float t1 = x[2] * cc1 + x[6] * ss1;
        float t2 = x[6] * cc1 - x[2] * ss1;
        float t3 = x[3] * cc3 + x[7] * ss3;
        float t4 = x[7] * cc3 - x[3] * ss3;
        const float t5 = t1 + t3, t6 = t2 + t4;
        t3 = t1 - t3;
        t4 = t2 - t4;
        q[2] = t6 - x[5];
        q[7] = x[5] + t6;
        q[6] = -x[1] - t3;
        q[3] = x[1] - t3;
        q[5] = x[0] - t5;
        q[0] = x[0] + t5;
        q[4] = x[4] - t4;
        q[1] = x[4] + t4;
    }
    {
        const float t1 = x[3] + x[2];
        p[3] = x[3] - x[2];
        p[2] = x[0] - t1;
        p[0] = x[0] + t1;
        p[1] = x[1];
        const float u1 = (float)((double)(x[6] + x[7]) * 0.70710678118654752440);
        const float u2 = (float)((double)(x[6] - x[7]) * 0.70710678118654752440);
        p[7] = x[5] - u1;
        p[6] = -x[5] - u1;
        p[5] = x[4] - u2;
        p[4] = x[4] + u2;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = isTw ? q[k] : p[k];
}

template <int S>
__device__ __forceinline__ void level16(float *workA, float *workB, const FftRegs16 &R)
{
    float xa[8], xb[8], oa[8], ob[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        xa[2 * q] = fft_at(workA, R.addr[S][q] & 0xffffu);
        xa[2 * q + 1] = fft_at(workA, R.addr[S][q] >> 16);
        xb[2 * q] = fft_at(workB, R.addr[S][q] & 0xffffu);
        xb[2 * q + 1] = fft_at(workB, R.addr[S][q] >> 16);
    }
    butterfly16(R.kind[S], R.tw[S], xa, oa);
    butterfly16(R.kind[S], R.tw[S], xb, ob);
    if (R.kind[S] != SEA_BF_NONE) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            fft_at(workA, R.addr[S][q] & 0xffffu) = oa[2 * q];
            fft_at(workA, R.addr[S][q] >> 16) = oa[2 * q + 1];
            fft_at(workB, R.addr[S][q] & 0xffffu) = ob[2 * q];
            fft_at(workB, R.addr[S][q] >> 16) = ob[2 * q + 1];
        }
    }
    wave_sync();
}

/* FFTtoPSD (NoiseSup.cpp:240-261) of one transformed frame: values lane, lane + 64 and (lane 0) 128 */
__device__ __forceinline__ void psd16(const float *work, const FftRegs16 &R, float *psd, int lane, bool act)
{
    float re0[2], re1[2], im0[2], im1[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        re0[q] = fft_at(work, R.psd[q][0] & 0xffffu);
        re1[q] = fft_at(work, R.psd[q][0] >> 16);
        im0[q] = fft_at(work, R.psd[q][1] & 0xffffu);
        im1[q] = fft_at(work, R.psd[q][1] >> 16);
    }
    const float ny = fft_at(work, R.nyq);
    asm volatile("" : "+v"(re0[0]), "+v"(re1[0]), "+v"(im0[0]), "+v"(im1[0]), "+v"(re0[1]), "+v"(re1[1]), "+v"(im0[1]), "+v"(im1[1]));
    if (act) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float p0 = (q == 0 && lane == 0) ? re0[q] * re0[q] : (re0[q] * re0[q] + im0[q] * im0[q]);
            const float p1 = re1[q] * re1[q] + im1[q] * im1[q];
            psd[lane + kLanes * q] = (float)((double)(p0 + p1) / 2.0);
        }
        if (lane == 0) psd[kSpec - 1] = ny * ny;
    }
}

/* ---------------------------------------------------------------------------------------------------------------
 * B0 / B1: everything recursive of one stage (NoiseSup.cpp:1207-1366 after the transform)
 * ------------------------------------------------------------------------------------------------------------- */
/* DoGamma's in-order sum of one window (MelProc.cpp:119-135): sum_i src[i] * coef[i], i = 0..127, products and additions in
 * that order, float; last: one more plain term src[128] (the 129-value sums that ride along) */
__device__ __forceinline__ float gamma_chain(const float *src, const float *coef, bool last)
{
    float sum = 0.0f;
#pragma unroll 8
    for (int q = 0; q < SEA16_GLEN / 4; ++q) {
        const float4 w = *reinterpret_cast<const float4 *>(src + 4 * q);
        const float4 c = *reinterpret_cast<const float4 *>(coef + 4 * q);
        sum += w.x * c.x;
        sum += w.y * c.y;
        sum += w.z * c.z;
        sum += w.w * c.w;
    }
    if (last) sum += src[SEA16_GLEN];
    return sum;
}
/* one row of DoGammaIDCT (MelProc.cpp:556-576): sum_f gam[f] * basis[f][row], in order; idct = this lane's row of the basis */
__device__ __forceinline__ float idct_row(const float *gam, const float (&idct)[SEA16_NGAM])
{
    float h = 0.0f;
#pragma unroll
    for (int f = 0; f < SEA16_NGAM; ++f) h += gam[f] * idct[f];
    return h;
}

template <int ST>
__device__ __forceinline__ void back16(StreamLds &L, const Tab &T, const float *psd, const float *buf, NsRegs &s, NsFd &fd, float eps,
                                       const float (&idct)[SEA16_NGAM], float irWin, int lane, float frameSum, float *dst,
                                       int &fdBits, float &gainOut, float &denTotal)
{
    /* PSDMean (:280-294) and FilterCalc (:440-553) per spectral value: b = lane, lane + 64, 128 */
    {
        int nb = s.nbFrame[ST];
        if (nb < 2147483647) nb++;
        s.nbFrame[ST] = nb;
    }
    /* _VAD_ (:350-421).  SEA16P_LOG_IN_S (round 4): the helper wave S, which idles ~3500 clk of a beat, took the log-energy of the
     * sum it left at intake; this wave -- the longest role -- only updates the VAD with it */
    if (ST == 0) vad_update(s, SEA16P_LOG_IN_S ? frameSum : vad_frame_energy(frameSum));
    const int nb16 = (int)(short)s.nbFrame[ST];
    float nSigv[3], Pv[3], noisev[3], denv[3], Wv[3];
    bool inDomain = true;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int b = lane + kLanes * k;
        const int bb = (b < kSpec) ? b : (kSpec - 1);
        const float nSig = psd[bb];
        nSigv[k] = nSig;
        Pv[k] = (L.bins[4 + ST][bb] + nSig) / 2.0f;
        noisev[k] = L.bins[ST][bb];
        denv[k] = L.bins[2 + ST][bb];
        inDomain &= ns_psd_in_domain(nSig) && noisev[k] <= 0x1p28f && noisev[k] >= 0x1p-15f;
    }
    /* the guarded fast-division domain of ns_core.h (ns_back), established per frame, wave-uniformly */
    const bool domainNow = __ballot(!inDomain) == 0ull;
    const bool fast = SEA_NS_FAST_DIV && domainNow && (s.psdOk[ST] != 0);
    s.psdOk[ST] = domainNow ? 1 : 0;
#pragma unroll
    for (int k = 0; k < 3; ++k)
        Wv[k] = fast ? filter_bin<ST, true>(Pv[k], nSigv[k], noisev[k], denv[k], nb16, s.flagVAD, eps)
                     : filter_bin<ST, false>(Pv[k], nSigv[k], noisev[k], denv[k], nb16, s.flagVAD, eps);
    wave_sync(); /* every lane has read value 128 before lane 0 overwrites it */
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int b = lane + kLanes * k;
        if (b < kSpec) {
            L.bins[4 + ST][b] = nSigv[k];
            L.bins[ST][b] = noisev[k];
            L.bins[2 + ST][b] = denv[k];
            L.W[ST][b] = Wv[k];
        }
    }
    wave_sync();
    /* DoGamma (MelProc.cpp:119-135): window c = lane < 25 over gains 0..127, in order; three plain in-order sums ride in
     * lanes 25..27 as "windows" of their own (x * 1.0f == x):  25 sum W (SpeechQVar :866-870), 26 sum W^2, 27 the sum of
     * denSigSE1 (first stage) / noiseSE2 (second), 129 terms (DoGainFact_IBM :648, :653) */
    float sum;
    {
        const float *src = (lane == 27) ? L.bins[ST == 0 ? 2 : 1] : L.W[ST];
        const float *coef = (lane < SEA16_NGAM) ? T.gammaC[lane] : ((lane == 26) ? L.W[ST] : T.ones);
        sum = gamma_chain(src, coef, lane == 27);
        if (lane < SEA16_NGAM) L.gam[ST][lane] = sum;
    }
    wave_sync();
    const float total = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sum), 27));
    if (ST == 0) { /* SpeechQVar on the first 128 gains (:852-893), then :1301-1311, :1362-1365 */
        const float mean = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sum), 25));
        const float var = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sum), 26));
        fdBits = fd_var_sums<SEA16_NFFT / 4>(fd, mean, var, nb16);
        fdBits |= fd_spec_mel(fd, L.gam[ST], nb16) << 1;
        fdBits |= (s.nbSpeech > 4) ? 8 : 0;
        denTotal = total;
    }
    /* DoGainFact_IBM (:634-698) */
    float g = L.gam[ST][(lane < SEA16_NGAM) ? lane : 0];
    if (ST == 1) {
        gain_fact_update(s, total); /* the caller has loaded s.denEn0..2 */
        g = (float)((double)(s.alfaGF * g) + (1.0 - (double)s.alfaGF) * 1.0);
        gainOut = g;
        wave_sync();
        if (lane < SEA16_NGAM) L.gam[ST][lane] = g;
        wave_sync();
    }
    /* DoGammaIDCT (MelProc.cpp:556-576), taps t = lane = 0..8, + DoFilterWindowing (:716-725) */
    float *fir = (ST == 0) ? L.fir[0] : dst;
    {
        const float h = idct_row(L.gam[ST], idct);
        const float tap = h * irWin;
        if (lane <= 8) {
            fir[8 + lane] = tap;
            fir[8 - lane] = tap;
        }
    }
    wave_sync();
    if (ST == 0) {
        /* ApplyWF (cur, prv, filterIR, out, 160, 8) (:317-331) on buf[160..319]: one sum over j = -8..8 in that order */
        float acc[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int j = -8; j <= 8; ++j) {
            const float fj = fir[j + 8];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int i = lane + kLanes * k;
                acc[k] += fj * buf[kHop + ((i < kHop) ? i : 0) - j];
            }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) dst[k] = acc[k]; /* the caller stores them into the stage-1 ring */
    }
}

/* S's one dependent instruction stream per beat: acc = fma(m, acc, x[n]), n = 0..159, per-lane m / source / start --
 *   lanes  0..15  the frame gate's FrameCheck = 0 + sum sq[n]        (m = 1: RN(1 * acc + x) == RN(acc + x))
 *   lanes 16..31  the VAD's frameEn = 64 + sum sq[n]
 *   lanes 32..63  the DC-offset recurrence y = 1023/1024 y + dif[n] (float-FMA form of ns_core.h's dc_filter)
 * The serial pass keeps every fifth value of the DC chain (lane 32 + j captures y[5j - 1] under a one-bit scalar mask);
 * lanes 32..63 then recompute their five outputs each in parallel and check the FMA form's exactness condition on the
 * registers they hold.  Ends with wave_sync(). */
__device__ __forceinline__ void chains160(const float *sq, const float *dif, float *out, float &gate, float &vad, float &y, int lane,
                                          bool &unsafe)
{
    const int g = lane >> 4;
    const float *src = (g < 2) ? sq : dif;
    const float m = (g >= 2) ? 0.9990234375f : 1.0f;
    float acc = (g == 0) ? 0.0f : ((g == 1) ? 64.0f : y);
    constexpr int kChunks = 8, kQ = kHop / 4 / kChunks, kSeg = 5; /* 8 chunks of 5 quads; 32 segments of 5 steps */
    float4 x[2][kQ];
    auto request = [&](int c, float4(&dstq)[kQ]) {
#pragma unroll
        for (int k = 0; k < kQ; ++k) dstq[k] = *reinterpret_cast<const float4 *>(src + 4 * (c * kQ + k));
    };
    const int seg = (lane - 32) & 31;
    float d5[kSeg];
#pragma unroll
    for (int k = 0; k < kSeg; ++k) d5[k] = dif[kSeg * seg + k];
    float cap = y;
    auto step = [&](float xv, int n) {
        float next;
        asm volatile("v_fma_f32 %0, %2, %1, %3" : "=&v"(next) : "v"(m), "v"(acc), "v"(xv));
        if (n > 0 && n % kSeg == 0) {
            const unsigned long long bit = 1ull << (32 + n / kSeg);
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
            if (lane >= 32) out[kSeg * seg + k] = v;
        }
        unsafe = __ballot(bad && lane >= 32) != 0ull;
    }
    gate = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc), 0));
    vad = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc), 16));
    y = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc), 32));
    wave_sync();
}

constexpr int kBlobBins = 2 * SEA16_BUF, kBlobScal16 = kBlobBins + 6 * kSpecPad;
static_assert(kBlobScal16 + 32 == kNs16StateFloats, "state blob layout");

} // namespace p16

/* Device-side check of the pieces the reference's own rfft.cpp + MelProc.cpp pin (tests/golden/aurora_golden.npz; VERDICT r03
 * "What's missing" 4): ONE wave runs, with the very functions the pipelined kernel's role waves call,
 *   - rfft (x, 512, 8) on nfft frames of 512 floats (the transform wave's register-resident start + five LDS levels, both work
 *     areas fed the same frame; results from each in the reference's order),
 *   - DoGamma on ngain vectors of 129 gains (25 window sums), and rows 0..8 of DoGammaIDCT of those sums. */
__global__ void __launch_bounds__(64) ns16k_selftest_kernel(const sea_ns16k_tables *t, const float *frames, int nfft, float *outA, float *outB,
                                                            const float *gains, int ngain, float *gamma25, float *idct9)
{
    using namespace p16;
    __shared__ Tab T;
    __shared__ __attribute__((aligned(16))) float work[2][SEA16_NFFT];
    __shared__ __attribute__((aligned(16))) float W[kSpecPad], gam[32];
    const int lane = threadIdx.x;
    for (int i = lane; i < SEA16_GLEN * SEA16_NGAM; i += kLanes) T.gammaC[i % SEA16_NGAM][i / SEA16_NGAM] = (&t->gammaT[0][0])[i];
    for (int i = lane; i < kSpecPad; i += kLanes) T.ones[i] = 1.0f;
    FftRegs16 R;
    load_fft16(R, &t->pipe, lane);
    float idct[SEA16_NGAM];
#pragma unroll
    for (int f = 0; f < SEA16_NGAM; ++f) idct[f] = t->idctT[f][lane & 15];
    wave_sync();
    constexpr unsigned char kSwz[16] = SEA16_SWZ;
    for (int n = 0; n < nfft; ++n) {
        float eA[8], eB[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) eA[j] = eB[j] = frames[(size_t)n * SEA16_NFFT + t->pipe.src8[j][lane]];
        head8(eA, work[0], R);
        head8(eB, work[1], R);
        wave_sync();
        level16<0>(work[0], work[1], R);
        level16<1>(work[0], work[1], R);
        level16<2>(work[0], work[1], R);
        level16<3>(work[0], work[1], R);
        level16<4>(work[0], work[1], R);
        for (int i = lane; i < SEA16_NFFT; i += kLanes) {
            const unsigned w = (unsigned)i ^ kSwz[(i >> 5) & 15];
            outA[(size_t)n * SEA16_NFFT + i] = work[0][w];
            outB[(size_t)n * SEA16_NFFT + i] = work[1][w];
        }
        wave_sync();
    }
    for (int n = 0; n < ngain; ++n) {
        for (int i = lane; i < kSpecPad; i += kLanes) W[i] = (i < kSpec) ? gains[(size_t)n * kSpec + i] : 0.0f;
        wave_sync();
        const float sum = gamma_chain(W, (lane < SEA16_NGAM) ? T.gammaC[lane] : T.ones, false);
        if (lane < SEA16_NGAM) {
            gam[lane] = sum;
            gamma25[n * SEA16_NGAM + lane] = sum;
        }
        wave_sync();
        const float h = idct_row(gam, idct);
        if (lane <= 8) idct9[n * 9 + lane] = h;
        wave_sync();
    }
}

__global__ void __launch_bounds__(256 * p16::kStreams, 4) ns16k_pipe_kernel(Ns16StreamArgs a)
{
    using namespace p16;
    __shared__ Tab T;
    __shared__ StreamLds LS[kStreams];
    const int lane = threadIdx.x & (kLanes - 1), wave = threadIdx.x >> 6;
    const int sidx = __builtin_amdgcn_readfirstlane(wave >> 2), role = __builtin_amdgcn_readfirstlane(wave & 3);
    const int tid = threadIdx.x & 255; /* within the stream's four waves */
    const sea_ns16k_tables *t = a.tables;
    for (int i = threadIdx.x; i < SEA16_GLEN * SEA16_NGAM; i += 256 * kStreams) T.gammaC[i % SEA16_NGAM][i / SEA16_NGAM] = (&t->gammaT[0][0])[i];
    for (int i = threadIdx.x; i < kSpecPad; i += 256 * kStreams) T.ones[i] = 1.0f;
    const int b0 = blockIdx.x * kStreams + sidx;
    const bool live = b0 < a.n_streams;
    const int b = live ? b0 : a.n_streams - 1; /* a padding stream repeats the last one's reads and stores nothing */
    StreamLds &L = LS[sidx];
    float *blob = a.state + (size_t)b * kNs16StateFloats;
    const float *bq = blob + kBlobScal16;
    const int *bqi = reinterpret_cast<const int *>(bq + 16);
    const int nframes = a.nframes;
    const int tick0 = a.reset ? 0 : bqi[5]; /* nbFramesInFirstStage: frames that passed the gate so far */

    /* ---- state in: the reference's linear 640-sample stage buffers hold the frames of ticks T-2, T-1, T at [0..479] ---- */
    for (int i = tid; i < 2 * (kCirc + kMirror); i += 256) (&L.circ[0][0])[i] = 0.0f;
    if (tid < kSlots) {
        L.frameEn[tid] = 0.0f;
        L.denSum[tid] = 0.0f;
    }
    if (tid < 2) {
        L.rin[tid].valid = 0;
        L.r01[tid].valid = 0;
        L.r23[tid].valid = 0;
        L.r34[tid].produced = 0;
    }
    block_sync();
    if (a.reset) {
        for (int i = tid; i < 6 * kSpecPad; i += 256) (&L.bins[0][0])[i] = (i < 2 * kSpecPad) ? t->eps : 0.0f;
    } else {
        for (int i = tid; i < 6 * kSpecPad; i += 256) (&L.bins[0][0])[i] = blob[kBlobBins + i];
        for (int i = tid; i < 2 * 3 * kHop; i += 256) {
            const int st = i / (3 * kHop), x = i - st * 3 * kHop, k = x / kHop, o = x - k * kHop; /* frame of tick T - 2 + k */
            const int slot = (tick0 - 2 + k) & (kSlots - 1);
            const float v = blob[st * SEA16_BUF + x];
            L.circ[st][slot * kHop + o] = v;
            if (slot < 3) L.circ[st][kCirc + slot * kHop + o] = v;
        }
        if (tid < 3) L.denSum[(tick0 - 2 + tid) & (kSlots - 1)] = bq[2 + tid]; /* denEn0..2 = the sums of ticks T-2, T-1, T */
    }
    block_sync();

    T16_DECL;
    const long long niter = (long long)nframes + kLag;
    if (role == 0) {
        /* ---------------- F ---------------- */
        FftRegs16 R;
        load_fft16(R, &t->pipe, lane);
        int v1 = 0, t1 = 0, v2 = 0, t2 = 0; /* (valid, tick) of the frames taken in one and two beats before the current one */
        for (long long i = 0; i < niter; ++i) {
            T16_BEGIN;
            const long long fA = i - 1, fB = i - 3;
            int vA = 0, tA = 0;
            if (fA >= 0 && fA < nframes) {
                vA = L.rin[fA & 1].valid;
                tA = L.rin[fA & 1].tick;
            }
            const int vB = v2, tB = t2; /* frame fB = fA - 2 */
            const bool actA = vA && tA >= 3;              /* nbFramesInFirstStage - nbFramesInSecondStage > 2 (:1212) */
            const bool actB = fB >= 0 && vB && tB >= 5;   /* nbFramesInSecondStage - nbFramesOutSecondStage > 2 (:1230) */
            RecPsd &rA = L.r01[fA & 1], &rB = L.r23[fB & 1];
            if (fA >= 0 && fA < nframes && lane == 0) {
                rA.valid = vA;
                rA.tick = tA;
            }
            if (fB >= 0 && fB < nframes && lane == 0) {
                rB.valid = vB;
                rB.tick = tB;
            }
            if (actA || actB) {
                float eA[8], eB[8];
                window8(L.circ[0] + window_base(tA), actA, R, eA);
                window8(L.circ[1] + window_base(tB), actB, R, eB);
                head8(eA, L.work[0], R);
                head8(eB, L.work[1], R);
                wave_sync();
                level16<0>(L.work[0], L.work[1], R);
                level16<1>(L.work[0], L.work[1], R);
                level16<2>(L.work[0], L.work[1], R);
                level16<3>(L.work[0], L.work[1], R);
                level16<4>(L.work[0], L.work[1], R);
                psd16(L.work[0], R, rA.psd, lane, actA);
                psd16(L.work[1], R, rB.psd, lane, actB);
                wave_sync();
            }
            v2 = v1, t2 = t1, v1 = vA, t1 = tA;
            T16_MID;
            block_sync();
            T16_END;
        }
        T16_FLUSH(0);
    } else if (role == 1) {
        /* ---------------- B0 ---------------- */
        NsRegs s;
        NsFd fd;
        regs_init(s, t->eps);
        fd_init(fd);
        if (!a.reset) {
            s.meanEn = bq[7];
            fd.melMean = bq[8]; fd.varMean = bq[9]; fd.accTest = bq[10]; fd.specMean = bq[11];
            fd.mel0 = bq[12]; fd.specValues = bq[13]; fd.speechInVADQ = bq[14];
            s.nbFrame[0] = bqi[0]; s.flagVAD = bqi[2]; s.hangOver = bqi[3]; s.nbSpeech = bqi[4];
            s.psdOk[0] = bqi[10] & 1;
        }
        const float eps = t->eps;
        float idct[SEA16_NGAM];
#pragma unroll
        for (int f = 0; f < SEA16_NGAM; ++f) idct[f] = t->idctT[f][lane & 15];
        const float irWin = t->irWin[lane & 15];
        for (long long i = 0; i < niter; ++i) {
            T16_BEGIN;
            const long long f = i - 2;
            if (f >= 0 && f < nframes) {
                const RecPsd &r = L.r01[f & 1];
                const int valid = r.valid, tk = r.tick;
                int fdBits = 0, counter = 0;
                if (valid && tk >= 3) {
                    float y3[3], unusedGain, denTotal = 0.0f;
                    back16<0>(L, T, r.psd, L.circ[0] + window_base(tk), s, fd, eps, idct, irWin, lane,
                              L.frameEn[(tk - 2) & (kSlots - 1)], y3, fdBits, unusedGain, denTotal);
                    slot_store3(L.circ[1], tk, lane, y3);
                    if (lane == 0) L.denSum[tk & (kSlots - 1)] = denTotal;
                    counter = s.nbFrame[0];
                }
                if (live && lane == 0) {
                    const size_t rec = (size_t)b * nframes + f;
                    if (a.flags) a.flags[rec] = (unsigned char)fdBits;
                    if (a.frame_counter) a.frame_counter[rec] = counter;
                }
            }
            T16_MID;
            block_sync();
            T16_END;
        }
        if (live && lane == 0) {
            float *q = blob + kBlobScal16;
            int *qi = reinterpret_cast<int *>(q + 16);
            q[7] = s.meanEn;
            q[8] = fd.melMean; q[9] = fd.varMean; q[10] = fd.accTest; q[11] = fd.specMean;
            q[12] = fd.mel0; q[13] = fd.specValues; q[14] = fd.speechInVADQ;
            qi[0] = s.nbFrame[0]; qi[2] = s.flagVAD; qi[3] = s.hangOver; qi[4] = s.nbSpeech;
        }
        if (lane == 0) L.psdOkEnd[0] = s.psdOk[0];
        T16_FLUSH(1);
    } else if (role == 2) {
        /* ---------------- B1 ---------------- */
        NsRegs s;
        NsFd fdUnused;
        regs_init(s, t->eps);
        fd_init(fdUnused);
        if (!a.reset) {
            s.lowSNRtrack = bq[5]; s.alfaGF = bq[6];
            s.nbFrame[1] = bqi[1];
            s.psdOk[1] = (bqi[10] >> 1) & 1;
        }
        const float eps = t->eps;
        float idct[SEA16_NGAM];
#pragma unroll
        for (int f = 0; f < SEA16_NGAM; ++f) idct[f] = t->idctT[f][lane & 15];
        const float irWin = t->irWin[lane & 15];
        for (long long i = 0; i < niter; ++i) {
            T16_BEGIN;
            const long long f = i - 4;
            if (f >= 0 && f < nframes) {
                const RecPsd &r = L.r23[f & 1];
                RecTaps &o = L.r34[f & 1];
                const int valid = r.valid, tk = r.tick;
                int produced = 0;
                if (valid && tk >= 5) {
                    int unusedBits = 0;
                    float gain = 0.0f, unusedDen;
                    s.denEn0 = L.denSum[(tk - 2) & (kSlots - 1)];
                    s.denEn1 = L.denSum[(tk - 1) & (kSlots - 1)];
                    s.denEn2 = L.denSum[tk & (kSlots - 1)];
                    back16<1>(L, T, r.psd, L.circ[1] + window_base(tk), s, fdUnused, eps, idct, irWin, lane, 0.0f, o.fir,
                              unusedBits, gain, unusedDen);
                    produced = 1;
                    if (live && a.wiener && lane < SEA16_NGAM)
                        a.wiener[((size_t)b * nframes + f) * SEA16_NGAM + lane] = gain; /* the line func_Wiener prints (:1319-1328) */
                }
                if (lane == 0) {
                    o.produced = produced;
                    o.tick = tk;
                }
            }
            T16_MID;
            block_sync();
            T16_END;
        }
        if (live && lane == 0) {
            float *q = blob + kBlobScal16;
            int *qi = reinterpret_cast<int *>(q + 16);
            q[5] = s.lowSNRtrack; q[6] = s.alfaGF;
            qi[1] = s.nbFrame[1];
        }
        if (lane == 0) L.psdOkEnd[1] = s.psdOk[1];
        T16_FLUSH(2);
    } else {
        /* ---------------- S ---------------- */
        const float *in = a.in + (size_t)b * nframes * kHop;
        float *out = a.out + (size_t)b * nframes * kHop;
        float dcX = a.reset ? 0.0f : bq[0], dcY = a.reset ? 0.0f : bq[1];
        int tick = tick0;
        float xn[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int i = lane + kLanes * k;
            xn[k] = (i < kHop && nframes > 0) ? in[i] : 0.0f;
        }
        if (!a.reset) {
            /* the VAD's frame sums of the frames of ticks T-1 and T (needed by B0 at ticks T+1, T+2) are not part of the blob:
             * recomputed from the ring, in order, 64 + sum x^2, once per launch */
#pragma unroll 1
            for (int k = 0; k < 2; ++k) {
                const int tk = tick0 - 1 + k;
                const float *fr = L.circ[0] + (tk & (kSlots - 1)) * kHop;
                float acc = 64.0f;
                for (int n = 0; n < kHop; ++n) acc += fr[n] * fr[n];
                const float accEn = SEA16P_LOG_IN_S ? vad_frame_energy(acc) : acc;
                if (lane == 0) L.frameEn[tk & (kSlots - 1)] = accEn;
            }
            wave_sync();
        }
        for (long long i = 0; i < niter; ++i) {
            T16_BEGIN;
            const long long fi = i, fo = i - kLag;
            const bool haveIn = fi < nframes, haveOut = fo >= 0;
            float x[3] = {xn[0], xn[1], xn[2]};
            if (haveIn) { /* the next frame's samples are requested now, first touched one beat later */
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int n = lane + kLanes * k;
                    xn[k] = (n < kHop && fi + 1 < nframes) ? in[(fi + 1) * kHop + n] : 0.0f;
                }
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int n = lane + kLanes * k;
                    if (n < kHop) L.ssq[n] = x[k] * x[k];
                }
            }
            bool produced = false;
            if (haveOut) {
                const RecTaps &r = L.r34[fo & 1];
                produced = r.produced != 0;
                if (produced) {
                    /* ApplyWF of the second stage (:317-331) on buf[160..319] of tick r.tick: lane l < 54 takes outputs 3l .. 3l+2
                     * (stride 3 over the banks: conflict-free), then the DC filter's differences d[n] = y[n] - y[n-1]
                     * (:168-184; y[-1] = the previous frame's last filter output) across lanes with a one-lane DPP shift */
                    const float *buf = L.circ[1] + window_base(r.tick) + kHop - 8;
                    float c[SEA_NTAP];
#pragma unroll
                    for (int k4 = 0; k4 < 16; k4 += 4) {
                        const float4 v = *reinterpret_cast<const float4 *>(&r.fir[k4]);
                        c[k4] = v.x, c[k4 + 1] = v.y, c[k4 + 2] = v.z, c[k4 + 3] = v.w;
                    }
                    c[16] = r.fir[16];
                    const int l3 = 3 * ((lane < 54) ? lane : 53);
                    float xw[19]; /* xw[m] = buf[160 - 8 + 3l + m]; output n = 3l + k: sum_j fir[j + 8] * buf[160 + n - j] */
#pragma unroll
                    for (int m = 0; m < 19; ++m) xw[m] = buf[l3 + m];
                    float y0 = 0.0f, y1 = 0.0f, y2 = 0.0f;
#pragma unroll
                    for (int k = 0; k < SEA_NTAP; ++k) { /* k = j + 8; buf index 160 + n - j = (152 + n) + (16 - k) */
                        y0 += c[k] * xw[16 - k];
                        y1 += c[k] * xw[17 - k];
                        y2 += c[k] * xw[18 - k];
                    }
                    const float below = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(dcX), __float_as_int(y2), 0x138 /* wave_shr:1 */, 0xf, 0xf, false));
                    const float d0 = y0 - below, d1 = y1 - y0, d2 = y2 - y1;
                    dcX = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(y0), 53)); /* y[159] */
                    if (lane < 54) {
                        L.sdif[l3] = d0;
                        if (lane < 53) {
                            L.sdif[l3 + 1] = d1;
                            L.sdif[l3 + 2] = d2;
                        }
                    }
                }
            }
            if (haveIn || produced) {
                wave_sync();
                float gate, vadSum, y = dcY;
                bool unsafe;
                chains160(L.ssq, L.sdif, L.sout, gate, vadSum, y, lane, unsafe);
                if (produced) {
                    if (unsafe) { /* exact path: double multiply-add, rounded to float per sample (never yet observed) */
                        y = dcY;
                        for (int n = 0; n < kHop; ++n) {
                            y = (float)__fma_rn(0.9990234375, (double)y, (double)L.sdif[n]);
                            L.sout[n] = y;
                        }
                        wave_sync();
                    }
                    dcY = y;
                }
                if (haveIn) {
                    int valid = 0;
                    if (!(gate < 1.0f)) { /* (int)FrameCheck != 0; NaN and overflow convert to INT_MIN on the reference's x86 */
                        valid = 1;
                        tick++;
                        slot_store3(L.circ[0], tick, lane, x);
                        const float vadEn = SEA16P_LOG_IN_S ? vad_frame_energy(vadSum) : vadSum;
                        if (lane == 0) L.frameEn[tick & (kSlots - 1)] = vadEn;
                    }
                    if (lane == 0) {
                        L.rin[fi & 1].valid = valid;
                        L.rin[fi & 1].tick = tick;
                    }
                }
            }
            if (haveOut && live) {
                if (produced) {
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const int n = lane + kLanes * k;
                        if (n < kHop) out[fo * kHop + n] = L.sout[n];
                    }
                }
                if (lane == 0) a.produced[(size_t)b * nframes + fo] = produced ? 1 : 0;
            }
            wave_sync();
            T16_MID;
            block_sync();
            T16_END;
        }
        if (live && lane == 0) {
            float *q = blob + kBlobScal16;
            int *qi = reinterpret_cast<int *>(q + 16);
            q[0] = dcX; q[1] = dcY;
            qi[5] = tick;
            qi[6] = tick > 2 ? tick - 2 : 0; /* nbFramesInSecondStage */
            qi[7] = tick > 4 ? tick - 4 : 0; /* nbFramesOutSecondStage */
        }
        if (lane == 0) L.tickEnd = tick;
        T16_FLUSH(3);
    }
    /* ---- state out: every wave has passed the last beat barrier; the linear stage buffers of the reference after its
     *      slide (:1372-1390) hold the frames of ticks T-2, T-1, T at [0..479] and the frame of tick T once more behind ---- */
    block_sync();
    if (live) {
        const int T1 = L.tickEnd;
        for (int i = tid; i < 2 * SEA16_BUF; i += 256) {
            const int st = i / SEA16_BUF, x = i - st * SEA16_BUF, k = x / kHop, o = x - k * kHop;
            const int tk = T1 - 2 + (k < 3 ? k : 2);
            blob[i] = L.circ[st][(tk & (kSlots - 1)) * kHop + o];
        }
        for (int i = tid; i < 6 * kSpecPad; i += 256) blob[kBlobBins + i] = (&L.bins[0][0])[i];
        float *q = blob + kBlobScal16;
        int *qi = reinterpret_cast<int *>(q + 16);
        if (tid < 3) q[2 + tid] = L.denSum[(T1 - 2 + tid) & (kSlots - 1)]; /* denEn0..2 */
        if (tid == 3) qi[10] = (L.psdOkEnd[0] ? 1 : 0) | (L.psdOkEnd[1] ? 2 : 0);
    }
}

} // namespace sea

#ifdef SEA16P_TIMING
extern "C" int sea_debug_ns16p_timing(unsigned long long *out16, int reset)
{
    if (reset) {
        unsigned long long z[16] = {0};
        return (int)hipMemcpyToSymbol(HIP_SYMBOL(sea::p16::g_ns16p_timing), z, sizeof z);
    }
    return (int)hipMemcpyFromSymbol(out16, HIP_SYMBOL(sea::p16::g_ns16p_timing), 16 * sizeof(unsigned long long));
}
#endif
