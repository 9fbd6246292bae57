/*
 * ns16k_kernel.hip -- the 16 k-native NoiseSup variant behind the reference's batch plug-in symbols:
 *     function/20141106_speech_enhancement/aurora_etsi/NoiseSup.cpp:1140-1407   etsi_denoise_mapping_func_Wiener
 *     function/20141106_speech_enhancement/aurora_etsi/NoiseSup.h:36-53         hop 160, window 480 at offset 80 of a
 *                                                                               640-sample stage buffer, 129 spectral
 *                                                                               values, NS_FFT_LENGTH 512 / ORDER 8
 *     function/20141106_speech_enhancement/aurora_etsi/MelProc.cpp:119-135,556-576   DoGamma, DoGammaIDCT
 *     function/20141106_speech_enhancement/aurora_etsi/rfft.cpp:46-181          rfft (x, 512, 8)
 *
 * One 64-lane wavefront per stream, the whole per-stream state in LDS between the frames of a push and in a
 * kNs16StateFloats-float blob in HBM between pushes.  This is a SIDE path (the reference's hot path is etsi/'s 8 kHz
 * framing, ns_pipe_kernel.hip): written for exactness and clarity, every division and square root on the exact path, the
 * transform as a table of independent butterflies per pass (sea_ns16k_tables).  The scalar recursions (VAD, gain
 * factorisation, SpeechQ*) are the ones of ns_core.h, which the two variants share word for word
 * (diff of etsi/cpp/NoiseSup.c against aurora_etsi/NoiseSup.cpp: sizes, the gammatone windows, the frame gate).
 * Deviation stated in oracle/ns16k_oracle.c: DoGainFact_IBM's log10 is evaluated in double, as in the C tree.
 */
#include "ns_core.h"

namespace sea {

namespace {

constexpr int kHop = SEA16_HOP, kBuf = SEA16_BUF, kIn = SEA16_BUF - SEA16_HOP /* NS_DATA_IN_BUFFER */, kSpec = SEA16_NSPEC;
constexpr int kSpecPad = 132;

#ifndef SEA16_PIPE_SUMS
#define SEA16_PIPE_SUMS 0 /* software-pipelined LDS requests in the frame-sum chain: 256 VGPRs + 150 AGPRs, slower */
#endif
#ifndef SEA16_PIPE_DC
#define SEA16_PIPE_DC 1 /* and in the DC-offset chain */
#endif

/* timing-only diagnostic (-DSEA16_TIMING, tools/ns16k_phases.sh): shader clocks workgroup 0 spends per phase */
#ifdef SEA16_TIMING
__device__ unsigned long long g_ns16_ck[24];
#define CK16_START unsigned long long ck_ = clock64()
#define CK16(k) do { const unsigned long long c_ = clock64(); if (blockIdx.x == 0 && threadIdx.x == 0) g_ns16_ck[k] += c_ - ck_; ck_ = c_; } while (0)
#else
#define CK16_START
#define CK16(k)
#endif

struct __attribute__((aligned(16))) Ns16Lds {
    float buf[2][kBuf];       /* First / SecondStageInFloatBuffer */
    float bins[6][kSpecPad];  /* noiseSE1/2, denSigSE1/2, the other slot of PSDMeanBuffer1/2 */
    float work[SEA16_NFFT];   /* transform workspace */
    float W[kSpecPad];        /* Wiener gains of the 129 values */
    float sq[kHop];           /* squares (frame gate, VAD), then the DC filter's differences */
    float outb[kHop];         /* second-stage output frame */
    float gam[32];            /* the 25 window outputs */
    float fir[20];            /* 17 taps */
    float sq2[kHop];          /* VAD: squares of the first stage's current frame */
};

/* constants staged once per launch and shared by the kNs16Waves streams of a workgroup */
struct __attribute__((aligned(16))) Ns16Tab {
    float ones[kSpecPad];     /* 1.0f: the "window" of the plain in-order sums that ride along with DoGamma */
    float gammaC[SEA16_NGAM][kSpecPad]; /* [c][i], rows 132 floats apart: the 25 lanes' float4 reads fall on different banks */
    float sigWindow[SEA16_NFFT];        /* the analysis window */
    unsigned short rev[SEA16_NFFT];     /* the digit-reversal places */
};
/* streams (= wavefronts) per workgroup: they only share the tables; 4 x 13 KB + 16.8 KB leaves room for two workgroups,
 * eight streams, per CU (one stream per workgroup: 29.8 KB each, five per CU) */
constexpr int kNs16Waves = kNs16StreamsPerGroup;

/* rfft (x, 512, 8) on L.work, elements already at their digit-reversed places.  A pass = up to three butterflies per
 * lane, one per slot, the kind of each slot fixed by the pass (sea_tables.h): straight-line code -- every lane reads the
 * operands of all its slots (an empty slot reads element 0 and discards), computes, and only the stores are predicated --
 * so the LDS round trips of the three butterflies overlap.  Butterflies of one pass touch disjoint elements. */
struct Ns16Fft { /* a lane's share of the schedule, loaded once per launch: its butterflies and their twiddles */
    unsigned slot[SEA16_FFT_PASSES][SEA16_FFT_SLOTS];
    float4 tw[SEA16_FFT_PASSES]; /* cc1, ss1, cc3, ss3 of the twiddled butterfly of passes 3..7 */
};
__device__ __forceinline__ void bf_len2(float *x, unsigned it)
{ /* rfft.cpp:86-91 */
    const int i = (int)(it & 0xffffu);
    const float a0 = x[i], a1 = x[i + 1];
    const float s0 = a0 + a1, s1 = a0 - a1;
    if (it >> 31) {
        x[i] = s0;
        x[i + 1] = s1;
    }
}
template <int N4>
__device__ __forceinline__ void bf_plain(float *x, unsigned it)
{ /* :113-120 */
    const int i1 = (int)(it & 0xffffu), i3 = i1 + 2 * N4, i4 = i1 + 3 * N4;
    const float x1 = x[i1], x3 = x[i3], x4 = x[i4];
    const float t1 = x4 + x3;
    const float r4 = x4 - x3, r3 = x1 - t1, r1 = x1 + t1;
    if (it >> 31) {
        x[i4] = r4;
        x[i3] = r3;
        x[i1] = r1;
    }
}
template <int N4>
__device__ __forceinline__ void bf_pi4(float *x, unsigned it)
{ /* :122-133: the sums are divided by sqrt 2 in double */
    const int i1 = (int)(it & 0xffffu) + N4 / 2, i2 = i1 + N4, i3 = i2 + N4, i4 = i3 + N4;
    const float x1 = x[i1], x2 = x[i2], x3 = x[i3], x4 = x[i4];
    /* == (float)((double)s / M_SQRT2) for every float s (sea_selftest_pi4 sweeps all 2^32) */
    const float t1 = (float)((double)(x3 + x4) * 0.70710678118654752440);
    const float t2 = (float)((double)(x3 - x4) * 0.70710678118654752440);
    const float r4 = x2 - t1, r3 = -x2 - t1, r2 = x1 - t2, r1 = x1 + t2;
    if (it >> 31) {
        x[i4] = r4;
        x[i3] = r3;
        x[i2] = r2;
        x[i1] = r1;
    }
}
template <int N4>
__device__ __forceinline__ void bf_twiddle(float *x, unsigned it, const float4 w)
{ /* :139-176 */
    const int i = (int)(it & 0xffffu), j = (int)((it >> 16) & 0xffu);
    const float cc1 = w.x, ss1 = w.y, cc3 = w.z, ss3 = w.w;
    const int i1 = i + j, i2 = i1 + N4, i3 = i2 + N4, i4 = i3 + N4;
    const int i5 = i + N4 - j, i6 = i5 + N4, i7 = i6 + N4, i8 = i7 + N4;
    const float x1 = x[i1], x2 = x[i2], x3 = x[i3], x4 = x[i4], x5 = x[i5], x6 = x[i6], x7 = x[i7], x8 = x[i8];
    float t1 = x3 * cc1 + x7 * ss1;
    float t2 = x7 * cc1 - x3 * ss1;
    float t3 = x4 * cc3 + x8 * ss3;
    float t4 = x8 * cc3 - x4 * ss3;
    const float t5 = t1 + t3, t6 = t2 + t4;
    t3 = t1 - t3;
    t4 = t2 - t4;
    const float r3 = t6 - x6, r8 = x6 + t6, r7 = -x2 - t3, r4 = x2 - t3, r6 = x1 - t5, r1 = x1 + t5, r5 = x5 - t4, r2 = x5 + t4;
    if (it >> 31) {
        x[i3] = r3;
        x[i8] = r8;
        x[i7] = r7;
        x[i4] = r4;
        x[i6] = r6;
        x[i1] = r1;
        x[i5] = r5;
        x[i2] = r2;
    }
}

template <int PASS>
__device__ __forceinline__ void ns16_fft_pass(float *x, const Ns16Fft &t)
{
    constexpr int N4 = (PASS == 0) ? 0 : (1 << (PASS - 1)); /* n2 = 2^(PASS+1) */
    const unsigned it0 = t.slot[PASS][0], it1 = t.slot[PASS][1];
    if constexpr (PASS == 0) {
        const unsigned it2 = t.slot[PASS][2];
        /* the three length-2 butterflies of a lane touch disjoint pairs: all reads first */
        const int i0 = (int)(it0 & 0xffffu), i1 = (int)(it1 & 0xffffu), i2 = (int)(it2 & 0xffffu);
        const float a0 = x[i0], a1 = x[i0 + 1], b0 = x[i1], b1 = x[i1 + 1], c0 = x[i2], c1 = x[i2 + 1];
        if (it0 >> 31) {
            x[i0] = a0 + a1;
            x[i0 + 1] = a0 - a1;
        }
        if (it1 >> 31) {
            x[i1] = b0 + b1;
            x[i1 + 1] = b0 - b1;
        }
        if (it2 >> 31) {
            x[i2] = c0 + c1;
            x[i2 + 1] = c0 - c1;
        }
    } else if constexpr (PASS == 1) {
        bf_plain<N4>(x, it0);
        bf_plain<N4>(x, it1);
    } else {
        bf_plain<N4>(x, it0);
        bf_pi4<N4>(x, it1);
        if constexpr (PASS >= 3) bf_twiddle<N4>(x, t.slot[PASS][2], t.tw[PASS]);
    }
    wave_sync(); /* passes are ordered */
}

__device__ __forceinline__ void ns16_fft(float *x, const Ns16Fft &t)
{
    ns16_fft_pass<0>(x, t);
    ns16_fft_pass<1>(x, t);
    ns16_fft_pass<2>(x, t);
    ns16_fft_pass<3>(x, t);
    ns16_fft_pass<4>(x, t);
    ns16_fft_pass<5>(x, t);
    ns16_fft_pass<6>(x, t);
    ns16_fft_pass<7>(x, t);
}

/* DCOffsetFil over one 160-sample frame (NoiseSup.cpp:168-184) in the float-FMA form of ns_core.h's dc_filter: the
 * exactness condition is checked for every sample afterwards, the frame redone in double where it fails */
__device__ __forceinline__ void ns16_dc(const float *dif, float *out, float &yState, int lane)
{
    const float y0 = yState;
    float y = y0;
#if SEA16_PIPE_DC
    constexpr int kQ = 8, kChunks = kHop / 4 / kQ;
    float4 d[2][kQ];
    auto request = [&](int c, float4(&dst)[kQ]) {
#pragma unroll
        for (int k = 0; k < kQ; ++k) dst[k] = *reinterpret_cast<const float4 *>(&dif[4 * (c * kQ + k)]);
    };
    request(0, d[0]);
#pragma unroll
    for (int c = 0; c < kChunks; ++c) {
        if (c + 1 < kChunks) request(c + 1, d[(c + 1) & 1]);
#pragma unroll
        for (int k = 0; k < kQ; ++k) {
            float4 o;
            y = __fmaf_rn(0.9990234375f, y, d[c & 1][k].x);
            o.x = y;
            y = __fmaf_rn(0.9990234375f, y, d[c & 1][k].y);
            o.y = y;
            y = __fmaf_rn(0.9990234375f, y, d[c & 1][k].z);
            o.z = y;
            y = __fmaf_rn(0.9990234375f, y, d[c & 1][k].w);
            o.w = y;
            *reinterpret_cast<float4 *>(&out[4 * (c * kQ + k)]) = o;
        }
        if (c + 1 < kChunks) __builtin_amdgcn_sched_barrier(0);
    }
#else
#pragma unroll 4
    for (int n = 0; n < kHop; n += 4) {
        const float4 d = *reinterpret_cast<const float4 *>(&dif[n]);
        float4 o;
        y = __fmaf_rn(0.9990234375f, y, d.x);
        o.x = y;
        y = __fmaf_rn(0.9990234375f, y, d.y);
        o.y = y;
        y = __fmaf_rn(0.9990234375f, y, d.z);
        o.z = y;
        y = __fmaf_rn(0.9990234375f, y, d.w);
        o.w = y;
        *reinterpret_cast<float4 *>(&out[n]) = o;
    }
#endif
    wave_sync();
    bool unsafe = false;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int n = lane + kLanes * k;
        if (n < kHop) unsafe |= !dc_step_ok(dif[n], n == 0 ? y0 : out[n - 1]);
    }
    if (__ballot(unsafe) != 0ull) {
        wave_sync();
        y = y0;
        for (int n = 0; n < kHop; ++n) {
            y = (float)__fma_rn(0.9990234375, (double)y, (double)dif[n]);
            out[n] = y;
        }
    }
    yState = y;
    wave_sync();
}

/* The two in-order sums of squares a frame starts with, advanced together, one per lane:
 *   lane 0  the frame gate's FrameCheck = 0 + sum in[i]^2 (:1160-1164)          over sq[0..159]
 *   lane 1  the VAD's frameEn = 64 + sum cur[i]^2 of the first stage (:373-376)   over sq2[0..159]
 * (a wave issues a dependent add every ~8 clk whatever the number of active lanes) */
__device__ __forceinline__ void ns16_frame_sums(const float *sq, const float *sq2, int lane, float &check, float &vadSum)
{
    const float *src = (lane == 1) ? sq2 : sq;
    float acc = (lane == 1) ? 64.0f : 0.0f;
#if SEA16_PIPE_SUMS
    /* chunks of 8 quads, chunk c + 1 requested before the chain of chunk c starts (an LDS round trip is ~100 clk) */
    constexpr int kQ = 8, kChunks = kHop / 4 / kQ;
    float4 v[2][kQ];
    auto request = [&](int c, float4(&dst)[kQ]) {
#pragma unroll
        for (int k = 0; k < kQ; ++k) dst[k] = *reinterpret_cast<const float4 *>(src + 4 * (c * kQ + k));
    };
    request(0, v[0]);
#pragma unroll
    for (int c = 0; c < kChunks; ++c) {
        if (c + 1 < kChunks) request(c + 1, v[(c + 1) & 1]);
#pragma unroll
        for (int k = 0; k < kQ; ++k) {
            acc += v[c & 1][k].x;
            acc += v[c & 1][k].y;
            acc += v[c & 1][k].z;
            acc += v[c & 1][k].w;
        }
        if (c + 1 < kChunks) __builtin_amdgcn_sched_barrier(0);
    }
#else
#pragma unroll 8
    for (int q = 0; q < kHop / 4; ++q) {
        const float4 v = *reinterpret_cast<const float4 *>(src + 4 * q);
        acc += v.x;
        acc += v.y;
        acc += v.z;
        acc += v.w;
    }
#endif
    check = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc), 0));
    vadSum = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc), 1));
}

/* one stage of one frame (:1207-1366) */
template <int ST>
__device__ __forceinline__ void ns16_stage(Ns16Lds &L, const Ns16Tab &T, NsRegs &s, NsFd &fd, const Ns16Fft &fft, float eps, const float (&idct)[SEA16_NGAM],
                                           float irWin, int lane, int &fdBits, float &gainOut, float vadSum)
{
    CK16_START;
    float *buf = L.buf[ST];
    /* window + zero padding (:209-222) straight to the digit-reversed places */
#pragma unroll
    for (int k = 0; k < SEA16_NFFT / kLanes; ++k) {
        const int i = lane + kLanes * k;
        const float v = (i < SEA16_WIN) ? buf[SEA16_AWIN + i] * T.sigWindow[i] : 0.0f;
        L.work[T.rev[i]] = v;
    }
    wave_sync();
    CK16(ST * 10 + 0);
    ns16_fft(L.work, fft);
    CK16(ST * 10 + 1);

    /* FFTtoPSD (:240-261), PSDMean (:280-294) and FilterCalc (:440-553) per spectral value: b = lane, lane + 64, 128 */
    {
        int nb = s.nbFrame[ST];
        if (nb < 2147483647) nb++;
        s.nbFrame[ST] = nb;
    }
    if (ST == 0) vad_update(s, vad_frame_energy(vadSum)); /* _VAD_ (:350-421); the frame sum comes from ns16_frame_sums */
    CK16(ST * 10 + 2);
    const int nb16 = (int)(short)s.nbFrame[ST];
    const float *x = L.work;
    /* b = lane, lane + 64, 128: the three values of a lane are independent chains */
    float nSigv[3], Pv[3], noisev[3], denv[3], Wv[3];
    bool inDomain = true;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int b = lane + kLanes * k;
        const int bb = (b < kSpec) ? b : (kSpec - 1);
        float nSig;
        if (bb == kSpec - 1) {
            nSig = x[256] * x[256];
        } else {
            const int j0 = 2 * bb, j1 = 2 * bb + 1;
            const float p0 = (bb == 0) ? x[0] * x[0] : (x[j0] * x[j0] + x[SEA16_NFFT - j0] * x[SEA16_NFFT - j0]);
            const float p1 = x[j1] * x[j1] + x[SEA16_NFFT - j1] * x[SEA16_NFFT - j1];
            nSig = (float)((double)(p0 + p1) / 2.0);
        }
        nSigv[k] = nSig;
        Pv[k] = (L.bins[4 + ST][bb] + nSig) / 2.0f;
        noisev[k] = L.bins[ST][bb];
        denv[k] = L.bins[2 + ST][bb];
        inDomain &= ns_psd_in_domain(nSig) && noisev[k] <= 0x1p28f && noisev[k] >= 0x1p-15f;
    }
    /* the guarded fast-division domain of ns_core.h (ns_back): this frame's and the previous frame's spectral values
     * in {0} u [2^-40, 2^48], the noise magnitudes in [2^-15, 2^28]; otherwise IEEE division and sqrtf throughout */
    const bool domainNow = __ballot(!inDomain) == 0ull;
    const bool fast = SEA_NS_FAST_DIV && domainNow && (s.psdOk[ST] != 0);
    s.psdOk[ST] = domainNow ? 1 : 0;
#pragma unroll
    for (int k = 0; k < 3; ++k)
        Wv[k] = fast ? filter_bin<ST, true>(Pv[k], nSigv[k], noisev[k], denv[k], nb16, s.flagVAD, eps)
                     : filter_bin<ST, false>(Pv[k], nSigv[k], noisev[k], denv[k], nb16, s.flagVAD, eps);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int b = lane + kLanes * k;
        if (b < kSpec) {
            L.bins[4 + ST][b] = nSigv[k];
            L.bins[ST][b] = noisev[k];
            L.bins[2 + ST][b] = denv[k];
            L.W[b] = Wv[k];
        }
    }
    wave_sync();
    CK16(ST * 10 + 3);
    CK16(ST * 10 + 4);
    /* DoGamma (MelProc.cpp:119-135): window c = lane < 25 over gains 0..127, in order.  Three plain in-order sums ride
     * along in lanes 25..27 as "windows" of their own -- same multiply-by-coefficient-then-add per step, x * 1.0f == x:
     *   lane 25  mean = sum W[i]          (SpeechQVar, :866-870)     coefficients 1
     *   lane 26  var  = sum W[i] * W[i]                               coefficients W itself
     *   lane 27  sum of denSigSE1 (first stage) / noiseSE2 (second): DoGainFact_IBM (:648, :653), 129 terms */
    float sum = 0.0f;
    {
        const float *src = (lane == 27) ? L.bins[ST == 0 ? 2 : 1] : L.W;
        const float *coef = (lane < SEA16_NGAM) ? T.gammaC[lane] : ((lane == 26) ? L.W : T.ones);
#pragma unroll 8
        for (int q = 0; q < SEA16_GLEN / 4; ++q) {
            const float4 w = *reinterpret_cast<const float4 *>(src + 4 * q);
            const float4 c = *reinterpret_cast<const float4 *>(coef + 4 * q);
            sum += w.x * c.x;
            sum += w.y * c.y;
            sum += w.z * c.z;
            sum += w.w * c.w;
        }
        if (lane == 27) sum += src[SEA16_GLEN]; /* the 129th spectral value */
        if (lane < SEA16_NGAM) L.gam[lane] = sum;
    }
    wave_sync();
    CK16(ST * 10 + 5);
    const float total = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sum), 27));
    if (ST == 0) { /* SpeechQVar on the first 128 gains (:852-893), then :1301-1311, :1362-1365 */
        const float mean = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sum), 25));
        const float var = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sum), 26));
        fdBits = fd_var_sums<SEA16_NFFT / 4>(fd, mean, var, nb16);
        fdBits |= fd_spec_mel(fd, L.gam, nb16) << 1;
        fdBits |= (s.nbSpeech > 4) ? 8 : 0;
    }
    /* DoGainFact_IBM (:634-698) */
    float g = L.gam[(lane < SEA16_NGAM) ? lane : 0];
    if (ST == 0) {
        s.denEn0 = s.denEn1;
        s.denEn1 = s.denEn2;
        s.denEn2 = total;
    } else {
        gain_fact_update(s, total);
        g = (float)((double)(s.alfaGF * g) + (1.0 - (double)s.alfaGF) * 1.0);
        gainOut = g;
        wave_sync();
        if (lane < SEA16_NGAM) L.gam[lane] = g;
        wave_sync();
    }
    CK16(ST * 10 + 6);
    /* DoGammaIDCT (MelProc.cpp:556-576), taps t = lane = 0..8, + DoFilterWindowing (:716-725) */
    {
        float h = 0.0f;
#pragma unroll
        for (int f = 0; f < SEA16_NGAM; ++f) h += L.gam[f] * idct[f];
        const float tap = h * irWin;
        if (lane <= 8) {
            L.fir[8 + lane] = tap;
            L.fir[8 - lane] = tap;
        }
    }
    wave_sync();
    CK16(ST * 10 + 7);
    /* ApplyWF (cur, prv, filterIR, out, 160, 8) (:317-331): predata[160 - j + i] is cur[i - j] in the contiguous stage
     * buffer, so both loops are one sum over j = -8..8 in that order */
    float *dst = (ST == 0) ? (L.buf[1] + kIn) : L.outb;
    {
        float acc[3] = {0.0f, 0.0f, 0.0f}; /* the three outputs of a lane are independent chains: interleaved */
#pragma unroll
        for (int j = -8; j <= 8; ++j) {
            const float fj = L.fir[j + 8];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int i = lane + kLanes * k;
                acc[k] += fj * buf[kHop + ((i < kHop) ? i : 0) - j];
            }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int i = lane + kLanes * k;
            if (i < kHop) dst[i] = acc[k];
        }
    }
    wave_sync();
    CK16(ST * 10 + 8);
}

/* slide a stage buffer by one hop (:1372-1390) */
__device__ __forceinline__ void ns16_slide(float *buf, int lane)
{
    float r[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int i = lane + kLanes * k;
        r[k] = (i < kIn) ? buf[i + kHop] : 0.0f;
    }
    wave_sync();
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int i = lane + kLanes * k;
        if (i < kIn) buf[i] = r[k];
    }
}

constexpr int kBlobBins = 2 * kBuf, kBlobScal16 = kBlobBins + 6 * kSpecPad;
static_assert(kBlobScal16 + 32 == kNs16StateFloats, "state blob layout");

} // namespace

__global__ void __launch_bounds__(64 * kNs16Waves, 2) ns16k_stream_kernel(Ns16StreamArgs a)
{
    __shared__ Ns16Tab T;
    __shared__ Ns16Lds LS[kNs16Waves];
    const int lane = threadIdx.x & (kLanes - 1), wave = threadIdx.x >> 6, b = blockIdx.x * kNs16Waves + wave;
    const sea_ns16k_tables *t = a.tables;
    for (int i = threadIdx.x; i < SEA16_GLEN * SEA16_NGAM; i += kLanes * kNs16Waves) T.gammaC[i % SEA16_NGAM][i / SEA16_NGAM] = (&t->gammaT[0][0])[i];
    for (int i = threadIdx.x; i < kSpecPad; i += kLanes * kNs16Waves) T.ones[i] = 1.0f;
    for (int i = threadIdx.x; i < SEA16_NFFT; i += kLanes * kNs16Waves) {
        T.sigWindow[i] = t->sigWindow[i];
        T.rev[i] = t->rev[i];
    }
    __syncthreads(); /* the only workgroup barrier: from here on every wavefront runs its own stream */
    if (b >= a.n_streams) return;
    Ns16Lds &L = LS[wave];
    NsRegs s;
    NsFd fd;
    float *blob = a.state + (size_t)b * kNs16StateFloats;
    Ns16Fft fft;
#pragma unroll
    for (int p = 0; p < SEA16_FFT_PASSES; ++p) {
#pragma unroll
        for (int k = 0; k < SEA16_FFT_SLOTS; ++k) fft.slot[p][k] = t->fftSlot[p][k][lane];
        fft.tw[p] = *reinterpret_cast<const float4 *>(t->fftTw[p][(fft.slot[p][2] >> 16) & 0xffu]);
    }
    const float eps = t->eps;
    float idct[SEA16_NGAM];
#pragma unroll
    for (int f = 0; f < SEA16_NGAM; ++f) idct[f] = t->idctT[f][lane & 15];
    const float irWin = t->irWin[lane & 15];

    if (a.reset) { /* etsi_denoise_mapping_thread_init (:937-1083) */
        regs_init(s, t->eps);
        fd_init(fd);
        for (int i = lane; i < 2 * kBuf; i += kLanes) (&L.buf[0][0])[i] = 0.0f;
        for (int i = lane; i < 6 * kSpecPad; i += kLanes) (&L.bins[0][0])[i] = (i < 2 * kSpecPad) ? t->eps : 0.0f;
    } else {
        for (int i = lane; i < 2 * kBuf; i += kLanes) (&L.buf[0][0])[i] = blob[i];
        for (int i = lane; i < 6 * kSpecPad; i += kLanes) (&L.bins[0][0])[i] = blob[kBlobBins + i];
        const float *q = blob + kBlobScal16;
        const int *qi = reinterpret_cast<const int *>(q + 16);
        regs_init(s, t->eps);
        s.dcX = q[0]; s.dcY = q[1]; s.denEn0 = q[2]; s.denEn1 = q[3]; s.denEn2 = q[4];
        s.lowSNRtrack = q[5]; s.alfaGF = q[6]; s.meanEn = q[7];
        fd.melMean = q[8]; fd.varMean = q[9]; fd.accTest = q[10]; fd.specMean = q[11];
        fd.mel0 = q[12]; fd.specValues = q[13]; fd.speechInVADQ = q[14];
        s.nbFrame[0] = qi[0]; s.nbFrame[1] = qi[1]; s.flagVAD = qi[2]; s.hangOver = qi[3];
        s.nbSpeech = qi[4]; s.nIn1 = qi[5]; s.nIn2 = qi[6]; s.nOut2 = qi[7];
        s.psdOk[0] = qi[10] & 1; s.psdOk[1] = (qi[10] >> 1) & 1;
    }
    wave_sync();

    const float *in = a.in + (size_t)b * a.nframes * kHop;
    float *out = a.out + (size_t)b * a.nframes * kHop;
    float xin[3], xnext[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int i = lane + kLanes * k;
        xin[k] = (i < kHop && a.nframes > 0) ? in[i] : 0.0f;
    }
    for (int f = 0; f < a.nframes; ++f, in += kHop, out += kHop) {
        const size_t rec = (size_t)b * a.nframes + f;
        CK16_START;
        /* the next frame's samples are requested now and first touched at the end of this frame, just before this
         * frame's stores are issued: a wait for them is then a wait for nothing else */
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int i = lane + kLanes * k;
            xnext[k] = (i < kHop && f + 1 < a.nframes) ? in[kHop + i] : 0.0f;
        }
        /* the frame gate (:1160-1171): float sum of squares in sample order, truncated to int */
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int i = lane + kLanes * k;
            if (i < kHop) {
                L.sq[i] = xin[k] * xin[k];
                const float c = L.buf[0][kHop + i];
                L.sq2[i] = c * c;
            }
        }
        wave_sync();
        float check, vadSum;
        ns16_frame_sums(L.sq, L.sq2, lane, check, vadSum);
        int produced = 0, counter = 0, fdBits = 0;
        float gain = 0.0f; /* lane c < 25: the c-th of the gains func_Wiener prints for this frame */
        CK16(20);
        if (!(check < 1.0f)) { /* (int)check != 0; NaN and overflow convert to INT_MIN on the reference's x86 */
            wave_sync();
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int i = lane + kLanes * k;
                if (i < kHop) L.buf[0][kIn + i] = xin[k];
            }
            wave_sync();
            s.nIn1++;
            if (s.nIn1 - s.nIn2 > 2) { /* :1212 */
                float unused;
                ns16_stage<0>(L, T, s, fd, fft, eps, idct, irWin, lane, fdBits, unused, vadSum);
                s.nIn2++;
                counter = s.nbFrame[0];
            }
            if (s.nIn2 - s.nOut2 > 2) { /* :1230 */
                int unused = 0;
                ns16_stage<1>(L, T, s, fd, fft, eps, idct, irWin, lane, unused, gain, 0.0f);
                s.nOut2++;
                produced = 1;
            }
#ifdef SEA16_TIMING
            ck_ = clock64();
#endif
            ns16_slide(L.buf[0], lane);
            if (s.nIn2) ns16_slide(L.buf[1], lane);
            wave_sync();
            CK16(21);
            if (s.nOut2 > 0) { /* DCOffsetFil on the frame just written (:1392-1395) */
                float d[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int i = lane + kLanes * k;
                    d[k] = (i < kHop) ? (L.outb[i] - ((i == 0) ? s.dcX : L.outb[i - 1])) : 0.0f;
                }
                s.dcX = L.outb[kHop - 1];
                wave_sync();
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int i = lane + kLanes * k;
                    if (i < kHop) L.sq[i] = d[k];
                }
                wave_sync();
                ns16_dc(L.sq, L.outb, s.dcY, lane);
                CK16(22);
            }
        }
        asm volatile("" : "+v"(xnext[0]), "+v"(xnext[1]), "+v"(xnext[2])); /* the wait for the next frame's samples sits here */
#pragma unroll
        for (int k = 0; k < 3; ++k) xin[k] = xnext[k];
        if (produced) { /* once the second stage runs it runs at every frame that passes the gate: nOut2 > 0 here */
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int i = lane + kLanes * k;
                if (i < kHop) out[i] = L.outb[i];
            }
            if (a.wiener && lane < SEA16_NGAM) a.wiener[rec * SEA16_NGAM + lane] = gain; /* the line func_Wiener prints (:1319-1328) */
        }
        if (lane == 0) {
            a.produced[rec] = produced;
            if (a.flags) a.flags[rec] = (unsigned char)fdBits;
            if (a.frame_counter) a.frame_counter[rec] = counter;
        }
        wave_sync();
    }

    for (int i = lane; i < 2 * kBuf; i += kLanes) blob[i] = (&L.buf[0][0])[i];
    for (int i = lane; i < 6 * kSpecPad; i += kLanes) blob[kBlobBins + i] = (&L.bins[0][0])[i];
    if (lane == 0) {
        float *q = blob + kBlobScal16;
        int *qi = reinterpret_cast<int *>(q + 16);
        q[0] = s.dcX; q[1] = s.dcY; q[2] = s.denEn0; q[3] = s.denEn1; q[4] = s.denEn2;
        q[5] = s.lowSNRtrack; q[6] = s.alfaGF; q[7] = s.meanEn;
        q[8] = fd.melMean; q[9] = fd.varMean; q[10] = fd.accTest; q[11] = fd.specMean;
        q[12] = fd.mel0; q[13] = fd.specValues; q[14] = fd.speechInVADQ;
        qi[0] = s.nbFrame[0]; qi[1] = s.nbFrame[1]; qi[2] = s.flagVAD; qi[3] = s.hangOver;
        qi[4] = s.nbSpeech; qi[5] = s.nIn1; qi[6] = s.nIn2; qi[7] = s.nOut2;
        qi[10] = (s.psdOk[0] ? 1 : 0) | (s.psdOk[1] ? 2 : 0);
    }
}

#ifdef SEA16_TIMING
extern "C" int sea_ns16k_timing(unsigned long long *out24, int reset)
{
    if (reset) {
        unsigned long long z[24] = {};
        return hipMemcpyToSymbol(HIP_SYMBOL(g_ns16_ck), z, sizeof z) != hipSuccess;
    }
    return hipMemcpyFromSymbol(out24, HIP_SYMBOL(g_ns16_ck), 24 * sizeof(unsigned long long)) != hipSuccess;
}
#endif

} // namespace sea
