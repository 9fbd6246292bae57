/*
 * ns_core.h -- device code of the two-stage mel-warped Wiener noise suppressor shared by the
 * NoiseSup kernels (ns_kernel.hip: one wavefront per utterance / stream; ns_pipe_kernel.hip: four
 * pipelined wavefronts per utterance).  Everything is written for ONE 64-lane wavefront.
 *
 * Reference path reproduced (results bit-identical up to libm log/log10):
 *   etsi/cpp/NoiseSup.c:1061-1440    DoNoiseSup          two stages, latency gates, buffers
 *   etsi/cpp/NoiseSup.c:182-669      DCOffsetFil .. DoFilterWindowing
 *   etsi/cpp/MelProc.c:82-104,357-378 DoMelFB, DoMelIDCT
 *   etsi/cpp/rfft.c:45-180           rfft (sea_device.h)
 */
#pragma once
#include "sea_device.h"
#include "sea_kernels.h"

namespace sea {

namespace {

constexpr int kRing = 320;
constexpr double kLn2 = 0.69314718055994530942; /* log(2.0) */

/* scratch of the "back" half of a stage (everything after the PSD) */
struct __attribute__((aligned(16))) BackLds {
    float wbuf[68];       /* Wiener gains W[65] */
    float sbuf[68];       /* spectrum to be summed in order (denSigSE1 / noiseSE2) */
    float sq[80];         /* VAD: squared samples; DC filter: differences */
    float mel[28];        /* 25 mel gains */
    float fir[20];        /* 17 filter taps */
};

/* LDS of the single-wave form (streaming kernel): both stages share the scratch */
struct __attribute__((aligned(16))) NsLds {
    float ring[2][kRing]; /* First/SecondStageInFloatBuffer, NoiseSup.c:98-99 */
    float work[256];      /* FFT workspace */
    float psd[68];        /* 65 PSD bins handed from the front half to the back half */
    BackLds back;
    float outb[80];       /* second-stage filter output, then DC-filtered output */
};

/* per-lane constants (sea_ns_tables columns) */
struct NsConst {
    FftRegs fft;
    float win[4];
    int melStart, melLen;
    float melW[SEA_MEL_TAPS];
    float idct[SEA_NMEL];
    float irWin;
    float eps;
};

/* per-utterance recursive state that is not in LDS */
#ifndef SEA_NS_FAST_SQRT
#define SEA_NS_FAST_SQRT 1
#endif
#if SEA_NS_FAST_SQRT
#define SEA_SQRT(x) ns_sqrt_fast(x)
#else
#define SEA_SQRT(x) sqrtf(x)
#endif
#ifndef SEA_NS_PAIR_BINS
#define SEA_NS_PAIR_BINS 1
#endif
#ifndef SEA_NS_FAST_DIV
#define SEA_NS_FAST_DIV 1
#endif
#ifndef SEA_NOISE_SAFE
#define SEA_NOISE_SAFE 1 /* skip the noise range test after a frame that ran inside the fast-division domain (ns_back) */
#endif
#ifndef SEA_P6_G1_PK
#define SEA_P6_G1_PK 1 /* six-wave forms' gain wave: lean square roots and packed gains inside the fast-division domain, as the four-wave forms */
#endif
#ifndef SEA_NS_STEADY
#define SEA_NS_STEADY 1 /* branch-free FilterCalc of (bin lane, bin 64) side by side in the forms without register pairs */
#endif

struct NsRegs {
    /* per bin: [stage]; "Lo" = bin lane (0..63), "Hi" = bin 64 (meaningful in lane 0) */
    float noiseLo[2], noiseHi[2];   /* noiseSE1/2 */
    float denLo[2], denHi[2];       /* denSigSE1/2 */
    float prevLo[2], prevHi[2];     /* other slot of PSDMeanBuffer1/2 = previous frame's PSD */
    /* wave-uniform scalars */
    float dcX, dcY;                 /* prevSamples */
    float denEn0, denEn1, denEn2, lowSNRtrack, alfaGF;
    float meanEn;
    int nbFrame[2];
    int flagVAD, hangOver, nbSpeech; /* X_INT16 in the reference */
    int nIn1, nIn2, nOut2;
    int onset;
    int psdOk[2];                   /* previous frame's PSD was inside the fast-division domain (ns_psd_in_domain) */
    int noiseSafe[2];               /* the previous frame ran inside the domain: its noise update cannot have left [2^-15, 2^28]
                                     * (ns_back); 0 after DoNoiseSupInit and after any reload of the state: the range is then tested */
};

__device__ __forceinline__ void regs_init(NsRegs &s, float eps)
{ /* DoNoiseSupInit, NoiseSup.c:884-968 */
#pragma unroll
    for (int st = 0; st < 2; ++st) {
        s.noiseLo[st] = s.noiseHi[st] = eps;
        s.denLo[st] = s.denHi[st] = 0.0f;
        s.prevLo[st] = s.prevHi[st] = 0.0f;
        s.nbFrame[st] = 0;
        s.psdOk[st] = 1; /* all-zero history is inside the domain */
        s.noiseSafe[st] = 0;
    }
    s.dcX = s.dcY = 0.0f;
    s.denEn0 = s.denEn1 = s.denEn2 = 0.0f;
    s.lowSNRtrack = 0.0f;
    s.alfaGF = (float)0.8;
    s.meanEn = 0.0f;
    s.flagVAD = s.hangOver = s.nbSpeech = 0;
    s.nIn1 = s.nIn2 = s.nOut2 = 0;
    s.onset = 0;
}

__device__ __forceinline__ float uniform_f(float v)
{
    return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}

/* Natural logarithm of a positive, finite, normal double (the only arguments the hot path has:
 * a float >= 64 promoted to double, and a float > 1e-5).  The reference calls libm's log()/log10()
 * (< 1 ulp) at two scalar sites per frame and immediately rounds the result of a short double
 * expression to float; any double log accurate to a few ulp yields the same float except when the
 * exact value sits within ~1e-16 (relative) of a float rounding boundary (probability ~1e-8 per
 * call) -- the same caveat the device library's own log carries against glibc.  This one is
 * ~3x shorter than the general-purpose library routine, which matters because both sites sit on
 * lane-redundant critical chains:  x = m 2^e, m in [sqrt(1/2), sqrt 2), f = (m-1)/(m+1),
 * ln m = 2 f (1 + f^2/3 + f^4/5 + ... + f^22/23), ln x = e ln2_hi + (e ln2_lo + ln m).
 * Measured against 40-digit references by tests/test_gpu_parity.py::test_selftest_log. */
__device__ __forceinline__ double ns_ln_series(double x)
{
    const long long bits = __double_as_longlong(x);
    int e = (int)((bits >> 52) & 0x7ff) - 1023;
    double m = __longlong_as_double((bits & 0x000fffffffffffffLL) | 0x3ff0000000000000LL); /* [1,2) */
    const bool big = m > 1.4142135623730951;
    m = big ? m * 0.5 : m;
    e = big ? e + 1 : e;
    /* f = (m-1)/(m+1) by reciprocal refinement (m+1 in [1.7, 2.42]: no scaling needed) */
    const double d = m + 1.0, n = m - 1.0;
    double r = __builtin_amdgcn_rcp(d);
    r = __fma_rn(__fma_rn(-d, r, 1.0), r, r);
    r = __fma_rn(__fma_rn(-d, r, 1.0), r, r);
    double f = n * r;
    f = __fma_rn(__fma_rn(-d, f, n), r, f);
    const double f2 = f * f;
    double p = 1.0 / 23.0;
    p = __fma_rn(p, f2, 1.0 / 21.0);
    p = __fma_rn(p, f2, 1.0 / 19.0);
    p = __fma_rn(p, f2, 1.0 / 17.0);
    p = __fma_rn(p, f2, 1.0 / 15.0);
    p = __fma_rn(p, f2, 1.0 / 13.0);
    p = __fma_rn(p, f2, 1.0 / 11.0);
    p = __fma_rn(p, f2, 1.0 / 9.0);
    p = __fma_rn(p, f2, 1.0 / 7.0);
    p = __fma_rn(p, f2, 1.0 / 5.0);
    p = __fma_rn(p, f2, 1.0 / 3.0);
    /* ln m = 2f + 2f * f2 * p  (the leading term kept separate: it carries almost all the value) */
    const double two_f = f + f;
    const double lnm = __fma_rn(two_f * f2, p, two_f);
    const double de = (double)e;
    /* ln2 split so that e * ln2_hi is exact for |e| < 2^10 */
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    return __fma_rn(de, ln2_hi, __fma_rn(de, ln2_lo, lnm));
}


/* The same logarithm, table-driven (round 3): both call sites sit on the two longest role waves of the pipelined
 * kernel and the series above is a chain of ~30 dependent double-precision instructions.  Here
 *     x = m 2^e, m in [sqrt 1/2, sqrt 2);   i = interval of m (256 intervals of 2^15 float patterns, ns_logtab.inc)
 *     r = m c_i - 1      exact: m is a float (both sites pass floats), c_i has 26 significant bits, |r| <= 2^-9
 *     ln x = e ln2 + (th_i + tl_i) + (r - r^2/2 + r^3/3 - r^4/4 + r^5/5 - r^6/6)
 * ~11 dependent double instructions and one 24-byte scalar table read.  th + tl = -ln c_i to 106 bits; the interval
 * around 1 has c = 1, th = tl = 0, so ln x keeps its relative accuracy where it vanishes.  Truncation r^7/7 is below
 * 2^-56 of r; measured against 40-digit references by test_selftest_log_accuracy, and both complete call sites are
 * swept over every float argument against the double-double slow path by sea_selftest_log_guard.
 * The argument must be a positive normal FLOAT (as a double). */
static __device__ const double kNsLogTab[256][3] = {
#include "ns_logtab.inc"
};
template <bool UNI = false> /* UNI: the argument is wave-uniform -> the table row comes through a scalar load */
__device__ __forceinline__ double ns_ln(double x)
{
#ifdef SEA_LN_SERIES
    return ns_ln_series(x);
#else
    const unsigned fb = __float_as_uint((float)x); /* exact: x is a float */
    int e = (int)(fb >> 23) - 127;
    unsigned mb = (fb & 0x007fffffu) | 0x3f800000u; /* m in [1, 2) */
    const bool big = mb > 0x3FB504F2u;               /* m >= sqrt 2 (as float patterns): halve it */
    mb = big ? mb - 0x00800000u : mb;
    e = big ? e + 1 : e;
    unsigned idx = (mb - 0x3F3504F3u) >> 15;         /* 0 .. 255 */
    if (UNI) idx = (unsigned)__builtin_amdgcn_readfirstlane((int)idx);
    const double *t = kNsLogTab[idx];
    const double c = t[0], th = t[1], tl = t[2];
    const double m = (double)__uint_as_float(mb);
    const double r = __fma_rn(m, c, -1.0);           /* exact */
    const double r2 = r * r;
    double p = -1.0 / 6.0;
    p = __fma_rn(p, r, 1.0 / 5.0);
    p = __fma_rn(p, r, -1.0 / 4.0);
    p = __fma_rn(p, r, 1.0 / 3.0);
    p = __fma_rn(p, r, -0.5);
    const double de = (double)e;
    /* ln2 split so that e * ln2_hi is exact for |e| < 2^10 */
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const double small = __fma_rn(r2, p, r) + __fma_rn(de, ln2_lo, tl);
    return __fma_rn(de, ln2_hi, th) + small;
#endif
}

/* ---- the guard of the lean log --------------------------------------------------------------------------
 * Both call sites round a short double expression of the log to float (NoiseSup.c:391, :607).  ns_ln is within
 * 0.69 ulp of ln; glibc's log (what the reference calls) within 0.52 ulp, its log10 within 2 ulp.  The float can
 * only differ when the double expression lands within a few double ulps of a float ROUNDING BOUNDARY (the low 29
 * bits of its significand = 2^28).  ns_near_float_boundary() tests exactly that, with a window K that covers
 * both logs' error bounds propagated through the expression (site 1: 1.21 ulp of the log -> <= 4.4 ulp of the
 * result, + 1.5 for the folded constants, K = 10; site 2: 4.5 ulp -> <= 13, + 1.5, K = 20).  Inside the window
 * (probability (2K+1) 2^-29 per call: 4e-8 / 8e-8) the logarithm is recomputed in double-double arithmetic, accurate to 2^-80, and rounded once --
 * the correctly rounded value, which is what glibc's log returns in all but ~4 % of such cases; at site 2 it is
 * put through the very formula glibc's log10 uses (fdlibm e_log10.c:  z = k log10_2lo + ivln10 log(x'),
 * z + k log10_2hi, verified equal to this image's log10 on 2 M arguments).  What remains is the reference libm's own
 * last-bit freedom (its log has CPU-specific FMA variants): ~1e-10 per call, against ~4e-9 without the guard.
 * sea_selftest_log sweeps EVERY float argument either site can see and reports the guard hits. */
__device__ __forceinline__ bool ns_near_float_boundary(double v, int K)
{
    const int low = (int)((unsigned)__double_as_longlong(v) & 0x1FFFFFFFu); /* bits below a float's 24-bit significand */
    const int d = low - 0x10000000;
    return (d <= K) && (d >= -K);
}

struct NsDD {
    double hi, lo;
};
__device__ __forceinline__ NsDD dd_fast_two_sum(double a, double b) /* |a| >= |b| */
{
    const double s = a + b;
    return NsDD{s, b - (s - a)};
}
__device__ __forceinline__ NsDD dd_two_sum(double a, double b)
{
    const double s = a + b, bb = s - a;
    return NsDD{s, (a - (s - bb)) + (b - bb)};
}
__device__ __forceinline__ NsDD dd_two_prod(double a, double b)
{
    const double p = a * b;
    return NsDD{p, __fma_rn(a, b, -p)};
}
__device__ __forceinline__ NsDD dd_add(NsDD a, NsDD b)
{
    NsDD s = dd_two_sum(a.hi, b.hi);
    const NsDD t = dd_two_sum(a.lo, b.lo);
    s = dd_fast_two_sum(s.hi, s.lo + t.hi);
    return dd_fast_two_sum(s.hi, s.lo + t.lo);
}
__device__ __forceinline__ NsDD dd_mul(NsDD a, NsDD b)
{
    NsDD p = dd_two_prod(a.hi, b.hi);
    p.lo += a.hi * b.lo + a.lo * b.hi;
    return dd_fast_two_sum(p.hi, p.lo);
}
/* natural logarithm of a positive normal double as an unevaluated sum hi + lo, relative error < 2^-80:
 * x = m 2^e, m in [sqrt(1/2), sqrt 2), f = (m-1)/(m+1) in double-double, ln m = 2 f sum_k f^2k/(2k+1) (k <= 17:
 * f^2 <= 0.0295; the terms k >= 6 are below 1e-10 and run in plain double), ln x = e ln2 + ln m. */
__device__ __attribute__((noinline)) NsDD ns_ln_dd(double x)
{
    const long long bits = __double_as_longlong(x);
    int e = (int)((bits >> 52) & 0x7ff) - 1023;
    double m = __longlong_as_double((bits & 0x000fffffffffffffLL) | 0x3ff0000000000000LL);
    if (m > 1.4142135623730951) {
        m *= 0.5;
        e += 1;
    }
    const double n = m - 1.0;            /* exact (Sterbenz) */
    const NsDD d = dd_two_sum(m, 1.0);   /* m + 1 may need 54 bits */
    /* f = n / d: two quotient digits, remainder taken exactly */
    const double q1 = n / d.hi;
    const NsDD p1 = dd_two_prod(q1, d.hi);
    const double r = ((n - p1.hi) - p1.lo) - q1 * d.lo;
    const double q2 = r / d.hi;
    const NsDD f = dd_fast_two_sum(q1, q2);
    const NsDD f2 = dd_mul(f, f);
    double t = 1.0 / 35.0;               /* sum_{k=6..17} f2^(k-6) / (2k+1), Horner in double (f2^6 < 7e-10) */
    t = __fma_rn(t, f2.hi, 1.0 / 33.0);
    t = __fma_rn(t, f2.hi, 1.0 / 31.0);
    t = __fma_rn(t, f2.hi, 1.0 / 29.0);
    t = __fma_rn(t, f2.hi, 1.0 / 27.0);
    t = __fma_rn(t, f2.hi, 1.0 / 25.0);
    t = __fma_rn(t, f2.hi, 1.0 / 23.0);
    t = __fma_rn(t, f2.hi, 1.0 / 21.0);
    t = __fma_rn(t, f2.hi, 1.0 / 19.0);
    t = __fma_rn(t, f2.hi, 1.0 / 17.0);
    t = __fma_rn(t, f2.hi, 1.0 / 15.0);
    t = __fma_rn(t, f2.hi, 1.0 / 13.0);
    /* S = 1 + f2 (1/3 + f2 (1/5 + f2 (1/7 + f2 (1/9 + f2 (1/11 + f2 t))))) in double-double; constants to 106 bits */
    NsDD S = dd_mul(f2, NsDD{t, 0.0});
    S = dd_mul(f2, dd_add(S, NsDD{0.090909090909090912, -2.5232341468753558e-18}));
    S = dd_mul(f2, dd_add(S, NsDD{0.1111111111111111, 6.1679056923619804e-18}));
    S = dd_mul(f2, dd_add(S, NsDD{0.14285714285714285, 7.9301644616082606e-18}));
    S = dd_mul(f2, dd_add(S, NsDD{0.20000000000000001, -1.1102230246251566e-17}));
    S = dd_mul(f2, dd_add(S, NsDD{0.33333333333333331, 1.8503717077085941e-17}));
    S = dd_add(S, NsDD{1.0, 0.0});
    NsDD lnm = dd_mul(f, S);
    lnm.hi *= 2.0;
    lnm.lo *= 2.0;
    /* e ln2, ln2 to 106 bits */
    const double de = (double)e;
    NsDD el = dd_two_prod(de, 0.69314718055994529);
    el.lo += de * 2.3190468138462996e-17;
    el = dd_fast_two_sum(el.hi, el.lo);
    return dd_add(el, lnm);
}
/* the double nearest to ln x (correctly rounded but for ties closer than 2^-80) */
__device__ __forceinline__ double ns_ln_cr(double x)
{
    const NsDD l = ns_ln_dd(x);
    return l.hi + l.lo;
}
/* log10 x as glibc computes it (sysdeps/ieee754/dbl-64/e_log10.c, the fdlibm formula) with the correctly
 * rounded log inside; x positive and normal */
__device__ __forceinline__ double ns_log10_slow(double x)
{
    const double ivln10 = 4.34294481903251816668e-01, log10_2hi = 3.01029995663611771306e-01,
                 log10_2lo = 3.69423907715893078616e-13;
    const long long bits = __double_as_longlong(x);
    const int k = (int)((bits >> 52) & 0x7ff) - 1023;
    const int i = (k < 0) ? 1 : 0;
    const double y = (double)(k + i);
    const double xr = __longlong_as_double((bits & 0x000fffffffffffffLL) | ((long long)(0x3ff - i) << 52));
    const double z = y * log10_2lo + ivln10 * ns_ln_cr(xr);
    return z + y * log10_2hi;
}

/* the two sites, complete: fast log, guard, slow path.  *hit (optional) reports that the guard fired.
 * The fast forms also fold the expression's constant divisions into ONE multiplication -- (L / ln2) * 16 = L * (16 / ln2),
 * (20 L log10e) / 3 = L * (20 log10e / 3): each differs from the reference's operation sequence by at most ~1.5 ulp
 * of the double result, which the guard windows (K = 10 / 20 double ulps around a float rounding boundary) cover
 * together with the logs' own error bounds; a double division is a ~15-instruction dependent chain on the two
 * longest role waves.  The slow forms keep the reference's literal operations.  sea_selftest_log_guard() runs BOTH
 * forms on every float argument and counts disagreements outside the window (none). */
__device__ __forceinline__ float ns_vad_energy_slow(float frameSum)
{
    return (float)(0.5 + (ns_ln_cr((double)frameSum / 64.0) / kLn2) * 16.0);
}
template <bool UNI = false>
__device__ __forceinline__ float ns_vad_energy_expr(float frameSum, bool *hit = nullptr)
{ /* NoiseSup.c:391 */
    const double v = __fma_rn(ns_ln<UNI>((double)frameSum * 0.015625), 23.083120654223414 /* 16 / ln 2 */, 0.5);
    const bool near = ns_near_float_boundary(v, 10);
    if (hit) *hit = near;
    if (__builtin_expect(near, 0)) return ns_vad_energy_slow(frameSum);
    return (float)v;
}
__device__ __forceinline__ float ns_aversnr_slow(float averSNR)
{
    return (float)((20 * ns_log10_slow((double)averSNR)) / 3.0);
}
template <bool UNI = false>
__device__ __forceinline__ float ns_aversnr_expr(float averSNR, bool *hit = nullptr)
{ /* NoiseSup.c:607; the caller has established (double)averSNR > 0.00001 */
    const double v = ns_ln<UNI>((double)averSNR) * 2.8952965460216789 /* 20 log10(e) / 3 */;
    const bool near = ns_near_float_boundary(v, 20);
    if (hit) *hit = near;
    if (__builtin_expect(near, 0)) return ns_aversnr_slow(averSNR);
    return (float)v;
}

/* FilterCalc (NoiseSup.c:449-563) for one PSD bin, in two pieces so that the pipelined kernel can
 * run them in different waves.  nb is the frame counter narrowed to int16 as the reference does
 * (SURVEY F9).
 *
 * Second-stage noise tracking in the energy domain (:486-517): P = 2-frame mean PSD. */
/* Correctly rounded square root for x = 0 or x in [2^-96, 2^126]: the compiler's sqrtf expansion (v_sqrt_f32, then the
 * neighbours one ulp below / above tested with an exact fma residual) without its input scaling for tiny
 * arguments and its zero / infinity fix-up, which do nothing on this range (x = 0: v_sqrt gives 0 and neither
 * neighbour is selected).  9 instead of 15 instructions; used inside the fast-division domain only, where
 * every argument is 0 or in [2^-41, 2^58]. */
__device__ __forceinline__ float ns_sqrt_fast(float x)
{
    const float s = __builtin_amdgcn_sqrtf(x);
    const float sm = __int_as_float(__float_as_int(s) - 1), sp = __int_as_float(__float_as_int(s) + 1);
    const float rm = __fmaf_rn(-sm, s, x), rp = __fmaf_rn(-sp, s, x);
    float r = (0.0f >= rm) ? sm : s;
    r = (0.0f < rp) ? sp : r;
    return r;
}

/* ---- IEEE division without the range scaffolding ------------------------------------------------------
 * The compiler expands a / b into  v_div_scale x2, v_rcp, fma, fma | mul, fma, fma, fma, v_div_fmas |
 * v_div_fixup  (11 vector instructions; a third of the BACK waves' instruction count was division).  For
 * operands with  b normal, 1/b normal, a == 0 or |a| >= 2^-102, and -126 < exponent(a) - exponent(b) < 96
 * both v_div_scale return their operand unchanged with VCC = 0, v_div_fmas is then a plain fma and
 * v_div_fixup returns the quotient as it is (a == 0: the sequence below yields +0, like the fixup).  Inside
 * that domain the SAME instruction sequence without the three scaffolding instructions gives the same bits,
 * and the denominator-only part (rcp + 2 fma) is shared by the quotients that have a common denominator:
 * 5 + 3 instructions instead of 11 per quotient.  ns_back() establishes the domain per frame
 * (ns_psd_in_domain + the bounds derived in its comment) and falls back to plain division outside it;
 * sea_selftest_nsdiv() compares both forms bit for bit over random and edge operands of the domain. */
struct NsRcp {
    float d, r;
};
__device__ __forceinline__ NsRcp ns_rcp(float d)
{
    const float r0 = __builtin_amdgcn_rcpf(d);
    const float e = __fmaf_rn(-d, r0, 1.0f);
    return NsRcp{d, __fmaf_rn(e, r0, r0)};
}
__device__ __forceinline__ float ns_div(float n, const NsRcp &R)
{
    float q = n * R.r;
    float e = __fmaf_rn(-R.d, q, n);
    q = __fmaf_rn(e, R.r, q);
    e = __fmaf_rn(-R.d, q, n);
    return __fmaf_rn(e, R.r, q);
}
/* 1.0 / d in double for 1 <= d < 2^512: the compiler's sequence (rcp, 4 fma, mul, fma, fmas) with the
 * multiply by the numerator 1.0 dropped (exact) */
__device__ __forceinline__ double ns_inv64(double d)
{
    double r = __builtin_amdgcn_rcp(d);
    double e = __fma_rn(-d, r, 1.0);
    r = __fma_rn(r, e, r);
    e = __fma_rn(-d, r, 1.0);
    r = __fma_rn(r, e, r);
    e = __fma_rn(-d, r, 1.0);
    return __fma_rn(e, r, r);
}
/* the per-frame domain test: a PSD value that is 0 or in [2^-40, 2^48] (NaN, Inf, negative: false); full-scale
 * int16 audio reaches (200 * 32768)^2 = 2^45.3 */
__device__ __forceinline__ bool ns_psd_in_domain(float v)
{
    const unsigned lo = 0x2B800000u /* 2^-40 */, hi = 0x57800000u /* 2^48 */;
    return (v == 0.0f) || ((__float_as_uint(v) - lo) <= (hi - lo));
}

template <bool FAST>
__device__ __forceinline__ void noise_track1(float P, float &noise, int nb, float eps)
{
    float n2 = noise * noise;
    if (nb < 11) {
        const float lambda = 1 - 1 / (float)nb;
        n2 = lambda * n2 + (1 - lambda) * P;
    } else {
        /* FAST: P in {0} u [2^-41, 2^48], n2 in [2^-30, 2^57] -> exponent differences within [-99, 78] */
        const float r1 = FAST ? ns_div(P, ns_rcp(P + n2)) : P / (P + n2);
        const float r2 = FAST ? ns_div(P, ns_rcp(n2)) : P / n2;
        const double inv = FAST ? ns_inv64(1.0 + 0.1 * (double)r2) : 1.0 / (1.0 + 0.1 * (double)r2);
        const float upd = (float)(0.9 + 0.1 * (double)r1 * (1.0 + inv));
        n2 *= upd;
    }
    n2 = FAST ? SEA_SQRT(n2) : sqrtf(n2);
    noise = (n2 < eps) ? eps : n2;
}

/* Wiener gain of one bin given the (already updated) noise magnitude (:522-526, :551-560).
 * Psqrt = sqrt of the mean PSD, nSigSqrt = sqrt of this frame's PSD.
 * FAST domain (see ns_back): Psqrt, nSigSqrt in {0} u [2^-20.5, 2^24], den in {0} u [2^-24, 2^24], noise in
 * [2^-15, 2^29]  ->  prio in {0} u [2^-54, 2^40] (post > 0 implies post >= 2^-23), W in {0} u [2^-55, 1],
 * W * Psqrt in {0} u [2^-75.5, 2^24]; every numerator is 0 or >= 2^-76, every exponent difference is within
 * [-105, 40].  The returned den = W2 * nSigSqrt with W2 in [0.0736, 1] is 0 or in [2^-24, 2^24] again. */
template <bool FAST>
__device__ __forceinline__ float gain_bin(float Psqrt, float nSigSqrt, float noise, float &den)
{
    const float beta = (float)0.98, rsbMin = (float)0.079432823;
    if (FAST) {
        const NsRcp rn = ns_rcp(noise);
        const float post = ns_div(Psqrt, rn) - 1;
        float prio = beta * ns_div(den, rn) + (1 - beta) * ((0 > post) ? 0 : post);
        float W = ns_div(prio, ns_rcp(1 + prio));
        prio = ns_div(W * Psqrt, rn);
        prio = (prio > rsbMin) ? prio : rsbMin;
        W = ns_div(prio, ns_rcp(1 + prio));
        den = W * nSigSqrt;
        return W;
    }
    const float post = (Psqrt / noise) - 1;
    float prio = beta * (den / noise) + (1 - beta) * ((0 > post) ? 0 : post);
    float W = prio / (1 + prio);
    prio = W * Psqrt / noise;
    prio = (prio > rsbMin) ? prio : rsbMin;
    W = prio / (1 + prio);
    den = W * nSigSqrt;
    return W;
}

/* the whole bin: P = 2-frame mean PSD, nSig = this frame's PSD */
/* ---- bins (lane, 64) as register pairs -------------------------------------------------------------------
 * Bin 64 is computed by every lane (one value for the whole wave), which used to repeat the whole per-bin
 * instruction sequence.  Inside the fast-division domain the gain computation is straight-line mul / add / fma
 * code, so the pair (bin lane, bin 64) goes through it as one two-component vector: v_pk_mul_f32 / v_pk_add_f32 /
 * v_pk_fma_f32 round each half exactly like the scalar instruction.  Only the reciprocals, the two selects and
 * the square roots stay per component. */
typedef float ns_v2f __attribute__((ext_vector_type(2)));
struct NsRcp2 {
    ns_v2f d, r;
};
__device__ __forceinline__ ns_v2f ns_fma2(ns_v2f a, ns_v2f b, ns_v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ NsRcp2 ns_rcp2(ns_v2f d)
{
    const ns_v2f r0 = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    const ns_v2f e = ns_fma2(-d, r0, ns_v2f{1.0f, 1.0f});
    return NsRcp2{d, ns_fma2(e, r0, r0)};
}
__device__ __forceinline__ ns_v2f ns_div2(ns_v2f n, const NsRcp2 &R)
{
    ns_v2f q = n * R.r;
    ns_v2f e = ns_fma2(-R.d, q, n);
    q = ns_fma2(e, R.r, q);
    e = ns_fma2(-R.d, q, n);
    return ns_fma2(e, R.r, q);
}
/* gain_bin<true> on the pair; same operations in the same order per component */
__device__ __forceinline__ ns_v2f gain_bin2(ns_v2f Psqrt, ns_v2f nSigSqrt, ns_v2f noise, ns_v2f &den)
{
    const float beta = (float)0.98, rsbMin = (float)0.079432823;
    const NsRcp2 rn = ns_rcp2(noise);
    const ns_v2f post = ns_div2(Psqrt, rn) - 1;
    const ns_v2f postc = {(0 > post.x) ? 0 : post.x, (0 > post.y) ? 0 : post.y};
    ns_v2f prio = beta * ns_div2(den, rn) + (1 - beta) * postc;
    ns_v2f W = ns_div2(prio, ns_rcp2(1 + prio));
    prio = ns_div2(W * Psqrt, rn);
    prio = ns_v2f{(prio.x > rsbMin) ? prio.x : rsbMin, (prio.y > rsbMin) ? prio.y : rsbMin};
    W = ns_div2(prio, ns_rcp2(1 + prio));
    den = W * nSigSqrt;
    return W;
}

/* filter_bin<ST, true> for the bins (lane, 64) together: noise tracking and square roots per component exactly
 * as filter_bin does them, the gain computation on the pair */
template <int ST>
__device__ __forceinline__ void filter_bins_fast(float PLo, float PHi, float nSigLo, float nSigHi, float &noiseLo,
                                                 float &noiseHi, float &denLo, float &denHi, int nb, int flagVAD,
                                                 float eps, float &WLo, float &WHi)
{
    if (ST == 1) { /* (packing the two quotients of the noise update as well measured no gain) */
        noise_track1<true>(PLo, noiseLo, nb, eps);
        noise_track1<true>(PHi, noiseHi, nb, eps);
    }
    const ns_v2f nSig = {SEA_SQRT(nSigLo), SEA_SQRT(nSigHi)};
    const ns_v2f P = {SEA_SQRT(PLo), SEA_SQRT(PHi)};
    if (ST == 0) { /* VAD-gated noise tracking in magnitude, :531-546 */
        const float lambda = (nb < 100) ? 1 - 1 / (float)nb : (float)0.99;
        if (flagVAD == 0) {
            const float nl = lambda * noiseLo + (1 - lambda) * P.x, nh = lambda * noiseHi + (1 - lambda) * P.y;
            noiseLo = (nl < eps) ? eps : nl;
            noiseHi = (nh < eps) ? eps : nh;
        }
    }
    ns_v2f den = {denLo, denHi};
    const ns_v2f W = gain_bin2(P, nSig, ns_v2f{noiseLo, noiseHi}, den);
    denLo = den.x;
    denHi = den.y;
    WLo = W.x;
    WHi = W.y;
}

template <int ST, bool FAST = false>
__device__ __forceinline__ float filter_bin(float P, float nSig, float &noise, float &den, int nb,
                                            int flagVAD, float eps)
{
    if (ST == 1) noise_track1<FAST>(P, noise, nb, eps);
    nSig = FAST ? SEA_SQRT(nSig) : sqrtf(nSig); /* :522-526, (float)sqrt((double)x) == correctly rounded sqrtf */
    P = FAST ? SEA_SQRT(P) : sqrtf(P);
    if (ST == 0) { /* VAD-gated noise tracking in magnitude, :531-546 */
        const float lambda = (nb < 100) ? 1 - 1 / (float)nb : (float)0.99;
        if (flagVAD == 0) {
            const float n = lambda * noise + (1 - lambda) * P;
            noise = (n < eps) ? eps : n;
        }
    }
    return gain_bin<FAST>(P, nSig, noise, den);
}

/* filter_bin<ST, true> without control flow (ST 1: valid for nb >= 11 only: noise_track1's steady-state branch), the reference's
 * `if (flagVAD == 0)` as a select of the same values: two or three such chains (bins 0..63 | bin 64; utterance a | b) written
 * back to back sit in ONE basic block and the scheduler interleaves them -- a wave issues a dependent instruction every
 * ~8 clk, an independent one every ~2. */
template <int ST>
__device__ __forceinline__ float filter_steady(float P, float nSig, float &noise, float &den, int nb, int flagVAD, float eps)
{
    if (ST == 1) { /* noise_track1<true>, nb >= 11 (NoiseSup.c:492-508) */
        float n2 = noise * noise;
        const float r1 = ns_div(P, ns_rcp(P + n2));
        const float r2 = ns_div(P, ns_rcp(n2));
        const double inv = ns_inv64(1.0 + 0.1 * (double)r2);
        const float upd = (float)(0.9 + 0.1 * (double)r1 * (1.0 + inv));
        n2 *= upd;
        n2 = SEA_SQRT(n2);
        noise = (n2 < eps) ? eps : n2;
    }
    nSig = SEA_SQRT(nSig);
    P = SEA_SQRT(P);
    if (ST == 0) { /* VAD-gated noise tracking in magnitude, :531-546 */
        const float lambda = (nb < 100) ? 1 - 1 / (float)nb : (float)0.99;
        const float n = lambda * noise + (1 - lambda) * P;
        const float nn = (n < eps) ? eps : n;
        noise = (flagVAD == 0) ? nn : noise;
    }
    return gain_bin<true>(P, nSig, noise, den);
}

/* VAD frame log-energy (NoiseSup.c:386-391) from 64 + sum of the 80 squared samples: depends on
 * the raw frame only, so the pipelined kernel computes it in its helper wave */
__device__ __forceinline__ float vad_frame_energy(float frameSum)
{
#ifdef SEA_LIBM_LOG
    return uniform_f((float)(0.5 + (log((double)frameSum / 64.0) / kLn2) * 16.0));
#else
    return uniform_f(ns_vad_energy_expr<true>(uniform_f(frameSum)));
#endif
}

/* squares of the raw frame frame[0..79] -> sq[0..79], then the in-order sum (all lanes) */
__device__ __forceinline__ float vad_frame_sum(const float *frame, float *sq, int lane)
{
    const float x = frame[lane];
    sq[lane] = x * x;
    if (lane < 16) {
        const float y = frame[64 + lane];
        sq[64 + lane] = y * y;
    }
    wave_sync();
#ifdef SEA_ABLATE_VADSUM
    return 64.0f + sq[0] + sq[79];
#else
    return serial_sum<80>(sq, 64.0f);
#endif
}

/* VAD for noise suppression, NoiseSup.c:359-430 (first stage only does work) */
__device__ __forceinline__ void vad_update(NsRegs &s, float frameEn)
{
    const int nb = s.nbFrame[0];
    const float lambdaLTE = (nb < 10) ? 1 - 1 / (float)nb : (float)0.97;
    float meanEn = s.meanEn;
    if (((frameEn - meanEn) < 20.0f) || (nb < 10)) {
        if ((frameEn < meanEn) || (nb < 10))
            meanEn += (1 - lambdaLTE) * (frameEn - meanEn);
        else
            meanEn += (1 - (float)0.99) * (frameEn - meanEn);
        if (meanEn < 80.0f) meanEn = 80.0f;
    }
    if (nb > 4) {
        if ((frameEn - meanEn) > 15.0f) {
            s.flagVAD = 1;
            s.nbSpeech = (short)(s.nbSpeech + 1);
        } else {
            if (s.nbSpeech > 4) s.hangOver = 15;
            s.nbSpeech = 0;
            if (s.hangOver != 0) {
                s.hangOver--;
                s.flagVAD = 1;
            } else
                s.flagVAD = 0;
        }
    }
    s.meanEn = meanEn;
}

/* second-stage gain factorisation scalars, NoiseSup.c:600-637; returns alfaGF */
__device__ __forceinline__ void gain_fact_update(NsRegs &s, float noiseEn)
{
    float averSNR = (s.denEn0 * s.denEn1 * s.denEn2) / (noiseEn * noiseEn * noiseEn);
    if ((double)averSNR > 0.00001)
#ifdef SEA_LIBM_LOG
        averSNR = (float)((20 * log10((double)averSNR)) / 3.0);
#else /* log10(y) = ln(y) * log10(e), guarded (ns_aversnr_expr) */
        averSNR = ns_aversnr_expr<true>(uniform_f(averSNR));
#endif
    else
        averSNR = (float)(-100.0 / 3.0);
    averSNR = uniform_f(averSNR);
    const int nb = s.nbFrame[1];
    if (((double)(averSNR - s.lowSNRtrack) < 10.0) || (nb < 10)) {
        float lambdaSNR;
        if (nb < 10)
            lambdaSNR = (float)(1.0 - 1.0 / (double)(float)nb);
        else
            lambdaSNR = (averSNR < s.lowSNRtrack) ? (float)0.95 : (float)0.99;
        s.lowSNRtrack =
            (float)((double)s.lowSNRtrack + (1.0 - (double)lambdaSNR) * (double)(averSNR - s.lowSNRtrack));
    }
    if (s.denEn2 > 100.0f) {
        if ((double)averSNR < ((double)s.lowSNRtrack + 3.5)) {
            s.alfaGF = (float)((double)s.alfaGF + 0.15);
            if ((double)s.alfaGF > 0.8) s.alfaGF = (float)0.8;
        } else {
            s.alfaGF = (float)((double)s.alfaGF - 0.3);
            if ((double)s.alfaGF < 0.1) s.alfaGF = (float)0.1;
        }
    }
}

/* ---- VAD for frame dropping: the measures taken inside the first stage -----------
 * SpeechQVar / SpeechQSpec / SpeechQMel (NoiseSup.c:672-839) and the VADNS flag (:1359-1365); their
 * four bits per frame are what DoVADProc (VAD.c:219) votes over.  State is wave-uniform; every lane
 * computes it.  Promotions follow the reference: float state, double literals, rounding to float on
 * assignment; fc = the frame counter narrowed to int16. */
struct NsFd {
    float melMean, varMean, accTest, specMean, mel0, specValues, speechInVADQ;
};
__device__ __forceinline__ void fd_init(NsFd &d) { d.melMean = d.varMean = d.accTest = d.specMean = d.mel0 = d.specValues = d.speechInVADQ = 0.0f; }

/* SpeechQVar given the two in-order sums over the first N = NS_FFT_LENGTH / 4 Wiener gains (mean = sum W, var = sum W^2) */
template <int N>
__device__ __forceinline__ int fd_var_sums(NsFd &d, float mean, float var, int fc)
{
    const float specVar = (var / (float)N) - mean * mean / (float)(N * N);
    if (fc < 15) d.varMean = (d.varMean > specVar) ? d.varMean : specVar;
    if ((double)specVar < (double)d.varMean * 1.5 && (double)specVar > (double)d.varMean * 0.85)
        d.varMean = (float)((double)d.varMean * 0.8 + (double)specVar * 0.2);
    if ((double)specVar <= (double)d.varMean * 0.25) d.varMean = (float)((double)d.varMean * 0.97 + (double)specVar * 0.03);
    return ((double)specVar > (double)d.varMean * 1.65) ? 1 : 0;
}

/* SpeechQVar: variance of the first N Wiener gains W (in LDS), two in-order float sums */
template <int N = 64>
__device__ __forceinline__ int fd_var(NsFd &d, const float *W, int fc)
{
    float mean = 0.0f, var = 0.0f;
#pragma unroll 4
    for (int i = 0; i < N; i += 4) {
        const float4 w = *reinterpret_cast<const float4 *>(&W[i]);
        mean += w.x; var += w.x * w.x;
        mean += w.y; var += w.y * w.y;
        mean += w.z; var += w.z * w.z;
        mean += w.w; var += w.w * w.w;
    }
    return fd_var_sums<N>(d, uniform_f(mean), uniform_f(var), fc);
}

/* SpeechQSpec + SpeechQMel given the in-order sum of the 25 mel-filtered gains and gains 1..3; returns spec | mel << 1 */
__device__ __forceinline__ int fd_spec_mel_sums(NsFd &d, float tempEn, float m1, float m2, float m3, int fc)
{
    d.specValues = (float)((double)(tempEn * tempEn) - 3.0);
    if (fc == 1) d.specMean = d.specValues;
    if (fc < 15) {
        d.accTest = (float)(1.1 * (double)(d.accTest * (float)(fc - 1) + d.specValues) / (double)(float)fc);
        const float acceleration = d.specValues / d.accTest;
        if ((double)acceleration > 2.5) d.speechInVADQ = 1.0f;
        if (d.speechInVADQ == 0.0f) d.specMean = (d.specMean > d.specValues) ? d.specMean : d.specValues;
    }
    if ((double)d.specValues < (double)d.specMean * 1.5 && (double)d.specValues > (double)d.specMean * 0.75)
        d.specMean = (float)((double)d.specMean * 0.8 + (double)d.specValues * 0.2);
    if ((double)d.specValues <= (double)d.specMean * 0.5)
        d.specMean = (float)((double)d.specMean * 0.97 + (double)d.specValues * 0.03);
    const int spec = ((double)d.specValues > (double)d.specMean * 1.65) ? 1 : 0;

    const float mel1 = (float)((double)(float)(m1 + m2 + m3) / 3.0);
    const float smoothMel = (float)(0.75 * (double)mel1 + 0.25 * (double)d.mel0);
    d.mel0 = mel1;
    if (fc < 15) d.melMean = (d.melMean > smoothMel) ? d.melMean : smoothMel;
    if ((double)smoothMel < (double)d.melMean * 1.5 && (double)smoothMel > (double)d.melMean * 0.75)
        d.melMean = (float)((double)d.melMean * 0.8 + (double)smoothMel * 0.2);
    if ((double)smoothMel <= (double)d.melMean * 0.5) d.melMean = (float)((double)d.melMean * 0.97 + (double)smoothMel * 0.03);
    const int melf = ((double)smoothMel > (double)d.melMean * 3.25) ? 1 : 0;
    return spec | (melf << 1);
}

/* SpeechQSpec + SpeechQMel on the 25 mel-filtered gains (in LDS); returns spec | mel << 1 */
__device__ __forceinline__ int fd_spec_mel(NsFd &d, const float *mel, int fc)
{
    float tempEn = 0.0f;
#pragma unroll
    for (int i = 0; i < SEA_NMEL; ++i) tempEn += mel[i];
    return fd_spec_mel_sums(d, uniform_f(tempEn), mel[1], mel[2], mel[3], fc);
}

/* What ns_back leaves in fdRec for the DEFERRED evaluation of the speech measures (four-wave fd kernel: their three
 * in-order sums ride in free lanes of the helper wave's chain, their scalar logic runs in the transform wave):
 * [0..63] the first 64 Wiener gains, [64..127] their squares, [128..152] the 25 mel-filtered gains, [153..155] zeros,
 * ints [156] the frame counter narrowed to int16, [157] nbSpeechFrames > 4 */
constexpr int kFdRecFloats = 160;

/* FRONT half of a stage: analysis window on buf[60..259] (buf = 320-sample stage buffer, zero
 * padded to 256: NoiseSup.c:218-231), 256-point rfft, FFTtoPSD (129 power bins averaged pairwise to
 * 65: NoiseSup.c:249-270).  Depends on the buffer only -- no recursive state -- which is what lets
 * the pipelined kernel give it a wavefront of its own.  Writes psd[0..64]; ends with wave_sync(). */
__device__ __forceinline__ void ns_front(const float *buf, float *work, float *psd, const FftRegs &fft,
                                         const float (&win)[4], int lane)
{
    const float e0 = buf[60 + lane] * win[0];
    const float e1 = buf[124 + lane] * win[1];
    const float e2 = buf[188 + lane] * win[2];
    const float e3 = (lane < 8) ? buf[252 + lane] * win[3] : 0.0f;
    rfft256(e0, e1, e2, e3, work, fft, lane);
    const float re0 = work[2 * lane], re1 = work[2 * lane + 1];
    const float im1 = work[255 - 2 * lane];
    const float im0 = (lane > 0) ? work[256 - 2 * lane] : 0.0f;
    const float p0 = (lane > 0) ? (re0 * re0 + im0 * im0) : (re0 * re0);
    const float p1 = re1 * re1 + im1 * im1;
    psd[lane] = (p0 + p1) * 0.5f; /* == (float)((p0+p1)/2.0) */
    if (lane == 0) {
        const float ny = work[128];
        psd[64] = ny * ny;
    }
    wave_sync();
}

/* FFTtoPSD (NoiseSup.c:249-270) of one transformed frame of the dual transform -> psd[0..64]
 * (addresses from the tables: the work area is swizzled).  All five operands are fetched unconditionally (lane 0's
 * unused Im(0) address is element 0, the Nyquist bin is a broadcast read) and made opaque before the arithmetic:
 * written with conditional loads the compiler builds a branch -- and an LDS latency -- per operand. */
struct PsdOps {
    float re0, re1, im1, im0, ny;
};
__device__ __forceinline__ void psd_load(const float *work, const Fft2Regs &R, PsdOps &o)
{
    o.re0 = fft_at(work, R.psdA[0] & 0xffffu), o.re1 = fft_at(work, R.psdA[0] >> 16);
    o.im1 = fft_at(work, R.psdA[1] & 0xffffu), o.im0 = fft_at(work, R.psdA[1] >> 16);
    o.ny = fft_at(work, R.nyq);
}
__device__ __forceinline__ void psd_store(const PsdOps &o, float *psd, int lane)
{
    const float p0 = (lane > 0) ? (o.re0 * o.re0 + o.im0 * o.im0) : (o.re0 * o.re0);
    const float p1 = o.re1 * o.re1 + o.im1 * o.im1;
    psd[lane] = (p0 + p1) * 0.5f;
    if (lane == 0) psd[64] = o.ny * o.ny;
}
__device__ __forceinline__ void psd_from_fft2(const float *work, float *psd, const Fft2Regs &R, int lane)
{
    PsdOps o;
    psd_load(work, R, o);
    asm volatile("" : "+v"(o.re0), "+v"(o.re1), "+v"(o.im1), "+v"(o.im0), "+v"(o.ny));
    psd_store(o, psd, lane);
}

/* FFTtoPSD of BOTH transforms of a dual transform straight from the registers of its last level (sea_device.h,
 * rfft256_dual_keep_last): slot s = lane & 31 of a transform holds both parts of the bins s + 1, 65 + s, 63 - s, 127 - s (slot 31: 0, 64,
 * 32, 96 and 128), so a PSD value (P(2l) + P(2l+1)) / 2 is the sum of one kind of power of two NEIGHBOURING slots: the even slots take
 * them -- from the slot below (cyclic within the transform's 32 lanes: wave_shr:1, slots 0 repaired from lanes 31 / 63) for the bins
 * below 32 and from 64 to 95, from the slot above (the same quad) for the others.  Same products, same sums as psd_store; no store of
 * the level's results, no reads of the spectrum.  (NoiseSup.c:249-270) */
#ifndef SEA_PSD_REGS
#define SEA_PSD_REGS 1
#endif
__device__ __forceinline__ void psd_from_last_level(const float (&o)[8], const Fft2Regs &R, float *psdA, bool actA, float *psdB,
                                                    bool actB, int lane)
{
    const bool pl = R.kind[SEA_FFT_LSTAGES - 1] == SEA_BF_PAIR;
    const float ia = pl ? 0.0f : o[7], ib = pl ? o[3] : o[6], ic = pl ? o[7] : o[3], id = pl ? o[6] : o[2];
    const float Pa = o[0] * o[0] + ia * ia; /* bin s + 1 | 0: x * x + 0 * 0 == x * x */
    const float Pb = o[1] * o[1] + ib * ib; /* bin 65 + s | 64 */
    const float Pc = o[4] * o[4] + ic * ic; /* bin 63 - s | 32 */
    const float Pd = o[5] * o[5] + id * id; /* bin 127 - s | 96 */
    const float Pe = o[2] * o[2];           /* slot 31: bin 128 */
    auto below = [&](float v) { /* the same value of slot s - 1, slot 0: of slot 31 */
        const float sh = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138 /* wave_shr:1 */, 0xf, 0xf, false));
        const float l31 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 31));
        const float l63 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
        return lane == 0 ? l31 : (lane == 32 ? l63 : sh);
    };
    auto above = [&](float v) { /* of slot s + 1 (even s: the same quad) */
        return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xF5 /* quad_perm:[1,1,3,3] */, 0xf, 0xf, false));
    };
    const float qa = (below(Pa) + Pa) * 0.5f, qb = (below(Pb) + Pb) * 0.5f;
    const float qc = (above(Pc) + Pc) * 0.5f, qd = (above(Pd) + Pd) * 0.5f;
    const bool hi = lane >= 32;
    float *psd = hi ? psdB : psdA;
    const bool act = hi ? actB : actA;
    const int t2 = (lane & 31) >> 1;
    if (act && (lane & 1) == 0) {
        psd[t2] = qa;
        psd[32 + t2] = qb;
        psd[31 - t2] = qc;
        psd[63 - t2] = qd;
    }
    if (act && pl) psd[64] = Pe;
}

/* Two FRONT halves in one wave: frame A (stage buffer bufA) and frame B (bufB) are windowed,
 * transformed side by side (rfft256_dual) and reduced to their 65-bin PSDs.  actA / actB are
 * wave-uniform; an inactive side is fed zeros and its PSD is not written.  work: 512 floats. */
/* the lane's eight windowed input elements for rfft256_head8: element n0 + 32 * bitrev3(j) of its transform's
 * frame (lanes 0..31: bufA, 32..63: bufB; analysis window on buf[60..259], zero beyond element 199 -- literal
 * zeros, as the reference pads, not products with a zero weight).  An inactive side is fed zeros.
 * The eight samples are fetched unconditionally (60 + 255 < 320: always inside the stage buffer) and made opaque
 * before the selects, so that they travel as ONE batch of LDS reads instead of a branch and a latency each. */
__device__ __forceinline__ void ns_window8(const float *bufA, bool actA, const float *bufB, bool actB,
                                           const float (&win8)[8], int lane, float (&e)[8])
{
    const int n0 = lane & 31;
    const bool hiHalf = lane >= 32;
    const float *buf = hiHalf ? bufB : bufA;
    const bool act = hiHalf ? actB : actA;
    constexpr int kRev3[8] = {0, 4, 2, 6, 1, 5, 3, 7};
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = buf[60 + n0 + 32 * kRev3[j]];
    asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int idx = n0 + 32 * kRev3[j];
        const float p = v[j] * win8[j];
        e[j] = (act && idx < SEA_WIN) ? p : 0.0f;
    }
}

template <bool ADDR_LDS>
__device__ __forceinline__ void ns_front_dual(const float *bufA, bool actA, float *psdA, const float *bufB,
                                              bool actB, float *psdB, float *work, const Fft2Regs &fft,
                                              const float (&win8)[8], int lane)
{
    float e[8];
    ns_window8(bufA, actA, bufB, actB, win8, lane, e);
#ifndef SEA_BIG_LAT
#define SEA_BIG_LAT 0 /* the table-in-LDS (large-batch, issue-bound) form keeps the throughput transform: 484 vs 468 M frames/s on the configs[4] shard */
#endif
    if (SEA_PSD_REGS) { /* the last level feeds the PSDs from registers */
        float o[8];
        if (!ADDR_LDS || SEA_BIG_LAT) { /* the latency form: levels chained */
            if (SEA_FFT_HEAD16 && ADDR_LDS) {
                rfft256_head16(e, work, fft);
                wave_sync();
                fft2_levels_keep_last<2, 5, ADDR_LDS>(work, fft, e, o);
            } else {
                rfft256_head8(e, work, fft);
                wave_sync();
                fft2_levels_keep_last<1, 5, ADDR_LDS>(work, fft, e, o);
            }
        } else {
            rfft256_dual_lo<ADDR_LDS>(e, work, fft);
            rfft256_dual_hi_keep_last<ADDR_LDS>(work, fft, o);
        }
        psd_from_last_level(o, fft, psdA, actA, psdB, actB, lane);
        wave_sync();
        return;
    }
    rfft256_dual<ADDR_LDS, (!ADDR_LDS || SEA_BIG_LAT)>(e, work, fft);
    /* both PSDs' operands in one batch of reads */
    PsdOps a, b;
    psd_load(work, fft, a);
    psd_load(work + 256, fft, b);
    asm volatile("" : "+v"(a.re0), "+v"(a.re1), "+v"(a.im1), "+v"(a.im0), "+v"(a.ny), "+v"(b.re0), "+v"(b.re1), "+v"(b.im1),
                 "+v"(b.im0), "+v"(b.ny));
    if (actA) psd_store(a, psdA, lane);
    if (actB) psd_store(b, psdB, lane);
    wave_sync();
}

/* DoMelFB (MelProc.c:82-104): 25 bands over the Wiener gains in B.wbuf, taps in order; lane = band */
__device__ __forceinline__ float ns_mel_fb(const BackLds &B, const NsConst &C, int lane)
{
    float melOut = 0.0f;
    if (lane < SEA_NMEL) {
#pragma unroll
        for (int i = 0; i < SEA_MEL_TAPS; ++i) {
            /* taps past the band's length carry weight 0 in the table: melOut + W * 0 == melOut exactly (W is
             * a finite gain, melOut >= +0), which keeps a select out of the dependent chain */
            const int idx = C.melStart + i;
            melOut = melOut + B.wbuf[idx < 65 ? idx : 64] * C.melW[i];
        }
    }
    return melOut;
}

/* DoMelIDCT rows 0..8 (MelProc.c:357-378), mirror + Hanning(17) (NoiseSup.c:660-669), then ApplyWF:
 * the 17-tap FIR over buf[80..159] with 8 samples of context either side (NoiseSup.c:324-340);
 * lanes 0..39 produce two outputs each into dst.  Ends with wave_sync().
 * LDSBASIS: the 9x25 basis sits in LDS ([f][16], lane = row) instead of 25 VGPRs per lane. */
template <bool LDSBASIS, bool RL = false>
__device__ __forceinline__ void ns_idct_taps(float melOut, BackLds &B, const NsConst &C, int lane, const float *idctLds,
                                             float *firOut = nullptr /* where the 17 taps go; default B.fir */)
{
    float *fir = firOut ? firOut : B.fir;
    /* RL (the latency-bound kernel forms): band f's gain sits in lane f and reaches the nine row lanes through
     * v_readlane (a scalar operand of the multiply) instead of an LDS store, a fence and seven broadcast reads.
     * The issue-bound forms keep the LDS route: 25 lane reads are 25 more vector instructions. */
    float mf[SEA_NMEL];
    if (RL) {
#pragma unroll
        for (int f = 0; f < SEA_NMEL; ++f) mf[f] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(melOut), f));
    } else {
        if (lane < SEA_NMEL) B.mel[lane] = melOut;
        wave_sync();
    }
    if (lane <= 8) {
        float h = 0.0f;
        if (RL) {
#pragma unroll
            for (int f = 0; f < SEA_NMEL; ++f) h += mf[f] * (LDSBASIS ? idctLds[f * 16 + lane] : C.idct[f]);
        } else {
#pragma unroll
            for (int f4 = 0; f4 < 24; f4 += 4) {
                const float4 m = *reinterpret_cast<const float4 *>(&B.mel[f4]);
                h += m.x * (LDSBASIS ? idctLds[(f4 + 0) * 16 + lane] : C.idct[f4]);
                h += m.y * (LDSBASIS ? idctLds[(f4 + 1) * 16 + lane] : C.idct[f4 + 1]);
                h += m.z * (LDSBASIS ? idctLds[(f4 + 2) * 16 + lane] : C.idct[f4 + 2]);
                h += m.w * (LDSBASIS ? idctLds[(f4 + 3) * 16 + lane] : C.idct[f4 + 3]);
            }
            h += B.mel[24] * (LDSBASIS ? idctLds[24 * 16 + lane] : C.idct[24]);
        }
        const float tap = h * C.irWin;
        fir[8 + lane] = tap;
        fir[8 - lane] = tap;
    }
    wave_sync();
}

/* DoMelIDCT split between two waves (the in-order sum h[t] = sum_f W[f] basis[f][t], MelProc.c:357-378, is one chain
 * of 25 dependent additions): the producer adds terms 0 .. K-1 and hands over the partial sums of rows 0..8 and the
 * remaining gains; the consumer continues with terms K .. 24 in the same order, then windows and mirrors the taps.
 * rec layout: [0..24] mel gains (only K.. are read), [28..36] partial sums of rows 0..8. */
/* bregs (optional): the lane's basis column in registers (bregs[f] = basis[f][min(lane, 8)]) instead of 25 LDS reads per frame */
template <int K>
__device__ __forceinline__ void ns_idct_head(float melOut, float *rec, int lane, const float *idctLds, const float *bregs = nullptr)
{
    float h = 0.0f;
    const int l = (lane <= 8) ? lane : 8;
#pragma unroll
    for (int f = 0; f < K; ++f)
        h += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(melOut), f)) * (bregs ? bregs[f] : idctLds[f * 16 + l]);
    if (lane <= 8) rec[28 + lane] = h;
    if (lane >= K && lane < SEA_NMEL) rec[lane] = melOut;
    wave_sync();
}
template <int K>
__device__ __forceinline__ void ns_idct_tail(const float *rec, const float *idctLds, float irWin, float *fir, int lane)
{
    const int l = (lane <= 8) ? lane : 8; /* every lane computes (no divergence), rows 0..8 store */
    float m[SEA_NMEL], b[SEA_NMEL];
#pragma unroll
    for (int f = K; f < SEA_NMEL; ++f) {
        m[f] = rec[f];
        b[f] = idctLds[f * 16 + l];
    }
    float h = rec[28 + l];
#pragma unroll
    for (int f = K; f < SEA_NMEL; ++f) h += m[f] * b[f];
    const float tap = h * irWin;
    if (lane <= 8) {
        fir[8 + lane] = tap;
        fir[8 - lane] = tap;
    }
    wave_sync();
}

/* DoMelIDCT rows 0..8 + mirror + Hanning(17) for a wave that did NOT compute the mel gains: mel[0..24] in LDS (28
 * floats, 16-byte aligned), basis idctLds[f][16], irWin = this lane's window weight; taps to fir[0..16].  Same
 * operations in the same order as ns_idct_taps.  Ends with wave_sync(). */
__device__ __forceinline__ void ns_idct_taps_from(const float *mel, const float *idctLds, float irWin, float *fir, int lane)
{
    float m[SEA_NMEL];
#pragma unroll
    for (int f4 = 0; f4 < 24; f4 += 4) {
        const float4 v = *reinterpret_cast<const float4 *>(&mel[f4]);
        m[f4] = v.x, m[f4 + 1] = v.y, m[f4 + 2] = v.z, m[f4 + 3] = v.w;
    }
    m[24] = mel[24];
    const int l = (lane <= 8) ? lane : 8; /* every lane computes (no divergence), rows 0..8 store */
    float h = 0.0f;
#pragma unroll
    for (int f = 0; f < SEA_NMEL; ++f) h += m[f] * idctLds[f * 16 + l];
    const float tap = h * irWin;
    if (lane <= 8) {
        fir[8 + lane] = tap;
        fir[8 - lane] = tap;
    }
    wave_sync();
}

/* ApplyWF (NoiseSup.c:324-340): the 17 taps fir[0..16] (wave-uniform: broadcast LDS reads) over
 * buf[80..159] with 8 samples of context either side; lanes 0..39 produce two outputs each into dst.
 * Ends with wave_sync(). */
__device__ __forceinline__ void ns_fir_apply(const float *fir, const float *buf, float *dst, int lane)
{
    float c[SEA_NTAP];
#pragma unroll
    for (int k4 = 0; k4 < 16; k4 += 4) {
        const float4 v = *reinterpret_cast<const float4 *>(&fir[k4]);
        c[k4] = v.x;
        c[k4 + 1] = v.y;
        c[k4 + 2] = v.z;
        c[k4 + 3] = v.w;
    }
    c[16] = fir[16];
    if (lane < 40) {
        float x[18];
        const float *src = buf + 72 + 2 * lane; /* x[m] = buf[72 + 2l + m] */
#pragma unroll
        for (int m = 0; m < 18; m += 2) {
            const float2 v = *reinterpret_cast<const float2 *>(src + m);
            x[m] = v.x;
            x[m + 1] = v.y;
        }
        float y0 = 0.0f, y1 = 0.0f;
        /* out[i] = sum_{j=-8..8} fir[j+8] * buf[80+i-j]; i = 2l -> buf index 72+2l+(8-j) */
#pragma unroll
        for (int k = 0; k < SEA_NTAP; ++k) {
            y0 += c[k] * x[16 - k];
            y1 += c[k] * x[17 - k];
        }
        *reinterpret_cast<float2 *>(dst + 2 * lane) = make_float2(y0, y1);
    }
    wave_sync();
}

/* ApplyWF for the helper wave: the same 17-tap sums as ns_fir_apply, left in registers (lanes 0..39: outputs 2l and
 * 2l + 1; other lanes: unspecified), together with what the DC-offset filter needs next -- the differences
 * d[n] = y[n] - y[n-1] (NoiseSup.c:190-194; y[-1] = lastIn, the previous frame's last filter output) taken across
 * lanes with a one-lane DPP shift instead of a trip through LDS.  Returns y[79] (the next lastIn) in all lanes. */
__device__ __forceinline__ float ns_fir_dif(const float *fir, const float *buf, int lane, float lastIn, float &d0, float &d1)
{
    float c[SEA_NTAP];
#pragma unroll
    for (int k4 = 0; k4 < 16; k4 += 4) {
        const float4 v = *reinterpret_cast<const float4 *>(&fir[k4]);
        c[k4] = v.x;
        c[k4 + 1] = v.y;
        c[k4 + 2] = v.z;
        c[k4 + 3] = v.w;
    }
    c[16] = fir[16];
    float x[18];
    const int l = (lane < 40) ? lane : 39; /* every lane computes (no divergence); lanes >= 40 repeat lane 39 */
    const float *src = buf + 72 + 2 * l;   /* x[m] = buf[72 + 2l + m] */
#pragma unroll
    for (int m = 0; m < 18; m += 2) {
        const float2 v = *reinterpret_cast<const float2 *>(src + m);
        x[m] = v.x;
        x[m + 1] = v.y;
    }
    float y0 = 0.0f, y1 = 0.0f;
#pragma unroll
    for (int k = 0; k < SEA_NTAP; ++k) {
        y0 += c[k] * x[16 - k];
        y1 += c[k] * x[17 - k];
    }
    /* y1 of the lane below; lane 0 keeps `old` = lastIn (wave_shr:1 has no source lane for it) */
    const float below = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(lastIn), __float_as_int(y1), 0x138 /* wave_shr:1 */,
                                                                   0xf, 0xf, false));
    d0 = y0 - below;
    d1 = y1 - y0;
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(y1), 39));
}

/* ---- the same taps and filter without their trips through LDS (round 4; the latency-bound four-wave form) ----------
 * An LDS round trip (store, fence, broadcast reads) costs a role ~200 clk of its chain.  The nine distinct taps sit one per
 * lane (lane k: tap[8 - k] = tap[8 + k]) and reach the multiplies as scalar operands through v_readlane; the filter's two
 * outputs per lane stay in registers for whoever consumes them.  Same operations in the same order. */
struct FirTaps {
    float t[9];
};
__device__ __forceinline__ FirTaps fir_taps_rl(float tap)
{
    FirTaps T;
#pragma unroll
    for (int k = 0; k < 9; ++k) T.t[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(tap), k));
    return T;
}
/* ApplyWF (NoiseSup.c:324-340) as ns_fir_apply computes it: lanes 0..39 get outputs 2l and 2l + 1 (lanes >= 40 repeat lane 39) */
__device__ __forceinline__ void ns_fir_regs(const FirTaps &T, const float *buf, int lane, float &y0, float &y1)
{
    float x[18];
    const int l = (lane < 40) ? lane : 39;
    const float *src = buf + 72 + 2 * l; /* x[m] = buf[72 + 2l + m] */
#pragma unroll
    for (int m = 0; m < 18; m += 2) {
        const float2 v = *reinterpret_cast<const float2 *>(src + m);
        x[m] = v.x;
        x[m + 1] = v.y;
    }
    y0 = 0.0f, y1 = 0.0f;
#pragma unroll
    for (int k = 0; k < SEA_NTAP; ++k) {
        const float c = T.t[k < 8 ? 8 - k : k - 8];
        y0 += c * x[16 - k];
        y1 += c * x[17 - k];
    }
}
/* DoMelIDCT rows 0..8 + Hanning(17) as ns_idct_taps<.., RL = true> computes them, the tap left in lane k (k = 0..8) */
__device__ __forceinline__ float ns_idct_tap_rl(float melOut, float irWin, int lane, const float *idctLds, const float *bregs = nullptr)
{
    const int l = (lane <= 8) ? lane : 8;
    float h = 0.0f;
#pragma unroll
    for (int f = 0; f < SEA_NMEL; ++f)
        h += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(melOut), f)) * (bregs ? bregs[f] : idctLds[f * 16 + l]);
    return h * irWin;
}
/* ns_idct_tail with the tap left in lane k instead of the 17 taps in LDS */
template <int K>
__device__ __forceinline__ float ns_idct_tail_rl(const float *rec, const float *idctLds, float irWin, int lane)
{
    const int l = (lane <= 8) ? lane : 8;
    float m[SEA_NMEL], b[SEA_NMEL];
#pragma unroll
    for (int f = K; f < SEA_NMEL; ++f) {
        m[f] = rec[f];
        b[f] = idctLds[f * 16 + l];
    }
    float h = rec[28 + l];
#pragma unroll
    for (int f = K; f < SEA_NMEL; ++f) h += m[f] * b[f];
    return h * irWin;
}
/* ns_fir_dif on register taps */
__device__ __forceinline__ float ns_fir_dif_rl(const FirTaps &T, const float *buf, int lane, float lastIn, float &d0, float &d1)
{
    float y0, y1;
    ns_fir_regs(T, buf, lane, y0, y1);
    const float below = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(lastIn), __float_as_int(y1), 0x138 /* wave_shr:1 */,
                                                                   0xf, 0xf, false));
    d0 = y0 - below;
    d1 = y1 - y0;
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(y1), 39));
}

template <bool LDSBASIS, bool RL = false>
__device__ __forceinline__ void ns_idct_fir(float melOut, BackLds &B, const NsConst &C, const float *buf,
                                            float *dst, int lane, const float *idctLds)
{
    ns_idct_taps<LDSBASIS, RL>(melOut, B, C, lane, idctLds);
    ns_fir_apply(B.fir, buf, dst, lane);
}

/* In-order sum over bins 0..64 of values that sit one per lane (bin 64 wave-uniform): each term arrives through
 * v_readlane as a scalar operand, no LDS staging.  (Every 16 terms the source is made to depend on the running sum, or
 * all 64 lane reads are hoisted and their SGPRs spill.)  A moving-accumulator form -- v_add_f32_dpp wave_shr:1, ONE
 * instruction per term instead of two -- was measured 3.5 % slower end to end: its dependent chain is longer. */
__device__ __forceinline__ float ns_lane_sum65(float vLo, float vHi)
{
    float total = 0.0f, src = vLo;
#pragma unroll
    for (int k = 0; k < 64; ++k) {
        if (k > 0 && (k & 15) == 0) asm("" : "+v"(src) : "v"(total));
        total += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(src), k));
    }
    return total + vHi;
}

/* part of the in-order sum of ns_lane_sum65: terms K0 .. K1-1 added to `total` */
template <int K0, int K1>
__device__ __forceinline__ float ns_lane_sum_part(float total, float vLo)
{
    float src = vLo;
#pragma unroll
    for (int k = K0; k < K1; ++k) {
        if (k > K0 && (k & 15) == 0) asm("" : "+v"(src) : "v"(total));
        total += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(src), k));
    }
    return total;
}

/* timing-only diagnostic (-DSEA_NS_TIMING -DSEA_NS_BACK_CK): shader clocks between checkpoints inside ns_back(),
 * accumulated by lane 0 of workgroup 0 in g_back_ck[ST * 8 + k]; read / reset through sea_debug_ns_back_ck().
 * Each checkpoint costs ~300 clk (s_memtime round trip), so the role totals of such a build are inflated. */
#ifdef SEA_NS_TIMING
__device__ unsigned long long g_back_ck[16];
#endif
#if defined(SEA_NS_TIMING) && defined(SEA_NS_BACK_CK)
#define NS_BACK_CK_START unsigned long long bt_ = clock64()
#define NS_BACK_CK(k) do { const unsigned long long c_ = clock64(); if (blockIdx.x == 0 && lane == 0) g_back_ck[ST * 8 + (k)] += c_ - bt_; bt_ = c_; } while (0)
#else
#define NS_BACK_CK_START
#define NS_BACK_CK(k)
#endif

/* BACK half of a stage (ST = 0 first, 1 second): everything recursive.  Consumes psd[0..64] and
 * the stage buffer (raw frame buf[80..159] for the VAD, buf[72..167] for the FIR), updates the
 * per-utterance state and deposits the 80 filtered samples in dst.  Ends with wave_sync().
 *
 * PIPE = true (pipelined kernel): the input-only / deferrable scalar chains run in a helper wave.
 *   ST 0: the VAD frame log-energy arrives in frameEnExt; the 65 denSigSE1 values go to spectOut
 *         (summed later by the helper), the denEn registers are not touched.
 *   ST 1: the caller has loaded s.denEn0..2.
 * DEFER_FIR: stop after the IDCT and deposit the 17 filter taps in dst[0..16]; the FIR itself is then
 *   run by the consumer wave (ns_fir_apply), which has cycles to spare.
 * IDCT_HEAD = K >= 0 (with DEFER_FIR): add only the first K terms of the IDCT's sums and deposit the partial sums and
 *   the remaining mel gains in dst (ns_idct_head); another wave finishes them into the taps (ns_idct_tail). */
template <int ST, bool PIPE, bool FD = false, bool DEFER_FIR = false, bool RL = false, int IDCT_HEAD = -1>
__device__ __forceinline__ void ns_back(const float *psd, const float *buf, BackLds &B, NsRegs &s,
                                        const NsConst &C, float *dst, int lane, float frameEnExt = 0.0f,
                                        float *spectOut = nullptr, const float *idctLds = nullptr,
                                        NsFd *fd = nullptr, int *fdFlags = nullptr, float *fdRec = nullptr,
                                        float *yRegs = nullptr /* RL, !DEFER_FIR: the filter's two outputs of this lane stay in
                                                                * yRegs[0..1] (ns_fir_regs), dst is not written */,
                                        const float *bregs = nullptr /* RL: the IDCT basis column in registers (ns_idct_head) */)
{
    NS_BACK_CK_START;
    const float nSigLo = psd[lane], nSigHi = psd[64];

    /* --- PSDMean over two frames (NoiseSup.c:289-303) --- */
    const float PLo = (s.prevLo[ST] + nSigLo) * 0.5f;
    const float PHi = (s.prevHi[ST] + nSigHi) * 0.5f;
    s.prevLo[ST] = nSigLo;
    s.prevHi[ST] = nSigHi;

    /* --- VAD (NoiseSup.c:359-430) --- */
    {
        int nb = s.nbFrame[ST];
        if (nb < 2147483647) nb++;
        s.nbFrame[ST] = nb;
    }
    if (ST == 0) {
        const float frameEn = PIPE ? frameEnExt : vad_frame_energy(vad_frame_sum(buf + 80, B.sq, lane));
        vad_update(s, frameEn);
    }

    /* --- FilterCalc (NoiseSup.c:449-563): bins 0..63 one per lane; bin 64 is computed by EVERY lane
     *     (same cost as one predicated lane, but branch-free, so the two independent chains
     *     interleave) --- */
    const int nb16 = (int)(short)s.nbFrame[ST];
    /* the fast-division domain (ns_div): this frame's and the previous frame's PSD in {0} u [2^-40, 2^48]
     * (so P is 0 or in [2^-41, 2^48], and den, left by the previous frame, 0 or in [2^-24, 2^24]) and the
     * noise magnitude in [2^-15, 2^28] (>= eps = 2^-14.4 by construction; each update keeps it below
     * max(noise, 1.05 sqrt(P))).  Wave-uniform; always true for int16 audio ((200 * 32768)^2 = 2^45.3). */
    const bool psdOk = __ballot(!(ns_psd_in_domain(nSigLo) && ns_psd_in_domain(nSigHi))) == 0ull;
    /* The noise range is an invariant of the fast path: a frame inside the domain has P <= 2^48, and its update leaves the noise in
     * [eps, max(noise, 1.05 sqrt(P))] (stage 0: a convex combination with sqrt(P), :531-546; stage 1: n2 * upd with upd <= 1 when
     * n2 >= P and n2 * upd < 1.1 P otherwise, :492-508), so after a fast frame the test is skipped (~10 vector instructions per stage
     * and frame); after anything else -- initial state, a reloaded state, a frame outside the domain -- it is made. */
    bool noiseOk = true;
    if (!SEA_NOISE_SAFE || !s.noiseSafe[ST])
        noiseOk = __ballot(!(s.noiseLo[ST] <= 0x1p28f && s.noiseHi[ST] <= 0x1p28f && s.noiseLo[ST] >= 0x1p-15f && s.noiseHi[ST] >= 0x1p-15f)) == 0ull;
    const bool fast = SEA_NS_FAST_DIV && psdOk && noiseOk && (s.psdOk[ST] != 0);
    s.noiseSafe[ST] = fast ? 1 : 0;
    s.psdOk[ST] = psdOk ? 1 : 0;
    float WLo, WHi;
    auto lane_sum = [&](float vLo, float vHi) { return ns_lane_sum65(vLo, vHi); };
#ifndef SEA_NS_NO_OVERLAP
    if constexpr (ST == 1 && PIPE && RL && DEFER_FIR && !FD) {
        /* The second stage of the four-wave form, fast-division domain: two chains that do not depend on each other
         * until the gain factor is applied --
         *   A  gains of the 65 bins -> LDS -> mel filter bank              (FilterCalc :522-560, DoMelFB)
         *   B  in-order sum of the 65 noise magnitudes -> DoGainFact's scalars with their log10   (:600-637)
         * written so that each half of B sits in the same basic block as a piece of A (the wave_sync between the gain
         * store and the mel reads is a scheduling barrier): the instruction scheduler interleaves them, and the
         * dependent-instruction latency of one chain hides behind the other's issue slots.  Same operations, same
         * order within each sum. */
        if (fast) {
            noise_track1<true>(PLo, s.noiseLo[1], nb16, C.eps);
            noise_track1<true>(PHi, s.noiseHi[1], nb16, C.eps);
            float total = ns_lane_sum_part<0, 32>(0.0f, s.noiseLo[1]);
            {
                const ns_v2f nSig = {SEA_SQRT(nSigLo), SEA_SQRT(nSigHi)};
                const ns_v2f P = {SEA_SQRT(PLo), SEA_SQRT(PHi)};
                ns_v2f den = {s.denLo[1], s.denHi[1]};
                const ns_v2f W = gain_bin2(P, nSig, ns_v2f{s.noiseLo[1], s.noiseHi[1]}, den);
                s.denLo[1] = den.x;
                s.denHi[1] = den.y;
                B.wbuf[lane] = W.x;
                if (lane == 0) B.wbuf[64] = W.y;
            }
            wave_sync();
            total = ns_lane_sum_part<32, 64>(total, s.noiseLo[1]) + s.noiseHi[1];
            gain_fact_update(s, total);
            float melOut = ns_mel_fb(B, C, lane);
            melOut = (float)((double)(s.alfaGF * melOut) + (1.0 - (double)s.alfaGF) * 1.0);
            if (IDCT_HEAD >= 0)
                ns_idct_head<(IDCT_HEAD >= 0 ? IDCT_HEAD : 0)>(melOut, dst, lane, idctLds, bregs);
            else
                ns_idct_taps<PIPE, RL>(melOut, B, C, lane, idctLds, dst); /* the 17 taps straight into the consumer's record */
            return;
        }
    }
#endif
    if (fast) {
        if (SEA_NS_PAIR_BINS && RL) { /* not in the 80-VGPR form: the pairs cost registers there (9 spills, -4 %) */
            filter_bins_fast<ST>(PLo, PHi, nSigLo, nSigHi, s.noiseLo[ST], s.noiseHi[ST], s.denLo[ST], s.denHi[ST],
                                 nb16, s.flagVAD, C.eps, WLo, WHi);
        } else if (SEA_NS_STEADY && (ST == 0 || nb16 >= 11)) { /* the two chains in one basic block (filter_steady) */
            WLo = filter_steady<ST>(PLo, nSigLo, s.noiseLo[ST], s.denLo[ST], nb16, s.flagVAD, C.eps);
            WHi = filter_steady<ST>(PHi, nSigHi, s.noiseHi[ST], s.denHi[ST], nb16, s.flagVAD, C.eps);
        } else {
            WLo = filter_bin<ST, true>(PLo, nSigLo, s.noiseLo[ST], s.denLo[ST], nb16, s.flagVAD, C.eps);
            WHi = filter_bin<ST, true>(PHi, nSigHi, s.noiseHi[ST], s.denHi[ST], nb16, s.flagVAD, C.eps);
        }
    } else {
        WLo = filter_bin<ST>(PLo, nSigLo, s.noiseLo[ST], s.denLo[ST], nb16, s.flagVAD, C.eps);
        WHi = filter_bin<ST>(PHi, nSigHi, s.noiseHi[ST], s.denHi[ST], nb16, s.flagVAD, C.eps);
    }
    NS_BACK_CK(0); /* PSD mean, VAD update, FilterCalc of 65 bins */
    B.wbuf[lane] = WLo;
    if (lane == 0) B.wbuf[64] = WHi;
    if (FD && ST == 0 && fdRec) { /* deferred speech measures (kFdRecFloats) */
        fdRec[lane] = WLo;
        fdRec[64 + lane] = WLo * WLo;
    }
    if (PIPE && ST == 0) { /* the helper wave sums denSigSE1 */
        spectOut[lane] = s.denLo[0];
        if (lane == 0) spectOut[64] = s.denHi[0];
    } else if (!RL) {
        B.sbuf[lane] = (ST == 0) ? s.denLo[0] : s.noiseLo[1];
        if (lane == 0) B.sbuf[64] = (ST == 0) ? s.denHi[0] : s.noiseHi[1];
    }
    wave_sync();

    int fdBits = 0;
    if (FD && ST == 0 && !fdRec) fdBits = fd_var(*fd, B.wbuf, nb16); /* NoiseSup.c:1255-1258, before DoMelFB */

    NS_BACK_CK(1); /* gains staged in LDS */
    float melOut = ns_mel_fb(B, C, lane);
    NS_BACK_CK(2); /* mel filter bank */

    if (FD && ST == 0 && fdRec) {
        if (lane < SEA_NMEL) fdRec[128 + lane] = melOut;
        if (lane == 0) {
            reinterpret_cast<int *>(fdRec)[156] = nb16;
            reinterpret_cast<int *>(fdRec)[157] = (s.nbSpeech > 4) ? 1 : 0;
        }
    } else if (FD && ST == 0) { /* NoiseSup.c:1268-1281, on the mel-filtered gains; VADNS :1359-1365 */
        if (lane < SEA_NMEL) B.mel[lane] = melOut;
        wave_sync();
        fdBits |= fd_spec_mel(*fd, B.mel, nb16) << 1;
        fdBits |= (s.nbSpeech > 4) ? 8 : 0;
        *fdFlags = fdBits; /* bit 0 Var, 1 Spec, 2 Mel, 3 VADNS */
        wave_sync();
    }

    /* --- DoGainFact (NoiseSup.c:581-642) --- */
    if (!(PIPE && ST == 0)) {
        /* the in-order sum over bins 0..64 (NoiseSup.c:597-601): through lane reads in the latency-bound
         * kernel forms (RL: -230 clk per frame on the second-stage wave), from the staged LDS copy in quads in the
         * issue-bound ones, where 64 lane reads are 64 more vector instructions */
        const float total = RL ? lane_sum((ST == 0) ? s.denLo[0] : s.noiseLo[1], (ST == 0) ? s.denHi[0] : s.noiseHi[1])
                               : serial_sum<65>(B.sbuf, 0.0f);
        if (ST == 0) {
            s.denEn0 = s.denEn1;
            s.denEn1 = s.denEn2;
            s.denEn2 = total;
        } else {
            gain_fact_update(s, total);
            melOut = (float)((double)(s.alfaGF * melOut) + (1.0 - (double)s.alfaGF) * 1.0);
        }
    }
    NS_BACK_CK(3); /* in-order sum, gain factor */
    if (DEFER_FIR && IDCT_HEAD >= 0) {
        ns_idct_head<(IDCT_HEAD >= 0 ? IDCT_HEAD : 0)>(melOut, dst, lane, idctLds, bregs);
    } else if (DEFER_FIR) { /* the consumer wave applies the filter (ns_fir_apply): hand over the 17 taps */
        ns_idct_taps<PIPE, RL>(melOut, B, C, lane, idctLds);
        if (lane < SEA_NTAP) dst[lane] = B.fir[lane];
        wave_sync();
    } else if (RL && PIPE && yRegs) {
        ns_fir_regs(fir_taps_rl(ns_idct_tap_rl(melOut, C.irWin, lane, idctLds, bregs)), buf, lane, yRegs[0], yRegs[1]);
    } else {
        ns_idct_fir<PIPE, RL>(melOut, B, C, buf, dst, lane, idctLds);
    }
    NS_BACK_CK(4); /* IDCT taps (+ FIR unless deferred) */
}

/* ---- the second-stage BACK half cut in two (six-wave kernel, ns_pipe6_kernel.hip) -------------------
 * ns_noise1 (wave N1): PSDMean, the non-VAD noise tracking of all 65 bins, the in-order sum of the
 *   noise spectrum and the gain-factor scalars (NoiseSup.c:289-303, :486-517, :600-637).  None of it
 *   depends on the Wiener gains, so it runs one frame ahead of ns_gain1.  Writes P[0..64] (mean PSD),
 *   noise[0..64] and returns alfaGF; the caller has loaded s.denEn0..2.
 * ns_gain1 (wave G1): the gains of all bins from (P, PSD, noise), mel filter bank, gain
 *   factorisation of the 25 mel gains, IDCT, FIR (:522-560, MelProc.c, :639-640, :324-340). */
__device__ __forceinline__ float ns_noise1(const float *psd, float *Pout, float *noiseOut, NsRegs &s, float eps,
                                           int lane)
{
    const float nSigLo = psd[lane], nSigHi = psd[64];
    const float PLo = (s.prevLo[1] + nSigLo) * 0.5f;
    const float PHi = (s.prevHi[1] + nSigHi) * 0.5f;
    s.prevLo[1] = nSigLo;
    s.prevHi[1] = nSigHi;
    {
        int nb = s.nbFrame[1];
        if (nb < 2147483647) nb++;
        s.nbFrame[1] = nb;
    }
    const int nb16 = (int)(short)s.nbFrame[1];
    /* the fast-division domain exactly as in ns_back (this wave sees every frame's PSD and owns the noise) */
    const bool psdOk = __ballot(!(ns_psd_in_domain(nSigLo) && ns_psd_in_domain(nSigHi))) == 0ull;
    const bool noiseOk = __ballot(!(s.noiseLo[1] <= 0x1p28f && s.noiseHi[1] <= 0x1p28f)) == 0ull;
    const bool fast = SEA_NS_FAST_DIV && psdOk && noiseOk && (s.psdOk[1] != 0);
    s.psdOk[1] = psdOk ? 1 : 0;
    if (fast) {
        noise_track1<true>(PLo, s.noiseLo[1], nb16, eps);
        noise_track1<true>(PHi, s.noiseHi[1], nb16, eps);
    } else {
        noise_track1<false>(PLo, s.noiseLo[1], nb16, eps);
        noise_track1<false>(PHi, s.noiseHi[1], nb16, eps);
    }
    Pout[lane] = PLo;
    noiseOut[lane] = s.noiseLo[1];
    if (lane == 0) {
        Pout[64] = PHi;
        noiseOut[64] = s.noiseHi[1];
    }
    wave_sync();
    const float total = ns_lane_sum65(s.noiseLo[1], s.noiseHi[1]); /* in-order sum of the noise spectrum */
    gain_fact_update(s, total);
    return s.alfaGF;
}

__device__ __forceinline__ void ns_gain1(const float *psd, const float *P, const float *noise, float alfaGF,
                                         const float *buf, BackLds &B, NsRegs &s, const NsConst &C, float *dst,
                                         int lane, const float *idctLds)
{
    /* the fast-division domain as in ns_back: this wave sees every frame's PSD too and owns den; the noise
     * magnitudes arrive from the N1 wave (>= eps by its clamp) */
    const float nSigLo = psd[lane], nSigHi = psd[64], nzLo = noise[lane], nzHi = noise[64];
    const bool psdOk = __ballot(!(ns_psd_in_domain(nSigLo) && ns_psd_in_domain(nSigHi))) == 0ull;
    const bool noiseOk = __ballot(!(nzLo <= 0x1p29f && nzHi <= 0x1p29f)) == 0ull;
    const bool fast = SEA_NS_FAST_DIV && psdOk && noiseOk && (s.psdOk[1] != 0);
    s.psdOk[1] = psdOk ? 1 : 0;
    float WLo, WHi;
    if (fast && SEA_P6_G1_PK) { /* as filter_bins_fast: the lean square root (valid on the domain), the pair (lane, 64) through the packed form */
        const ns_v2f nSig = {SEA_SQRT(nSigLo), SEA_SQRT(nSigHi)};
        const ns_v2f Pq = {SEA_SQRT(P[lane]), SEA_SQRT(P[64])};
        ns_v2f den = {s.denLo[1], s.denHi[1]};
        const ns_v2f W = gain_bin2(Pq, nSig, ns_v2f{nzLo, nzHi}, den);
        s.denLo[1] = den.x;
        s.denHi[1] = den.y;
        WLo = W.x;
        WHi = W.y;
    } else if (fast) {
        WLo = gain_bin<true>(sqrtf(P[lane]), sqrtf(nSigLo), nzLo, s.denLo[1]);
        WHi = gain_bin<true>(sqrtf(P[64]), sqrtf(nSigHi), nzHi, s.denHi[1]);
    } else {
        WLo = gain_bin<false>(sqrtf(P[lane]), sqrtf(nSigLo), nzLo, s.denLo[1]);
        WHi = gain_bin<false>(sqrtf(P[64]), sqrtf(nSigHi), nzHi, s.denHi[1]);
    }
    B.wbuf[lane] = WLo;
    if (lane == 0) B.wbuf[64] = WHi;
    wave_sync();
    float melOut = ns_mel_fb(B, C, lane);
    melOut = (float)((double)(alfaGF * melOut) + (1.0 - (double)alfaGF) * 1.0);
    ns_idct_fir<true, true>(melOut, B, C, buf, dst, lane, idctLds);
}

/* ns_gain1 for the six-wave forms since round 4: the taps as scalar operands, the filter's outputs in registers, and what
 * leaves the wave is the DC-offset filter's input differences d[n] = y[n] - y[n-1] (ns_fir_dif_rl) instead of y itself: the helper
 * wave's chain starts from them without a pass of its own over the frame.  lastIn = y[79] of the previous filtered frame (this
 * wave's state now; prevSamples, NoiseSup.c:908). */
__device__ __forceinline__ void ns_gain1_dif(const float *psd, const float *P, const float *noise, float alfaGF,
                                             const float *buf, BackLds &B, NsRegs &s, const NsConst &C, float *dif,
                                             int lane, const float *idctLds, float &lastIn)
{
    const float nSigLo = psd[lane], nSigHi = psd[64], nzLo = noise[lane], nzHi = noise[64];
    const bool psdOk = __ballot(!(ns_psd_in_domain(nSigLo) && ns_psd_in_domain(nSigHi))) == 0ull;
    const bool noiseOk = __ballot(!(nzLo <= 0x1p29f && nzHi <= 0x1p29f)) == 0ull;
    const bool fast = SEA_NS_FAST_DIV && psdOk && noiseOk && (s.psdOk[1] != 0);
    s.psdOk[1] = psdOk ? 1 : 0;
    float WLo, WHi;
    if (fast && SEA_P6_G1_PK) { /* as filter_bins_fast: the lean square root (valid on the domain), the pair (lane, 64) through the packed form */
        const ns_v2f nSig = {SEA_SQRT(nSigLo), SEA_SQRT(nSigHi)};
        const ns_v2f Pq = {SEA_SQRT(P[lane]), SEA_SQRT(P[64])};
        ns_v2f den = {s.denLo[1], s.denHi[1]};
        const ns_v2f W = gain_bin2(Pq, nSig, ns_v2f{nzLo, nzHi}, den);
        s.denLo[1] = den.x;
        s.denHi[1] = den.y;
        WLo = W.x;
        WHi = W.y;
    } else if (fast) {
        WLo = gain_bin<true>(sqrtf(P[lane]), sqrtf(nSigLo), nzLo, s.denLo[1]);
        WHi = gain_bin<true>(sqrtf(P[64]), sqrtf(nSigHi), nzHi, s.denHi[1]);
    } else {
        WLo = gain_bin<false>(sqrtf(P[lane]), sqrtf(nSigLo), nzLo, s.denLo[1]);
        WHi = gain_bin<false>(sqrtf(P[64]), sqrtf(nSigHi), nzHi, s.denHi[1]);
    }
    B.wbuf[lane] = WLo;
    if (lane == 0) B.wbuf[64] = WHi;
    wave_sync();
    float melOut = ns_mel_fb(B, C, lane);
    melOut = (float)((double)(alfaGF * melOut) + (1.0 - (double)alfaGF) * 1.0);
    float d0, d1;
    lastIn = ns_fir_dif_rl(fir_taps_rl(ns_idct_tap_rl(melOut, C.irWin, lane, idctLds)), buf, lane, lastIn, d0, d1);
    if (lane < 40) *reinterpret_cast<float2 *>(dif + 2 * lane) = make_float2(d0, d1);
    wave_sync();
}

/* One whole stage on the single-wave form: stage 0 deposits its 80 output samples in
 * ring[1][240..319], stage 1 in outb[0..79]. */
template <int ST, bool FD = false>
__device__ __forceinline__ void ns_stage(NsLds &L, NsRegs &s, const NsConst &C, int lane, NsFd *fd = nullptr,
                                         int *fdFlags = nullptr)
{
    ns_front(L.ring[ST], L.work, L.psd, C.fft, C.win, lane);
    ns_back<ST, false, FD>(L.psd, L.ring[ST], L.back, s, C, (ST == 0) ? (L.ring[1] + 240) : L.outb, lane, 0.0f, nullptr,
                           nullptr, fd, fdFlags);
}

/* DCOffsetFil over one frame (NoiseSup.c:182-198): y[n] = float( double(d[n]) + 0.9990234375 *
 * double(y[n-1]) ), d[n] = x[n] - x[n-1] already in dif[0..79]; writes y to out[0..79], updates yState.
 * An 80-step serial recurrence (all lanes compute it redundantly).
 *
 * The reference rounds twice per sample (sum to double, then to float).  Whenever the exact value
 * d + c*y (c = 1023/1024, so c*y has at most 34 significant bits) fits a double exactly, that
 * equals ONE rounding to float, which a float FMA delivers with a 3x shorter dependency chain.
 * "Fits exactly" holds when the exponents of d and y are within [-16, +26] of each other (or
 * either is 0); that is verified for all 80 samples in parallel afterwards, and the frame is
 * recomputed on the exact double path in the (never yet observed) case that a sample fails. */
__device__ __forceinline__ bool dc_filter(const float *dif, float *out, float &yState, int lane)
{
    const float y0 = yState;
    float y = y0;
#ifdef SEA_ABLATE_DC
    for (int n = 0; n < 4; n += 4) {
#else
#pragma unroll 5
    for (int n = 0; n < SEA_HOP; n += 4) {
#endif
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
    wave_sync();
    bool unsafe = false;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int n = lane + 64 * k;
        if (n < SEA_HOP) {
            const float ad = fabsf(dif[n]);
            const float ay = fabsf(n == 0 ? y0 : out[n - 1]);
            const bool ok = (ad == 0.0f) || (ay == 0.0f) ||
                            (ay >= ad * 0x1p-16f && ay <= ad * 0x1p26f && ad < 0x1p100f && ad > 0x1p-100f);
            unsafe |= !ok;
        }
    }
    const bool redo = __ballot(unsafe) != 0ull;
    if (redo) { /* exact path: double multiply-add, rounded to float per sample */
        wave_sync();
        y = y0;
        for (int n = 0; n < SEA_HOP; ++n) {
            y = (float)__fma_rn(0.9990234375, (double)y, (double)dif[n]);
            out[n] = y;
        }
    }
    yState = y;
    wave_sync();
    return redo;
}

/* The three in-order chains of the helper wave -- the VAD sum (64 + sum sq[0..79]), the denSigSE1 sum
 * (den[0..64]) and the DC-offset recurrence (dif[0..79] -> out[0..79], float-FMA form of dc_filter) --
 * advanced by ONE instruction stream, each in its own group of lanes: every step is
 *     acc = fma(m, acc, x[n])      m = 1 for the sums (RN(1*acc + x) == RN(acc + x)), 1023/1024 for DC
 * with per-lane source pointers (lanes 0-15 | 16-31 | 32-63).  A wave64 issues an instruction that depends on
 * the one before it every ~10 clk whatever the number of active lanes, so the three 65..80-step chains cost 80
 * dependent FMAs instead of 225 dependent operations.  den[65..67] must be zero (they are: the record is
 * cleared once and only [0..64] is ever written); zero4: four zero floats.  All three are always computed; the
 * caller discards what it does not need.  Ends with wave_sync().
 *
 * The 80 intermediate values of the DC chain are needed too (they are the output samples).  Anything issued
 * between two FMAs of the chain costs its full issue time (tools/fma_probe.hip: FMA + one independent
 * instruction = 14 clk per step, FMA alone 10), so the serial pass keeps only every fifth value -- lane 32 + j
 * captures y[5j - 1], the value segment j starts from, under a one-bit scalar mask -- and lanes 32..47 then
 * recompute their five outputs each, in parallel, with the same FMA on the same operands. */
/* the exactness condition of the float-FMA form of the DC recurrence for one step (see dc_filter) */
__device__ __forceinline__ bool dc_step_ok(float d, float yPrev)
{
    const float ad = fabsf(d), ay = fabsf(yPrev);
    return (ad == 0.0f) || (ay == 0.0f) || (ay >= ad * 0x1p-16f && ay <= ad * 0x1p26f && ad < 0x1p100f && ad > 0x1p-100f);
}

/* *unsafe (optional): set when some step of the DC chain fails dc_step_ok -- checked by the sixteen recomputing lanes
 * on the values they hold in registers (start value, five inputs, five outputs); the caller then redoes the frame on
 * the exact path (dc_redo_exact).  nullptr: the caller verifies through LDS (dc_verify). */
/* FDCH (four-wave fd kernel): three more sums in lanes 48..50, which otherwise repeat the DC chain -- the mean and the
 * sum of squares of the first 64 Wiener gains (SpeechQVar) and the sum of the 25 mel-filtered gains (SpeechQSpec) of
 * the frame whose record fdRec is (kFdRecFloats); returned in fdSums[0..2]. */
template <int CHUNKS, bool FDCH = false>
__device__ __forceinline__ void helper_chains(const float *sq, const float *den, const float *dif, float *out,
                                              const float *zero4, float &vadSum, float &denSum,
                                              float &y, int lane, bool *unsafe = nullptr, const float *fdRec = nullptr,
                                              float *fdSums = nullptr)
{
    const int g = lane >> 4;
    const bool fdLane = FDCH && lane >= 48 && lane <= 50;
    const float *src = (g == 0) ? sq : ((g == 1) ? den : dif);
    if (FDCH && fdLane) src = fdRec + 64 * (lane - 48);
    const float *tail = (g == 1) ? zero4 : src; /* the den chain runs out after 65 terms: x = 0 from n = 68 on */
    const int fdLim = FDCH ? ((lane == 50) ? 28 : (fdLane ? 64 : SEA_HOP)) : SEA_HOP; /* an fd chain's first quad of zeros */
    const float m = (g >= 2 && !fdLane) ? 0.9990234375f : 1.0f;
    float acc = (g == 0) ? 64.0f : ((g == 1 || fdLane) ? 0.0f : y);
    /* the 20 quads are requested in CHUNKS chunks (4: 2 x 20 VGPRs in flight; 10: 2 x 8, for the 80-VGPR kernel
     * forms), chunk c + 1 before the chain of chunk c starts (an LDS round trip is ~60 clk) */
    constexpr int kQ = SEA_HOP / 4 / CHUNKS;
    constexpr int kSeg = 5; /* 16 segments of 5 steps */
    float4 x[2][kQ];
    auto request = [&](int c, float4(&dstq)[kQ]) {
#pragma unroll
        for (int k = 0; k < kQ; ++k) {
            const int n = 4 * (c * kQ + k);
            const float *p = (n >= 68) ? tail + ((g == 1) ? 0 : n) : src + n;
            if (FDCH && n >= 28) p = (n >= fdLim) ? zero4 : p;
            dstq[k] = *reinterpret_cast<const float4 *>(p);
        }
    };
    /* the segment inputs of the recomputing lanes (stride 5 floats across 16 lanes: 16 different banks) */
    const int seg = (lane - 32) & 15;
    float d5[kSeg];
#pragma unroll
    for (int k = 0; k < kSeg; ++k) d5[k] = dif[kSeg * seg + k];
    float cap = y; /* lane 32 (segment 0) starts from the incoming state */
    auto step = [&](float xv, int n) {
        float next; /* volatile asm keeps FMA n ahead of the capture of y[n - 1], which reads the FMA's input */
        asm volatile("v_fma_f32 %0, %2, %1, %3" : "=&v"(next) : "v"(m), "v"(acc), "v"(xv));
        if (n > 0 && n % kSeg == 0) {
            const unsigned long long bit = 1ull << (32 + n / kSeg);
            asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(cap) : "v"(acc), "s"(bit));
        }
        acc = next;
    };
    request(0, x[0]);
#pragma unroll
    for (int c = 0; c < CHUNKS; ++c) {
        if (c + 1 < CHUNKS) request(c + 1, x[(c + 1) & 1]);
#pragma unroll
        for (int k = 0; k < kQ; ++k) {
            const float4 v = x[c & 1][k];
            const int n = 4 * (c * kQ + k);
            step(v.x, n);
            step(v.y, n + 1);
            step(v.z, n + 2);
            step(v.w, n + 3);
        }
        if (c + 1 < CHUNKS) __builtin_amdgcn_sched_barrier(0);
    }
    /* sixteen lanes redo their five steps from the captured start values (the inputs are made opaque here: nothing that
     * depends on them -- the exactness checks -- may be scheduled into the serial chain above, where every extra
     * instruction costs its full issue time) */
    asm volatile("" : "+v"(d5[0]), "+v"(d5[1]), "+v"(d5[2]), "+v"(d5[3]), "+v"(d5[4]), "+v"(cap));
    {
        float v = cap;
        bool bad = false;
#pragma unroll
        for (int k = 0; k < kSeg; ++k) {
            if (unsafe) bad |= !dc_step_ok(d5[k], v);
            v = __fmaf_rn(0.9990234375f, v, d5[k]);
            if (lane >= 32 && lane < 48) out[kSeg * seg + k] = v;
        }
        if (unsafe) *unsafe = __ballot(bad && lane >= 32 && lane < 48) != 0ull;
    }
    vadSum = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc), 0));
    denSum = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc), 16));
    y = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc), 32));
    if (FDCH && fdSums) {
        fdSums[0] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc), 48));
        fdSums[1] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc), 49));
        fdSums[2] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc), 50));
    }
    wave_sync();
}

/* the exact path of the DC recurrence over one frame (double multiply-add, rounded to float per sample), for the
 * frames on which the FMA form's exactness condition failed */
__device__ __forceinline__ void dc_redo_exact(const float *dif, float *out, float y0, float &y)
{
    wave_sync();
    y = y0;
    for (int n = 0; n < SEA_HOP; ++n) {
        y = (float)__fma_rn(0.9990234375, (double)y, (double)dif[n]);
        out[n] = y;
    }
    wave_sync();
}

/* second half of dc_filter(): the exactness guard of the float-FMA recurrence over one frame and,
 * if it fails, the exact double recomputation.  y0 = state before the frame, y = after (updated). */
__device__ __forceinline__ bool dc_verify(const float *dif, float *out, float y0, float &y, int lane)
{
    bool unsafe = false;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int n = lane + 64 * k;
        if (n < SEA_HOP) {
            const float ad = fabsf(dif[n]);
            const float ay = fabsf(n == 0 ? y0 : out[n - 1]);
            const bool ok = (ad == 0.0f) || (ay == 0.0f) ||
                            (ay >= ad * 0x1p-16f && ay <= ad * 0x1p26f && ad < 0x1p100f && ad > 0x1p-100f);
            unsafe |= !ok;
        }
    }
    const bool redo = __ballot(unsafe) != 0ull;
    if (redo) {
        wave_sync();
        y = y0;
        for (int n = 0; n < SEA_HOP; ++n) {
            y = (float)__fma_rn(0.9990234375, (double)y, (double)dif[n]);
            out[n] = y;
        }
    }
    wave_sync();
    return redo;
}

/* dc_verify with the output frame handed back in registers: lane l < 40 checks samples 2l and 2l + 1 and returns them
 * (what the int16 cast and the store want next), all its operands fetched in one batch of LDS reads -- the check and
 * the store used to be two round trips.  Lanes >= 40 repeat lane 39. */
__device__ __forceinline__ float2 dc_verify_take(const float *dif, float *out, float y0, float &y, int lane)
{
    const int l = (lane < 40) ? lane : 39;
    const float2 d = *reinterpret_cast<const float2 *>(dif + 2 * l);
    float2 v = *reinterpret_cast<const float2 *>(out + 2 * l);
    const float below = out[(l > 0) ? 2 * l - 1 : 0];
    const bool unsafe = !(dc_step_ok(d.x, (l == 0) ? y0 : below) && dc_step_ok(d.y, v.x));
    if (__ballot(unsafe) != 0ull) {
        wave_sync();
        y = y0;
        for (int n = 0; n < SEA_HOP; ++n) {
            y = (float)__fma_rn(0.9990234375, (double)y, (double)dif[n]);
            out[n] = y;
        }
        wave_sync();
        v = *reinterpret_cast<const float2 *>(out + 2 * l);
    }
    wave_sync();
    return v;
}

/* DoNoiseSup (NoiseSup.c:1061-1440) for one 80-sample frame.  Lanes 0..39 pass samples 2l and
 * 2l+1.  Returns true when L.outb[0..79] holds a DC-filtered output frame. */
template <bool FD = false>
__device__ __forceinline__ bool ns_tick(NsLds &L, NsRegs &s, const NsConst &C, int lane, float x0, float x1,
                                        NsFd *fd = nullptr, int *fdFlags = nullptr)
{
    if (lane < 40) *reinterpret_cast<float2 *>(&L.ring[0][240 + 2 * lane]) = make_float2(x0, x1);
    wave_sync();
    s.nIn1++;
    if (s.nIn1 - s.nIn2 > 2) { /* NoiseSup.c:1152 */
        ns_stage<0, FD>(L, s, C, lane, fd, fdFlags);
        s.nIn2++;
    }
    if (s.nIn2 - s.nOut2 > 2) { /* NoiseSup.c:1178 */
        ns_stage<1>(L, s, C, lane);
        s.nOut2++;
    }
    /* slide both buffers by one hop (NoiseSup.c:1372-1390) */
    {
        float4 r0, r1;
        if (lane < 60) {
            r0 = *reinterpret_cast<const float4 *>(&L.ring[0][80 + 4 * lane]);
            r1 = *reinterpret_cast<const float4 *>(&L.ring[1][80 + 4 * lane]);
        }
        wave_sync();
        if (lane < 60) {
            *reinterpret_cast<float4 *>(&L.ring[0][4 * lane]) = r0;
            *reinterpret_cast<float4 *>(&L.ring[1][4 * lane]) = r1;
        }
    }
    if (s.nOut2 <= 0) {
        wave_sync();
        return false;
    }
    /* DCOffsetFil (NoiseSup.c:182-198): differences in parallel, recurrence in dc_filter() */
    {
        const float xm1 = (lane == 0) ? s.dcX : L.outb[lane - 1];
        const float d0 = L.outb[lane] - xm1;
        float d1 = 0.0f;
        if (lane < 16) d1 = L.outb[64 + lane] - L.outb[63 + lane];
        s.dcX = L.outb[79];
        wave_sync();
        L.back.sq[lane] = d0;
        if (lane < 16) L.back.sq[64 + lane] = d1;
    }
    wave_sync();
    dc_filter(L.back.sq, L.outb, s.dcY, lane);
    return true;
}

__device__ __forceinline__ void load_ns_const(NsConst &C, const sea_ns_tables *t, int lane)
{
    load_fft_regs(C.fft, &t->fft, lane);
#pragma unroll
    for (int k = 0; k < 4; ++k) C.win[k] = t->win[k][lane];
    C.melStart = t->melStart[lane];
    C.melLen = t->melLen[lane];
#pragma unroll
    for (int i = 0; i < SEA_MEL_TAPS; ++i) C.melW[i] = t->melW[i][lane];
#pragma unroll
    for (int f = 0; f < SEA_NMEL; ++f) C.idct[f] = t->idct[f][lane];
    C.irWin = t->irWin[lane];
    C.eps = t->eps;
}

/* ---- state blob of a stream (sea_ns_stream_*): [640 ring][12 x 64 per-lane][32 scalars] ---- */
constexpr int kBlobLane = 2 * kRing, kBlobScal = kBlobLane + 12 * 64;

__device__ __forceinline__ void state_store(float *blob, const NsLds &L, const NsRegs &s, int lane,
                                            const NsFd *fd = nullptr, int fdBits = 0)
{
    for (int i = lane; i < 2 * kRing; i += kLanes) blob[i] = (&L.ring[0][0])[i];
    float *p = blob + kBlobLane + lane;
#pragma unroll
    for (int st = 0; st < 2; ++st) {
        p[(0 + st) * 64] = s.noiseLo[st];
        p[(2 + st) * 64] = s.noiseHi[st];
        p[(4 + st) * 64] = s.denLo[st];
        p[(6 + st) * 64] = s.denHi[st];
        p[(8 + st) * 64] = s.prevLo[st];
        p[(10 + st) * 64] = s.prevHi[st];
    }
    if (lane == 0) {
        float *q = blob + kBlobScal;
        int *qi = reinterpret_cast<int *>(q + 16);
        q[0] = s.dcX; q[1] = s.dcY; q[2] = s.denEn0; q[3] = s.denEn1; q[4] = s.denEn2;
        q[5] = s.lowSNRtrack; q[6] = s.alfaGF; q[7] = s.meanEn;
        if (fd) {
            q[8] = fd->melMean; q[9] = fd->varMean; q[10] = fd->accTest; q[11] = fd->specMean;
            q[12] = fd->mel0; q[13] = fd->specValues; q[14] = fd->speechInVADQ;
            qi[9] = fdBits;
        }
        qi[0] = s.nbFrame[0]; qi[1] = s.nbFrame[1]; qi[2] = s.flagVAD; qi[3] = s.hangOver;
        qi[4] = s.nbSpeech; qi[5] = s.nIn1; qi[6] = s.nIn2; qi[7] = s.nOut2; qi[8] = s.onset;
        qi[10] = (s.psdOk[0] ? 1 : 0) | (s.psdOk[1] ? 2 : 0);
    }
}

__device__ __forceinline__ void state_load(const float *blob, NsLds &L, NsRegs &s, int lane, NsFd *fd = nullptr,
                                           int *fdBits = nullptr)
{
    for (int i = lane; i < 2 * kRing; i += kLanes) (&L.ring[0][0])[i] = blob[i];
    const float *p = blob + kBlobLane + lane;
#pragma unroll
    for (int st = 0; st < 2; ++st) {
        s.noiseLo[st] = p[(0 + st) * 64];
        s.noiseHi[st] = p[(2 + st) * 64];
        s.denLo[st] = p[(4 + st) * 64];
        s.denHi[st] = p[(6 + st) * 64];
        s.prevLo[st] = p[(8 + st) * 64];
        s.prevHi[st] = p[(10 + st) * 64];
    }
    const float *q = blob + kBlobScal;
    const int *qi = reinterpret_cast<const int *>(q + 16);
    s.dcX = q[0]; s.dcY = q[1]; s.denEn0 = q[2]; s.denEn1 = q[3]; s.denEn2 = q[4];
    s.lowSNRtrack = q[5]; s.alfaGF = q[6]; s.meanEn = q[7];
    if (fd) {
        fd->melMean = q[8]; fd->varMean = q[9]; fd->accTest = q[10]; fd->specMean = q[11];
        fd->mel0 = q[12]; fd->specValues = q[13]; fd->speechInVADQ = q[14];
        if (fdBits) *fdBits = qi[9];
    }
    s.nbFrame[0] = qi[0]; s.nbFrame[1] = qi[1]; s.flagVAD = qi[2]; s.hangOver = qi[3];
    s.nbSpeech = qi[4]; s.nIn1 = qi[5]; s.nIn2 = qi[6]; s.nOut2 = qi[7]; s.onset = qi[8];
    s.psdOk[0] = qi[10] & 1; s.psdOk[1] = (qi[10] >> 1) & 1;
    s.noiseSafe[0] = s.noiseSafe[1] = 0; /* a reloaded state: the noise range is tested on the first frame (ns_back) */
}

} // namespace

} // namespace sea
