/*
 * oracle/resynth_oracle.c -- TEST INFRASTRUCTURE ONLY (see sea_oracle.h).
 *
 * Plain-C restatement of the Hu-Wang 64-channel gammatone analysis -> time-reverse -> re-filter ->
 * mask-weighted raised-cosine overlap-add -> channel-sum resynthesis:
 *   resyth_64sub_ori/cpp/extractwav.cpp:9-131   resynth()      (ratio mask)
 *   resyth_64sub_IBM/cpp/extractwav.cpp:97-99                  (binary mask: > 0.5 -> 1.0)
 *   extractwav.cpp:167-211  gammaToneFilter(),  :133-166 / :258-278 middle-ear table,
 *   HuWang.h:9-17,69-70 constants.  hairCell() is not restated: its output is dead.
 *
 * The reference is C++ built with g++ (float <cmath> overloads for exp/cos/sin of float
 * arguments, double pow); every promotion is spelled out here in C.
 *
 * PARITY UNPINNED.  The reference resynth cannot be compiled in this image (extractwav.cpp needs the
 * private asdk::CWave header; writing a stand-in header is not allowed), and the reference holds no fixture
 * or recorded output for it.  The only number available is the known answer SURVEY.md 8(c) recorded --
 * out[8000..8009] and the weighted checksum -2456454 on the seeded 48000-sample soft-mask case -- which came
 * from a survey-time build that used a stand-in Wave.h and therefore does not count as a pin; this file
 * reproduces it exactly (tests/test_oracle.py::test_survey_known_answer_resynth) and keeps it as a regression anchor.
 * The binary (IBM) branch, ora_haircell / ora_subband64 and the L/160 frame-count mode have no recorded
 * reference output at all.  Everything the GPU tests claim for the resynthesis half is "bit-identical to this
 * restatement", checked line by line against extractwav.cpp:9-211, not "to the reference".
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "sea_oracle.h"

#define PI_HW (3.1415926535897932384626433832795) /* HuWang.h:7 */
enum { NCHAN = 64, FS = 16000, WINDOW = FS / 50, OFFSET = FS / 100 };

/* BS3383 table 1, extractwav.cpp:133-166.  Written as double literals converted to float, as the
 * reference's assignments do (decimal -> double -> float). */
static const float ME_F[29] = {20.0,   25.0,   31.5,   40.0,   50.0,   63.0,   80.0,  100.0,
                               125.0,  160.0,  200.0,  250.0,  315.0,  400.0,  500.0, 630.0,
                               800.0,  1000.0, 1250.0, 1600.0, 2000.0, 2500.0, 3150.0, 4000.0,
                               5000.0, 6300.0, 8000.0, 10000.0, 12500.0};
static const float ME_AF[29] = {2.347, 2.190, 2.050, 1.879, 1.724, 1.579, 1.512, 1.466,
                                1.426, 1.394, 1.372, 1.344, 1.304, 1.256, 1.203, 1.135,
                                1.062, 1.000, 0.967, 0.943, 0.932, 0.933, 0.937, 0.952,
                                0.974, 1.027, 1.135, 1.266, 1.501};
static const float ME_BF[29] = {0.00561,  0.00527,  0.00481,  0.00404,  0.00383,  0.00286,
                                0.00259,  0.00257,  0.00256,  0.00255,  0.00254,  0.00248,
                                0.00229,  0.00201,  0.00162,  0.00111,  0.00052,  0.00000,
                                -0.00039, -0.00067, -0.00092, -0.00105, -0.00104, -0.00088,
                                -0.00055, 0.00000,  0.00089,  0.00211,  0.00488};
static const float ME_TF[29] = {74.3, 65.0, 56.3, 48.4, 41.7, 35.5, 29.8, 25.1, 20.7, 16.8,
                                13.8, 11.2, 8.9,  7.2,  6.0,  5.0,  4.4,  4.2,  3.7,  2.6,
                                1.0,  -1.2, -3.6, -3.9, -1.1, 6.6,  15.3, 16.4, 11.6};

/* extractwav.cpp:258-278 */
static float loudness_phons(float freq)
{
    int i = 0;
    float ratio, afy, bfy, tfy;
    while (ME_F[i] < freq) i++;
    ratio = (freq - ME_F[i - 1]) / (ME_F[i] - ME_F[i - 1]);
    afy = ME_AF[i - 1] + ratio * (ME_AF[i] - ME_AF[i - 1]);
    bfy = ME_BF[i - 1] + ratio * (ME_BF[i] - ME_BF[i - 1]);
    tfy = ME_TF[i - 1] + ratio * (ME_TF[i] - ME_TF[i - 1]);
    return (float)(4.2 + afy * (60.0 - tfy) / (1.0 + bfy * (60.0 - tfy)));
}

/* extractwav.cpp:41-54 */
void ora_resynth_channels(float *cf64, float *bw64, float *midEar64)
{
    float lowerERB = (float)(21.4 * log10(50 * 0.00437 + 1.0));
    float upperERB = (float)(21.4 * log10(8000 * 0.00437 + 1.0));
    float spaceERB = (upperERB - lowerERB) / (NCHAN - 1);
    int c;
    for (c = 0; c < NCHAN; c++) {
        float cf = (float)((pow(10, (lowerERB + c * spaceERB) / 21.4) - 1) / 0.00437);
        float phon = (float)(loudness_phons(cf) - 60.0);
        cf64[c] = cf;
        bw64[c] = (float)(24.7 * (cf * 0.00437 + 1.0) * 1.019);
        midEar64[c] = (float)pow(10, (double)(phon / 20));
    }
}

/* filter coefficients of one channel: extractwav.cpp:176-183 (g++: expf/cosf/sinf, double pow) */
static void gt_coeffs(float cf, float bw, float midEar, float *gain, float *f1, float *f2)
{
    float dt = 1 / (float)FS;
    float twoPiT = (float)(2 * PI_HW * dt);
    float z = expf(-twoPiT * bw);
    *gain = (float)(midEar * pow((double)(twoPiT * bw), 4.0) / 3.0);
    *f1 = cosf(cf * twoPiT) * z;
    *f2 = sinf(cf * twoPiT) * z;
}

/* extractwav.cpp:167-211 */
void ora_gammatone(const float *in, float *out, float cf, float bw, float midEar, long L)
{
    float gain, f1, f2, p[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0}, x[4], y[4];
    long n;
    int i;
    gt_coeffs(cf, bw, midEar, &gain, &f1, &f2);
    for (n = 0; n < L; n++) {
        out[n] = p[3] * gain;
        for (i = 0; i < 4; i++) {
            x[i] = f1 * p[i] - f2 * q[i];
            y[i] = f2 * p[i] + f1 * q[i];
        }
        p[0] = in[n] * f1 + x[0];
        q[0] = in[n] * f2 + y[0];
        p[1] = p[0] + x[1];
        q[1] = q[0] + y[1];
        p[2] = p[1] + x[1] + x[2];
        q[2] = q[1] + y[1] + y[2];
        p[3] = p[2] + x[1] + 2 * x[2] + x[3];
        q[3] = q[2] + y[1] + 2 * y[2] + y[3];
    }
}

int ora_resynth64(const short *in, long L, const float *mask, int F, int binary, short *out)
{
    float cf[NCHAN], bw[NCHAN], me[NCHAN];
    float *input, *g, *rev, *w, *acc;
    long n;
    int c, f;
    /* binary: bit 0 = ideal-binary-mask variant; bit 1 = the older driver's frame count
     * numFrame = L/160 (1dnn_resynth/extractwav.cpp:67; one more frame, whose falling half covers the
     * last hop) instead of (L-320)/160+1 -- PARITY UNPINNED for that mode (no recorded output) */
    const int alt = (binary >> 1) & 1;
    binary &= 1;
    if (alt ? (L < OFFSET || F != (int)(L / OFFSET)) : (L < WINDOW || F != (int)((L - WINDOW) / OFFSET + 1))) return 1;
    ora_resynth_channels(cf, bw, me);
    input = (float *)malloc(L * sizeof(float));
    g = (float *)malloc(L * sizeof(float));
    rev = (float *)malloc(L * sizeof(float));
    w = (float *)malloc(L * sizeof(float));
    acc = (float *)calloc(L, sizeof(float));
    for (n = 0; n < L; n++) input[n] = (float)in[n];

    for (c = 0; c < NCHAN; c++) {
        ora_gammatone(input, g, cf[c], bw[c], me[c], L);          /* :60-64 */
        for (n = 0; n < L; n++) rev[L - n - 1] = g[n] / me[c];    /* :86-87 */
        ora_gammatone(rev, g, cf[c], bw[c], me[c], L);            /* :88 */
        for (n = 0; n < L; n++) rev[L - n - 1] = g[n] / me[c];    /* :89-90 */
        for (n = 0; n < L; n++) w[n] = 0.0f;
        for (f = 0; f < F; f++) {                                 /* :93-107 */
            float m = mask[(long)f * NCHAN + c];
            if (binary ? (m > 0.5) : (m > 0)) {
                if (binary) m = (float)1.0;
                if (f > 0)
                    for (n = 0; n < OFFSET; n++)
                        w[(f - 1) * OFFSET + n] =
                            (float)(w[(f - 1) * OFFSET + n] + 0.5 * (1.0 + cos(n * PI_HW / (OFFSET) + PI_HW)) * m);
                for (n = OFFSET; n < WINDOW; n++)
                    w[(f - 1) * OFFSET + n] =
                        (float)(w[(f - 1) * OFFSET + n] + 0.5 * (1.0 + cos((n - OFFSET) * PI_HW / (OFFSET))) * m);
            }
        }
        for (n = 0; n < L; n++) acc[n] += w[n] * rev[n];          /* :108-112 */
    }
    for (n = 0; n < L; n++) {                                      /* :120-121, (short) cast */
        float v = acc[n];
        out[n] = (v > -2147483648.0f && v < 2147483648.0f) ? (short)(int)v : 0;
    }
    free(input);
    free(g);
    free(rev);
    free(w);
    free(acc);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * the analysis half on its own -- subbband(): gammatone + Meddis hair cell ->
 * 64 int16 streams (enhancement_extract_test/cpp/extractwav.cpp:40-101; hairCell
 * resyth_64sub_ori/cpp/extractwav.cpp:212-257; constants HuWang.h:36-44).
 * PARITY UNPINNED: the reference cannot be built here (asdk Wave.h) and no reference output of
 * this function was recorded; the restatement follows the C++ promotions literally (MED_* are
 * double literals, dt is float).
 * ---------------------------------------------------------------------------------------- */
void ora_haircell(const float *input, float *output, long L)
{
    const double Y = 5.05, G = 2000.0, Lc = 2500.0, R = 6580.0, X = 66.31, A = 3.0, B = 300.0, H = 48000.0, M = 1.0;
    const float dt = 1 / (float)FS;
    const float ymdt = (float)(Y * M * dt), xdt = (float)(X * dt), ydt = (float)(Y * dt);
    const float lplusrdt = (float)((Lc + R) * dt), rdt = (float)(R * dt), gdt = (float)(G * dt), hdt = (float)H;
    float kt = (float)(G * A / (A + B));
    float c = (float)(M * Y * kt / (Lc * kt + Y * (Lc + R)));
    float q = (float)(c * (Lc + R) / kt);
    float w = (float)(c * R / X);
    long n;
    for (n = 0; n < L; n++) {
        float replenish, eject, reuptakeandloss, reuptake, reprocess;
        kt = ((input[n] + A) > 0.0) ? (float)(gdt * (input[n] + A) / (input[n] + A + B)) : (float)0;
        replenish = (q < M) ? (ymdt - ydt * q) : 0;
        eject = kt * q;
        reuptakeandloss = lplusrdt * c;
        reuptake = rdt * c;
        reprocess = xdt * w;
        q = q + replenish - eject + reprocess;
        if (q < 0.0) q = 0.0;
        c = c + eject - reuptakeandloss;
        if (c < 0.0) c = 0.0;
        w = w + reuptake - reprocess;
        if (w < 0.0) w = 0.0;
        output[n] = hdt * c;
    }
}

/* out is [64][L] int16: channel c's stream at out + c*L */
int ora_subband64(const short *in, long L, short *out)
{
    float cf[NCHAN], bw[NCHAN], me[NCHAN];
    float *input, *g, *h;
    long n;
    int c;
    if (L <= 0) return 1;
    ora_resynth_channels(cf, bw, me);
    input = (float *)malloc(L * sizeof(float));
    g = (float *)malloc(L * sizeof(float));
    h = (float *)malloc(L * sizeof(float));
    for (n = 0; n < L; n++) input[n] = (float)in[n];
    for (c = 0; c < NCHAN; c++) {
        ora_gammatone(input, g, cf[c], bw[c], me[c], L);
        ora_haircell(g, h, L);
        for (n = 0; n < L; n++) {
            float v = h[n];
            out[(long)c * L + n] = (v > -2147483648.0f && v < 2147483648.0f) ? (short)(int)v : 0;
        }
    }
    free(input);
    free(g);
    free(h);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * SURVEY 8(f) rank 2 -- the ideal-ratio-mask TARGET of make_single_IBM
 * (enhancement_extract_test/cpp/show_IBM.cpp:105-169): for each of the 64 subband streams of the clean and
 * of the noise signal (the int16 outputs of subbband()), frames of WINDOW = 320 samples every OFFSET = 160,
 * a 512-point power spectrum per frame (asdk::SpecInfo with m_nFFT = 512, :137-144), its first 64 bins
 * summed in float in bin order (:154-158), IRM = sum_pure / (sum_pure + sum_noise) (:165).
 *
 * PARITY UNPINNED.  asdk::SpecInfo is third-party code that is neither in the reference tree nor in this
 * image (version directory asdk_20130701; call sites show_IBM.cpp:41-43, :137-144) and no reference test or
 * recorded output pins it, so what its GetSpecInfo does between "320 samples" and "power spectrum" -- the
 * analysis window, a possible pre-emphasis or scaling -- is not known.  Restated here with the window as a
 * parameter (0 rectangular, 1 Hamming 0.54 - 0.46 cos(2 pi n / 319), 2 Hanning 0.5 - 0.5 cos(2 pi n / 319);
 * 1 is the default: SpecInfo is the front half of the SDK's MFCC extractor), zero padding to 512, an
 * unnormalised |X|^2 (any constant factor cancels in the ratio).  The spectrum is evaluated as a direct DFT in
 * double and rounded to float per bin.  pure / noise: [64][pitch] int16, irm: [F][64], F = (L-320)/160+1.
 * ---------------------------------------------------------------------------------------- */
static double irm_window(int kind, int n)
{
    if (kind == 1) return 0.54 - 0.46 * cos(2.0 * PI_HW * n / (WINDOW - 1));
    if (kind == 2) return 0.5 - 0.5 * cos(2.0 * PI_HW * n / (WINDOW - 1));
    return 1.0;
}

int ora_irm_target(const short *pure, const short *noise, long L, long pitch, int window, float *irm)
{
    long F, i;
    int c, j, n;
    static double cs[512], sn[512];
    float w[WINDOW];
    if (L < WINDOW || pitch < L) return 1;
    F = (L - WINDOW) / OFFSET + 1;
    for (n = 0; n < 512; n++) {
        cs[n] = cos(2.0 * PI_HW * n / 512.0);
        sn[n] = sin(2.0 * PI_HW * n / 512.0);
    }
    for (n = 0; n < WINDOW; n++) w[n] = (float)irm_window(window, n);
    for (c = 0; c < NCHAN; c++)
        for (i = 0; i < F; i++) {
            float sum[2];
            int which;
            for (which = 0; which < 2; which++) {
                const short *x = (which ? noise : pure) + (long)c * pitch + i * OFFSET;
                float acc = 0.0f;
                for (j = 0; j < NCHAN; j++) { /* bins 0..63 of the 512-point spectrum */
                    double re = 0.0, im = 0.0;
                    for (n = 0; n < WINDOW; n++) {
                        const double v = (double)(w[n] * (float)x[n]);
                        re += v * cs[(j * n) & 511];
                        im -= v * sn[(j * n) & 511];
                    }
                    acc += (float)(re * re + im * im);
                }
                sum[which] = acc;
            }
            irm[i * NCHAN + c] = sum[0] / (sum[0] + sum[1]);
        }
    return 0;
}

