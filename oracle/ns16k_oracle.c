/*
 * oracle/ns16k_oracle.c -- TEST INFRASTRUCTURE ONLY (see sea_oracle.h).
 *
 * Plain-C restatement of the 16 k-native NoiseSup variant behind the reference's batch plug-in symbols
 * (SURVEY 8(f) #4):
 *     function/20141106_speech_enhancement/aurora_etsi/NoiseSup.cpp:912-1407   (= resyth_64sub_ori/cpp/NoiseSup.cpp)
 *     function/20141106_speech_enhancement/aurora_etsi/NoiseSup.h              hop 160, window 480, buffer 640,
 *                                                                              NS_FFT_LENGTH 512 with NS_FFT_ORDER 8,
 *                                                                              129 spectral values
 *     function/20141106_speech_enhancement/aurora_etsi/MelProc.cpp:66-76,119-135,269-341,464-514,556-576
 *                                                                              25 gammatone-shaped windows over the first
 *                                                                              128 gains + their inverse DCT
 *     function/20141106_speech_enhancement/aurora_etsi/rfft.cpp:46-181         called as rfft(x, 512, 8): the last
 *                                                                              split-radix level never runs
 *
 * PARITY.  rfft.cpp and MelProc.cpp compile on their own (g++, oracle/Makefile target `aurora`): the transform as the
 * variant calls it, the 25 windows, the IDCT basis, DoGamma and DoGammaIDCT below are checked bit for bit against that
 * build (tests/test_oracle.py).  NoiseSup.cpp itself needs aurora/aurora_include.h, which the reference tree does not
 * hold, and no output of it exists anywhere in the tree: the frame loop below is PARITY UNPINNED.  It follows the
 * text under C++ overload rules with X_FLOAT32 = float, X_INT16 = short, X_INT32 = int and PIx2 =
 * 6.28318530717958647692 (MelProc.cpp:25; the value NoiseSup.cpp takes from the absent header), with ONE stated
 * deviation: DoGainFact_IBM's `log10 (averSNR)` (:667) on a float selects the float overload in C++, whose value is
 * the C library's own (glibc's log10f is not correctly rounded, MSVC's -- the tree ships Visual Studio projects --
 * differs from it); here, as in the C tree the file was derived from (etsi/cpp/NoiseSup.c:611), it is the double
 * log10 rounded to float once, after the /3.0.
 * What C++ changes against the C tree and IS followed: cos / sin of a float in rfft.cpp are cosf / sinf.
 * The 25 gains func_Wiener prints per second-stage frame (`fprintf (fp, "%f ", W[i])`, :1321-1328) are returned as
 * floats.
 *
 * Build: gcc -O2 -ffp-contract=off (oracle/Makefile).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "sea_oracle.h"

#define PI2_D 6.28318530717958647692

enum { HOP = 160, WIN = 480, NFFT = 512, NSPEC = 129, NGAM = 25, GLEN = 128, NBUF = 640, DATAIN = 480, AWIN = 80, NTAP = 17, HALF = 8 };

/* The transform: ora16_rfft (ns_oracle.c) = the butterfly schedule of the etsi/ restatement with the twiddles of the C++
 * build; the variant calls it with (512, 8) -- the index patterns of a 512-point transform, eight of its nine levels. */

/* ---- tables -------------------------------------------------------------------------------------------------- */
typedef struct {
    int ready;
    float sigWindow[WIN], irWindow[NTAP];
    int gstart[NGAM];
    float gamma[NGAM][GLEN];
    float idct[NGAM][NGAM];
    float eps;
} tables16;
static tables16 T;

static float hz_to_erb(float hz) { return (float)(21.4 * log10(hz * 0.00437 + 1.0)); }         /* MelProc.cpp:505-508 */
static float erb_to_hz(float r) { return (float)((pow(10, r / 21.4) - 1) / 0.00437); }          /* :510-513 */

static void build_gamma(void)
{ /* InitGammawindows (First, 80.0, 16000.0f, 256, 25, 1): MelProc.cpp:269-341 */
    const float st = 80.0f, smpl = 16000.0f;
    const int nfft = 2 * (NSPEC - 1);
    float cf[NGAM], erb[NGAM];
    float lo = hz_to_erb(st), hi = hz_to_erb(smpl / 2);
    float step = (hi - lo) / (NGAM - 1);
    int i, j;
    for (i = 0; i < NGAM; i++) {
        cf[i] = erb_to_hz(lo + step * i);
        erb[i] = (float)(1.019 * 24.7 * (4.37 * cf[i] / 1000 + 1));
    }
    for (i = 0; i < NGAM; i++) {
        float norm = 0.0f;
        T.gstart[i] = (int)((int)cf[i] * nfft / smpl); /* (int)cf * FFTLength is an int product, / SmplFreq a float division */
        for (j = 0; j < GLEN; j++) {
            /* :306: a / erb * a / erb associates as ((a / erb) * a) / erb */
            T.gamma[i][j] = (float)(1.0 / pow((1.0 + (j * 1.0 / nfft * smpl - cf[i]) / erb[i] * (j * 1.0 / nfft * smpl - cf[i]) / erb[i]), 2));
            norm += T.gamma[i][j];
        }
        for (j = 0; j < GLEN; j++) T.gamma[i][j] /= norm;
    }
}

static void build_idct(void)
{ /* InitGammaIDCTbasis (basis, First, 25, 16000, 256): MelProc.cpp:464-503 */
    const int fs = 16000;
    float lin = fs / (float)(2 * (NSPEC - 1));
    float cf[NGAM], df[NGAM];
    int i, j;
    for (j = 0; j < NGAM; j++) cf[j] = T.gstart[j] * lin;
    for (j = 0; j < NGAM; j++) {
        if (j == 0)
            df[j] = (cf[1] - cf[0]) / fs;
        else if (j == NGAM - 1)
            df[j] = (cf[j] - cf[j - 1]) / fs;
        else
            df[j] = (cf[j + 1] - cf[j - 1]) / fs;
    }
    for (i = 0; i < NGAM; i++)
        for (j = 0; j < NGAM; j++) T.idct[i][j] = (float)(df[j] * cos(PI2_D * i * cf[j] / fs));
}

static void tables_init(void)
{
    int i;
    if (T.ready) return;
    for (i = 0; i < WIN; i++) /* NoiseSup.cpp:1034-1037 */
        T.sigWindow[i] = (float)(0.5 - 0.5 * cos((PI2_D * ((float)i + 0.5)) / (float)(short)WIN));
    for (i = 0; i < NTAP; i++) /* :1040-1043 */
        T.irWindow[i] = (float)(0.5 - 0.5 * cos((PI2_D * ((float)i + 0.5)) / (float)(short)NTAP));
    build_gamma();
    build_idct();
    T.eps = (float)exp(-10.0);
    T.ready = 1;
}

void ora16_tables(float *sigWindow480, float *irWindow17, int *gammaStart25, float *gamma25x128, float *idct25x25)
{
    tables_init();
    memcpy(sigWindow480, T.sigWindow, sizeof T.sigWindow);
    memcpy(irWindow17, T.irWindow, sizeof T.irWindow);
    memcpy(gammaStart25, T.gstart, sizeof T.gstart);
    memcpy(gamma25x128, T.gamma, sizeof T.gamma);
    memcpy(idct25x25, T.idct, sizeof T.idct);
}

/* DoGamma, MelProc.cpp:119-135: every window starts at gain 0 and is 128 long */
void ora16_do_gamma(float *W)
{
    float sum[NGAM];
    int i, j;
    tables_init();
    for (j = 0; j < NGAM; j++) {
        sum[j] = 0.0f;
        for (i = 0; i < GLEN; i++) sum[j] += W[i] * T.gamma[j][i];
    }
    for (j = NGAM - 1; j >= 0; j--) W[j] = sum[j];
}

/* DoGammaIDCT (W, basis, 25, 25), MelProc.cpp:556-576 */
void ora16_idct(float *W)
{
    float out[NGAM];
    int t, f;
    tables_init();
    for (t = 0; t < NGAM; t++) {
        out[t] = 0.0f;
        for (f = 0; f < NGAM; f++) out[t] += W[f] * T.idct[t][f];
    }
    for (t = 0; t < NGAM; t++) W[t] = out[t];
}

/* ---- state (NoiseSup.cpp:45-139) --------------------------------------------------------------------------------- */
struct ora16_state {
    float buf[2][NBUF];
    int nIn1, nIn2, nOut2;
    float nSig[2][NSPEC], noise[2][NSPEC], den[2][NSPEC];
    float psdPrev[2][NSPEC];
    float dcX, dcY;
    float denEn[3], lowSNRtrack, alfaGF;
    int nbFrame[2];
    short flagVAD, hangOver, nbSpeechFrames;
    float meanEn;
    float fdMelMean, fdVarMean, fdAccTest, fdSpecMean, fdMelValues[2], fdSpecValues, fdSpeechInVADQ;
    float tmp[NFFT]; /* nsTmp.tmpMem: the scratch whose aliasing the frame loop relies on */
};

ora16_state *ora16_new(void)
{ /* etsi_denoise_mapping_thread_init, :937-1083 */
    ora16_state *s;
    int i, st;
    tables_init();
    s = (ora16_state *)calloc(1, sizeof *s);
    if (!s) return NULL;
    s->alfaGF = (float)0.8;
    for (st = 0; st < 2; st++)
        for (i = 0; i < NSPEC; i++) s->noise[st][i] = T.eps;
    return s;
}

void ora16_free(ora16_state *s) { free(s); }

static void vad16(ora16_state *s, int st, const float *frame)
{ /* _VAD_, :350-421 */
    int nb = s->nbFrame[st], i;
    float frameEn, lambdaLTE, meanEn = s->meanEn;
    short flagVAD = s->flagVAD, hangOver = s->hangOver, nbSpeech = s->nbSpeechFrames;
    if (nb < 2147483647) nb++;
    s->nbFrame[st] = nb;
    if (st == 1) return;
    lambdaLTE = (nb < 10) ? 1 - 1 / (float)nb : (float)0.97;
    frameEn = 64.0f;
    for (i = 0; i < HOP; i++) frameEn += frame[i] * frame[i];
    frameEn = (float)(0.5 + (log(frameEn / 64.0) / log(2.0)) * 16.0);
    if (((frameEn - meanEn) < (short)20) || (nb < (short)10)) {
        if ((frameEn < meanEn) || (nb < (short)10))
            meanEn += (1 - lambdaLTE) * (frameEn - meanEn);
        else
            meanEn += (1 - (float)0.99) * (frameEn - meanEn);
        if (meanEn < (float)80.0) meanEn = (float)80.0;
    }
    if (nb > 4) {
        if ((frameEn - meanEn) > (short)15) {
            flagVAD = 1;
            nbSpeech++;
        } else {
            if (nbSpeech > (short)4) hangOver = 15;
            nbSpeech = 0;
            if (hangOver != 0) {
                hangOver--;
                flagVAD = 1;
            } else
                flagVAD = 0;
        }
    }
    s->meanEn = meanEn;
    s->flagVAD = flagVAD;
    s->hangOver = hangOver;
    s->nbSpeechFrames = nbSpeech;
}

static void filter_calc16(ora16_state *s, int st, float *P, float *W)
{ /* FilterCalc, :440-553; the frame counter is narrowed to X_INT16 there */
    float *nSig = s->nSig[st], *noise = s->noise[st], *den = s->den[st];
    short nb = (short)s->nbFrame[st];
    const float beta = (float)0.98, rsbMin = (float)0.079432823;
    float lambda;
    int i;
    if (st == 1) {
        for (i = 0; i < NSPEC; i++) noise[i] *= noise[i];
        if (nb < 11) {
            lambda = 1 - 1 / (float)nb;
            for (i = 0; i < NSPEC; i++) noise[i] = lambda * noise[i] + (1 - lambda) * P[i];
        } else {
            for (i = 0; i < NSPEC; i++) {
                float upd = (float)(0.9 + 0.1 * (P[i] / (P[i] + noise[i])) * (1.0 + 1.0 / (1.0 + 0.1 * (P[i] / noise[i]))));
                noise[i] *= upd;
            }
        }
        for (i = 0; i < NSPEC; i++) {
            noise[i] = sqrtf(noise[i]);
            if (noise[i] < T.eps) noise[i] = T.eps;
        }
    }
    for (i = 0; i < NSPEC; i++) {
        nSig[i] = sqrtf(nSig[i]);
        P[i] = sqrtf(P[i]);
    }
    if (st == 0) {
        lambda = (nb < (short)100) ? 1 - 1 / (float)nb : (float)0.99;
        if (s->flagVAD == 0)
            for (i = 0; i < NSPEC; i++) {
                noise[i] = lambda * noise[i] + (1 - lambda) * P[i];
                if (noise[i] < T.eps) noise[i] = T.eps;
            }
    }
    for (i = 0; i < NSPEC; i++) {
        float post = (P[i] / noise[i]) - 1;
        float prio = beta * (den[i] / noise[i]) + (1 - beta) * ((0 > post) ? 0 : post);
        W[i] = prio / (1 + prio);
        prio = W[i] * P[i] / noise[i];
        prio = (prio > rsbMin) ? prio : rsbMin;
        W[i] = prio / (1 + prio);
        den[i] = W[i] * nSig[i];
    }
}

static void gain_fact16(ora16_state *s, int st, float *W)
{ /* DoGainFact_IBM, :634-698 */
    int i;
    if (st == 0) {
        s->denEn[0] = s->denEn[1];
        s->denEn[1] = s->denEn[2];
        s->denEn[2] = 0.0f;
        for (i = 0; i < NSPEC; i++) s->denEn[2] += s->den[0][i];
    } else {
        float noiseEn = 0.0f, averSNR, lambdaSNR;
        for (i = 0; i < NSPEC; i++) noiseEn += s->noise[1][i];
        averSNR = (s->denEn[0] * s->denEn[1] * s->denEn[2]) / (noiseEn * noiseEn * noiseEn);
        if (averSNR > 0.00001)
            averSNR = (float)((20 * log10((double)averSNR)) / 3.0); /* the stated deviation: see the header */
        else
            averSNR = (float)(-100.0 / 3.0);
        if (((averSNR - s->lowSNRtrack) < 10.0) || (s->nbFrame[1] < (short)10)) {
            if (s->nbFrame[1] < (short)10)
                lambdaSNR = (float)(1.0 - 1.0 / (float)s->nbFrame[1]);
            else
                lambdaSNR = (averSNR < s->lowSNRtrack) ? (float)0.95 : (float)0.99;
            s->lowSNRtrack = (float)(s->lowSNRtrack + (1.0 - lambdaSNR) * (averSNR - s->lowSNRtrack));
        }
        if (s->denEn[2] > 100) {
            if (averSNR < (s->lowSNRtrack + 3.5)) {
                s->alfaGF = (float)(s->alfaGF + 0.15);
                if (s->alfaGF > 0.8) s->alfaGF = (float)0.8;
            } else {
                s->alfaGF = (float)(s->alfaGF - 0.3);
                if (s->alfaGF < 0.1) s->alfaGF = (float)0.1;
            }
        }
        for (i = 0; i < NGAM; i++) W[i] = (float)(s->alfaGF * W[i] + (1.0 - s->alfaGF) * 1.0);
    }
}

#define FMAXF(a, b) (((a) > (b)) ? (a) : (b))

static int speech_q_var16(ora16_state *s, const float *W)
{ /* SpeechQVar, :852-893: ssize = NS_FFT_LENGTH / 4 = 128 */
    const short ssize = NFFT / 4, fc = (short)s->nbFrame[0];
    float var = 0.0f, mean = 0.0f, specVar;
    int i;
    for (i = 0; i < ssize; i++) {
        mean += W[i];
        var += W[i] * W[i];
    }
    specVar = (var / ssize) - mean * mean / (ssize * ssize);
    if (fc < 15) s->fdVarMean = FMAXF(s->fdVarMean, specVar);
    if (specVar < s->fdVarMean * 1.5 && specVar > s->fdVarMean * 0.85) s->fdVarMean = (s->fdVarMean * 0.8 + specVar * 0.2);
    if (specVar <= s->fdVarMean * 0.25) s->fdVarMean = (s->fdVarMean * 0.97 + specVar * 0.03);
    return (specVar > s->fdVarMean * 1.65) ? 1 : 0;
}

static int speech_q_spec16(ora16_state *s)
{ /* SpeechQSpec, :742-783 */
    const short fc = (short)s->nbFrame[0];
    if (fc == 1) s->fdSpecMean = s->fdSpecValues;
    if (fc < 15) {
        float acceleration;
        s->fdAccTest = 1.1 * (s->fdAccTest * (float)(fc - 1) + s->fdSpecValues) / (float)(fc);
        acceleration = s->fdSpecValues / s->fdAccTest;
        if (acceleration > 2.5) s->fdSpeechInVADQ = 1;
        if (s->fdSpeechInVADQ == 0) s->fdSpecMean = FMAXF(s->fdSpecMean, s->fdSpecValues);
    }
    if (s->fdSpecValues < s->fdSpecMean * 1.5 && s->fdSpecValues > s->fdSpecMean * 0.75)
        s->fdSpecMean = (s->fdSpecMean * 0.8 + s->fdSpecValues * 0.2);
    if (s->fdSpecValues <= s->fdSpecMean * 0.5) s->fdSpecMean = (s->fdSpecMean * 0.97 + s->fdSpecValues * 0.03);
    return (s->fdSpecValues > s->fdSpecMean * 1.65) ? 1 : 0;
}

static int speech_q_mel16(ora16_state *s)
{ /* SpeechQMel, :799-836 */
    const short fc = (short)s->nbFrame[0];
    float smoothMel = 0.75 * s->fdMelValues[1] + 0.25 * s->fdMelValues[0];
    s->fdMelValues[0] = s->fdMelValues[1];
    if (fc < 15) s->fdMelMean = FMAXF(s->fdMelMean, smoothMel);
    if (smoothMel < s->fdMelMean * 1.5 && smoothMel > s->fdMelMean * 0.75) s->fdMelMean = (s->fdMelMean * 0.8 + smoothMel * 0.2);
    if (smoothMel <= s->fdMelMean * 0.5) s->fdMelMean = (s->fdMelMean * 0.97 + smoothMel * 0.03);
    return (smoothMel > s->fdMelMean * 3.25) ? 1 : 0;
}

/* etsi_denoise_mapping_func_Wiener, :1140-1407.  Entries of out / the flag arrays that the reference leaves untouched
 * are left untouched.  wiener (may be NULL) receives 25 floats per second-stage frame, *wiener_rows their count.
 * Returns the number of frames consumed (dataNum / 160). */
long ora16_push(ora16_state *s, const float *in, long dataNum, float *out, int *var, int *spec, int *mel, int *vadns,
                int *counter, float *wiener, long *wiener_rows)
{
    const long frames = dataNum / HOP;
    long n, rows = 0;
    for (n = 0; n < frames; n++, in += HOP, out += HOP) {
        float check = 0.0f;
        float *W = s->tmp + NSPEC, *filterIR = s->tmp, *signalIn = s->tmp, *signalOut = s->tmp + NSPEC, *P = s->tmp;
        int st, i, j;
        for (i = 0; i < HOP; i++) check += in[i] * in[i]; /* :1160-1171 */
        if (0 == (int)check) continue;
        for (i = 0; i < HOP; i++) s->buf[0][DATAIN + i] = in[i];
        s->nIn1++;
        for (st = 0; st < 2; st++) {
            float *buf = s->buf[st], *nSig = s->nSig[st];
            const float *cur = buf + HOP, *prv = buf;
            if (st == 0 && !((s->nIn1 - s->nIn2) > 2)) continue;
            if (st == 1 && !((s->nIn2 - s->nOut2) > 2)) continue;
            for (i = 0; i < WIN; i++) signalIn[i] = buf[AWIN + i];
            for (i = 0; i < WIN; i++) signalIn[i] = signalIn[i] * T.sigWindow[i]; /* DoSigWindowing, :209-222 */
            for (i = WIN; i < NFFT; i++) signalIn[i] = 0.0f;
            ora16_rfft(signalIn, NFFT, 8);
            /* FFTtoPSD (signalIn, nSigSE, 512), :240-261 */
            signalIn[0] = signalIn[0] * signalIn[0];
            for (i = 1, j = NFFT - 1; i < NFFT / 2; i++, j--) signalIn[i] = (signalIn[i] * signalIn[i] + signalIn[j] * signalIn[j]);
            signalIn[i] = signalIn[i] * signalIn[i];
            for (i = 0, j = 0; i < NSPEC - 1; i++, j += 2) nSig[i] = (float)((signalIn[j] + signalIn[j + 1]) / 2.0);
            nSig[i] = signalIn[j];
            /* PSDMean, :280-294 */
            for (i = 0; i < NSPEC; i++) {
                P[i] = (s->psdPrev[st][i] + nSig[i]) / (short)2;
                s->psdPrev[st][i] = nSig[i];
            }
            vad16(s, st, cur);
            filter_calc16(s, st, P, W);
            if (st == 0) var[n] = speech_q_var16(s, W);
            ora16_do_gamma(W);
            if (st == 0) { /* :1301-1311 */
                float tempEn = 0.0f;
                for (i = 0; i < NGAM; i++) tempEn += (float)W[i];
                s->fdSpecValues = tempEn * tempEn - 3.0;
                spec[n] = speech_q_spec16(s);
                s->fdMelValues[1] = (float)(W[1] + W[2] + W[3]) / 3.0;
                mel[n] = speech_q_mel16(s);
            }
            gain_fact16(s, st, W);
            if (st == 1) { /* :1319-1328 */
                if (wiener) memcpy(wiener + NGAM * rows, W, NGAM * sizeof(float));
                rows++;
            }
            ora16_idct(W);
            for (i = 1; i < NGAM; i++) W[2 * NGAM - 1 - i] = W[i];
            for (i = HALF, j = 0; i < NTAP; i++, j++) { /* DoFilterWindowing, :716-725 */
                filterIR[HALF + j] = W[j] * T.irWindow[i];
                filterIR[HALF - j] = filterIR[HALF + j];
            }
            for (i = 0; i < HOP; i++) { /* ApplyWF (cur, prv, filterIR, signalOut, 160, 8), :317-331 */
                signalOut[i] = 0.0f;
                for (j = -HALF; j <= (i <= HALF ? i : HALF); j++) signalOut[i] += (filterIR[j + HALF] * cur[i - j]);
                for (j = i + 1; j <= HALF; j++) signalOut[i] += (filterIR[j + HALF] * prv[HOP - j + i]);
            }
            if (st == 0) {
                for (i = 0; i < HOP; i++) s->buf[1][DATAIN + i] = signalOut[i];
                s->nIn2++;
                vadns[n] = (s->nbSpeechFrames > 4) ? 1 : 0;
            } else {
                for (i = 0; i < HOP; i++) out[i] = signalOut[i];
                s->nOut2++;
            }
            counter[n] = s->nbFrame[0];
        }
        if (s->nIn1) memmove(s->buf[0], s->buf[0] + HOP, DATAIN * sizeof(float));
        if (s->nIn2) memmove(s->buf[1], s->buf[1] + HOP, DATAIN * sizeof(float));
        if (s->nOut2 > 0) { /* DCOffsetFil (pOutData, prevSamples, 160), :168-184 */
            for (i = 0; i < HOP; i++) {
                float aux = out[i];
                out[i] = (float)(out[i] - s->dcX + 0.9990234375 * s->dcY);
                s->dcX = aux;
                s->dcY = out[i];
            }
        }
    }
    if (wiener_rows) *wiener_rows = rows;
    return frames;
}
