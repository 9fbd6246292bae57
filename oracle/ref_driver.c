/*
 * oracle/ref_driver.c -- TEST INFRASTRUCTURE ONLY (never shipped, never on the product path).
 *
 * Thin driver that is compiled TOGETHER WITH the reference's own C sources, where they lie
 * under /root/reference/etsi/cpp, into oracle/_ref/libetsi_ref.so (see oracle/Makefile).
 * Nothing from the reference is copied into this repository: this file only calls the
 * reference's public entry points and peeks at its state struct, which it can see because it
 * textually includes the reference translation unit NoiseSup.c at build time (include path
 * -I/root/reference/etsi/cpp).  Must be built as C, gnu dialect (F12), without FMA
 * contraction (F11).
 *
 * Exposed (all plain C ABI, used through ctypes by oracle/oracle.py):
 *   ref_etsi_denoise      -> etsi/cpp/AdvFrontEnd.c:125  etsi_denoise()
 *   ref_rfft              -> etsi/cpp/rfft.c:45          rfft()
 *   ref_ns_trace          -> drives DoAdvProcess (etsi/cpp/ParmInterface.c:208) frame by frame,
 *                            records the float denoised stream, the per-frame NoiseSup state and
 *                            runs DoCompCeps exactly as the commented-out block
 *                            etsi/cpp/ParmInterface.c:275-293 would.
 *   ref_ns_stream_f32     -> fe->DoNoiseSup (etsi/cpp/NoiseSup.c:1061) on float frames
 *   ref_afe_trace         -> the whole commented-out chain (WaveProc, CompCeps, PostProc, VAD, flush)
 */
#include "NoiseSup.c" /* the reference TU itself: gives access to struct NoiseSupStructX */

#include "AdvFrontEnd.h"
#include "BufferIn.h"
#include "CompCepsExports.h"

FILE *fp_denoised = NULL; /* ParmInterface.c:39 declares it extern; nothing uses it */

int ref_etsi_denoise(short *in, short *out, long n) { return etsi_denoise(in, out, n); }

void ref_rfft(float *x, int n, int m) { rfft(x, n, m); }

/* number of float scalars recorded per input frame in `scal` */
#define REF_TRACE_NSCAL 16

/*
 * in[n]           int16 samples
 * out_i16[n]      what etsi_denoise would write (frames of 80; caller pre-fills)
 * den_f32         float NoiseSup output stream, 80 per TRUE output (capacity (n/80)*80)
 * ceps            14 floats per cepstral frame (capacity (n/80)*14)
 * scal            REF_TRACE_NSCAL floats per input frame (capacity (n/80)*REF_TRACE_NSCAL), may be NULL
 * spec            4*65 floats per input frame: noiseSE1, noiseSE2, denSigSE1, denSigSE2; may be NULL
 * counts[0]       number of TRUE NoiseSup outputs, counts[1] number of cepstral frames
 */
long ref_ns_trace(const short *in, long n, short *out_i16, float *den_f32, float *ceps,
                  float *scal, float *spec, long *counts)
{
    FEParamsX *fe = AdvProcessAlloc(8000);
    short sig[80], den[80];
    float feat[NUM_CEP_COEFF + 2];
    float frameBuf[FRAME_BUF_SIZE + HP16k_MEL_USED];
    long nfr = n / 80, f, nout = 0, nceps = 0;
    int i;

    fe->Noc0 = 0;
    AdvProcessInit(fe);
    memset(den, 0, sizeof den);

    for (f = 0; f < nfr; f++) {
        NoiseSupStructX *NSX = fe->NSX;
        long before = NSX->nsVar.buffers.nbFramesOutSecondStage;
        for (i = 0; i < 80; i++) sig[i] = in[f * 80 + i];
        /* return value deliberately ignored: SURVEY F4 */
        DoAdvProcess(sig, den, feat, fe);
        for (i = 0; i < 80; i++) out_i16[f * 80 + i] = den[i];

        if (NSX->nsVar.buffers.nbFramesOutSecondStage > before) {
            BufInGetLast(fe->denoisedBuf, frameBuf, FRAME_BUF_SIZE);
            for (i = 0; i < 80; i++) den_f32[nout * 80 + i] = frameBuf[FRAME_BUF_SIZE - 80 + i];
            nout++;
            if (nout >= 3) { /* offsetDenoisedFrame: -200 -> -120 -> -40 -> +40 */
                fe->DoCompCeps(frameBuf + 1, feat, fe);
                for (i = 0; i < 14; i++) ceps[nceps * 14 + i] = feat[i];
                nceps++;
            }
        }
        if (scal) {
            float *s = scal + f * REF_TRACE_NSCAL;
            s[0] = (float)NSX->nsVar.vadNS.nbFrame[0];
            s[1] = (float)NSX->nsVar.vadNS.nbFrame[1];
            s[2] = (float)NSX->nsVar.vadNS.flagVAD;
            s[3] = (float)NSX->nsVar.vadNS.hangOver;
            s[4] = (float)NSX->nsVar.vadNS.nbSpeechFrames;
            s[5] = NSX->nsVar.vadNS.meanEn;
            s[6] = NSX->nsVar.gainFact.alfaGF;
            s[7] = NSX->nsVar.gainFact.lowSNRtrack;
            s[8] = NSX->nsVar.gainFact.denEn1[0];
            s[9] = NSX->nsVar.gainFact.denEn1[1];
            s[10] = NSX->nsVar.gainFact.denEn1[2];
            s[11] = NSX->nsVar.prevSamples.lastSampleIn;
            s[12] = NSX->nsVar.prevSamples.lastDCOut;
            s[13] = (float)NSX->nsVar.buffers.nbFramesInFirstStage;
            s[14] = (float)NSX->nsVar.buffers.nbFramesInSecondStage;
            s[15] = (float)NSX->nsVar.buffers.nbFramesOutSecondStage;
        }
        if (spec) {
            float *p = spec + f * 4 * 65;
            memcpy(p, NSX->nsVar.spectrum.noiseSE1, 65 * sizeof(float));
            memcpy(p + 65, NSX->nsVar.spectrum.noiseSE2, 65 * sizeof(float));
            memcpy(p + 130, NSX->nsVar.spectrum.denSigSE1, 65 * sizeof(float));
            memcpy(p + 195, NSX->nsVar.spectrum.denSigSE2, 65 * sizeof(float));
        }
    }
    AdvProcessDelete(&fe);
    counts[0] = nout;
    counts[1] = nceps;
    return nfr;
}

/* fe->DoNoiseSup on float frames (the frame-level plugin slot, ParmInterface.h:120-178), no gate */
long ref_ns_stream_f32(const float *in, long nframes, float *out, int *produced)
{
    FEParamsX *fe = AdvProcessAlloc(8000);
    long f, nout = 0;
    int i;
    AdvProcessInit(fe);
    for (f = 0; f < nframes; f++) {
        float cur[80], y[80];
        int ok;
        for (i = 0; i < 80; i++) cur[i] = in[f * 80 + i];
        ok = fe->DoNoiseSup(cur, y, fe);
        produced[f] = ok ? 1 : 0;
        if (ok) {
            for (i = 0; i < 80; i++) out[nout * 80 + i] = y[i];
            nout++;
        }
    }
    AdvProcessDelete(&fe);
    return nout;
}

/*
 * ref_afe_trace -- the per-frame chain the reference author commented out
 * (etsi/cpp/ParmInterface.c:274-311): WaveProc -> CompCeps -> PostProc -> VAD on every NoiseSup
 * output once the denoised buffer holds a frame, then FlushAdvProcess (ParmInterface.c:348-354) at
 * the end of the input, exactly in that order, calling the reference's own functions through the
 * slots AdvProcessAlloc wired (ParmInterface.c:58-82).  All-zero frames before the first non-zero
 * one take the null-feature branch of DoAdvProcess itself (ParmInterface.c:314-329).
 *
 *   flags     5 ints per input frame after DoAdvProcess: SpeechFoundVar, SpeechFoundSpec,
 *             SpeechFoundMel, SpeechFoundVADNS, FrameCounter
 *   feat_cc   14 floats per cepstral frame after WaveProc + CompCeps
 *   feat_pp   14 floats per cepstral frame after PostProc
 *   vad_out   15 floats per EMITTED feature frame (14 features + the VAD flag), in emission order
 *   counts    [0] NoiseSup outputs, [1] cepstral frames, [2] emitted feature frames
 */
long ref_afe_trace(const short *in, long n, int *flags, float *feat_cc, float *feat_pp, float *vad_out,
                   long *counts)
{
    FEParamsX *fe = AdvProcessAlloc(8000);
    short sig[80], den[80];
    float feat[NUM_CEP_COEFF + 2];
    float frameBuf[FRAME_BUF_SIZE + HP16k_MEL_USED];
    long nfr = n / 80, f, nout = 0, nceps = 0, nvad = 0;
    int i;

    fe->Noc0 = 0;
    AdvProcessInit(fe);
    memset(den, 0, sizeof den);
    memset(feat, 0, sizeof feat); /* DoVADFlush may return TRUE without writing it (VAD.c:421-428) */

    for (f = 0; f < nfr; f++) {
        NoiseSupStructX *NSX = fe->NSX;
        long before = NSX->nsVar.buffers.nbFramesOutSecondStage;
        long zeros_before = fe->ZeroFrameCounter;
        for (i = 0; i < 80; i++) sig[i] = in[f * 80 + i];
        DoAdvProcess(sig, den, feat, fe);
        if (fe->ZeroFrameCounter > zeros_before) { /* null MFCC vector, VAD = NON_SPEECH, returned TRUE */
            for (i = 0; i < 14; i++) vad_out[nvad * 15 + i] = feat[i];
            vad_out[nvad * 15 + 14] = 0.0f;
            nvad++;
        }
        if (NSX->nsVar.buffers.nbFramesOutSecondStage > before) {
            nout++;
            if (fe->offsetDenoisedFrame < 0) fe->offsetDenoisedFrame += fe->FrameShift;
            if (fe->offsetDenoisedFrame >= 0) {
                BufInGetLast(fe->denoisedBuf, frameBuf, fe->FrameLength + fe->offsetDenoisedFrame + 1);
                fe->DoWaveProc(frameBuf + 1, fe);
                fe->DoCompCeps(frameBuf + 1, feat, fe);
                for (i = 0; i < 14; i++) feat_cc[nceps * 14 + i] = feat[i];
                fe->DoPostProc(feat, fe);
                for (i = 0; i < 14; i++) feat_pp[nceps * 14 + i] = feat[i];
                nceps++;
                if (fe->DoVADProc(feat, fe)) {
                    for (i = 0; i < 15; i++) vad_out[nvad * 15 + i] = feat[i];
                    nvad++;
                }
            }
        }
        if (flags) {
            int *p = flags + f * 5;
            p[0] = fe->SpeechFoundVar;
            p[1] = fe->SpeechFoundSpec;
            p[2] = fe->SpeechFoundMel;
            p[3] = fe->SpeechFoundVADNS;
            p[4] = fe->FrameCounter;
        }
    }
    while (FlushAdvProcess(feat, fe)) {
        for (i = 0; i < 15; i++) vad_out[nvad * 15 + i] = feat[i];
        nvad++;
    }
    AdvProcessDelete(&fe);
    counts[0] = nout;
    counts[1] = nceps;
    counts[2] = nvad;
    return nfr;
}

/* Table dumps, so the restatement's constants can be checked entry by entry. */
void ref_ns_tables(float *sigWindow200, float *irWindow17, float *idct25x25,
                   int *melStart25, int *melLen25, float *melData /* 25*16 */)
{
    FEParamsX *fe = AdvProcessAlloc(8000);
    MelFB_Window *p;
    int j = 0, i;
    fe->Noc0 = 0;
    AdvProcessInit(fe);
    memcpy(sigWindow200, fe->NSX->nsTmp.sigWindow, 200 * sizeof(float));
    memcpy(irWindow17, fe->NSX->nsTmp.IRWindow, 17 * sizeof(float));
    for (i = 0; i < 25; i++) memcpy(idct25x25 + 25 * i, fe->NSX->nsTmp.melIDCTbasis[i], 25 * sizeof(float));
    for (p = fe->NSX->nsTmp.FirstWindow; p; p = p->Next, j++) {
        melStart25[j] = p->StartingPoint;
        melLen25[j] = p->Length;
        for (i = 0; i < p->Length && i < 16; i++) melData[j * 16 + i] = p->Data[i];
    }
    AdvProcessDelete(&fe);
}
