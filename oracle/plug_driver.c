/*
 * oracle/plug_driver.c -- TEST INFRASTRUCTURE ONLY (never shipped, never on the product path).
 *
 * Runs the MI355X engine INSIDE the reference's own driver: compiled together with the reference's
 * etsi/cpp/*.c where they lie (all eleven translation units, unmodified) and with SeaPlugin.c -- the file
 * INTEGRATION.md section 2 tells a maintainer to add, extracted from that document at build time -- into
 * oracle/_ref/libetsi_ref_plugged.so, linked against speech_enhancement_amd/libsea_mi355x.so (oracle/Makefile,
 * target `plugged`).  Nothing from the reference is copied into this repository.
 *
 *   plug_trace(in, n, plugged, ...)   AdvProcessAlloc (plugged = 0) or SeaAdvProcessAlloc (plugged = 1), then the
 *                                     reference's DoAdvProcess (etsi/cpp/ParmInterface.c:208-330) once per 80-sample
 *                                     frame; after each NoiseSup output from the third on, DoCompCeps through the
 *                                     FEParamsX slot exactly as the block the author commented out did
 *                                     (ParmInterface.c:275-293).  With plugged = 1 the NoiseSup and CompCeps slots
 *                                     are the engine's (sea_ns_stream_push, sea_compceps_frame); everything else --
 *                                     the int16 <-> float casts, the zero-frame gate, the denoised-sample shift
 *                                     register (BufferIn.c) -- is the reference's own code in both runs.
 */
#include <stdio.h>
#include <string.h>

#include "ParmInterface.h"
#include "BufferIn.h"
#include "16kHzProcExports.h"

FILE *fp_denoised = NULL; /* ParmInterface.c declares it extern; nothing uses it */

FEParamsX *SeaAdvProcessAlloc(int fs); /* SeaPlugin.c */

/* counts the TRUE returns of whatever sits in the DoNoiseSup slot (DoAdvProcess's own return value is not usable:
 * it falls off the end of a non-void function once the feature chain is commented out) */
static BOOLEAN (*g_inner)(X_FLOAT32 *, X_FLOAT32 *, FEParamsX *);
static long g_produced;
static BOOLEAN counting_ns(X_FLOAT32 *in, X_FLOAT32 *out, FEParamsX *This)
{
    const BOOLEAN r = g_inner(in, out, This);
    if (r) g_produced++;
    return r;
}

/*
 * in[n] int16; out_i16[(n/80)*80] what the driver copies out per frame (caller pre-fills: frames before the first
 * NoiseSup output keep the previous DenoiseBuffer content = zeros, as etsi_denoise's loop sees them);
 * ceps: 14 floats per cepstral frame (capacity (n/80)*14); counts[0] NoiseSup outputs, counts[1] cepstral frames.
 */
long plug_trace(const short *in, long n, int plugged, short *out_i16, float *ceps, long *counts)
{
    FEParamsX *fe = plugged ? SeaAdvProcessAlloc(8000) : AdvProcessAlloc(8000);
    short sig[80], den[80];
    float feat[NUM_CEP_COEFF + 2];
    float frameBuf[FRAME_BUF_SIZE + HP16k_MEL_USED];
    long nfr = n / 80, f, nceps = 0;
    int i;

    fe->Noc0 = 0;
    AdvProcessInit(fe);
    g_inner = fe->DoNoiseSup;
    g_produced = 0;
    fe->DoNoiseSup = counting_ns;
    memset(den, 0, sizeof den);
    for (f = 0; f < nfr; f++) {
        const long before = g_produced;
        for (i = 0; i < 80; i++) sig[i] = in[f * 80 + i];
        DoAdvProcess(sig, den, feat, fe);
        for (i = 0; i < 80; i++) out_i16[f * 80 + i] = den[i];
        if (g_produced > before && g_produced >= 3) {
            BufInGetLast(fe->denoisedBuf, frameBuf, FRAME_BUF_SIZE);
            fe->DoCompCeps(frameBuf + 1, feat, fe);
            for (i = 0; i < 14; i++) ceps[nceps * 14 + i] = feat[i];
            nceps++;
        }
    }
    fe->DoNoiseSup = g_inner;
    AdvProcessDelete(&fe);
    counts[0] = g_produced;
    counts[1] = nceps;
    return nfr;
}
