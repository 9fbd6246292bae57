/*
 * oracle/ref_driver_aurora.cpp -- TEST INFRASTRUCTURE ONLY.
 *
 * C entry points over the two files of the 16 k-native NoiseSup variant that compile on their own
 * (function/20141106_speech_enhancement/aurora_etsi/rfft.cpp and MelProc.cpp, compiled where they lie by
 * oracle/Makefile's `aurora` target into oracle/_ref/libaurora_ref.so; the variant's NoiseSup.cpp needs the absent
 * aurora/aurora_include.h and is not built).  They pin ns16k_oracle.c's transform, tables, DoGamma and DoGammaIDCT.
 * Nothing of the reference is copied: its headers are included from the reference tree.
 */
#include <stdlib.h>
#include <string.h>

#include "MelProcExports.h" /* aurora_etsi/MelProcExports.h */
#include "rfft.h"           /* aurora_etsi/rfft.h */

namespace {
Gamma_Window *g_first = NULL;
float *g_basis[WF_MEL_ORDER];

void ensure()
{
    if (g_first) return;
    g_first = CGammaAlloc();
    /* the arguments of NoiseSup.cpp:1054 and :1079 (SampFreq 16000, 2 * (NS_SPEC_ORDER - 1) = 256) */
    InitGammawindows(g_first, 80.0, (float)16000, 256, WF_MEL_ORDER, 1);
    for (int i = 0; i < WF_MEL_ORDER; i++) g_basis[i] = (float *)malloc(sizeof(float) * WF_MEL_ORDER);
    InitGammaIDCTbasis(g_basis, g_first, WF_MEL_ORDER, 16000, 256);
}
} // namespace

extern "C" {

void ref16_rfft(float *x, int n, int m) { rfft(x, n, m); }

void ref16_tables(int *gammaStart25, int *gammaLen25, float *gamma25x128, float *idct25x25)
{
    ensure();
    int j = 0;
    for (Gamma_Window *p = g_first; p && j < WF_MEL_ORDER; p = p->Next, j++) {
        gammaStart25[j] = p->StartingPoint;
        gammaLen25[j] = p->Length;
        memcpy(gamma25x128 + 128 * j, p->Data, sizeof(float) * (p->Length < 128 ? p->Length : 128));
    }
    for (int i = 0; i < WF_MEL_ORDER; i++) memcpy(idct25x25 + WF_MEL_ORDER * i, g_basis[i], sizeof(float) * WF_MEL_ORDER);
}

void ref16_do_gamma(float *W)
{
    ensure();
    DoGamma(W, g_first);
}

void ref16_idct(float *W)
{
    ensure();
    DoGammaIDCT(W, g_basis, WF_MEL_ORDER, WF_MEL_ORDER);
}
}
