/*
 * oracle/ref_driver_cc.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Second driver TU for oracle/_ref/libetsi_ref.so: textually includes the reference's CompCeps.c
 * (where it lies, -I/root/reference/etsi/cpp) so that the opaque CompCepsStructX tables
 * (Hamming window etsi/cpp/CompCeps.c:87-94, DCT matrix :153-173, 23 mel triangles
 * etsi/cpp/MelProc.c:402-522) can be dumped and compared entry by entry with the restatement.
 * Also exposes a stand-alone DoCompCeps call on a caller-provided 201-sample frame.
 */
#include "CompCeps.c"

/* data201[0] is Data[-1]; data201[1..200] the frame.  coef14 = c1..c12, c0, logE */
void ref_compceps_frame(const float *data201, float *coef14)
{
    FEParamsX *fe = AdvProcessAlloc(8000);
    float buf[201];
    int i;
    fe->Noc0 = 0;
    AdvProcessInit(fe);
    for (i = 0; i < 201; i++) buf[i] = data201[i];
    fe->DoCompCeps(buf + 1, coef14, fe);
    AdvProcessDelete(&fe);
}

void ref_cc_tables(float *hamming100, float *dct12x23, int *melStart23, int *melLen23,
                   float *melData /* 23*32 */)
{
    FEParamsX *fe = AdvProcessAlloc(8000);
    MelFB_Window *p;
    int j = 0, i;
    fe->Noc0 = 0;
    AdvProcessInit(fe);
    for (i = 0; i < 100; i++) hamming100[i] = fe->CCX->HammingWindow[i];
    for (i = 0; i < 12 * 23; i++) dct12x23[i] = fe->CCX->pDCTMatrix[i];
    for (p = fe->CCX->FirstWindow; p; p = p->Next, j++) {
        melStart23[j] = p->StartingPoint;
        melLen23[j] = p->Length;
        for (i = 0; i < p->Length && i < 32; i++) melData[j * 32 + i] = p->Data[i];
    }
    AdvProcessDelete(&fe);
}
