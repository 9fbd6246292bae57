/*
 * oracle/sea_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement ("oracle") of the hot path of guokiddo1/speech_enhancement:
 *   - etsi/ two-stage Wiener NoiseSup + rfft + CompCeps   (ns_oracle.c)
 *   - resyth_64sub_{ori,IBM} 64-band gammatone resynthesis (resynth_oracle.c)
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product (speech_enhancement_amd/) never links, imports or calls it.
 *
 * Parity pins:
 *   NoiseSup/rfft/CompCeps: checked bit-for-bit against the reference C compiled from
 *     /root/reference/etsi/cpp (oracle/_ref/libetsi_ref.so) and against the known answers of
 *     SURVEY.md 8(c); golden vectors from that build are committed under tests/golden/.
 *   resynth: the reference cannot be built here (needs the private asdk Wave.h; no stand-ins
 *     allowed); pinned by the reference outputs SURVEY.md 8(c) recorded (10 samples + checksum).
 */
#ifndef SEA_ORACLE_H
#define SEA_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define ORA_TRACE_NSCAL 16

/* etsi/cpp/AdvFrontEnd.c:125-210 */
int ora_etsi_denoise(const short *in, short *out, long n);
/* etsi/cpp/rfft.c:45-180 (n == 1<<m, any m >= 2) */
void ora_rfft(float *x, int n, int m);
/* same contract as ref_ns_trace() in ref_driver.c */
long ora_ns_stream_f32(const float *in, long nframes, float *out, int *produced);
long ora_afe_trace(const short *in, long n, int *flags, float *feat_cc, float *feat_pp, float *vad_out,
                   long *counts);
long ora_ns_trace(const short *in, long n, short *out_i16, float *den_f32, float *ceps,
                  float *scal, float *spec, long *counts);
/* etsi/cpp/CompCeps.c:309-318,368-549; data201[0] is Data[-1] */
void ora_compceps_frame(const float *data201, float *coef14);
void ora_ns_tables(float *sigWindow200, float *irWindow17, float *idct25x25, int *melStart25,
                   int *melLen25, float *melData /* 25*16 */);
void ora_cc_tables(float *hamming100, float *dct12x23, int *melStart23, int *melLen23,
                   float *melData /* 23*32 */);

/* resyth_64sub_ori/cpp/extractwav.cpp:9-131 (binary=0) and resyth_64sub_IBM (binary=1).
 * mask is [F][64] row-major with F = (L-320)/160+1.  Returns 0, or 1 on bad arguments. */
int ora_resynth64(const short *in, long L, const float *mask, int F, int binary, short *out);
/* extractwav.cpp:167-211: one channel's 4th-order gammatone */
void ora_gammatone(const float *in, float *out, float cf, float bw, float midEar, long L);
/* extractwav.cpp:41-54: per-channel constants cf, bw, midEarCoeff (64 each) */
void ora_resynth_channels(float *cf64, float *bw64, float *midEar64);

/* SURVEY 8(f) rank 1 -- subbband(): enhancement_extract_test/cpp/extractwav.cpp:40-101 and
 * hairCell resyth_64sub_ori/cpp/extractwav.cpp:212-257.  out is [64][L].  Parity unpinned. */
void ora_haircell(const float *input, float *output, long L);
int ora_subband64(const short *in, long L, short *out);
/* SURVEY 8(f) rank 2 -- the IRM target of make_single_IBM (enhancement_extract_test/cpp/show_IBM.cpp:105-169) on
 * the 64 subband streams of the clean and the noise signal ([64][pitch] int16 each); irm is [F][64],
 * F = (L-320)/160+1; window 0 rectangular / 1 Hamming / 2 Hanning (asdk::SpecInfo is absent: parity unpinned). */
int ora_irm_target(const short *pure, const short *noise, long L, long pitch, int window, float *irm);

/* SURVEY 8(f) rank 4 -- the 16 k-native NoiseSup variant behind etsi_denoise_mapping_* (ns16k_oracle.c; its frame
 * loop is parity unpinned, its transform / windows / IDCT are pinned against oracle/_ref/libaurora_ref.so) */
typedef struct ora16_state ora16_state;
ora16_state *ora16_new(void);
void ora16_free(ora16_state *s);
long ora16_push(ora16_state *s, const float *in, long dataNum, float *out, int *var, int *spec, int *mel, int *vadns,
                int *counter, float *wiener /* [rows][25], may be NULL */, long *wiener_rows);
void ora16_rfft(float *x, int n, int m); /* aurora_etsi/rfft.cpp as C++: the variant calls it with (512, 8) */
void ora16_tables(float *sigWindow480, float *irWindow17, int *gammaStart25, float *gamma25x128, float *idct25x25);
void ora16_do_gamma(float *W /* >= 128 gains in, 25 out */);
void ora16_idct(float *W /* 25 */);

#ifdef __cplusplus
}
#endif
#endif
