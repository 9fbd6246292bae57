/*
 * sea_kernels.h -- kernel argument blocks and launch prototypes (internal to the library).
 */
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sea_tables.h"

namespace sea {

/* Batch of utterances packed in one int16 buffer.  offsets[u] (in samples, multiple of 8) locates
 * utterance u in `in`, `out` and `out_f32`; lengths[u] is its sample count. */
struct NsBatchArgs {
    const int16_t *in;
    int16_t *out;
    float *out_f32;            /* optional: float NoiseSup output stream, same indexing as out */
    const long long *offsets;
    const long long *lengths;
    const int *order;          /* optional: block b processes utterance order[b] (longest first) */
    int *first_out;            /* optional: per utterance, frame index of the first NS output (-1: none) */
    const sea_ns_tables *tables;
    int n_utt;
    /* frame-dropping VAD inputs (ns_denoise_pipe_fd_kernel only): */
    unsigned char *flags_out;  /* per output frame fo of utterance u, at [offsets[u]/8 + 10*fo]: bit 0 SpeechFoundVar,
                                * 1 Spec, 2 Mel, 3 VADNS of the tick that produced the frame */
    int *onset_out;            /* per utterance: index of the first non-zero frame (number of frames if none) */
    int prio_row;              /* > 0: workgroups [k * prio_row, (k+1) * prio_row) get issue priority 3 - k - prio_base (four-wave form only) */
    int prio_base;             /* rows to skip: a later chunk of a batch launched in pieces (hostpipe.hip) starts below the first */
    /* An utterance processed in TIME SLICES, one launch per slice (four-wave forms without speech flags only): state != nullptr
     * makes every workgroup store its recursion at the end of the launch -- kNsPipeStateFloats floats at state + u * that --
     * and, with resume != 0, start from what the previous slice stored instead of DoNoiseSupInit's state.  in / out /
     * offsets / lengths describe the slice; frame_base = frames of the utterance before this slice (first_out is absolute). */
    float *state;
    int resume;
    int frame_base;
    int perm6;                 /* six-wave form: wave -> role map, three bits per wave, wave 0 lowest (0: the kernel's default) */
};
/* [2 x 640 stage buffers as 8-slot rings][12 x 64 per-lane spectra: noise, den, previous PSD of (lane, 64) x 2 stages]
 * [8 frame energies][8 denSigSE1 sums][8 speech-flag words][40 scalars] */
constexpr int kNsPipeStateFloats = 2 * 640 + 12 * 64 + 3 * 8 + 40;

/* B independent streams, nframes frames of 80 floats each, state blobs of kNsStateFloats floats */
constexpr int kNsStateFloats = 2 * 320 + 12 * 64 + 32;
struct NsStreamArgs {
    const float *in;           /* [B][nframes][80] */
    float *out;                /* [B][nframes][80], written where produced */
    int *produced;             /* [B][nframes] */
    float *state;              /* [B][kNsStateFloats] */
    const sea_ns_tables *tables;
    int nframes;
    int reset;                 /* 1: start from DoNoiseSupInit state instead of loading */
    unsigned char *flags;      /* optional [B][nframes]: bit 0 SpeechFoundVar, 1 Spec, 2 Mel, 3 VADNS of the tick */
    int *frame_counter;        /* optional [B][nframes]: FEParamsX::FrameCounter after the tick */
};

/* the 16 k-native variant (ns16k_kernel.hip): B independent streams, nframes frames of 160 floats each */
constexpr int kNs16StateFloats = 2 * SEA16_BUF + 6 * 132 + 32;
struct Ns16StreamArgs {
    const float *in;           /* [B][nframes][160] */
    float *out;                /* [B][nframes][160], written where produced */
    int *produced;             /* [B][nframes]: the second stage ran (outData was written) */
    unsigned char *flags;      /* optional [B][nframes]: bit 0 SpeechFoundVar, 1 Spec, 2 Mel, 3 VADNS; 0 where the first stage did not run */
    int *frame_counter;        /* optional [B][nframes]: pFrameCounter, 0 where the first stage did not run */
    float *wiener;             /* optional [B][nframes][25]: the gains func_Wiener prints, written where produced */
    float *state;              /* [B][kNs16StateFloats] */
    const sea_ns16k_tables *tables;
    int nframes;
    int reset;
    int n_streams;
};
constexpr int kNs16StreamsPerGroup = 4; /* wavefronts = streams per workgroup of ns16k_stream_kernel */

struct CepsArgs {
    const float *den_f32;      /* float NoiseSup stream written by ns_denoise_kernel */
    const long long *offsets;  /* as above */
    const long long *lengths;
    const int *first_out;
    const long long *ceps_cum; /* n_utt+1 prefix sums of the per-utterance frame capacity */
    float *ceps;               /* [ceps_cum[n_utt]][14] */
    int *n_ceps;               /* optional: valid cepstral frames per utterance */
    const sea_cc_tables *tables;
    int n_utt;
};

/* WaveProc -> CompCeps -> PostProc -> VAD (+ flush) on the float NoiseSup stream */
struct AfeArgs {
    const float *den_f32;          /* float NoiseSup stream (ns_denoise_pipe_fd_kernel) */
    const unsigned char *flags;    /* its speech flags, [offsets[u]/8 + 10*fo] */
    const long long *offsets;
    const long long *lengths;
    const int *first_out;          /* frame index of the first NoiseSup output, -1: none */
    const int *onset;              /* index of the first non-zero frame */
    const long long *ceps_cum;     /* n_utt+1 prefix sums of cepstral-frame capacities (>= lengths/80 - 6) */
    float *feat_cc;                /* [ceps_cum[n_utt]][14]: after WaveProc + CompCeps */
    float *feat_pp;                /* optional, same shape: after PostProc */
    const long long *feat_cum;     /* n_utt+1 prefix sums of emitted-frame capacities (>= lengths/80 + 6) */
    float *feat15;                 /* [feat_cum[n_utt]][15]: emitted feature frames + VAD flag */
    int *n_feat;                   /* emitted frames per utterance */
    int *n_ceps;                   /* optional */
    const sea_cc_tables *tables;
    int n_utt;
};

struct ResynthArgs {
    const int16_t *in;
    int16_t *out;
    const long long *offsets;      /* samples, multiple of 8 */
    const long long *lengths;
    const float *mask;             /* rows of 64 floats */
    const long long *mask_offsets; /* in rows */
    float *inter;                  /* intermediate [sum(lengths padded)][64] floats */
    const int *order;
    const sea_gt_tables *tables;
    int n_utt;
    int binary;
};

/* subbband(): utterance u's 64 int16 streams form a [64][pitch] block at out + offsets[u]*64,
 * pitch = lengths[u] rounded up to 8 samples */
struct SubbandArgs {
    const int16_t *in;
    int16_t *out;
    const long long *offsets;
    const long long *lengths;
    const int *order;
    const sea_gt_tables *tables;
    int n_utt;
};

/* IRM target from the subband streams of the clean and the noise signal (irm_kernel.hip) */
struct IrmArgs {
    const int16_t *pure, *noise;   /* [64][pitch] blocks at offsets[u] * 64, pitch = lengths[u] rounded up to 8 */
    const long long *offsets;
    const long long *lengths;
    const long long *row_offsets;  /* first mask row of utterance u; rows = (lengths[u] - 320) / 160 + 1 */
    float *irm;                    /* [rows][64] */
    const sea_fft_tables *fft;
    int n_utt;
    int window;                    /* 0 rectangular, 1 Hamming, 2 Hanning */
};
__global__ void irm_target_kernel(IrmArgs a);      /* round 4: per-lane codelet, lane = (polyphase component, frame) */
__global__ void irm_target_dual_kernel(IrmArgs a); /* rounds 2-3: LDS dual transform per frame (A/B) */

__global__ void subband_kernel(SubbandArgs a);
__global__ void ns_denoise_kernel(NsBatchArgs a);
__global__ void ns_denoise_pipe_kernel(NsBatchArgs a);
__global__ void ns_denoise_pipe_big_kernel(NsBatchArgs a); /* lower-register form for > 4 utterances per CU */
__global__ void ns_denoise_pipe_pair_kernel(NsBatchArgs a);
/* two utterances per workgroup, lane-sparse phases packed (ns_pipe2_kernel.hip: experiment) */
int ns_pair_threads(); /* threads per workgroup of that kernel */
__global__ void ns_denoise_pipe_fd_kernel(NsBatchArgs a);
__global__ void ns_denoise_pipe_slice_kernel(NsBatchArgs a);     /* time slices: state in / out (NsBatchArgs::state) */
__global__ void ns_denoise_pipe_big_slice_kernel(NsBatchArgs a);
__global__ void ns_denoise_pipe6_kernel(NsBatchArgs a);    /* six waves per utterance (ns_pipe6_kernel.hip) */
__global__ void ns_denoise_pipe6_dense_kernel(NsBatchArgs a);
__global__ void ns_denoise_wave_kernel(NsBatchArgs a);     /* one wave per utterance, several per workgroup (ns_wave_kernel.hip) */
int ns_wave_utts_per_block(); /* the same compiled for seven waves per SIMD: four workgroups per CU co-reside */
__global__ void ns_denoise_pipe6_fd_kernel(NsBatchArgs a); /* + speech flags for the frame-dropping VAD */
__global__ void ns_stream_kernel(NsStreamArgs a);
__global__ void ns16k_stream_kernel(Ns16StreamArgs a);
__global__ void ns16k_pipe_kernel(Ns16StreamArgs a); /* four pipelined waves per stream, two streams per workgroup */
constexpr int kNs16PipeStreamsPerGroup = 2;
__global__ void ns16k_selftest_kernel(const sea_ns16k_tables *t, const float *frames, int nfft, float *outA, float *outB, const float *gains,
                                      int ngain, float *gamma25, float *idct9);
__global__ void ns_stream_fd_kernel(NsStreamArgs a);
__global__ void selftest_pi4_kernel(unsigned long long *mismatches);
__global__ void selftest_dc_kernel(const float *dif, const float *y0, float *out, int *fellback, int ncases);
__global__ void selftest_log_kernel(const float *x, double *out, int n);
__global__ void selftest_log_dd_kernel(const double *x, double *hi, double *lo, int n);
__global__ void selftest_log_sites_kernel(const float *x, float *site1, float *site2, int n);
__global__ void selftest_log_guard_kernel(int site, unsigned long long *stats, float *hits, int cap);
__global__ void selftest_nsdiv_kernel(unsigned long long *out, int iters);
__global__ void selftest_div_kernel(const sea_gt_tables *t, unsigned long long *mismatches);
__global__ void rfft256_kernel(const float *in, float *out, long long nframes, const sea_fft_tables *t);
__global__ void rfft_any_kernel(float *x, const unsigned *sched, long long nframes);
__global__ void compceps_kernel(CepsArgs a);
__global__ void afe_ceps_kernel(AfeArgs a); /* WaveProc + CompCeps, one wave per cepstral frame */
__global__ void afe_vad_kernel(AfeArgs a);  /* PostProc + frame-dropping VAD + flush, one wave per utterance */
__global__ void compceps_frames_kernel(const float *data201, float *coef14, long long nframes,
                                       const sea_cc_tables *t);
__global__ void resynth_fwd_kernel(ResynthArgs a);
__global__ void resynth_bwd_kernel(ResynthArgs a);
__global__ void resynth_fused_kernel(ResynthArgs a); /* both passes of an utterance in one workgroup */
__global__ void gammatone_kernel(const float *in, float *out, int chan, long long L, const sea_gt_tables *t);

} // namespace sea
