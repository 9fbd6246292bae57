/*
 * sea_mi355x.h -- C ABI of libsea_mi355x.so, the MI355X (gfx950) engine for the per-frame
 * noise-suppression hot path of guokiddo1/speech_enhancement.
 *
 * Plain C: pointers and sizes only, no C++/torch types.  Three groups of entry points:
 *
 *  (1) DROP-INS with the reference's own names and signatures, so the reference's callers link
 *      against this library unchanged (host buffers in, host buffers out; the library does the
 *      H2D/D2H itself):
 *          etsi_denoise*            etsi/cpp/AdvFrontEnd.h:13-17  (AdvFrontEnd.c:125-329)
 *          rfft                     etsi/cpp/rfft.h:19            (rfft.c:45-180)
 *  (2) HANDLE-BASED equivalents of the FEParamsX plug-in slots (etsi/cpp/ParmInterface.h:120-178;
 *      wired in etsi/cpp/ParmInterface.c:58-82).  INTEGRATION.md shows the 4-line shims that
 *      install them into the reference's vtable:
 *          sea_ns_stream_*          DoNoiseSupAlloc/Init/DoNoiseSup/Delete  (NoiseSup.c:859-1440)
 *          sea_compceps_frame       DoCompCeps                              (CompCeps.c:309-318)
 *          sea_gammatone_filter     gammaToneFilter   resyth_64sub_ori/cpp/HuWang.h:49
 *          sea_resynth64            resynth()         resyth_64sub_ori/cpp/extractwav.h:4
 *                                                     (C form: the reference's takes asdk::CWave)
 *  (3) BATCH entry points on DEVICE pointers (HBM-resident data, caller's HIP stream): what the
 *      reference's only parallel harness -- a thread pool over files,
 *      function/20141106_speech_enhancement/aurora_speech_enhancement/aurora_speech_enhancement.cpp:82-230
 *      -- becomes on a GPU: one launch over a packed batch of utterances.
 *
 * Return convention follows the reference (AdvFrontEnd.c:205-209): 0 (FALSE) = success, non-zero
 * = fault; sea_last_error() describes the most recent fault on the calling thread.
 * There is NO CPU fallback: without a usable gfx950 device every call fails.
 */
#ifndef SEA_MI355X_H
#define SEA_MI355X_H

#ifdef __cplusplus
extern "C" {
#endif

/* ----------------------------------------------------------------------------------------------
 * (1) drop-ins
 * -------------------------------------------------------------------------------------------- */
/* etsi/cpp/AdvFrontEnd.h:13 -- 8 kHz-mode two-stage Wiener denoiser on i_frame samples.
 * p_denoised[0 .. 80*(i_frame/80)) is written (first 320 samples after the first non-zero frame
 * are 0: SURVEY F7/F8); the trailing i_frame%80 samples are left untouched, as in the reference. */
int etsi_denoise(short *p_data, short *p_denoised, long i_frame);
/* etsi/cpp/AdvFrontEnd.h:14 -- reference behaviour kept: runs etsi_denoise into a scratch buffer
 * and copies only on fault, i.e. never writes p_denoised on success. */
int etsi_denoise_synchronization(short *p_data, short *p_denoised, long i_frame);
/* etsi/cpp/AdvFrontEnd.h:16 -- the reference's 16 kHz-mode entry point reads 80 shorts past a heap
 * buffer per frame and nothing calls it; exported so callers link, it returns 1 (fault)
 * without touching p_denoised. */
int etsi_denoise_16k(short *p_data, short *p_denoised, long i_frame);
/* etsi/cpp/AdvFrontEnd.h:17 -- AdvFrontEnd.c:316-329 calls the 8 kHz-mode etsi_denoise (not the _16k one): the same
 * behaviour as etsi_denoise_synchronization, returns 0 and never writes p_denoised on success. */
int etsi_denoise_16k_synchronization(short *p_data, short *p_denoised, long i_frame);
/* etsi/cpp/rfft.h:19 -- in-place real split-radix FFT, output Re(0..n/2), Im(n/2-1..1), for every size the
 * reference's routine takes (etsi/cpp/rfft.c:45-180): n a power of two (here up to 16384) and any order m with
 * 2^m <= n -- incl. the 16 k-native variant's rfft (x, 512, 8), order 8 on length 512.  A size the routine cannot
 * take (or a missing device) leaves x untouched, prints the reason to stderr and sets sea_last_error(); nothing
 * abort()s the caller. */
void rfft(float *x, int n, int m);

/* ----------------------------------------------------------------------------------------------
 * library state
 * -------------------------------------------------------------------------------------------- */
int sea_init(int device);            /* device < 0: keep the current HIP device; >= 0: make it the calling thread's device */
int sea_device_count(void);          /* usable HIP devices (0 when there is none); one host thread per device is the
                                      * multi-GPU model: utterances are independent, nothing crosses between devices */
const char *sea_last_error(void);
const char *sea_version(void);
/* Device-resident constant tables, for callers that want to inspect them (tests). */
int sea_tables_host(float *sigWindow200, float *irWindow17, float *idct25x25, int *melStart25,
                    int *melLen25, float *melData25x16, float *hamming100, float *dct12x23,
                    int *ccStart23, int *ccLen23, float *ccData23x32);
int sea_gammatone_channels(float *cf64, float *bw64, float *midEar64);

/* ----------------------------------------------------------------------------------------------
 * (3) batch entry points, device pointers.  `stream` is a hipStream_t (NULL = default stream).
 * Utterance u occupies samples [offsets[u], offsets[u]+lengths[u]) of d_in / d_out / d_out_f32;
 * offsets must be multiples of 8 samples.
 * -------------------------------------------------------------------------------------------- */
/* NoiseSup over a batch (etsi_denoise semantics per utterance).
 *   d_out_f32    optional: float NoiseSup output (pre-cast), same indexing as d_out
 *   d_order      optional: launch order (utterance indices, longest first balances the tail; with more than one utterance per
 *                CU the kernels also set their issue priority from the frames each utterance has left relative to the FIRST
 *                utterance of this order, so that the utterances sharing a CU finish together -- results do not depend on it)
 *   d_first_out  optional: per utterance, frame index of the first output frame, -1 if none */
int sea_ns_denoise_batch(const short *d_in, short *d_out, float *d_out_f32,
                         const long long *d_offsets, const long long *d_lengths, const int *d_order,
                         int *d_first_out, int n_utt, void *stream);
/* rfft on [nframes][256] floats (d_out may equal d_in) */
int sea_rfft256_batch(const float *d_in, float *d_out, long long nframes, void *stream);
/* rfft (x, n, m) in place on [nframes][n] floats, any size the reference's routine takes (see rfft above); (256, 8)
 * goes to the streaming kernel, every other size to a one-workgroup-per-frame schedule walker */
int sea_rfft_batch(float *d_x, int n, int m, long long nframes, void *stream);
/* DoCompCeps on [nframes][201] floats (Data[-1..199]) -> [nframes][14] = c1..c12, c0, logE */
int sea_compceps_frames(const float *d_data201, float *d_coef14, long long nframes, void *stream);
/* CompCeps straight from the float NoiseSup stream of sea_ns_denoise_batch.  d_ceps_cum holds
 * n_utt+1 prefix sums of per-utterance capacities (>= lengths/80 - 6 each); cepstral frame j of
 * utterance u lands at d_ceps[(d_ceps_cum[u] + j) * 14]; d_n_ceps[u] receives the valid count. */
int sea_compceps_batch(const float *d_den_f32, const long long *d_offsets, const long long *d_lengths,
                       const int *d_first_out, const long long *d_ceps_cum, long long total_frames,
                       float *d_ceps, int *d_n_ceps, int n_utt, void *stream);
/* sea_ns_denoise_batch picks one of several forms of the same kernel by batch size (all bit-identical):
 * 3 = six waves per utterance (up to 3 utterances per CU: shortest frame period), 6 = the same compiled so that four
 * workgroups co-reside on a CU (up to 4 per CU; round 4), 4 = four waves, lower register use (larger batches), 2 = four waves
 * (the form for up to 4 per CU until round 4; the time-slice launches run on it), 1 = one wave per utterance; 5 = two utterances per
 * workgroup with their lane-sparse phases packed into one wave: 22 % fewer vector instructions per frame and slower; 7 = one wave per
 * utterance running every role in sequence (no workgroup barrier; register-bound at twelve utterances per CU: slower)
 * (5 and 7: experiments kept under the parity tests, never chosen by batch size).
 * sea_ns_kernel_form(f) forces form f for later calls (0 = by batch size again; the SEA_NS_KERNEL
 * environment variable = single | pipe | pipe6 | pipe6d | big | pair | wave sets the initial value); returns the previous one. */
int sea_ns_kernel_form(int form);
/* The same for one TIME SLICE of every utterance: a batch may be cut along the time axis and run as one launch per
 * slice, so that a caller can upload slice k + 1 and download slice k - 1 while slice k is on the device
 * (sea_denoise_utterances does).  d_in / d_out / d_offsets / d_lengths describe THIS slice (each utterance's frames of
 * the slice, packed like a batch of their own); d_state holds sea_ns_slice_state_floats() floats per utterance --
 * utterance u of every slice is the same utterance -- and carries the recursion (DoNoiseSup's state, the pipeline's
 * sample rings, the DC filter, the zero-frame gate) from launch to launch; resume = 0 for the first slice.  The slices
 * of an utterance must be whole frames except the last; frame_base = the utterance's frames before this slice
 * (d_first_out, optional, is the absolute frame index of the first NoiseSup output).  Results are those of the one
 * launch: tests/test_gpu_parity.py::test_ns_time_slices_equal_one_launch. */
int sea_ns_denoise_batch_slice(const short *d_in, short *d_out, float *d_out_f32, const long long *d_offsets,
                               const long long *d_lengths, const int *d_order, int *d_first_out, float *d_state, int n_utt,
                               int frame_base, int resume, void *stream);
int sea_ns_slice_state_floats(void);

/* SURVEY 8(f) #3 -- the feature chain the reference keeps commented out (etsi/cpp/ParmInterface.c:274-311):
 * WaveProc -> CompCeps -> PostProc -> VAD, then FlushAdvProcess (:348-354).
 * Step 1: NoiseSup that also stores what the frame-dropping VAD votes over.  d_flags: one byte per
 * output frame fo of utterance u at [d_offsets[u]/8 + 10*fo] (buffer of total_padded_samples/8
 * bytes): bit 0 SpeechFoundVar, 1 SpeechFoundSpec, 2 SpeechFoundMel, 3 SpeechFoundVADNS
 * (NoiseSup.c:1255-1281, :1359-1365).  d_onset[u]: index of the first non-zero frame. */
int sea_ns_denoise_batch_fd(const short *d_in, short *d_out, float *d_out_f32,
                            const long long *d_offsets, const long long *d_lengths, const int *d_order,
                            int *d_first_out, unsigned char *d_flags, int *d_onset, int n_utt, void *stream);
/* Step 2: features.  d_ceps_cum / d_feat_cum: n_utt+1 prefix sums of per-utterance capacities
 * (>= lengths/80 - 6 cepstral frames, >= lengths/80 + 6 emitted frames).  d_feat_cc (and optionally
 * d_feat_pp) receive 14 floats per cepstral frame after WaveProc+CompCeps (after PostProc);
 * d_feat15 receives, per utterance and in emission order, the frames DoAdvProcess / FlushAdvProcess
 * would hand to the recogniser: c1..c12, c0, logE and the VAD flag (null vectors for the all-zero
 * lead, ParmInterface.c:314-329); d_n_feat[u] their number. */
int sea_afe_features_batch(const float *d_den_f32, const unsigned char *d_flags, const long long *d_offsets,
                           const long long *d_lengths, const int *d_first_out, const int *d_onset,
                           const long long *d_ceps_cum, long long total_ceps, float *d_feat_cc, float *d_feat_pp,
                           const long long *d_feat_cum, float *d_feat15, int *d_n_feat, int *d_n_ceps, int n_utt,
                           void *stream);
/* 64-band gammatone resynthesis over a batch.  mask rows (64 floats) of utterance u start at row
 * d_mask_offsets[u] and number (lengths[u]-320)/160+1.  d_inter is scratch of
 * sea_resynth_scratch_bytes(total padded samples of the batch, n_utt) bytes (the [time][64]
 * float intermediate between the two passes, ~256 B per sample: it is what 288 GB of HBM is
 * for).  binary: bit 0 selects the ideal-binary-mask variant (resyth_64sub_IBM); bit 1 the older
 * driver's frame count, lengths[u]/160 mask rows per utterance instead of (lengths[u]-320)/160+1
 * (1dnn_resynth/extractwav.cpp:67: one more frame, whose falling half covers the last hop). */
int sea_resynth64_batch(const short *d_in, short *d_out, const long long *d_offsets,
                        const long long *d_lengths, const float *d_mask,
                        const long long *d_mask_offsets, float *d_inter, const int *d_order,
                        int n_utt, int binary, void *stream);
long long sea_resynth_scratch_bytes(long long total_padded_samples, int n_utt);

/* ----------------------------------------------------------------------------------------------
 * (2) handle-based plug-in equivalents and host-buffer conveniences
 * -------------------------------------------------------------------------------------------- */
/* Many utterances at once from host memory -- what etsi/cpp/main.cpp:43-67 does per file and the batch tool
 * function/20141106_speech_enhancement/aurora_speech_enhancement/aurora_speech_enhancement.cpp:25-80 from a thread
 * pool.  A copy / compute pipeline (csrc/hostpipe.hip) over TIME SLICES of the list: slice k = the frames
 * [B_k, B_k+1) of every utterance that has them (SEA_HOST_SLICES slices of equal sample count, default 8), one launch
 * per slice with the recursion carried per utterance (sea_ns_denoise_batch_slice); a pool of SEA_HOST_THREADS (default
 * min(8, cores - 1)) host threads packs slice k+1 into pinned staging and unpacks slice k-1 while slice k's upload,
 * launch and download run on a stream each.  SEA_HOST_MODE=chunks: chunks of whole utterances of about
 * SEA_HOST_CHUNK_MB MB instead (default a third of the list).  out[u][0 .. 80*(lengths[u]/80)) is written, exactly as
 * etsi_denoise does; results do not depend on either cut. */
int sea_denoise_utterances(const short *const *in, short *const *out, const long *lengths, int n_utt);
int sea_host_threads(void); /* size of that pool */
/* NoiseSup from PINNED staging the caller fills and reads -- no pack / unpack copies (csrc/hostpipe.hip).  For a caller that
 * produces its samples itself (a file reader) and consumes the results itself (a file writer):
 *   p = sea_packed_create();                          a reusable staging set (pinned, portable across devices)
 *   sea_packed_plan(p, lengths, n_utt);               lays the list out in time slices (as sea_denoise_utterances does inside)
 *   k = sea_packed_segments(p, u, in, out, count, max);   utterance u = k pieces in time order (k <= sea_packed_slices(p)):
 *                                                     write its samples into in[i][0 .. count[i]), i = 0..k-1
 *   sea_packed_denoise(p);                            on a thread bound to a device (sea_init): 0, or 1 = fault
 *   ... out[i][0 .. count[i]) is etsi_denoise's output for every whole frame of utterance u; the trailing lengths[u] % 80
 *   samples belong to no piece (etsi_denoise never writes them).  A set may be planned again after it has been read. */
typedef struct sea_packed sea_packed;
sea_packed *sea_packed_create(void);
void sea_packed_destroy(sea_packed *p);
int sea_packed_plan(sea_packed *p, const long *lengths, int n_utt);
int sea_packed_slices(const sea_packed *p);
int sea_packed_segments(const sea_packed *p, int u, short **in_seg, short **out_seg, long *count, int max_seg);
int sea_packed_denoise(sea_packed *p);
/* NoiseSup + CompCeps from host buffers, the chain ParmInterface.c:275-293 ran before its author commented it out
 * (SURVEY 8(d) Config 1: a 4-s utterance gives 800 NoiseSup frames, 796 outputs, 794 cepstral frames): out as above;
 * ceps[u] receives n_ceps[u] rows of 14 floats (c1..c12, c0, logE), capacity max(lengths[u]/80 - 6, 0) rows. */
int sea_denoise_ceps_utterances(const short *const *in, short *const *out, float *const *ceps, int *n_ceps,
                                const long *lengths, int n_utt);

/* DoCompCeps(Data, Coef, This): Data[-1] must be valid (host pointers) */
int sea_compceps_frame(const float *Data, float *Coef14);

/* resynth(): in/out L samples, mask [F][64] with F=(L-320)/160+1 (host pointers) */
int sea_resynth64(const short *in, long L, const float *mask, int F, int binary, short *out);
/* many utterances at once from host memory (masks[u] is [F_u][64]): the same pipeline; a chunk is also bounded by
 * its share of the HBM scratch budget (60 % of the free HBM over four streams; SEA_RESYNTH_SCRATCH_MB overrides) */
int sea_resynth_utterances(const short *const *in, const long *lengths, const float *const *masks, int binary,
                           short *const *out, int n_utt);
/* subbband() (enhancement_extract_test/cpp/extractwav.cpp:40-101): gammatone + Meddis hair cell
 * (resyth_64sub_ori/cpp/extractwav.cpp:212-257) + (short) cast -> 64 int16 streams.
 * Host form: out is [64][L].  Batch form (device pointers): utterance u's streams form a [64][pitch]
 * block at d_out + d_offsets[u]*64 with pitch = lengths[u] rounded up to 8; d_out holds 64x the
 * packed input size. */
int sea_subband64(const short *in, long L, short *out);
int sea_subband64_batch(const short *d_in, short *d_out, const long long *d_offsets, const long long *d_lengths,
                        const int *d_order, int n_utt, void *stream);
/* SURVEY 8(f) #2 -- the ideal-ratio-mask TARGET of make_single_IBM (enhancement_extract_test/cpp/show_IBM.cpp:105-169):
 * from the 64 subband streams of the clean and of the noise signal (two sea_subband64 outputs), frames of 320
 * samples every 160, 512-point power spectrum, first 64 bins summed, IRM = pure / (pure + noise); one row of 64
 * floats per frame = the mask matrix sea_resynth64 takes (and sea_mask_text_write prints).  The reference's
 * spectrum routine, asdk::SpecInfo, is absent third-party code: its analysis window is the `window` parameter here
 * (0 rectangular, 1 Hamming, 2 Hanning) and parity is unpinned.  Host form: streams [64][L], irm [F][64],
 * F = (L-320)/160+1.  Batch form: blocks as sea_subband64_batch writes them; utterance u's rows start at
 * d_row_offsets[u] (the d_mask_offsets of sea_resynth64_batch). */
int sea_irm_target(const short *pure64, const short *noise64, long L, int window, float *irm);
int sea_irm_target_batch(const short *d_pure64, const short *d_noise64, const long long *d_offsets,
                         const long long *d_lengths, const long long *d_row_offsets, float *d_irm, int window,
                         int n_utt, void *stream);
/* gammaToneFilter(input, output, fChan, sigLength) for channel `chan` of the 64-band bank */
int sea_gammatone_filter(const float *input, float *output, int chan, long sigLength);

/* DoNoiseSupAlloc / DoNoiseSupInit / DoNoiseSup / DoNoiseSupDelete on a device-resident state */
typedef struct sea_ns_stream sea_ns_stream;
sea_ns_stream *sea_ns_stream_alloc(void);
void sea_ns_stream_init(sea_ns_stream *s);
/* consumes 80 float samples, returns 1 (TRUE) when out80 was produced, 0 during the latency */
int sea_ns_stream_push(sea_ns_stream *s, const float *in80, float *out80);
void sea_ns_stream_delete(sea_ns_stream *s);
/* batched form on device pointers: n_streams independent streams x nframes frames of 80 floats,
 * [stream][frame][80]; d_state holds sea_ns_state_floats() floats per stream and carries the
 * recursion from call to call (reset != 0 starts from the DoNoiseSupInit state). */
int sea_ns_streams_push(const float *d_in, float *d_out, int *d_produced, float *d_state, int n_streams,
                        int nframes, int reset, void *stream);
int sea_ns_state_floats(void);
/* the same, additionally reporting what the reference's batch plug-in shape hands back per frame
 * (esti_denoise_out of function/20141106_speech_enhancement/aurora_etsi/NoiseSupExports.h:19-27, filled by
 * etsi_denoise_mapping_func_Wiener): d_flags[stream][frame] bit 0 SpeechFoundVar, 1 Spec, 2 Mel,
 * 3 VADNS; d_frame_counter[stream][frame] = FrameCounter after the tick.  In that API's terms:
 * sea_init ~ ..._global_init, one state blob ~ ..._thread_init, one call ~ ..._func_Wiener on
 * dataNum = 80*nframes samples (this engine implements the etsi/ arithmetic, 80-sample frames). */
int sea_ns_streams_push_fd(const float *d_in, float *d_out, int *d_produced, unsigned char *d_flags,
                           int *d_frame_counter, float *d_state, int n_streams, int nframes, int reset, void *stream);

/* The 16 k-native NoiseSup variant (function/20141106_speech_enhancement/aurora_etsi/NoiseSup.cpp:1140-1407,
 * NoiseSup.h:36-53 -- 160-sample frames, window 480, NS_FFT_LENGTH 512 transformed with NS_FFT_ORDER 8, 25 gammatone-
 * shaped windows) on device pointers: n_streams independent streams x nframes frames, d_in / d_out
 * [stream][frame][160]; d_state holds sea_ns16k_state_floats() floats per stream (reset != 0: thread_init's state).
 * The frame gate of func_Wiener (:1160-1171) is applied inside.  Per frame: d_produced = outData was written;
 * d_flags (optional) bit 0 SpeechFoundVar, 1 Spec, 2 Mel, 3 VADNS and d_frame_counter (optional) = pFrameCounter, both 0
 * where the first stage did not run; d_wiener (optional) [stream][frame][25] = the gains func_Wiener prints, written
 * where produced.  Parity: see oracle/ns16k_oracle.c (transform / windows / IDCT pinned against the reference's own
 * rfft.cpp + MelProc.cpp compiled here, the frame loop unpinned). */
int sea_ns16k_streams_push(const float *d_in, float *d_out, int *d_produced, unsigned char *d_flags, int *d_frame_counter,
                           float *d_wiener, float *d_state, int n_streams, int nframes, int reset, void *stream);
int sea_ns16k_state_floats(void);
/* kernel form of sea_ns16k_streams_push for later calls: 0 = four pipelined wavefronts per stream (default), 1 = one
 * wavefront per stream (round 3's form, kept for A/B; SEA_NS16K_KERNEL=single makes it the initial form).  Same arithmetic,
 * same state blob: a stream may change forms between two pushes.  form < 0 only reads; returns the previous form. */
int sea_ns16k_kernel_form(int form);
/* its host-side tables as the reference's init code lays them out (for checks against the oracle) */
int sea_ns16k_tables_host(float *sigWindow480, float *irWindow17, int *gammaStart25, float *gamma25x128, float *idct25x25);
/* and its table-driven transform schedule (what the kernel walks) run on the host, in place on 512 floats: rfft (x, 512, 8) */
void sea_ns16k_fft_host(float *x512);

/* The reference's batch plug-in symbols (function/20141106_speech_enhancement/aurora_etsi/NoiseSupExports.h:35-42;
 * INSTANCE / PINSTANCE / int32s of the absent aurora/aurora_include.h = void*, void**, int), as adapters over
 * sea_init / one state blob per thread instance / sea_ns16k_streams_push (csrc/mapping.hip).  in_ins points to
 * {float *inData; int dataNum}, out_ins to {float *outData; int *pSpeechFoundVar, *pSpeechFoundSpec, *pSpeechFoundMel,
 * *pSpeechFoundVADNS, *pFrameCounter} (NoiseSupExports.h:14-27).  With sm_glb_res == NULL (the reference's caller,
 * resyth_64sub_ori/cpp/aurora_etsi_test.cpp:20) they run the 16 k-native variant the reference builds behind these
 * names: one call consumes dataNum / 160 frames, zero frames are skipped (aurora_etsi/NoiseSup.cpp:1160-1171), and
 * the FILE* argument of func_Wiener, when not NULL, receives the line of 25 gains per second-stage frame (:1319-1328).
 * sm_glb_res is ignored whatever it points to, as the reference ignores it (NoiseSup.cpp:913-922).  Extension, opt-in
 * through the environment only: SEA_MAPPING_8K=1 at global_init selects the etsi/ arithmetic on 80-sample frames
 * instead (sea_ns_streams_push_fd; nothing is printed).
 * global_init / thread_init return 1 on success, func_Wiener / func return 0. */
int etsi_denoise_mapping_global_init(void **sm_glb_pins, void *sm_glb_res);
int etsi_denoise_mapping_thread_init(void **sm_thd_pins, void *sm_glb_ins);
int etsi_denoise_mapping_func_Wiener(void *sm_glb_ins, void *sm_thd_ins, void *in_ins, void *out_ins, void *fp_Wiener);
int etsi_denoise_mapping_func(void *sm_glb_ins, void *sm_thd_ins, void *in_ins, void *out_ins);
void etsi_denoise_mapping_thread_release(void **sm_thd_pins);
void etsi_denoise_mapping_global_release(void **sm_glb_pins);

/* ----------------------------------------------------------------------------------------------
 * device self-tests of the places where a kernel takes a cheaper route than the reference's
 * literal arithmetic (each proven or guarded, see csrc/sea_device.h, csrc/ns_core.h and DESIGN.md section 3)
 * -------------------------------------------------------------------------------------------- */
/* all 2^32 floats s: (float)((double)s * (1/sqrt2)) vs (float)((double)s / sqrt2); count of differences */
int sea_selftest_pi4(unsigned long long *n_mismatch);
/* the resynthesis kernels' 3-instruction division by the per-channel middle-ear gain: every float
 * inside its domain (2^-100 <= |a| <= 2^100) x all 64 divisors against the IEEE quotient.
 * out2[0] = mismatches, out2[1] = patterns tested per divisor */
int sea_selftest_div(unsigned long long *out2);
/* the NoiseSup gain / noise-tracking divisions inside their per-frame guarded domain (csrc/ns_core.h, ns_div:
 * the compiler's IEEE sequence without v_div_scale / v_div_fixup, reciprocal shared per denominator) against
 * plain division on 2^30 pseudo-random + edge-mantissa operand pairs spanning the domain, and the double
 * reciprocal of the noise update.  out4[0] = pairs tested, out4[1] = float mismatches, out4[2] = double;
 * out4[3] = mismatches of the lean correctly rounded square root used in the same domain (ns_sqrt_fast: the
 * sqrtf expansion without its tiny-argument scaling) against sqrtf over EVERY float in [2^-96, 2^126] and 0 */
int sea_selftest_nsdiv(unsigned long long *out4);
/* DC-offset recurrence on ncases frames of 80 differences (host pointers): output and whether the
 * exact double path had to be taken */
int sea_selftest_dc(const float *dif, const float *y0, float *out, int *fellback, int ncases);
/* the kernels' own double-precision natural log (positive normal arguments) on n floats (host pointers) */
int sea_selftest_log(const float *x, double *ln_out, int n);
/* the guard around that log (csrc/ns_core.h, ns_near_float_boundary): both call sites round a double expression
 * of the log to float (NoiseSup.c:391, :607); when the expression lands within a few ulps of a float rounding
 * boundary the log is redone in double-double arithmetic (site 2: through the fdlibm log10 formula glibc uses).
 *   sea_selftest_log_dd     the double-double log itself on n doubles (host pointers): hi + lo
 *   sea_selftest_log_sites  both complete sites on n floats: site1 = VAD frame log-energy of frameSum = x
 *                           (x >= 64), site2 = averSNR of x (x > 1e-5); NaN outside a site's range
 *   sea_selftest_log_guard  sweeps EVERY float argument of site 1 (every finite float >= 64) or 2 (every float > 1e-5) through
 *                           the fast AND the slow form: stats8 = {arguments, guard hits, hits where the slow form
 *                           changed the float, hits recorded, arguments outside the guard window on which the two
 *                           forms disagree (must be 0), 0, 0, 0}; hits3 receives up to cap triples (argument, fast
 *                           float, returned float) */
int sea_selftest_log_dd(const double *x, double *hi, double *lo, int n);
int sea_selftest_log_sites(const float *x, float *site1, float *site2, int n);
int sea_selftest_log_guard(int site, unsigned long long *stats8, float *hits3, int cap);
/* host pipelines (csrc/hostpipe.hip): from the nth hipEventQuery of the process on, every query reports a device fault
 * (0: off).  The pipelines must then return 1 -- the reference's fault code -- instead of polling for ever. */
int sea_selftest_hostpipe_fault(long long nth_query);
/* the 16 k-native variant's pieces that the reference's own rfft.cpp + MelProc.cpp pin (tests/golden/aurora_golden.npz), run on
 * the device by the very functions the pipelined kernel calls (host pointers): rfft (x, 512, 8) of nfft frames from both of the
 * transform wave's work areas ([nfft][512] each), DoGamma of ngain vectors of 129 gains ([ngain][25]) and rows 0..8 of
 * DoGammaIDCT of those ([ngain][9], before the filter window) */
int sea_selftest_ns16k_pieces(const float *frames512, int nfft, float *fft512_a, float *fft512_b, const float *gains129, int ngain,
                              float *gamma25, float *idct9);

#ifdef __cplusplus
}
#endif
#endif
