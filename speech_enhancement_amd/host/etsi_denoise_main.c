/*
 * etsi_denoise_main.c -- file-in/file-out driver with the reference's command line and cfg
 * (etsi/deal.sh builds it as ./etsi_denoise <cfg>; etsi/cpp/main.cpp:17-71):
 *
 *   cfg   10 positional "key= value" lines: purewavDictionary, purewavlist, numMix,
 *         outputDictionary, save_noisy_dir, save_noisy_ebm_dir, save_noisy_sirm_dir,
 *         save_resynth_e_dir, save_resynth_i_dir, Log
 *   in    <outputDictionary><save_noisy_dir><id>_noisy.wav      for every id of purewavlist
 *   out   <outputDictionary><save_resynth_e_dir><id>_e_resynth.wav   (16 kHz mono PCM16)
 *
 * Difference to the reference loop: the utterances of the list are denoised TOGETHER, one GPU
 * launch per chunk of the list (one workgroup per utterance), through the C ABI of
 * libsea_mi355x.so; each utterance's result equals etsi_denoise() on it alone.  The reference
 * leaves the trailing len%80 samples of its new[]'ed buffer uninitialised; here they are 0.
 *
 * Shape: the reference's only parallel harness is a pool of threads that pull file indices from a shared counter
 * (function/20141106_speech_enhancement/aurora_speech_enhancement/aurora_speech_enhancement.cpp:111-121, 311-327).
 * Here, one step coarser and with the file I/O taken off the GPU's critical path:
 *   reader threads   pull CHUNKS of the list from a shared counter, read their WAVs          -> queue
 *   device threads   one per device (sea_device_count(), or SEA_DEVICES=n; more threads than devices share them
 *                    round robin): sea_init(dev), then one launch pipeline per chunk         -> queue
 *   writer threads   write the chunk's WAVs, free it
 * so chunk k+1 is being read and chunk k-1 written while chunk k is on a GPU; no data crosses between devices.
 * Round 4: the samples travel WITHOUT pack / unpack copies (sea_packed_*): a reader probes the chunk's WAV headers, lets the
 * library lay the chunk out in its pinned staging (sea_packed_plan) and freads every file straight into the pieces
 * sea_packed_segments names; the device thread runs sea_packed_denoise; a writer fwrites the output pieces.  (--ceps keeps
 * the pointer-array entry point sea_denoise_ceps_utterances; SEA_HOST_PACKED=0 switches the plain path back too.)
 *
 *   --dry-run   parse cfg/list/WAVs and report, no GPU work, nothing written
 *   --ceps      also write <id>_e_resynth.ceps next to each output WAV: the cepstra DoCompCeps gives on the denoised
 *               stream (etsi/cpp/ParmInterface.c:275-293, commented out in the reference): int32 rows, int32 14, then
 *               rows x 14 float32 (c1..c12, c0, logE)
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/sea_mi355x.h"
#include "sea_host.h"

#define CHUNK_MAX 4096
#define CHUNK_MIN 64
#define MAX_THREADS 64

typedef struct {
    const sea_cfg *opts;
    char **ids;
    int n_ids, chunk, dry, n_dev, ceps, packed;
    FILE *Log;
    pthread_mutex_t mu; /* the shared chunk counter, stdout / Log lines, rc */
    int next, rc;
    sea_queue to_device, to_writer;
} job_t;

typedef struct {
    job_t *job;
    int index;
} worker_t;

static void set_rc(job_t *J, int rc)
{
    pthread_mutex_lock(&J->mu);
    if (!J->rc) J->rc = rc;
    pthread_mutex_unlock(&J->mu);
}

static void *reader(void *arg)
{
    job_t *J = ((worker_t *)arg)->job;
    const sea_cfg *opts = J->opts;
    char path[4 * SEA_FILE_LEN];
    for (;;) {
        int first, n, u;
        sea_chunk *c;
        pthread_mutex_lock(&J->mu);
        first = J->next;
        if (J->rc || first >= J->n_ids) {
            pthread_mutex_unlock(&J->mu);
            break;
        }
        J->next = first + J->chunk;
        pthread_mutex_unlock(&J->mu);
        n = (J->n_ids - first < J->chunk) ? J->n_ids - first : J->chunk;
        c = sea_chunk_new(first, n, 0, J->ceps);
        if (!c) {
            set_rc(J, 1);
            break;
        }
        if (J->packed) {
            c->data_off = (long *)calloc((size_t)n, sizeof(long));
            c->channels = (int *)calloc((size_t)n, sizeof(int));
        }
        for (u = 0; u < n; u++) {
            int fs = 0;
            const char *id = J->ids[first + u];
            snprintf(path, sizeof path, "%s%s%s_noisy.wav", opts->outputDictionary, opts->save_noisy_dir, id);
            pthread_mutex_lock(&J->mu);
            printf("%s\n", id);
            if (J->Log) fprintf(J->Log, "%s\n ", id);
            printf("%s %d\n", path, first + u);
            pthread_mutex_unlock(&J->mu);
            if (J->packed) { /* header only: the samples are read once the library has laid the chunk out */
                if (sea_wav_probe(path, &c->len[u], &fs, &c->data_off[u], &c->channels[u])) {
                    fprintf(stderr, "ERROR:   cannot read %s\n", path);
                    c->rc = 3;
                    c->len[u] = 0;
                }
                continue;
            }
            if (sea_wav_read(path, &c->in[u], &c->len[u], &fs)) {
                fprintf(stderr, "ERROR:   cannot read %s\n", path);
                c->rc = 3;
                c->len[u] = 0;
                continue;
            }
            c->out[u] = (short *)calloc(c->len[u] ? c->len[u] : 1, sizeof(short));
            if (J->ceps) {
                const long cap = c->len[u] / 80 - 6;
                c->ceps[u] = (float *)calloc((size_t)(cap > 0 ? cap : 1) * 14, sizeof(float));
            }
            if (J->dry) printf("  %ld samples, %d Hz\n", c->len[u], fs);
        }
        if (J->packed && !c->rc) {
            /* the library's pinned staging for this chunk; every file read straight into its pieces */
            c->packed = sea_packed_create();
            if (!c->packed || sea_packed_plan(c->packed, c->len, n)) {
                fprintf(stderr, "ERROR:   %s\n", sea_last_error());
                c->rc = 1;
            } else {
                const int K = sea_packed_slices(c->packed) > 0 ? sea_packed_slices(c->packed) : 1;
                short **seg = (short **)malloc((size_t)K * sizeof *seg);
                long *cnt = (long *)malloc((size_t)K * sizeof *cnt);
                for (u = 0; u < n && seg && cnt; u++) {
                    const int k = sea_packed_segments(c->packed, u, seg, NULL, cnt, K);
                    if (k <= 0) continue; /* shorter than one frame: nothing to denoise */
                    snprintf(path, sizeof path, "%s%s%s_noisy.wav", opts->outputDictionary, opts->save_noisy_dir, J->ids[first + u]);
                    if (sea_wav_read_segs(path, c->data_off[u], c->channels[u], seg, cnt, k)) {
                        fprintf(stderr, "ERROR:   cannot read %s\n", path);
                        c->rc = 3;
                    }
                }
                if (!seg || !cnt) c->rc = 1;
                free(seg);
                free(cnt);
            }
        }
        if (c->rc) set_rc(J, c->rc);
        sea_queue_put(&J->to_device, c);
    }
    sea_queue_producer_done(&J->to_device);
    return NULL;
}

static void *device_thread(void *arg)
{
    worker_t *w = (worker_t *)arg;
    job_t *J = w->job;
    sea_chunk *c;
    int ok = 1;
    if (!J->dry && sea_init(J->n_dev > 0 ? w->index % J->n_dev : -1)) {
        fprintf(stderr, "ERROR:   %s\n", sea_last_error());
        set_rc(J, 1);
        ok = 0;
    }
    while ((c = sea_queue_get(&J->to_device)) != NULL) {
        if (ok && !J->dry && !c->rc) {
            const int bad = c->packed ? sea_packed_denoise(c->packed) : J->ceps ? sea_denoise_ceps_utterances((const short *const *)c->in, c->out, c->ceps, c->n_ceps, c->len, c->n)
                                    : sea_denoise_utterances((const short *const *)c->in, c->out, c->len, c->n);
            if (bad) {
                fprintf(stderr, "ERROR:   %s\n", sea_last_error());
                c->rc = 1;
                set_rc(J, 1);
            }
        } else if (!ok)
            c->rc = 1;
        sea_queue_put(&J->to_writer, c);
    }
    sea_queue_producer_done(&J->to_writer);
    return NULL;
}

static void *writer(void *arg)
{
    job_t *J = ((worker_t *)arg)->job;
    const sea_cfg *opts = J->opts;
    char path[4 * SEA_FILE_LEN];
    sea_chunk *c;
    while ((c = sea_queue_get(&J->to_writer)) != NULL) {
        int u, rc = c->rc;
        for (u = 0; u < c->n && !rc && !J->dry; u++) {
            snprintf(path, sizeof path, "%s%s%s_e_resynth.wav", opts->outputDictionary, opts->save_resynth_e_dir,
                     J->ids[c->first + u]);
            if (c->packed) { /* the output pieces in time order, then the trailing partial frame as zeros */
                const int K = sea_packed_slices(c->packed) > 0 ? sea_packed_slices(c->packed) : 1;
                short **seg = (short **)malloc((size_t)K * sizeof *seg);
                long *cnt = (long *)malloc((size_t)K * sizeof *cnt);
                const int k = (seg && cnt) ? sea_packed_segments(c->packed, u, NULL, seg, cnt, K) : -1;
                if (k < 0 || sea_wav_write_segs(path, seg, cnt, k, c->len[u] - c->len[u] / 80 * 80, 16000)) rc = 4;
                free(seg);
                free(cnt);
            } else if (sea_wav_write(path, c->out[u], c->len[u], 16000)) rc = 4;
            if (!rc && J->ceps) {
                FILE *f;
                const int hdr[2] = {c->n_ceps[u], 14};
                snprintf(path, sizeof path, "%s%s%s_e_resynth.ceps", opts->outputDictionary, opts->save_resynth_e_dir,
                         J->ids[c->first + u]);
                f = fopen(path, "wb");
                if (!f || fwrite(hdr, sizeof hdr, 1, f) != 1 ||
                    (c->n_ceps[u] > 0 && fwrite(c->ceps[u], 14 * sizeof(float), (size_t)c->n_ceps[u], f) != (size_t)c->n_ceps[u]))
                    rc = 4;
                if (f) fclose(f);
            }
        }
        if (rc) set_rc(J, rc);
        if (c->packed) {
            sea_packed_destroy(c->packed);
            c->packed = NULL;
        }
        sea_chunk_free(c);
    }
    return NULL;
}

int main(int argc, char *argv[])
{
    sea_cfg opts;
    char path[4 * SEA_FILE_LEN], **ids = NULL;
    int n_ids, n_dev = 0, n_thr, n_read, n_write, n_chunks, k, a;
    job_t J;
    pthread_t rd[MAX_THREADS], dv[MAX_THREADS], wr[MAX_THREADS];
    worker_t wk[MAX_THREADS];
    const char *e;
    if (argc < 2) {
        fprintf(stderr, "usage: %s <cfg> [--dry-run] [--ceps]\n", argv[0]);
        return 2;
    }
    memset(&J, 0, sizeof J);
    for (a = 2; a < argc; a++) {
        if (!strcmp(argv[a], "--dry-run")) J.dry = 1;
        else if (!strcmp(argv[a], "--ceps")) J.ceps = 1;
        else {
            fprintf(stderr, "usage: %s <cfg> [--dry-run] [--ceps]\n", argv[0]);
            return 2;
        }
    }
    if (sea_read_cfg(argv[1], 1, &opts)) return 2;
    n_ids = sea_read_list(opts.purewavlist, &ids);
    if (n_ids < 0) {
        fprintf(stderr, "Open %s file error!\n", opts.purewavlist);
        return 2;
    }
    if (!J.dry) n_dev = sea_device_count();
    J.packed = !J.dry && !J.ceps && !((e = getenv("SEA_HOST_PACKED")) && e[0] == '0');
    n_thr = n_dev > 0 ? n_dev : 1;
    if ((e = getenv("SEA_DEVICES")) && atoi(e) > 0) n_thr = atoi(e);
    if (n_thr > MAX_THREADS) n_thr = MAX_THREADS;
    J.opts = &opts;
    J.ids = ids;
    J.n_ids = n_ids;
    J.n_dev = n_dev;
    /* chunks: about four per device thread so that the shared counter balances unequal chunks, within [64, 4096] utterances */
    J.chunk = (n_ids + 4 * n_thr - 1) / (4 * n_thr);
    if (J.chunk < CHUNK_MIN) J.chunk = CHUNK_MIN;
    if (J.chunk > CHUNK_MAX) J.chunk = CHUNK_MAX;
    n_chunks = (n_ids + J.chunk - 1) / J.chunk;
    if (n_thr > n_chunks) n_thr = n_chunks;
    if (n_thr < 1) n_thr = 1;
    /* file I/O: a few threads each side keep one device busy (a chunk is read in ~ms per file, on the GPU in ~7 ms per
     * thousand files); SEA_IO_THREADS overrides */
    n_read = 2 * n_thr;
    if ((e = getenv("SEA_IO_THREADS")) && atoi(e) > 0) n_read = atoi(e);
    if (n_read > n_chunks) n_read = n_chunks;
    if (n_read > MAX_THREADS) n_read = MAX_THREADS;
    if (n_read < 1) n_read = 1;
    n_write = n_read;
    snprintf(path, sizeof path, "%s%s", opts.outputDictionary, opts.Log);
    J.Log = J.dry ? NULL : fopen(path, "a+");
    pthread_mutex_init(&J.mu, NULL);
    sea_queue_init(&J.to_device, 2 * n_thr, n_read);
    sea_queue_init(&J.to_writer, 2 * n_thr, n_thr);
    for (k = 0; k < MAX_THREADS; k++) {
        wk[k].job = &J;
        wk[k].index = k;
    }
    for (k = 0; k < n_write; k++)
        if (pthread_create(&wr[k], NULL, writer, &wk[k])) {
            fprintf(stderr, "ERROR:   cannot start writer thread %d\n", k);
            return 1;
        }
    for (k = 0; k < n_thr; k++)
        if (pthread_create(&dv[k], NULL, device_thread, &wk[k])) {
            fprintf(stderr, "ERROR:   cannot start device thread %d\n", k);
            return 1;
        }
    for (k = 0; k < n_read; k++)
        if (pthread_create(&rd[k], NULL, reader, &wk[k])) {
            fprintf(stderr, "ERROR:   cannot start reader thread %d\n", k);
            return 1;
        }
    for (k = 0; k < n_read; k++) pthread_join(rd[k], NULL);
    for (k = 0; k < n_thr; k++) pthread_join(dv[k], NULL);
    for (k = 0; k < n_write; k++) pthread_join(wr[k], NULL);
    sea_queue_destroy(&J.to_device);
    sea_queue_destroy(&J.to_writer);
    pthread_mutex_destroy(&J.mu);
    if (J.Log) fclose(J.Log);
    sea_free_list(ids, n_ids);
    return J.rc;
}
