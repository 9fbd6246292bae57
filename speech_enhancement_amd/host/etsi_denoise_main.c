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
 * Several GPUs: the reference's only parallel harness is a pool of threads that pull file indices from a
 * shared counter (function/20141106_speech_enhancement/aurora_speech_enhancement/aurora_speech_enhancement.cpp:
 * 111-121, 311-327).  Same shape here, one step coarser: one host thread per device (sea_device_count(), or
 * SEA_DEVICES=n; more threads than devices share them round robin) pulls CHUNKS of the list from a shared
 * counter and runs each on its own device -- no data crosses between devices.
 *   --dry-run   parse cfg/list/WAVs and report, no GPU work, nothing written
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/sea_mi355x.h"
#include "sea_host.h"

#define CHUNK_MAX 4096
#define CHUNK_MIN 64

typedef struct {
    const sea_cfg *opts;
    char **ids;
    int n_ids, chunk, dry, n_dev;
    FILE *Log;
    pthread_mutex_t mu; /* the shared chunk counter, stdout / Log lines */
    int next, rc;
} job_t;

typedef struct {
    job_t *job;
    int index;
} worker_t;

static void *worker(void *arg)
{
    worker_t *w = (worker_t *)arg;
    job_t *J = w->job;
    const sea_cfg *opts = J->opts;
    char path[4 * SEA_FILE_LEN];
    if (!J->dry && sea_init(J->n_dev > 0 ? w->index % J->n_dev : -1)) {
        fprintf(stderr, "ERROR:   %s\n", sea_last_error());
        pthread_mutex_lock(&J->mu);
        J->rc = 1;
        pthread_mutex_unlock(&J->mu);
        return NULL;
    }
    for (;;) {
        int first, rc = 0;
        pthread_mutex_lock(&J->mu);
        first = J->next;
        if (J->rc || first >= J->n_ids) {
            pthread_mutex_unlock(&J->mu);
            break;
        }
        J->next = first + J->chunk;
        pthread_mutex_unlock(&J->mu);
        {
        char **ids = J->ids;
        FILE *Log = J->Log;
        const int n_ids = J->n_ids, dry = J->dry, CHUNK = J->chunk;
        int n = (n_ids - first < CHUNK) ? n_ids - first : CHUNK, u;
        short **in = (short **)calloc(n, sizeof(short *)), **out = (short **)calloc(n, sizeof(short *));
        long *len = (long *)calloc(n, sizeof(long));
        for (u = 0; u < n; u++) {
            int fs = 0;
            const char *id = ids[first + u];
            snprintf(path, sizeof path, "%s%s%s_noisy.wav", opts->outputDictionary, opts->save_noisy_dir, id);
            pthread_mutex_lock(&J->mu);
            printf("%s\n", id);
            if (Log) fprintf(Log, "%s\n ", id);
            printf("%s %d\n", path, first + u);
            pthread_mutex_unlock(&J->mu);
            if (sea_wav_read(path, &in[u], &len[u], &fs)) {
                fprintf(stderr, "ERROR:   cannot read %s\n", path);
                rc = 3;
                len[u] = 0;
                continue;
            }
            out[u] = (short *)calloc(len[u] ? len[u] : 1, sizeof(short));
            if (dry) printf("  %ld samples, %d Hz\n", len[u], fs);
        }
        if (!dry && !rc) {
            if (sea_denoise_utterances((const short *const *)in, out, len, n)) {
                fprintf(stderr, "ERROR:   %s\n", sea_last_error());
                rc = 1;
            }
            for (u = 0; u < n && !rc; u++) {
                snprintf(path, sizeof path, "%s%s%s_e_resynth.wav", opts->outputDictionary, opts->save_resynth_e_dir,
                         ids[first + u]);
                if (sea_wav_write(path, out[u], len[u], 16000)) rc = 4;
            }
        }
        for (u = 0; u < n; u++) {
            free(in[u]);
            free(out[u]);
        }
        free(in);
        free(out);
        free(len);
        }
        if (rc) {
            pthread_mutex_lock(&J->mu);
            if (!J->rc) J->rc = rc;
            pthread_mutex_unlock(&J->mu);
        }
    }
    return NULL;
}

int main(int argc, char *argv[])
{
    sea_cfg opts;
    char path[4 * SEA_FILE_LEN], **ids = NULL;
    int n_ids, dry = 0, n_dev = 0, n_thr, k;
    job_t J;
    pthread_t thr[64];
    worker_t wk[64];
    const char *e;
    if (argc < 2) {
        fprintf(stderr, "usage: %s <cfg> [--dry-run]\n", argv[0]);
        return 2;
    }
    dry = argc > 2 && !strcmp(argv[2], "--dry-run");
    if (sea_read_cfg(argv[1], 1, &opts)) return 2;
    n_ids = sea_read_list(opts.purewavlist, &ids);
    if (n_ids < 0) {
        fprintf(stderr, "Open %s file error!\n", opts.purewavlist);
        return 2;
    }
    if (!dry) n_dev = sea_device_count();
    n_thr = n_dev > 0 ? n_dev : 1;
    if ((e = getenv("SEA_DEVICES")) && atoi(e) > 0) n_thr = atoi(e);
    if (n_thr > 64) n_thr = 64;
    memset(&J, 0, sizeof J);
    J.opts = &opts;
    J.ids = ids;
    J.n_ids = n_ids;
    J.dry = dry;
    J.n_dev = n_dev;
    /* chunks: about four per thread so that the shared counter balances unequal chunks, within [64, 4096] utterances */
    J.chunk = (n_ids + 4 * n_thr - 1) / (4 * n_thr);
    if (J.chunk < CHUNK_MIN) J.chunk = CHUNK_MIN;
    if (J.chunk > CHUNK_MAX) J.chunk = CHUNK_MAX;
    snprintf(path, sizeof path, "%s%s", opts.outputDictionary, opts.Log);
    J.Log = dry ? NULL : fopen(path, "a+");
    pthread_mutex_init(&J.mu, NULL);
    if (n_thr > (n_ids + J.chunk - 1) / J.chunk) n_thr = (n_ids + J.chunk - 1) / J.chunk;
    if (n_thr < 1) n_thr = 1;
    for (k = 0; k < n_thr; k++) {
        wk[k].job = &J;
        wk[k].index = k;
        if (k > 0 && pthread_create(&thr[k], NULL, worker, &wk[k])) {
            fprintf(stderr, "ERROR:   cannot start worker thread %d\n", k);
            n_thr = k;
            break;
        }
    }
    worker(&wk[0]);
    for (k = 1; k < n_thr; k++) pthread_join(thr[k], NULL);
    pthread_mutex_destroy(&J.mu);
    if (J.Log) fclose(J.Log);
    sea_free_list(ids, n_ids);
    return J.rc;
}
