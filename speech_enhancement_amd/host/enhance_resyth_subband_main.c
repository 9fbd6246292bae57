/*
 * enhance_resyth_subband_main.c -- file-in/file-out driver of the 64-band resynthesis with the
 * reference's command line (resyth_64sub_ori/deal.sh builds ./enhance_resyth_subband <cfg>;
 * resyth_64sub_ori/cpp/main.cpp:50-148).  Built twice: plain (ratio mask, resyth_64sub_ori) and
 * with -DSEA_IBM=1 (ideal binary mask, resyth_64sub_IBM).
 *
 *   cfg   9 positional "key= value" lines (as etsi's, without numMix)
 *   in    <outputDictionary>result.txt : Kaldi-style text matrices, "<utt> [" then rows of 64
 *         floats, the last row closed by "]" (resyth_64sub_ori/cpp/main.cpp:84-145); the k-th
 *         matrix belongs to the k-th id of purewavlist;
 *         <outputDictionary><save_noisy_dir><id>_noisy.wav
 *   out   <outputDictionary><save_resynth_e_dir><id>_e_resynth.wav
 *
 * Shape (the etsi driver's, host/etsi_denoise_main.c; the reference's parallel harness is the shared-counter thread
 * pool of aurora_speech_enhancement.cpp:111-121, 311-327):
 *   one reader thread   walks result.txt (one stream of text: inherently serial), reads each id's WAV, hands over
 *                       CHUNKS of utterances                                                        -> queue
 *   device threads      one per device (sea_device_count(), or SEA_DEVICES=n): sea_init(dev), then one
 *                       sea_resynth_utterances per chunk (which bounds its 256-B-per-sample HBM scratch by the free
 *                       device memory)                                                              -> queue
 *   writer threads      write the chunk's WAVs, free it
 * Host memory stays bounded for lists of any length (configs[4]: 100 000 utterances): at most 2 chunks per device
 * thread wait in each queue.
 *   --dry-run   parse cfg / list / result.txt / WAVs and report, no GPU work, nothing written
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/sea_mi355x.h"
#include "sea_host.h"

#ifndef SEA_IBM
#define SEA_IBM 0
#endif
#ifndef CHUNK
#define CHUNK 1024
#endif
#define MAX_THREADS 64

typedef struct {
    const sea_cfg *opts;
    char **ids;
    int n_ids, dry, n_dev, chunk;
    FILE *Log, *erm;
    pthread_mutex_t mu;
    int rc;
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

/* result.txt in order: a line containing '[' opens the next id's matrix, every other line is one row of it */
static void *reader(void *arg)
{
    job_t *J = ((worker_t *)arg)->job;
    const sea_cfg *opts = J->opts;
    char path[4 * SEA_FILE_LEN], buf[64 * 32];
    int count = 0, row = 0, rows_needed = 0, rc = 0;
    sea_chunk *c = NULL;
    while (!rc && fgets(buf, sizeof buf, J->erm)) {
        if (strstr(buf, "[")) { /* next matrix -> next id of the list */
            int fs = 0, u;
            if (count >= J->n_ids) break;
            if (c && c->n == J->chunk) { /* the chunk is full and its last matrix is complete: hand it over */
                sea_queue_put(&J->to_device, c);
                c = NULL;
            }
            if (!c) {
                c = sea_chunk_new(count, J->chunk, 1, 0); /* room for a whole chunk; n counts what is in it */
                if (!c || !c->in || !c->out || !c->len || !c->used || !c->mask) {
                    rc = 1;
                    break;
                }
                c->n = 0;
            }
            u = c->n;
            pthread_mutex_lock(&J->mu);
            printf("%s\n", J->ids[count]);
            if (J->Log) fprintf(J->Log, "%s\n ", J->ids[count]);
            pthread_mutex_unlock(&J->mu);
            snprintf(path, sizeof path, "%s%s%s_noisy.wav", opts->outputDictionary, opts->save_noisy_dir, J->ids[count]);
            if (sea_wav_read(path, &c->in[u], &c->len[u], &fs) || c->len[u] < 320) {
                fprintf(stderr, "ERROR:   cannot use %s\n", path);
                free(c->in[u]); /* a WAV shorter than one mask frame was read but is not kept */
                c->in[u] = NULL;
                rc = 3;
                break;
            }
            rows_needed = (int)((c->len[u] - 320) / 160 + 1); /* numFrame, main.cpp:110 */
            printf("%d\n", rows_needed);
            c->mask[u] = (float *)calloc((size_t)rows_needed * 64, sizeof(float));
            c->out[u] = (short *)calloc(c->len[u], sizeof(short));
            c->used[u] = count;
            if (J->dry) printf("  %s: %ld samples, %d mask rows\n", J->ids[count], c->len[u], rows_needed);
            row = 0;
            count++;
            c->n++;
        } else if (c && c->n > 0 && row < rows_needed) {
            char *p = buf, *end;
            float *m = c->mask[c->n - 1];
            int j;
            for (j = 0; j < 64; j++) {
                float v = strtof(p, &end);
                if (end == p) break;
                m[(size_t)row * 64 + j] = v;
                p = end;
            }
            row++;
        }
    }
    if (rc) {
        set_rc(J, rc);
        sea_chunk_free(c); /* nothing of a failed run is resynthesised or reported */
    } else if (c) {
        if (c->n > 0) sea_queue_put(&J->to_device, c);
        else sea_chunk_free(c);
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
        if (ok && !J->dry) {
            pthread_mutex_lock(&J->mu);
            if (J->Log) fprintf(J->Log, "resynth\n ");
            pthread_mutex_unlock(&J->mu);
            if (sea_resynth_utterances((const short *const *)c->in, c->len, (const float *const *)c->mask, SEA_IBM, c->out, c->n)) {
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
                     J->ids[c->used[u]]);
            if (sea_wav_write(path, c->out[u], c->len[u], 16000)) rc = 4;
        }
        if (rc) set_rc(J, rc);
        sea_chunk_free(c);
    }
    return NULL;
}

int main(int argc, char *argv[])
{
    sea_cfg opts;
    char path[4 * SEA_FILE_LEN], **ids = NULL;
    int n_ids, n_dev = 0, n_thr, n_write, k;
    job_t J;
    pthread_t rd, dv[MAX_THREADS], wr[MAX_THREADS];
    worker_t wk[MAX_THREADS];
    const char *e;
    if (argc < 2) {
        fprintf(stderr, "usage: %s <cfg> [--dry-run]\n", argv[0]);
        return 2;
    }
    memset(&J, 0, sizeof J);
    J.dry = argc > 2 && !strcmp(argv[2], "--dry-run");
    if (sea_read_cfg(argv[1], 0, &opts)) return 2;
    snprintf(path, sizeof path, "%s%s", opts.outputDictionary, opts.Log);
    J.Log = J.dry ? NULL : fopen(path, "a+");
    n_ids = sea_read_list(opts.purewavlist, &ids);
    snprintf(path, sizeof path, "%sresult.txt", opts.outputDictionary);
    printf("%s\n", path);
    J.erm = fopen(path, "r");
    if (n_ids < 0 || !J.erm) {
        fprintf(stderr, "Open %s file error!\n", n_ids < 0 ? opts.purewavlist : path);
        return 2;
    }
    if (!J.dry) n_dev = sea_device_count();
    n_thr = n_dev > 0 ? n_dev : 1;
    if ((e = getenv("SEA_DEVICES")) && atoi(e) > 0) n_thr = atoi(e);
    if (n_thr > MAX_THREADS) n_thr = MAX_THREADS;
    J.opts = &opts;
    J.ids = ids;
    J.n_ids = n_ids;
    J.n_dev = n_dev;
    /* chunks: about four per device thread, within [16, CHUNK] utterances (SEA_CHUNK overrides: tests) */
    J.chunk = (n_ids + 4 * n_thr - 1) / (4 * n_thr);
    if (J.chunk < 16) J.chunk = 16;
    if (J.chunk > CHUNK) J.chunk = CHUNK;
    if ((e = getenv("SEA_CHUNK")) && atoi(e) > 0) J.chunk = atoi(e);
    n_write = n_thr;
    pthread_mutex_init(&J.mu, NULL);
    sea_queue_init(&J.to_device, 2 * n_thr, 1);
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
    if (pthread_create(&rd, NULL, reader, &wk[0])) {
        fprintf(stderr, "ERROR:   cannot start the reader thread\n");
        return 1;
    }
    pthread_join(rd, NULL);
    for (k = 0; k < n_thr; k++) pthread_join(dv[k], NULL);
    for (k = 0; k < n_write; k++) pthread_join(wr[k], NULL);
    fclose(J.erm);
    sea_queue_destroy(&J.to_device);
    sea_queue_destroy(&J.to_writer);
    pthread_mutex_destroy(&J.mu);
    if (J.Log) fclose(J.Log);
    sea_free_list(ids, n_ids);
    return J.rc;
}
