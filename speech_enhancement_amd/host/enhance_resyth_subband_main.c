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
 * Utterances are collected and resynthesised together on the GPU (one workgroup per utterance), CHUNK of
 * them at a time: a chunk's WAVs are written and its buffers freed before the next chunk is read, so host
 * memory stays bounded for lists of any length (configs[4]: 100 000 utterances); inside a chunk the library
 * bounds its 256-B-per-sample HBM scratch by the free device memory (sea_resynth_utterances).
 */
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

/* resynthesise and write the n collected utterances, then free them */
static int flush_chunk(const sea_cfg *opts, char **ids, const int *used, short **in, short **out, float **mask,
                       long *len, int n, int dry, FILE *Log)
{
    char path[4 * SEA_FILE_LEN];
    int rc = 0, u;
    if (dry)
        for (u = 0; u < n; u++) printf("  %s: %ld samples, %ld mask rows\n", ids[used[u]], len[u], (len[u] - 320) / 160 + 1);
    if (!dry && n > 0) {
        if (Log) fprintf(Log, "resynth\n ");
        if (sea_resynth_utterances((const short *const *)in, len, (const float *const *)mask, SEA_IBM, out, n)) {
            fprintf(stderr, "ERROR:   %s\n", sea_last_error());
            rc = 1;
        }
        for (u = 0; u < n && !rc; u++) {
            snprintf(path, sizeof path, "%s%s%s_e_resynth.wav", opts->outputDictionary, opts->save_resynth_e_dir,
                     ids[used[u]]);
            if (sea_wav_write(path, out[u], len[u], 16000)) rc = 4;
        }
    }
    for (u = 0; u < n; u++) {
        free(in[u]);
        free(out[u]);
        free(mask[u]);
        in[u] = out[u] = NULL;
        mask[u] = NULL;
    }
    return rc;
}

int main(int argc, char *argv[])
{
    sea_cfg opts;
    char path[4 * SEA_FILE_LEN], buf[64 * 32], **ids = NULL;
    int n_ids, dry, count = 0, rc = 0, row = 0, rows_needed = 0, cap = 0, n = 0;
    short **in = NULL, **out = NULL;
    float **mask = NULL;
    long *len = NULL;
    int *used = NULL;
    FILE *Log, *erm;
    if (argc < 2) {
        fprintf(stderr, "usage: %s <cfg> [--dry-run]\n", argv[0]);
        return 2;
    }
    dry = argc > 2 && !strcmp(argv[2], "--dry-run");
    if (sea_read_cfg(argv[1], 0, &opts)) return 2;
    snprintf(path, sizeof path, "%s%s", opts.outputDictionary, opts.Log);
    Log = dry ? NULL : fopen(path, "a+");
    n_ids = sea_read_list(opts.purewavlist, &ids);
    snprintf(path, sizeof path, "%sresult.txt", opts.outputDictionary);
    printf("%s\n", path);
    erm = fopen(path, "r");
    if (n_ids < 0 || !erm) {
        fprintf(stderr, "Open %s file error!\n", n_ids < 0 ? opts.purewavlist : path);
        return 2;
    }
    cap = CHUNK;
    in = (short **)calloc(cap, sizeof *in);
    out = (short **)calloc(cap, sizeof *out);
    mask = (float **)calloc(cap, sizeof *mask);
    len = (long *)calloc(cap, sizeof *len);
    used = (int *)calloc(cap, sizeof *used);
    while (!rc && fgets(buf, sizeof buf, erm)) {
        if (strstr(buf, "[")) { /* next matrix -> next id of the list */
            int fs = 0;
            if (count >= n_ids) break;
            if (n == cap) { /* the chunk is full and its last matrix is complete: run it */
                rc = flush_chunk(&opts, ids, used, in, out, mask, len, n, dry, Log);
                n = 0;
                if (rc) break;
            }
            printf("%s\n", ids[count]);
            if (Log) fprintf(Log, "%s\n ", ids[count]);
            snprintf(path, sizeof path, "%s%s%s_noisy.wav", opts.outputDictionary, opts.save_noisy_dir, ids[count]);
            if (sea_wav_read(path, &in[n], &len[n], &fs) || len[n] < 320) {
                fprintf(stderr, "ERROR:   cannot use %s\n", path);
                free(in[n]); /* a WAV shorter than one mask frame was read but is not kept */
                in[n] = NULL;
                rc = 3;
                break;
            }
            rows_needed = (int)((len[n] - 320) / 160 + 1); /* numFrame, main.cpp:110 */
            printf("%d\n", rows_needed);
            mask[n] = (float *)calloc((size_t)rows_needed * 64, sizeof(float));
            out[n] = (short *)calloc(len[n], sizeof(short));
            used[n] = count;
            row = 0;
            count++;
            n++;
        } else if (n > 0 && row < rows_needed) {
            char *p = buf, *end;
            int j;
            for (j = 0; j < 64; j++) {
                float v = strtof(p, &end);
                if (end == p) break;
                mask[n - 1][(size_t)row * 64 + j] = v;
                p = end;
            }
            row++;
        }
    }
    fclose(erm);
    if (!rc)
        rc = flush_chunk(&opts, ids, used, in, out, mask, len, n, dry, Log);
    else
        (void)flush_chunk(&opts, ids, used, in, out, mask, len, n, 1 /* free only */, NULL);
    free(in); free(out); free(mask); free(len); free(used);
    if (Log) fclose(Log);
    sea_free_list(ids, n_ids);
    return rc;
}
