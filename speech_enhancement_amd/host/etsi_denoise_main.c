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
 * launch per chunk of the list (one wavefront per utterance), through the C ABI of
 * libsea_mi355x.so; each utterance's result equals etsi_denoise() on it alone.  The reference
 * leaves the trailing len%80 samples of its new[]'ed buffer uninitialised; here they are 0.
 *   --dry-run   parse cfg/list/WAVs and report, no GPU work, nothing written
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/sea_mi355x.h"
#include "sea_host.h"

#define CHUNK 4096

int main(int argc, char *argv[])
{
    sea_cfg opts;
    char path[4 * SEA_FILE_LEN], **ids = NULL;
    int n_ids, dry = 0, first, rc = 0;
    FILE *Log;
    if (argc < 2) {
        fprintf(stderr, "usage: %s <cfg> [--dry-run]\n", argv[0]);
        return 2;
    }
    dry = argc > 2 && !strcmp(argv[2], "--dry-run");
    if (sea_read_cfg(argv[1], 1, &opts)) return 2;
    snprintf(path, sizeof path, "%s%s", opts.outputDictionary, opts.Log);
    Log = dry ? NULL : fopen(path, "a+");
    n_ids = sea_read_list(opts.purewavlist, &ids);
    if (n_ids < 0) {
        fprintf(stderr, "Open %s file error!\n", opts.purewavlist);
        return 2;
    }
    for (first = 0; first < n_ids && !rc; first += CHUNK) {
        int n = (n_ids - first < CHUNK) ? n_ids - first : CHUNK, u;
        short **in = (short **)calloc(n, sizeof(short *)), **out = (short **)calloc(n, sizeof(short *));
        long *len = (long *)calloc(n, sizeof(long));
        for (u = 0; u < n; u++) {
            int fs = 0;
            const char *id = ids[first + u];
            printf("%s\n", id);
            if (Log) fprintf(Log, "%s\n ", id);
            snprintf(path, sizeof path, "%s%s%s_noisy.wav", opts.outputDictionary, opts.save_noisy_dir, id);
            printf("%s %d\n", path, first + u);
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
                snprintf(path, sizeof path, "%s%s%s_e_resynth.wav", opts.outputDictionary, opts.save_resynth_e_dir,
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
    if (Log) fclose(Log);
    sea_free_list(ids, n_ids);
    return rc;
}
