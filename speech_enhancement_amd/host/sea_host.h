/*
 * sea_host.h -- file-in/file-out plumbing of the thin C host: RIFF/PCM16 WAV reader/writer (the
 * reference used the private asdk::CWave, which is not available: etsi/cpp/main.cpp:7,53-64),
 * the positional "key= value" cfg parser (etsi/cpp/main.cpp:73-141, resyth_64sub_ori/cpp/main.cpp:
 * 150-214) and the utterance-id list reader with the reference's chop() (etsi/cpp/main.cpp:144-148).
 */
#ifndef SEA_HOST_H
#define SEA_HOST_H

#define SEA_FILE_LEN 1024

typedef struct {
    char purewavDictionary[SEA_FILE_LEN];
    char purewavlist[SEA_FILE_LEN];
    int numMix;
    char outputDictionary[SEA_FILE_LEN];
    char save_noisy_dir[SEA_FILE_LEN];
    char save_noisy_ebm_dir[SEA_FILE_LEN];
    char save_noisy_sirm_dir[SEA_FILE_LEN];
    char save_resynth_e_dir[SEA_FILE_LEN];
    char save_resynth_i_dir[SEA_FILE_LEN];
    char Log[SEA_FILE_LEN];
} sea_cfg;

/* has_numMix = 1: the 10-line etsi cfg; 0: the 9-line resynth cfg.  Returns 0 on success. */
int sea_read_cfg(const char *path, int has_numMix, sea_cfg *cfg);

/* Reads the list of utterance ids (one per line, last character dropped like chop()).
 * Returns the number of ids; *ids is a malloc'ed array of malloc'ed strings. */
int sea_read_list(const char *path, char ***ids);
void sea_free_list(char **ids, int n);

/* Left channel of a PCM16 RIFF file.  Returns 0 and a malloc'ed buffer, or non-zero. */
int sea_wav_read(const char *path, short **data, long *n, int *fs);
int sea_wav_write(const char *path, const short *data, long n, int fs);

/* SURVEY 8(f) #2 -- the on-disk contract between feature extraction, the external DNN and resynth:
 * Kaldi-style text matrices of 64 columns.  sea_mask_text_write emits exactly what make_single_IBM
 * prints (enhancement_extract_test/cpp/show_IBM.cpp:194-208): "<id> [\n", rows of "%.7f " joined
 * by "\n ", "]\n" after the last row.  sea_mask_text_read parses the next matrix the way
 * resyth_64sub_ori/cpp/main.cpp:84-145 does (a line containing '[' opens a matrix, every other line
 * contributes up to 64 floats, text after them -- the closing ']' -- is ignored); it fills at most
 * max_rows rows and returns the number of rows read, -1 at end of file. */
#include <stdio.h>
int sea_mask_text_write(FILE *fp, const char *id, const float *mask64, long rows);
long sea_mask_text_read(FILE *fp, char *id_out /* SEA_FILE_LEN, may be NULL */, float *mask64, long max_rows);

#endif
