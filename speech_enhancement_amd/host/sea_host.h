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

#endif
