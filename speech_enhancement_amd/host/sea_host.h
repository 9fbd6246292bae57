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
/* The same in pieces, for staging the library lays out (sea_packed_segments): sea_wav_probe parses the header only (sample
 * count per channel, rate, channel count, byte offset of the data); sea_wav_read_segs reads the first sum(cnt) samples of the
 * left channel straight into seg[0][0..cnt[0]), seg[1][..], ... (mono files: one fread per piece, no intermediate buffer);
 * sea_wav_write_segs writes a mono PCM16 file of sum(cnt) + n_zero_tail samples from such pieces. */
int sea_wav_probe(const char *path, long *n, int *fs, long *data_off, int *channels);
int sea_wav_read_segs(const char *path, long data_off, int channels, short *const *seg, const long *cnt, int nseg);
int sea_wav_write_segs(const char *path, short *const *seg, const long *cnt, int nseg, long n_zero_tail, int fs);

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


/* ---- the drivers' pipeline: reader thread(s) -> device thread(s) -> writer thread(s) ------------------------------
 * A chunk of the utterance list travels through two bounded queues, so that chunk k+1's WAVs are read and chunk
 * k-1's written while chunk k is on a GPU (the reference reads, processes and writes one file at a time,
 * etsi/cpp/main.cpp:43-67; its batch tool does that from a pool of threads, aurora_speech_enhancement.cpp:311-327). */
#include <pthread.h>
typedef struct sea_chunk {
    int first, n;          /* ids[first .. first + n) of the list (resynth: ids[used[u]]) */
    int *used;             /* resynth: list index of utterance u of the chunk */
    short **in, **out;
    float **mask;          /* resynth: [rows][64] per utterance */
    float **ceps;          /* etsi --ceps: [n_ceps][14] per utterance */
    int *n_ceps;
    long *len;
    int rc;
    struct sea_packed *packed; /* etsi: the library's pinned staging for this chunk (sea_packed_*), or NULL; the driver destroys it
                                * before sea_chunk_free (this file does not link against the library) */
    long *data_off;            /* etsi, packed: byte offset of the samples in the WAV file */
    int *channels;
    struct sea_chunk *next;
} sea_chunk;
sea_chunk *sea_chunk_new(int first, int n, int with_mask, int with_ceps);
void sea_chunk_free(sea_chunk *c);

typedef struct {
    pthread_mutex_t mu;
    pthread_cond_t can_put, can_get;
    sea_chunk *head, *tail;
    int count, cap, producers; /* closed when producers reaches 0 */
} sea_queue;
void sea_queue_init(sea_queue *q, int cap, int producers);
void sea_queue_put(sea_queue *q, sea_chunk *c);        /* blocks while the queue is full */
sea_chunk *sea_queue_get(sea_queue *q);                /* blocks; NULL once every producer is done and the queue is empty */
void sea_queue_producer_done(sea_queue *q);
void sea_queue_destroy(sea_queue *q);

#endif
