/* sea_host.c -- see sea_host.h */
#include "sea_host.h"

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static void chop_last(char *s)
{ /* the reference's chop(): drops the last character whatever it is */
    size_t n = strlen(s);
    if (n) s[n - 1] = '\0';
}

static int cfg_line(FILE *fp, char *value, int *ivalue)
{
    char line[2 * SEA_FILE_LEN], key[SEA_FILE_LEN];
    if (!fgets(line, sizeof line, fp)) return 1;
    chop_last(line);
    if (value) {
        value[0] = '\0';
        sscanf(line, "%1023s %1023s", key, value);
    } else
        sscanf(line, "%1023s %d", key, ivalue);
    return 0;
}

int sea_read_cfg(const char *path, int has_numMix, sea_cfg *c)
{
    FILE *fp = fopen(path, "r");
    int bad = 0;
    memset(c, 0, sizeof *c);
    if (!fp) {
        printf("Open %s file error!\n", path);
        return 1;
    }
    bad |= cfg_line(fp, c->purewavDictionary, NULL);
    bad |= cfg_line(fp, c->purewavlist, NULL);
    if (has_numMix) bad |= cfg_line(fp, NULL, &c->numMix);
    bad |= cfg_line(fp, c->outputDictionary, NULL);
    bad |= cfg_line(fp, c->save_noisy_dir, NULL);
    bad |= cfg_line(fp, c->save_noisy_ebm_dir, NULL);
    bad |= cfg_line(fp, c->save_noisy_sirm_dir, NULL);
    bad |= cfg_line(fp, c->save_resynth_e_dir, NULL);
    bad |= cfg_line(fp, c->save_resynth_i_dir, NULL);
    bad |= cfg_line(fp, c->Log, NULL);
    fclose(fp);
    return bad;
}

int sea_read_list(const char *path, char ***ids)
{
    FILE *fp = fopen(path, "r");
    char line[SEA_FILE_LEN];
    int n = 0, cap = 0;
    *ids = NULL;
    if (!fp) return -1;
    while (fgets(line, sizeof line, fp)) {
        chop_last(line);
        if (n == cap) {
            cap = cap ? 2 * cap : 64;
            *ids = (char **)realloc(*ids, cap * sizeof(char *));
        }
        (*ids)[n++] = strdup(line);
    }
    fclose(fp);
    return n;
}

void sea_free_list(char **ids, int n)
{
    int i;
    for (i = 0; i < n; i++) free(ids[i]);
    free(ids);
}

static uint32_t rd32(const unsigned char *p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
static uint16_t rd16(const unsigned char *p) { return (uint16_t)(p[0] | (p[1] << 8)); }

int sea_wav_read(const char *path, short **data, long *n, int *fs)
{
    FILE *fp = fopen(path, "rb");
    unsigned char h[12], ck[8], fmt[16];
    int channels = 0, bits = 0, format = 0, have_fmt = 0;
    *data = NULL;
    *n = 0;
    if (!fp) return 1;
    if (fread(h, 1, 12, fp) != 12 || memcmp(h, "RIFF", 4) || memcmp(h + 8, "WAVE", 4)) goto bad;
    while (fread(ck, 1, 8, fp) == 8) {
        uint32_t size = rd32(ck + 4);
        if (!memcmp(ck, "fmt ", 4)) {
            if (size < 16 || fread(fmt, 1, 16, fp) != 16) goto bad;
            format = rd16(fmt);
            channels = rd16(fmt + 2);
            *fs = (int)rd32(fmt + 4);
            bits = rd16(fmt + 14);
            have_fmt = 1;
            fseek(fp, (long)(size - 16 + (size & 1)), SEEK_CUR);
        } else if (!memcmp(ck, "data", 4)) {
            long frames, i;
            unsigned char *raw;
            if (!have_fmt || format != 1 || bits != 16 || channels < 1) goto bad;
            frames = (long)(size / (2u * (unsigned)channels));
            raw = (unsigned char *)malloc(size ? size : 1);
            frames = (long)(fread(raw, 2u * (unsigned)channels, (size_t)frames, fp));
            *data = (short *)malloc((size_t)(frames ? frames : 1) * sizeof(short));
            for (i = 0; i < frames; i++) (*data)[i] = (short)rd16(raw + (size_t)i * 2u * (unsigned)channels);
            free(raw);
            *n = frames;
            fclose(fp);
            return 0;
        } else
            fseek(fp, (long)(size + (size & 1)), SEEK_CUR);
    }
bad:
    fclose(fp);
    return 2;
}

int sea_wav_probe(const char *path, long *n, int *fs, long *data_off, int *channels)
{
    FILE *fp = fopen(path, "rb");
    unsigned char h[12], ck[8], fmt[16];
    int ch = 0, bits = 0, format = 0, have_fmt = 0;
    *n = 0;
    if (!fp) return 1;
    if (fread(h, 1, 12, fp) != 12 || memcmp(h, "RIFF", 4) || memcmp(h + 8, "WAVE", 4)) goto bad;
    while (fread(ck, 1, 8, fp) == 8) {
        uint32_t size = rd32(ck + 4);
        if (!memcmp(ck, "fmt ", 4)) {
            if (size < 16 || fread(fmt, 1, 16, fp) != 16) goto bad;
            format = rd16(fmt);
            ch = rd16(fmt + 2);
            *fs = (int)rd32(fmt + 4);
            bits = rd16(fmt + 14);
            have_fmt = 1;
            fseek(fp, (long)(size - 16 + (size & 1)), SEEK_CUR);
        } else if (!memcmp(ck, "data", 4)) {
            long here, end;
            if (!have_fmt || format != 1 || bits != 16 || ch < 1) goto bad;
            here = ftell(fp);
            fseek(fp, 0, SEEK_END);
            end = ftell(fp);
            if ((long)size > end - here) size = (uint32_t)(end - here); /* a truncated file holds what it holds */
            *n = (long)(size / (2u * (unsigned)ch));
            *data_off = here;
            *channels = ch;
            fclose(fp);
            return 0;
        } else
            fseek(fp, (long)(size + (size & 1)), SEEK_CUR);
    }
bad:
    fclose(fp);
    return 2;
}

int sea_wav_read_segs(const char *path, long data_off, int channels, short *const *seg, const long *cnt, int nseg)
{
    FILE *fp = fopen(path, "rb");
    int k, rc = 0;
    if (!fp) return 1;
    if (fseek(fp, data_off, SEEK_SET)) rc = 2;
    for (k = 0; k < nseg && !rc; k++) {
        if (channels == 1) { /* little-endian host: the samples go where they are used, one read per piece */
            if (fread(seg[k], sizeof(short), (size_t)cnt[k], fp) != (size_t)cnt[k]) rc = 2;
        } else {
            long i;
            unsigned char *raw = (unsigned char *)malloc((size_t)(cnt[k] ? cnt[k] : 1) * 2u * (unsigned)channels);
            if (!raw || fread(raw, 2u * (unsigned)channels, (size_t)cnt[k], fp) != (size_t)cnt[k]) rc = 2;
            for (i = 0; i < cnt[k] && !rc; i++) seg[k][i] = (short)rd16(raw + (size_t)i * 2u * (unsigned)channels);
            free(raw);
        }
    }
    fclose(fp);
    return rc;
}

static void wr32(unsigned char *p, uint32_t v) { p[0] = v & 255; p[1] = (v >> 8) & 255; p[2] = (v >> 16) & 255; p[3] = (v >> 24) & 255; }
static void wr16(unsigned char *p, uint16_t v) { p[0] = v & 255; p[1] = (v >> 8) & 255; }

int sea_wav_write(const char *path, const short *data, long n, int fs)
{
    FILE *fp = fopen(path, "wb");
    unsigned char h[44];
    long i;
    if (!fp) {
        fprintf(stderr, "Cannot write in the file %s\n", path);
        return 1;
    }
    memcpy(h, "RIFF", 4);
    wr32(h + 4, (uint32_t)(36 + 2 * n));
    memcpy(h + 8, "WAVEfmt ", 8);
    wr32(h + 16, 16);
    wr16(h + 20, 1);
    wr16(h + 22, 1);
    wr32(h + 24, (uint32_t)fs);
    wr32(h + 28, (uint32_t)fs * 2u);
    wr16(h + 32, 2);
    wr16(h + 34, 16);
    memcpy(h + 36, "data", 4);
    wr32(h + 40, (uint32_t)(2 * n));
    fwrite(h, 1, 44, fp);
    for (i = 0; i < n; i++) {
        unsigned char s[2];
        wr16(s, (uint16_t)data[i]);
        fwrite(s, 1, 2, fp);
    }
    fclose(fp);
    return 0;
}

int sea_wav_write_segs(const char *path, short *const *seg, const long *cnt, int nseg, long n_zero_tail, int fs)
{
    FILE *fp = fopen(path, "wb");
    unsigned char h[44];
    long n = n_zero_tail, i;
    int k, rc = 0;
    static const short zeros[80] = {0};
    if (!fp) {
        fprintf(stderr, "Cannot write in the file %s\n", path);
        return 1;
    }
    for (k = 0; k < nseg; k++) n += cnt[k];
    memcpy(h, "RIFF", 4);
    wr32(h + 4, (uint32_t)(36 + 2 * n));
    memcpy(h + 8, "WAVEfmt ", 8);
    wr32(h + 16, 16);
    wr16(h + 20, 1);
    wr16(h + 22, 1);
    wr32(h + 24, (uint32_t)fs);
    wr32(h + 28, (uint32_t)fs * 2u);
    wr16(h + 32, 2);
    wr16(h + 34, 16);
    memcpy(h + 36, "data", 4);
    wr32(h + 40, (uint32_t)(2 * n));
    if (fwrite(h, 1, 44, fp) != 44) rc = 1;
    for (k = 0; k < nseg && !rc; k++)
        if (fwrite(seg[k], sizeof(short), (size_t)cnt[k], fp) != (size_t)cnt[k]) rc = 1;
    for (i = n_zero_tail; i > 0 && !rc; i -= 80)
        if (fwrite(zeros, sizeof(short), (size_t)(i < 80 ? i : 80), fp) != (size_t)(i < 80 ? i : 80)) rc = 1;
    if (fclose(fp)) rc = 1;
    return rc;
}

/* ---- Kaldi-style 64-column text matrices (show_IBM.cpp:194-208 writer, main.cpp:84-145 reader) ---- */
int sea_mask_text_write(FILE *fp, const char *id, const float *mask64, long rows)
{
    long f;
    int c;
    if (!fp || rows < 1) return 1;
    fprintf(fp, "%s [\n", id);
    for (f = 0; f < rows - 1; f++) {
        for (c = 0; c < 64; c++) fprintf(fp, "%.7f ", mask64[f * 64 + c]);
        fprintf(fp, "\n ");
    }
    for (c = 0; c < 64; c++) fprintf(fp, "%.7f ", mask64[(rows - 1) * 64 + c]);
    fprintf(fp, "]\n");
    return ferror(fp) ? 1 : 0;
}

long sea_mask_text_read(FILE *fp, char *id_out, float *mask64, long max_rows)
{
    char buf[64 * 32];
    long row = -1;
    while (1) {
        long pos = ftell(fp);
        if (!fgets(buf, sizeof buf, fp)) break;
        if (strstr(buf, "[")) {
            if (row >= 0) { /* the next matrix begins: leave its header for the next call */
                fseek(fp, pos, SEEK_SET);
                break;
            }
            if (id_out) {
                size_t n = strcspn(buf, " \t[");
                if (n >= SEA_FILE_LEN) n = SEA_FILE_LEN - 1;
                memcpy(id_out, buf, n);
                id_out[n] = 0;
            }
            row = 0;
        } else if (row >= 0 && row < max_rows) {
            char *p = buf, *end;
            int j;
            for (j = 0; j < 64; j++) {
                float v = strtof(p, &end);
                if (end == p) break;
                mask64[row * 64 + j] = v;
                p = end;
            }
            if (j > 0) row++;
        }
    }
    return row;
}

/* ---- chunk + bounded queue (see sea_host.h) ---- */
sea_chunk *sea_chunk_new(int first, int n, int with_mask, int with_ceps)
{
    sea_chunk *c = (sea_chunk *)calloc(1, sizeof *c);
    if (!c) return NULL;
    c->first = first;
    c->n = n;
    c->in = (short **)calloc(n > 0 ? n : 1, sizeof *c->in);
    c->out = (short **)calloc(n > 0 ? n : 1, sizeof *c->out);
    c->len = (long *)calloc(n > 0 ? n : 1, sizeof *c->len);
    c->used = (int *)calloc(n > 0 ? n : 1, sizeof *c->used);
    if (with_mask) c->mask = (float **)calloc(n > 0 ? n : 1, sizeof *c->mask);
    if (with_ceps) {
        c->ceps = (float **)calloc(n > 0 ? n : 1, sizeof *c->ceps);
        c->n_ceps = (int *)calloc(n > 0 ? n : 1, sizeof *c->n_ceps);
    }
    return c;
}

void sea_chunk_free(sea_chunk *c)
{
    int u;
    if (!c) return;
    for (u = 0; u < c->n; u++) {
        if (c->in) free(c->in[u]);
        if (c->out) free(c->out[u]);
        if (c->mask) free(c->mask[u]);
        if (c->ceps) free(c->ceps[u]);
    }
    free(c->in);
    free(c->out);
    free(c->len);
    free(c->used);
    free(c->mask);
    free(c->ceps);
    free(c->n_ceps);
    free(c->data_off);
    free(c->channels);
    free(c);
}

void sea_queue_init(sea_queue *q, int cap, int producers)
{
    memset(q, 0, sizeof *q);
    pthread_mutex_init(&q->mu, NULL);
    pthread_cond_init(&q->can_put, NULL);
    pthread_cond_init(&q->can_get, NULL);
    q->cap = cap > 0 ? cap : 1;
    q->producers = producers;
}

void sea_queue_put(sea_queue *q, sea_chunk *c)
{
    pthread_mutex_lock(&q->mu);
    while (q->count >= q->cap) pthread_cond_wait(&q->can_put, &q->mu);
    c->next = NULL;
    if (q->tail) q->tail->next = c; else q->head = c;
    q->tail = c;
    q->count++;
    pthread_cond_signal(&q->can_get);
    pthread_mutex_unlock(&q->mu);
}

sea_chunk *sea_queue_get(sea_queue *q)
{
    sea_chunk *c;
    pthread_mutex_lock(&q->mu);
    while (q->count == 0 && q->producers > 0) pthread_cond_wait(&q->can_get, &q->mu);
    c = q->head;
    if (c) {
        q->head = c->next;
        if (!q->head) q->tail = NULL;
        q->count--;
        pthread_cond_signal(&q->can_put);
    }
    pthread_mutex_unlock(&q->mu);
    return c;
}

void sea_queue_producer_done(sea_queue *q)
{
    pthread_mutex_lock(&q->mu);
    if (q->producers > 0) q->producers--;
    if (q->producers == 0) pthread_cond_broadcast(&q->can_get);
    pthread_mutex_unlock(&q->mu);
}

void sea_queue_destroy(sea_queue *q)
{
    pthread_mutex_destroy(&q->mu);
    pthread_cond_destroy(&q->can_put);
    pthread_cond_destroy(&q->can_get);
}
