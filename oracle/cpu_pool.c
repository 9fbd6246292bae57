/*
 * oracle/cpu_pool.c -- TEST / BENCH INFRASTRUCTURE ONLY (bench.py's cpu_baseline leg).
 *
 * Runs ONE etsi_denoise-shaped function -- int fn(short *in, short *out, long n), the signature of
 * etsi/cpp/AdvFrontEnd.h:13 -- over a list of utterances from T host threads that pull utterance indices from a
 * shared counter: the shape of the reference's own parallel harness
 * (function/20141106_speech_enhancement/aurora_speech_enhancement/aurora_speech_enhancement.cpp:111-121, 311-327: N
 * workers, one shared index under a lock).  The function timed is the oracle's restatement (oracle/libsea_oracle.so:
 * ora_etsi_denoise) or the reference C compiled here (oracle/_ref/libetsi_ref.so: ref_etsi_denoise); this file only
 * supplies the threads, so that a 256-core host is not paced by a Python dispatcher.  Never part of the product.
 */
#include <pthread.h>
#include <stdlib.h>
#include <time.h>

typedef int (*denoise_fn)(short *, short *, long);

struct job {
    denoise_fn fn;
    short **in, **out;
    const long *len;
    long n, passes;
    long next; /* shared counter over passes * n work items */
};

static void *worker(void *p)
{
    struct job *j = (struct job *)p;
    for (;;) {
        const long k = __atomic_fetch_add(&j->next, 1, __ATOMIC_RELAXED);
        if (k >= j->n * j->passes) break;
        const long u = k % j->n;
        j->fn(j->in[u], j->out[u], j->len[u]);
    }
    return NULL;
}

/* returns the wall-clock seconds of `passes` passes over the n utterances on `threads` threads (< 0: thread
 * creation failed).  out[u] must have room for len[u] samples; with passes > 1 an utterance may be processed by two
 * threads at once, both writing the same values. */
double sea_cpu_pool_run(void *fn, short **in, short **out, const long *len, long n, int threads, int passes)
{
    struct job j = {(denoise_fn)fn, in, out, len, n, passes, 0};
    if (threads < 1) threads = 1;
    pthread_t *th = (pthread_t *)calloc((size_t)threads, sizeof *th);
    if (!th) return -1.0;
    struct timespec a, b;
    clock_gettime(CLOCK_MONOTONIC, &a);
    int started = 0;
    for (; started < threads; ++started)
        if (pthread_create(&th[started], NULL, worker, &j)) break;
    if (started == 0) {
        free(th);
        return -1.0;
    }
    for (int i = 0; i < started; ++i) pthread_join(th[i], NULL);
    clock_gettime(CLOCK_MONOTONIC, &b);
    free(th);
    return (double)(b.tv_sec - a.tv_sec) + 1e-9 * (double)(b.tv_nsec - a.tv_nsec);
}
