/* round 6 probe: what the first touch of a fresh 560 MB destination costs a multi-threaded host copy (the download of a CSC into a
 * vector the caller has just allocated), and what huge pages / MADV_POPULATE_WRITE make of it.  gcc -O2 -pthread r6_prefault.c */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + t.tv_nsec * 1e-6; }
typedef struct { char *dst; const char *src; size_t n; int mode; } job_t;
static void *work(void *p) {
    job_t *j = (job_t *)p;
    if (j->mode == 2) madvise(j->dst, j->n, MADV_POPULATE_WRITE);
    memcpy(j->dst, j->src, j->n);
    return 0;
}
int main(int argc, char **argv) {
    const size_t n = (size_t)560 << 20;
    const int T = argc > 1 ? atoi(argv[1]) : 8;
    char *src = mmap(0, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    memset(src, 1, n);
    FILE *f = fopen("/sys/kernel/mm/transparent_hugepage/enabled", "r");
    char buf[128] = "";
    if (f) { if (!fgets(buf, sizeof buf, f)) buf[0] = 0; fclose(f); }
    printf("THP: %s", buf);
    for (int mode = 0; mode < 4; mode++) {
        for (int rep = 0; rep < 3; rep++) {
            char *dst = mmap(0, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            const double t0 = now();
            if (mode == 1 || mode == 3) madvise(dst, n, MADV_HUGEPAGE);
            pthread_t th[64];
            job_t jobs[64];
            for (int t = 0; t < T; t++) {
                const size_t a = n / T * t, b = t == T - 1 ? n : n / T * (t + 1);
                jobs[t] = (job_t){dst + a, src + a, b - a, mode == 3 ? 2 : mode};
                pthread_create(&th[t], 0, work, &jobs[t]);
            }
            for (int t = 0; t < T; t++) pthread_join(th[t], 0);
            const double t1 = now();
            const double tt0 = now();
            for (int t = 0; t < T; t++) {
                jobs[t].mode = 0;
                pthread_create(&th[t], 0, work, &jobs[t]);
            }
            for (int t = 0; t < T; t++) pthread_join(th[t], 0);
            printf("mode %d (%s) threads %d: first copy %.1f ms, second copy %.1f ms\n", mode,
                   mode == 0 ? "plain" : mode == 1 ? "MADV_HUGEPAGE" : mode == 2 ? "POPULATE_WRITE per thread" : "HUGEPAGE + POPULATE_WRITE", T, t1 - t0, now() - tt0);
            munmap(dst, n);
        }
    }
    return 0;
}
