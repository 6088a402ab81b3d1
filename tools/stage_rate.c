/* stage_rate.c -- rate of the staged per-entry boundary (what the Julia shim's push loop drives): a C producer loop fills the
 * pinned chunk of esp_stage_begin entry by entry (stencil-like stream) and commits chunk after chunk; esp_commit packs on the
 * host and lets the transfer run behind the next chunk's filling loop.  usage: stage_rate [entries, default 100e6] [chunk]
 * build: gcc -O2 -std=c99 -I include tools/stage_rate.c -o gpurun_out/stage_rate -L extendablesparse.jl_amd -l:libesparse_hip.so -Wl,-rpath,$PWD/extendablesparse.jl_amd -Wl,-rpath,/opt/rocm/lib */
#define _POSIX_C_SOURCE 200809L
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#include "esparse_hip.h"

static double now(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int main(int argc, char **argv) {
    const int64_t total = argc > 1 ? (int64_t)atof(argv[1]) : 100000000;
    const int64_t want = argc > 2 ? atoll(argv[2]) : (1 << 16);
    const int64_t n = 16777216;
    esp_handle *h = NULL;
    int64_t *rows, *cols, cap = 0, fill = 0, done = 0, z = 0;
    double *vals, t0, t1, t2;
    uint8_t *kinds;
    int32_t changed = 0;
    if (esp_create(n, n, 0, total, &h) != ESP_OK) return 1;
    if (esp_stage_begin(h, want, &rows, &cols, &vals, &kinds, &cap) != ESP_OK) return 1;
    t0 = now();
    while (done < total) {
        const int64_t l = 1 + (done / 7) % (n - 300);
        rows[fill] = l + (done % 7) * 40;
        cols[fill] = l;
        vals[fill] = 1.0 + (double)(done & 15);
        kinds[fill] = ESP_UPDATE;
        done++;
        if (++fill == cap) {
            if (esp_commit(h, fill, -1, ESP_OP_ADD) != ESP_OK) {
                fprintf(stderr, "commit: %s\n", esp_last_error(h));
                return 1;
            }
            fill = 0;
        }
    }
    if (fill && esp_commit(h, fill, -1, ESP_OP_ADD) != ESP_OK) return 1;
    esp_synchronize(h);
    t1 = now();
    if (esp_flush(h, ESP_FLUSH_ROUTED, &z, &changed) != ESP_OK) return 1;
    esp_synchronize(h);
    t2 = now();
    printf("stage_rate: %lld entries in chunks of %lld: fill + commit %.1f ms = %.3g entries/s; flush %.1f ms, nnz %lld\n", (long long)total,
           (long long)cap, (t1 - t0) * 1e3, (double)total / (t1 - t0), (t2 - t1) * 1e3, (long long)z);
    esp_destroy(h);
    return 0;
}
