/* Latency of one synthesis of the reference's bench circuit (benches/digest.rs: 56-byte message, max 1,024 B,
 * input range checks, 9 columns at k = 17) through the C ABI alone -- no Python in the timed path.
 *   gcc -O2 -std=c99 -I include tools/region_latency.c -L halo2-dynamic-sha256_amd -lhsw -Wl,-rpath,$PWD/halo2-dynamic-sha256_amd -o /tmp/region_latency
 * Prints median / min microseconds of hsw_gadget_reset and of hsw_gadget_digest, whole-region and block-stream contexts. */
#define _POSIX_C_SOURCE 200809L
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "hsw.h"

static double now_us(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}
static int cmp(const void *a, const void *b) { return (*(const double *)a > *(const double *)b) - (*(const double *)a < *(const double *)b); }

static void run(int whole, size_t max_size, size_t msg_len, int split, unsigned k, int mont, int helpers) {
    hsw_engine *eng = NULL;
    if (hsw_engine_create_ex(0, NULL, 8, 2, whole ? HSW_MODE_HALO2_INTERNALS : HSW_MODE_DEFAULT, &eng) != HSW_OK) exit(1);
    hsw_engine_set_option(eng, "split", split);
    hsw_engine_set_option(eng, "helpers", helpers);
    size_t sizes[64];
    for (int i = 0; i < 64; i++) sizes[i] = max_size;
    hsw_gadget *g = NULL;
    if (hsw_gadget_create_ex(eng, sizes, whole ? 1 : 64, 1, whole ? HSW_GADGET_WHOLE_DIGEST : 0, &g) != HSW_OK) exit(2);
    uint64_t columns = 0;
    if (whole && hsw_gadget_set_columns(g, (1u << k) - 9, &columns) != HSW_OK) exit(3);
    if (mont && hsw_gadget_set_repr(g, HSW_REPR_MONTGOMERY) != HSW_OK) exit(6);
    uint8_t *msg = malloc(msg_len + 1);
    memset(msg, 1, msg_len);
    hsw_hash_result r;
    enum { N = 300 };
    static double td[N], tr[N];
    for (int i = 0; i < N + 20; i++) {
        double t0 = now_us();
        if (hsw_gadget_reset(g) != HSW_OK) exit(4);
        double t1 = now_us();
        if (hsw_gadget_digest(g, msg, msg_len, 0, &r) != HSW_OK) { fprintf(stderr, "%s\n", hsw_last_error(eng)); exit(5); }
        double t2 = now_us();
        if (i >= 20) { tr[i - 20] = t1 - t0; td[i - 20] = t2 - t1; }
    }
    qsort(td, N, sizeof(double), cmp);
    qsort(tr, N, sizeof(double), cmp);
    hsw_launch_info li;
    hsw_last_launch(eng, &li);
    printf("%-13s %s max %5zu B (%2zu blocks) split=%2d helpers=%d -> kernel split %u, grid %llu: digest median %6.1f us  min %6.1f us  p90 %6.1f us | reset median %4.1f us\n",
           whole ? "whole region" : "block streams", mont ? "Montgomery" : "canonical ", max_size, max_size / 64, split, helpers, li.split, (unsigned long long)li.grid,
           td[N / 2], td[0], td[N * 9 / 10], tr[N / 2]);
    free(msg);
    hsw_gadget_destroy(g);
    hsw_engine_destroy(eng);
}

/* BASELINE configs[4] substitute: the advice-column image of a k = 20 region -- 8 digests of 15 blocks (120 blocks),
 * input range checks, columns of 2^20 - 9 rows -- synthesised by ONE hsw_gadget_digest_batch call. */
static void run_k20(int mont) {
    hsw_engine *eng = NULL;
    if (hsw_engine_create_ex(0, NULL, 8, 2, HSW_MODE_HALO2_INTERNALS, &eng) != HSW_OK) exit(1);
    size_t sizes[8];
    for (int i = 0; i < 8; i++) sizes[i] = 960;
    hsw_gadget *g = NULL;
    if (hsw_gadget_create_ex(eng, sizes, 8, 1, HSW_GADGET_WHOLE_DIGEST, &g) != HSW_OK) exit(2);
    uint64_t columns = 0;
    if (hsw_gadget_set_columns(g, (1u << 20) - 9, &columns) != HSW_OK) exit(3);
    if (mont && hsw_gadget_set_repr(g, HSW_REPR_MONTGOMERY) != HSW_OK) exit(6);
    static uint8_t msg[8][900];
    const uint8_t *inputs[8];
    size_t lens[8], pre[8];
    for (int i = 0; i < 8; i++) { memset(msg[i], i + 1, sizeof msg[i]); inputs[i] = msg[i]; lens[i] = 900; pre[i] = 0; }
    hsw_hash_result r[8];
    enum { N = 200 };
    static double td[N];
    for (int i = 0; i < N + 10; i++) {
        if (hsw_gadget_reset(g) != HSW_OK) exit(4);
        double t1 = now_us();
        if (hsw_gadget_digest_batch(g, 8, inputs, lens, pre, r) != HSW_OK) { fprintf(stderr, "%s\n", hsw_last_error(eng)); exit(5); }
        double t2 = now_us();
        if (i >= 10) td[i - 10] = t2 - t1;
    }
    qsort(td, N, sizeof(double), cmp);
    hsw_launch_info li;
    hsw_last_launch(eng, &li);
    printf("k = 20 region %s 8 x 960 B (120 blocks), %llu columns -> kernel split %u, grid %llu: batch median %6.1f us  min %6.1f us  p90 %6.1f us\n",
           mont ? "Montgomery" : "canonical ", (unsigned long long)columns, li.split, (unsigned long long)li.grid, td[N / 2], td[0], td[N * 9 / 10]);
    hsw_gadget_destroy(g);
    hsw_engine_destroy(eng);
}

int main(int argc, char **argv) {
    if (argc > 1) {      /* sweep of the "helpers" option: waves per workgroup of the small-batch kernel */
        for (int mont = 0; mont < 2; mont++)
            for (int blocks = 1; blocks <= 32; blocks = blocks == 1 ? 2 : blocks * 2)
                for (int h = 1; h <= (argc > 2 ? atoi(argv[2]) : 4); h++) {      /* `sweep 6` against a build with more helper waves */
                    if (h == 3 || h == 5) continue;
                    run(1, 64 * (size_t)blocks, 3, -1, blocks > 16 ? 18 : 17, mont, h);
                }
        return 0;
    }
    run(1, 1024, 56, -1, 17, 0, 0);
    run(1, 1024, 56, 0, 17, 0, 0);
    run(0, 1024, 56, -1, 17, 0, 0);
    run(0, 1024, 56, 0, 17, 0, 0);
    run(1, 128, 3, -1, 17, 0, 0);
    run(1, 2048, 2000, -1, 18, 0, 0);
    run(1, 2048, 2000, 0, 18, 0, 0);
    run(0, 64, 3, -1, 17, 0, 0);
    run(1, 1024, 56, -1, 17, 1, 0);
    run(1, 1024, 56, 0, 17, 1, 0);
    run(0, 1024, 56, -1, 17, 1, 0);
    run(1, 128, 3, -1, 17, 1, 0);
    run(1, 2048, 2000, -1, 18, 1, 0);
    run_k20(0);
    run_k20(1);
    return 0;
}
