/* Plain C99 user of the C ABI: the reference's test_sha256_correct1 flow
 * (src/lib.rs:496-527: TestCircuit hashing "abc" and "" with max 128 bytes each)
 * through include/hsw.h.  Build:
 *   gcc -std=c99 -Iinclude examples/digest_abc.c -Lhalo2-dynamic-sha256_amd -lhsw \
 *       -Wl,-rpath,$PWD/halo2-dynamic-sha256_amd -o digest_abc
 * Prints the two digests and a few stream facts; exit code 0 on success. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "hsw.h"

static void die(const char *what, int rc, const hsw_engine *e) {
    fprintf(stderr, "%s: %s (%s)\n", what, hsw_strerror(rc), e ? hsw_last_error(e) : "");
    exit(1);
}

int main(void) {
    hsw_engine *eng = NULL;
    int rc = hsw_engine_create(0, NULL, /*num_bits_lookup*/ 8, /*num_advice_columns*/ 2, &eng);   /* lib.rs:421-428 */
    if (rc != HSW_OK) die("hsw_engine_create", rc, NULL);

    const size_t max_sizes[2] = {128, 128};                        /* lib.rs:488-489 */
    hsw_gadget *g = NULL;
    rc = hsw_gadget_create(eng, max_sizes, 2, /*is_input_range_check*/ 1, &g);
    if (rc != HSW_OK) die("hsw_gadget_create", rc, eng);

    hsw_hash_result r0, r1;
    rc = hsw_gadget_digest(g, (const uint8_t *)"abc", 3, 0, &r0);  /* lib.rs:455-459 */
    if (rc != HSW_OK) die("digest(abc)", rc, eng);
    rc = hsw_gadget_digest(g, NULL, 0, 0, &r1);                    /* lib.rs:462-466 */
    if (rc != HSW_OK) die("digest(\"\")", rc, eng);

    static const char *want0 = "ba7816bf8f01cfea414140de5dae2223b00361a396177a9cb410ff61f20015ad";
    static const char *want1 = "e3b0c44298fc1c149afbf4c8996fb92427ae41e4649b934ca495991b7852b855";
    char hex0[65], hex1[65];
    for (int i = 0; i < 32; i++) {
        sprintf(hex0 + 2 * i, "%02x", r0.output_bytes[i]);
        sprintf(hex1 + 2 * i, "%02x", r1.output_bytes[i]);
    }
    printf("sha256(\"abc\") = %s\nsha256(\"\")    = %s\n", hex0, hex1);
    if (strcmp(hex0, want0) != 0 || strcmp(hex1, want1) != 0) { fprintf(stderr, "digest mismatch\n"); return 1; }

    hsw_shape shape;
    hsw_gadget_view view;
    hsw_engine_shape(eng, &shape);
    hsw_gadget_streams(g, &view);
    printf("blocks assigned: %zu (2 per hash: the maximum is always synthesised, lib.rs:180)\n", view.blocks_done);
    printf("gate cells per block: %u, chip rows per block: %u, spread cursor now: %llu\n",
           shape.gate_cells_per_block, shape.limb_calls_per_block / shape.num_advice_columns,
           (unsigned long long)view.num_limb_sum);

    /* first gate row of block 0: [sum=0, byte=0x80, 2^0, 0x80]  (compression.rs:34-41; "abc" + 0x80 padding) */
    uint64_t row[4][4];
    rc = hsw_download(eng, row, view.d_gate, sizeof row);
    if (rc != HSW_OK) die("hsw_download", rc, eng);
    printf("first gate row: [%llu, %llu, %llu, %llu]\n", (unsigned long long)row[0][0], (unsigned long long)row[1][0],
           (unsigned long long)row[2][0], (unsigned long long)row[3][0]);
    if (row[0][0] != 0 || row[1][0] != 0x80 || row[2][0] != 1 || row[3][0] != 0x80) return 1;

    hsw_gadget_destroy(g);
    hsw_engine_destroy(eng);
    puts("ok");
    return 0;
}
