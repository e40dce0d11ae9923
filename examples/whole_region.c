/* Plain C99: the reference's BENCH circuit (benches/digest.rs:103-129 -- one 56-byte message,
 * max 1024 bytes, input range checks, k = 17) as the literal advice-column image of its region:
 * every cell Sha256DynamicConfig::digest allocates, laid out as the 9 FlexGate columns the bench
 * configures (NUM_ADVICE = 9), plus the lookup-advice column.  SURVEY 8 f2 + f4; cell layout per
 * DESIGN.md assumptions A1-A4.  Build like examples/digest_abc.c. */
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
    int rc = hsw_engine_create_ex(0, NULL, 8, 2, HSW_MODE_HALO2_INTERNALS, &eng);    /* benches/digest.rs:60-69 */
    if (rc != HSW_OK) die("hsw_engine_create_ex", rc, NULL);

    const size_t max_size = 1024;                                                    /* MAX_BYTE_SIZE1 */
    hsw_gadget *g = NULL;
    rc = hsw_gadget_create_ex(eng, &max_size, 1, /*is_input_range_check*/ 1, HSW_GADGET_WHOLE_DIGEST, &g);
    if (rc != HSW_OK) die("hsw_gadget_create_ex", rc, eng);
    const uint64_t max_rows = (1u << 17) - 9;                                        /* usable rows at k = 17 */
    uint64_t columns = 0;
    rc = hsw_gadget_set_columns(g, max_rows, &columns);
    if (rc != HSW_OK) die("hsw_gadget_set_columns", rc, eng);

    uint8_t msg[56];
    memset(msg, 1, sizeof msg);                                                      /* benches/digest.rs:129 */
    hsw_hash_result r;
    for (int pass = 0; pass < 2; pass++) {                                           /* keygen pass, proving pass */
        if ((rc = hsw_gadget_reset(g)) != HSW_OK) die("hsw_gadget_reset", rc, eng);
        if ((rc = hsw_gadget_digest(g, msg, sizeof msg, 0, &r)) != HSW_OK) die("hsw_gadget_digest", rc, eng);
    }
    hsw_gadget_view v;
    hsw_gadget_streams(g, &v);
    printf("advice columns: %llu x %llu rows; gate cells %llu, lookup cells %llu\n", (unsigned long long)v.columns,
           (unsigned long long)v.max_rows, (unsigned long long)v.gate_cells, (unsigned long long)v.lookup_cells);
    if (columns != 9 || v.gate_cells != 1116315 || v.lookup_cells != 53059) return 1;

    /* the library's own MockProver-style check of the whole region, on the device */
    hsw_verify_report rep;
    rc = hsw_gadget_verify(g, &rep);
    if (rc != HSW_OK) die("hsw_gadget_verify", rc, eng);
    printf("verified on the device: %llu constraints, %llu violations\n", (unsigned long long)rep.checks,
           (unsigned long long)rep.violations);
    if (rep.violations != 0) return 1;

    /* AssignedHashResult.output_bytes: the 32 load_witness cells of the epilogue (lib.rs:317-324) */
    char hex[65];
    for (int w = 0; w < 8; w++) {
        for (int i = 0; i < 4; i++) {
            const uint64_t cell = r.epilogue_cell + 76 * (r.n_blocks + 1) + 36 * (uint64_t)w + 5 * (uint64_t)i;
            uint64_t col, row, val[4];
            hsw_gadget_cell_position(g, cell, &col, &row);
            rc = hsw_download(eng, val, (const uint8_t *)v.d_gate + (col * v.max_rows + row) * HSW_CELL_BYTES, sizeof val);
            if (rc != HSW_OK) die("hsw_download", rc, eng);
            if (val[0] != r.output_bytes[4 * w + i] || val[1] || val[2] || val[3]) return 1;
            sprintf(hex + 2 * (4 * w + i), "%02x", (unsigned)val[0]);
        }
    }
    printf("digest read back from the advice columns: %s\n", hex);
    uint64_t col, row;
    hsw_gadget_cell_position(g, r.end_cell - 1, &col, &row);
    printf("last cell of the region: column %llu, row %llu\n", (unsigned long long)col, (unsigned long long)row);

    hsw_gadget_destroy(g);
    hsw_engine_destroy(eng);
    puts("ok");
    return 0;
}
