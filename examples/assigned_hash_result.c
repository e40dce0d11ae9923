/* Plain C99 mirror of what the Rust shim's digest() returns: AssignedHashResult { input_len, input_bytes,
 * output_bytes } (reference src/lib.rs:31-36, 342-346) resolved to (column, row) positions of the FlexGate
 * advice image and read back -- for the reference's TestCircuit (2 x 128 B, NUM_ADVICE = 3, lib.rs:455-466,
 * 487-494; its output-byte cells go to the instance column, lib.rs:480-482) and for its bench circuit
 * (1 x 1,024 B, NUM_ADVICE = 9, benches/digest.rs:102-109,129).  Cell layout per DESIGN.md A1-A4.
 * Build like examples/digest_abc.c. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "hsw.h"

static void die(const char *what, int rc, const hsw_engine *e) {
    fprintf(stderr, "%s: %s (%s)\n", what, hsw_strerror(rc), e ? hsw_last_error(e) : "");
    exit(1);
}

/* the cell at FlexGate (column, row), as four little-endian limbs: image column = column - origin_column */
static void read_cell(hsw_engine *eng, const hsw_gadget_view *v, const uint64_t pos[2], uint64_t out[4]) {
    const int rc = hsw_download(eng, out, (const uint8_t *)v->d_gate +
                                ((pos[0] - v->origin_column) * v->max_rows + pos[1]) * HSW_CELL_BYTES, 32);
    if (rc != HSW_OK) die("hsw_download", rc, eng);
}

/* The same cell the way the Rust shim takes it (rust/reference-patch/src/hsw.rs): only the region's DISTINCT values
 * cross PCIe, the input-independent tape says which of them -- or which gate constant -- stream cell i holds. */
static void tape_cell(const hsw_region_tape *t, const uint64_t (*distinct)[4], uint64_t stream_cell, uint64_t out[4]) {
    const uint32_t code = t->gate_code[stream_cell];
    const uint64_t *src = (code & HSW_TAPE_CONST) ? ((const uint64_t (*)[4])t->consts)[code & ~HSW_TAPE_CONST] : distinct[code];
    memcpy(out, src, 32);
}

/* origin: where the circuit's Context stands when it calls digest() for the first time -- (0, 0, fresh) for the
 * reference's own circuits, anything for a circuit that has used the gate / range chips before */
typedef struct { uint64_t column, row, lookups_queued; int zero_cell_loaded; } origin_t;

static int check_circuit(const char *name, const size_t *sizes, size_t n, const uint8_t *const *msgs, const size_t *lens,
                         uint64_t want_columns, origin_t o) {
    hsw_engine *eng = NULL;
    int rc = hsw_engine_create_ex(0, NULL, 8, 2, HSW_MODE_HALO2_INTERNALS, &eng);
    if (rc != HSW_OK) die("hsw_engine_create_ex", rc, NULL);
    hsw_gadget *g = NULL;
    if ((rc = hsw_gadget_create_ex(eng, sizes, n, 1, HSW_GADGET_WHOLE_DIGEST, &g)) != HSW_OK) die("hsw_gadget_create_ex", rc, eng);
    uint64_t columns = 0;
    if ((rc = hsw_gadget_set_origin(g, o.column, o.row, o.zero_cell_loaded, o.lookups_queued)) != HSW_OK) die("hsw_gadget_set_origin", rc, eng);
    if ((rc = hsw_gadget_set_columns(g, (1u << 17) - 9, &columns)) != HSW_OK) die("hsw_gadget_set_columns", rc, eng);
    if (columns != want_columns) { fprintf(stderr, "%s: %llu columns\n", name, (unsigned long long)columns); return 1; }
    hsw_hash_result r[4];
    for (size_t i = 0; i < n; i++)          /* sequential digest() calls, like lib.rs:455-466 */
        if ((rc = hsw_gadget_digest(g, msgs[i], lens[i], 0, &r[i])) != HSW_OK) die("hsw_gadget_digest", rc, eng);
    hsw_gadget_view v;
    hsw_gadget_streams(g, &v);
    /* the region, checked on the device against the constraint structure, wherever it starts */
    hsw_verify_report rep;
    if ((rc = hsw_gadget_verify(g, &rep)) != HSW_OK) die("hsw_gadget_verify", rc, eng);
    if (rep.violations) { fprintf(stderr, "%s: %llu violations\n", name, (unsigned long long)rep.violations); return 1; }
    /* the first cell of the first digest sits where the Context stood; its lookups follow the queued ones */
    if (r[0].prologue_lookup != o.lookups_queued) return 1;
    uint64_t c0 = 0, r0 = 0;
    if ((rc = hsw_gadget_cell_position(g, 0, &c0, &r0)) != HSW_OK) die("cell_position", rc, eng);
    if (o.row + 1 < v.max_rows && (c0 != o.column || r0 != o.row)) { fprintf(stderr, "%s: origin\n", name); return 1; }
    /* distinct-value delivery: tape (built once per circuit) + the new witnesses in pinned memory */
    hsw_region_tape tape;
    if ((rc = hsw_gadget_region_tape(g, &tape)) != HSW_OK) die("hsw_gadget_region_tape", rc, eng);
    void *dist_mem = NULL;
    if ((rc = hsw_host_alloc(tape.distinct_capacity * HSW_CELL_BYTES, &dist_mem)) != HSW_OK) die("hsw_host_alloc", rc, eng);
    const uint64_t (*distinct)[4] = (const uint64_t (*)[4])dist_mem;
    size_t n_distinct = 0;
    if ((rc = hsw_gadget_download_region_distinct(g, dist_mem, tape.distinct_capacity, &n_distinct)) != HSW_OK) die("download_region_distinct", rc, eng);
    if (tape.gate_cells != v.gate_cells || n_distinct != tape.n_distinct || 2 * n_distinct > tape.gate_cells) return 1;
    for (size_t i = 0; i < n; i++) {
        hsw_result_cells rc_;
        if ((rc = hsw_gadget_result_cells(g, i, &rc_)) != HSW_OK) die("hsw_gadget_result_cells", rc, eng);
        uint64_t cell[4], via_tape[4];
        /* input_len (lib.rs:124-125) */
        read_cell(eng, &v, rc_.input_len_pos, cell);
        if (cell[0] != lens[i] || cell[1] || cell[2] || cell[3]) { fprintf(stderr, "%s: input_len cell\n", name); return 1; }
        tape_cell(&tape, distinct, rc_.input_len_cell, via_tape);
        if (memcmp(cell, via_tape, 32) != 0) { fprintf(stderr, "%s: input_len through the tape\n", name); return 1; }
        /* input_bytes (lib.rs:170-173): the padded message, max_variable_byte_size cells */
        if (rc_.n_input_bytes != sizes[i]) return 1;
        uint8_t *padded = calloc(sizes[i], 1);
        size_t plen = 0;
        if ((rc = hsw_gadget_input_bytes(g, i, padded, sizes[i], &plen)) != HSW_OK || plen != sizes[i]) die("hsw_gadget_input_bytes", rc, eng);
        if (memcmp(padded, msgs[i], lens[i]) != 0 || padded[lens[i]] != 0x80) return 1;
        for (size_t b = 0; b < sizes[i]; b += (b < 70 ? 1 : 37)) {            /* the first bytes and a stride through the rest */
            uint64_t pos[2];
            if ((rc = hsw_gadget_cell_position(g, rc_.input_bytes_cell0 + b, &pos[0], &pos[1])) != HSW_OK) die("cell_position", rc, eng);
            if (b == 0 && (pos[0] != rc_.input_bytes_pos0[0] || pos[1] != rc_.input_bytes_pos0[1])) return 1;
            read_cell(eng, &v, pos, cell);
            if (cell[0] != padded[b] || cell[1] || cell[2] || cell[3]) { fprintf(stderr, "%s: input byte %zu\n", name, b); return 1; }
            tape_cell(&tape, distinct, rc_.input_bytes_cell0 + b, via_tape);
            if (memcmp(cell, via_tape, 32) != 0) { fprintf(stderr, "%s: input byte %zu through the tape\n", name, b); return 1; }
        }
        free(padded);
        /* output_bytes (lib.rs:317-324): what constrain_instance ties to the instance column */
        char hex[65];
        for (int k = 0; k < 32; k++) {
            read_cell(eng, &v, rc_.output_byte_pos[k], cell);
            if (cell[0] != r[i].output_bytes[k] || cell[1] || cell[2] || cell[3]) { fprintf(stderr, "%s: output byte %d\n", name, k); return 1; }
            tape_cell(&tape, distinct, rc_.output_byte_cells[k], via_tape);
            if (memcmp(cell, via_tape, 32) != 0) { fprintf(stderr, "%s: output byte %d through the tape\n", name, k); return 1; }
            sprintf(hex + 2 * k, "%02x", (unsigned)cell[0]);
        }
        printf("%s digest %zu: input_len at (%llu, %llu), input_bytes from (%llu, %llu), output_bytes[0] at (%llu, %llu): %s\n", name, i,
               (unsigned long long)rc_.input_len_pos[0], (unsigned long long)rc_.input_len_pos[1],
               (unsigned long long)rc_.input_bytes_pos0[0], (unsigned long long)rc_.input_bytes_pos0[1],
               (unsigned long long)rc_.output_byte_pos[0][0], (unsigned long long)rc_.output_byte_pos[0][1], hex);
    }
    hsw_host_free(dist_mem);
    hsw_gadget_destroy(g);
    hsw_engine_destroy(eng);
    return 0;
}

int main(void) {
    /* TestCircuit, test_sha256_correct1 (lib.rs:497-527): "abc" and "" */
    const size_t s1[2] = {128, 128}, l1[2] = {3, 0};
    const uint8_t *m1[2] = {(const uint8_t *)"abc", (const uint8_t *)""};
    const origin_t fresh = {0, 0, 0, 0};
    if (check_circuit("TestCircuit", s1, 2, m1, l1, 3, fresh)) return 1;
    /* the same two digests in a circuit that has already assigned 131,000 rows of FlexGate column 2, loaded the
     * Context's zero cell and queued 77 lookups: the first column break falls inside the first prologue */
    const origin_t later = {2, 131000, 77, 1};
    if (check_circuit("TestCircuit at (2, 131000)", s1, 2, m1, l1, 4, later)) return 1;
    /* bench circuit (benches/digest.rs:93,102-109,129): one 56-byte message of 0x01 */
    uint8_t msg[56];
    memset(msg, 1, sizeof msg);
    const size_t s2[1] = {1024}, l2[1] = {56};
    const uint8_t *m2[1] = {msg};
    if (check_circuit("bench circuit", s2, 1, m2, l2, 9, fresh)) return 1;
    puts("ok");
    return 0;
}
