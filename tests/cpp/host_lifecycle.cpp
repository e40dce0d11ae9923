// Lifetimes and copy bounds of libhsw's host side under AddressSanitizer + UBSan, against the stand-in HIP
// runtime of hip_stub.cpp ("device" memory = heap memory, launches do nothing): engines and gadgets created,
// used and destroyed in every order the ABI allows, every host delivery into EXACT-size destination buffers,
// geometry changes between synthesis passes, pinned-pointer validation.  What a kernel would have written is
// irrelevant here -- the sanitizer checks that no copy, free or handle use goes where it must not.
// Built and run by tests/test_host_sanitizers.py.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/hsw.h"

extern "C" {
int hip_stub_launches();
size_t hip_stub_live_device_allocations();
size_t hip_stub_live_pinned_allocations();
size_t hip_stub_live_events();
}

#define CHECK(cond)                                                        \
    do {                                                                   \
        if (!(cond)) {                                                     \
            std::fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            std::exit(1);                                                  \
        }                                                                  \
    } while (0)

static void *dev(size_t bytes) { void *p = nullptr; CHECK(hipMalloc(&p, bytes) == hipSuccess); return p; }

// raw block batches: both kernels, packed columns, host delivery into exact-size pageable buffers
static void block_batches() {
    for (uint32_t mode : {HSW_MODE_DEFAULT, HSW_MODE_HALO2_INTERNALS}) {
        hsw_engine *e = nullptr;
        CHECK(hsw_engine_create_ex(0, nullptr, 8, 2, mode, &e) == HSW_OK);
        hsw_shape s;
        CHECK(hsw_engine_shape(e, &s) == HSW_OK);
        for (size_t n : {3u, 200u}) {                               // small-batch kernel / streaming kernel
            for (uint64_t cursor : {0ull, 4121ull}) {
                const size_t rows = (size_t)hsw_chip_rows(&s, cursor, n);
                void *blocks = dev(n * 64), *pre = dev(n * 32), *next = dev(n * 32);
                void *gate = dev(n * s.gate_cells_per_block * 32), *cd = dev(2 * rows * 32), *cs = dev(2 * rows * 32);
                const int before = hip_stub_launches();
                CHECK(hsw_witness_blocks(e, (const uint8_t *)blocks, (const uint32_t *)pre, n, cursor, gate, cd, cs, rows,
                                         (uint32_t *)next, HSW_REPR_MONTGOMERY) == HSW_OK);
                CHECK(hip_stub_launches() > before);
                CHECK(hsw_witness_blocks(e, (const uint8_t *)blocks, (const uint32_t *)pre, n, cursor, gate, cd, cs, rows - 1,
                                         (uint32_t *)next, 0) == HSW_ERR_INVALID_ARG);      // stride too small: refused, not launched
                CHECK(hsw_engine_synchronize(e) == HSW_OK);
                for (void *p : {blocks, pre, next, gate, cd, cs}) CHECK(hipFree(p) == hipSuccess);
            }
        }
        // host delivery (two staging slots, 128-block chunks): exact-size pageable destinations
        for (uint64_t cursor : {0ull, 7ull}) {
            const size_t n = 300, rows = (size_t)hsw_chip_rows(&s, cursor, n);
            std::vector<uint8_t> blocks(n * 64, 1);
            std::vector<uint32_t> pre(n * 8, 2), next(n * 8);
            std::vector<uint64_t> gate(n * (size_t)s.gate_cells_per_block * 4), cd(2 * rows * 4), cs(2 * rows * 4);
            CHECK(hsw_witness_blocks_host(e, blocks.data(), pre.data(), n, cursor, gate.data(), cd.data(), cs.data(), rows,
                                          next.data(), HSW_HOST_REGISTER) == HSW_OK);       // flag accepted, ignored
            std::vector<uint64_t> gate8(n * (size_t)s.gate_cells_per_block), cd8(2 * rows), cs8(2 * rows);
            CHECK(hsw_witness_blocks_host(e, blocks.data(), pre.data(), n, cursor, gate8.data(), cd8.data(), cs8.data(), rows,
                                          nullptr, HSW_REPR_COMPACT64) == HSW_OK);
        }
        hsw_engine_destroy(e);                                      // with the pipeline's staging, events and stream alive
    }
}

// hsw_device_alloc: ranges of several physical pieces, freed in any order, a foreign pointer refused
static void device_ranges() {
    void *a = nullptr, *b = nullptr;
    CHECK(hsw_device_alloc(0, 10 * 4096 + 1, 3 * 4096, &a) == HSW_OK && a);        // 11 pages in pieces of 3: 4 physical allocations
    CHECK(hsw_device_alloc(0, 1, 0, &b) == HSW_OK && b);
    void *c = nullptr;
    CHECK(hsw_device_alloc(0, 0, 0, &c) == HSW_ERR_INVALID_ARG);
    CHECK(hsw_device_alloc(3, 4096, 0, &c) == HSW_ERR_NO_DEVICE && c == nullptr);
    std::memset(a, 0x5a, 10 * 4096 + 1);                                            // (heap memory under the stub)
    int on_stack = 0;
    CHECK(hsw_device_free(&on_stack) == HSW_ERR_INVALID_ARG);
    CHECK(hsw_device_free(a) == HSW_OK && hsw_device_free(a) == HSW_ERR_INVALID_ARG);
    CHECK(hsw_device_free(b) == HSW_OK && hsw_device_free(nullptr) == HSW_OK);
}

// the whole-region gadget through two synthesis passes with a geometry change in between
static void whole_region() {
    hsw_engine *e = nullptr;
    CHECK(hsw_engine_create_ex(0, nullptr, 8, 2, HSW_MODE_HALO2_INTERNALS, &e) == HSW_OK);
    const size_t sizes[2] = {128, 128};
    hsw_gadget *g = nullptr;
    CHECK(hsw_gadget_create_ex(e, sizes, 2, 1, HSW_GADGET_WHOLE_DIGEST, &g) == HSW_OK);
    hsw_hash_result r[2];
    auto digest_both = [&] {
        CHECK(hsw_gadget_digest(g, (const uint8_t *)"abc", 3, 0, &r[0]) == HSW_OK);
        CHECK(hsw_gadget_digest(g, nullptr, 0, 0, &r[1]) == HSW_OK);
        CHECK(hsw_gadget_digest(g, nullptr, 0, 0, &r[1]) == HSW_ERR_INVALID_ARG);      // a third digest: lib.rs:86 would panic
    };
    auto deliver = [&] {
        hsw_gadget_view v;
        CHECK(hsw_gadget_streams(g, &v) == HSW_OK);
        const size_t gate_cells = v.max_rows ? (size_t)(v.max_rows * v.columns) : (size_t)v.gate_cells;
        const size_t chip_cells = 2 * v.chip_col_stride;
        std::vector<uint64_t> gate(gate_cells * 4), lookup((size_t)v.lookup_cells * 4), cd(chip_cells * 4), cs(chip_cells * 4);
        hsw_region_host dst = {gate.data(), lookup.data(), cd.data(), cs.data()};
        CHECK(hsw_gadget_download_region(g, &dst) == HSW_OK);
        std::vector<uint64_t> gate8(gate_cells), lookup8((size_t)v.lookup_cells), cd8(chip_cells), cs8(chip_cells);
        std::vector<hsw_wide_cell> wide(4096);
        hsw_region_compact c = {gate8.data(), lookup8.data(), cd8.data(), cs8.data(), wide.data(), wide.size(), 0};
        CHECK(hsw_gadget_download_region_compact(g, &c) == HSW_OK);
        std::vector<uint64_t> widened(gate_cells * 4);
        CHECK(hsw_region_widen(gate8.data(), gate_cells, HSW_STREAM_GATE, wide.data(), c.n_wide, widened.data()) == HSW_OK);
        // distinct-value delivery: tape, packed witnesses into an exact-size buffer, replay into exact-size images
        hsw_region_tape tape;
        CHECK(hsw_gadget_region_tape(g, &tape) == HSW_OK);
        CHECK(tape.gate_cells == v.gate_cells && tape.lookup_cells + v.origin_lookups == v.lookup_cells);
        CHECK(tape.n_distinct > tape.gate_cells / 4 && tape.n_distinct < tape.gate_cells / 2 && tape.n_distinct <= tape.distinct_capacity);
        for (uint64_t i = 0; i < tape.gate_cells; i++) {
            const uint32_t code = tape.gate_code[i];
            CHECK((code & HSW_TAPE_CONST) ? (code & ~HSW_TAPE_CONST) < tape.n_consts : code < tape.n_distinct);
        }
        std::vector<uint64_t> distinct(tape.n_distinct * 4);
        size_t nd = 0;
        CHECK(hsw_gadget_download_region_distinct(g, distinct.data(), tape.n_distinct - 1, &nd) == HSW_ERR_TOO_LARGE && nd == tape.n_distinct);
        CHECK(hsw_gadget_download_region_distinct(g, distinct.data(), tape.n_distinct, &nd) == HSW_OK);
        for (unsigned threads : {1u, 5u}) CHECK(hsw_gadget_replay_region(g, distinct.data(), &dst, threads) == HSW_OK);
        hsw_verify_report rep;
        CHECK(hsw_gadget_verify(g, &rep) == HSW_OK);
        for (size_t h = 0; h < 2; h++) {
            hsw_result_cells rc;
            CHECK(hsw_gadget_result_cells(g, h, &rc) == HSW_OK);
            CHECK(rc.n_input_bytes == 128 && rc.output_byte_pos[31][0] >= v.origin_column);
        }
    };
    // placement: three candidate allocations of the chip columns, two freed again (LeakSanitizer watches)
    float ms_each[3];
    unsigned kept = 99;
    CHECK(hsw_gadget_place(g, 3, ms_each, &kept) == HSW_OK && kept < 3);
    CHECK(hsw_gadget_place(g, 0, nullptr, nullptr) == HSW_ERR_INVALID_ARG && hsw_gadget_place(g, 17, nullptr, nullptr) == HSW_ERR_INVALID_ARG);
    // pass 1: linear stream (the compact staging is sized for it)
    digest_both();
    CHECK(hsw_gadget_place(g, 2, nullptr, nullptr) == HSW_ERR_INVALID_ARG);           // digests assigned
    deliver();
    // pass 2: the FlexGate image of a Context that stands at (2, 131000) -- a larger geometry than pass 1
    // (the staging was sized once and a later, larger image overflowed it)
    CHECK(hsw_gadget_reset(g) == HSW_OK);
    uint64_t columns = 0;
    CHECK(hsw_gadget_set_columns(g, (1u << 17) - 9, &columns) == HSW_OK && columns == 3);
    CHECK(hsw_gadget_set_origin(g, 2, 131000, 1, 77) == HSW_OK);
    hsw_gadget_view v;
    CHECK(hsw_gadget_streams(g, &v) == HSW_OK && v.columns == 4 && v.origin_row == 131000 && v.lookup_cells == 77);
    CHECK(hsw_gadget_set_repr(g, HSW_REPR_MONTGOMERY) == HSW_OK);
    digest_both();
    CHECK(hsw_gadget_set_origin(g, 0, 0, 0, 0) == HSW_ERR_INVALID_ARG);                // digests assigned in this pass
    CHECK(r[0].prologue_lookup == 77 && r[0].block_cell == r[0].prologue_cell + 46 + 128 * 5);   // no zero cell
    CHECK(hsw_gadget_set_repr(g, HSW_REPR_CANONICAL) == HSW_OK);                       // (compact delivery packs canonical cells)
    deliver();
    // pass 3: back to the origin, taller columns
    CHECK(hsw_gadget_reset(g) == HSW_OK);
    CHECK(hsw_gadget_set_origin(g, 0, 0, 0, 0) == HSW_OK);
    CHECK(hsw_gadget_set_columns(g, 200000, &columns) == HSW_OK && columns == 2);
    digest_both();
    deliver();
    // pass 4: the same stream from another row and column: the tape is kept, only the witnesses' image positions
    // are worked out again (hsw_replay.cpp drop_region_tape_positions)
    hsw_region_tape before, after;
    CHECK(hsw_gadget_region_tape(g, &before) == HSW_OK);
    CHECK(hsw_gadget_reset(g) == HSW_OK);
    CHECK(hsw_gadget_set_origin(g, 1, 500, 0, 5) == HSW_OK);
    digest_both();
    deliver();
    CHECK(hsw_gadget_region_tape(g, &after) == HSW_OK);
    CHECK(after.gate_code == before.gate_code && after.distinct_capacity == before.distinct_capacity);
    CHECK(hsw_gadget_seek(g, 1) == HSW_OK);
    CHECK(hsw_gadget_digest(g, nullptr, 0, 0, &r[1]) == HSW_OK);
    hsw_gadget_destroy(g);
    // a second gadget on the same engine, destroyed AFTER the engine (its buffers do not need it)
    CHECK(hsw_gadget_create_ex(e, sizes, 2, 0, 0, &g) == HSW_OK);
    CHECK(hsw_gadget_digest(g, (const uint8_t *)"abc", 3, 0, &r[0]) == HSW_OK);
    CHECK(hsw_gadget_set_origin(g, 0, 1, 0, 0) == HSW_ERR_INVALID_ARG);                // block-stream context: no region
    hsw_shape s;
    CHECK(hsw_engine_shape(e, &s) == HSW_OK);
    std::vector<uint64_t> gate(2 * (size_t)s.gate_cells_per_block * 4), cd(2 * 4120 * 2 * 4), cs(2 * 4120 * 2 * 4);
    hsw_region_host dst = {gate.data(), nullptr, cd.data(), cs.data()};
    CHECK(hsw_gadget_download_region(g, &dst) == HSW_OK);
    hsw_engine_destroy(e);
    hsw_gadget_destroy(g);
}

// hsw_witness_digests (public entry): host_next_states must be pinned, device-mapped memory on EVERY call
// (a cached translation let a foreign pointer within 64 KiB of an earlier pinned one through)
static void pinned_pointer_validation() {
    hsw_engine *e = nullptr;
    CHECK(hsw_engine_create_ex(0, nullptr, 8, 2, HSW_MODE_HALO2_INTERNALS, &e) == HSW_OK);
    hsw_shape s;
    CHECK(hsw_engine_shape(e, &s) == HSW_OK);
    hsw_frame_shape fs;
    CHECK(hsw_frame_query(&s, 64, 0, &fs) == HSW_OK);
    const size_t cells = (size_t)fs.digest_cells + 1, rows = (size_t)hsw_chip_rows(&s, 0, 1);
    void *blocks = dev(64), *pre = dev(32), *next = dev(32), *gate = dev(cells * 32), *lookup = dev((size_t)fs.digest_lookups * 32);
    void *cd = dev(2 * rows * 32), *cs = dev(2 * rows * 32);
    hsw_frame_desc d;
    std::memset(&d, 0, sizeof d);
    d.input_len = 3; d.n_blocks = 1; d.num_round = 1;
    d.prologue_cell = 0; d.zero_cell = fs.prologue_cells;
    d.epilogue_cell = fs.prologue_cells + 1 + s.gate_cells_per_block;
    d.prologue_lookup = 0; d.epilogue_lookup = fs.prologue_lookups + s.lookup_cells_per_block;
    hsw_digests_args a;
    std::memset(&a, 0, sizeof a);
    a.blocks.d_blocks = (const uint8_t *)blocks; a.blocks.d_pre_states = (const uint32_t *)pre; a.blocks.n_blocks = 1;
    a.blocks.d_gate = (uint8_t *)gate + (fs.prologue_cells + 1) * 32;
    a.blocks.d_chip_dense = cd; a.blocks.d_chip_spread = cs; a.blocks.chip_col_stride = rows;
    a.blocks.d_next_states = (uint32_t *)next;
    a.blocks.d_lookup = (uint8_t *)lookup + fs.prologue_lookups * 32;
    a.descs = &d; a.n_digests = 1;
    a.d_blocks0 = (const uint8_t *)blocks; a.d_pre_states0 = (const uint32_t *)pre; a.d_next_states0 = (const uint32_t *)next;
    a.d_gate0 = gate; a.d_lookup0 = lookup;
    void *pinned = nullptr;
    CHECK(hsw_host_alloc(4096, &pinned) == HSW_OK);
    a.host_next_states = (uint32_t *)pinned + 64;                      // inside the pinned allocation: fine
    CHECK(hsw_witness_digests(e, &a) == HSW_OK);
    std::vector<uint32_t> heap(8);
    a.host_next_states = heap.data();                                 // ordinary heap memory: refused ...
    CHECK(hsw_witness_digests(e, &a) == HSW_ERR_INVALID_ARG);
    hsw_host_free(pinned);
    a.host_next_states = (uint32_t *)pinned + 64;                      // ... and so is the freed pinned buffer
    CHECK(hsw_witness_digests(e, &a) == HSW_ERR_INVALID_ARG);
    a.host_next_states = nullptr;
    CHECK(hsw_witness_digests(e, &a) == HSW_OK);
    hsw_engine_destroy(e);
    for (void *p : {blocks, pre, next, gate, lookup, cd, cs}) CHECK(hipFree(p) == hipSuccess);
}

int main() {
    block_batches();
    device_ranges();
    whole_region();
    pinned_pointer_validation();
    // everything the library allocated is gone with its engines and gadgets
    CHECK(hip_stub_live_device_allocations() == 0 && hip_stub_live_pinned_allocations() == 0 && hip_stub_live_events() == 0);
    std::puts("host lifecycle ok");
    return 0;
}
