// Host-only entry points of libhsw under AddressSanitizer + UBSan (CPU build only:
// GPU sanitizers are not available on the pool).  Built and run by
// tests/test_host_sanitizers.py; exits non-zero on any mismatch, and the
// sanitizers abort on any memory / UB error.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/hsw.h"

#define CHECK(cond)                                                        \
    do {                                                                   \
        if (!(cond)) {                                                     \
            std::fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            std::exit(1);                                                  \
        }                                                                  \
    } while (0)

int main() {
    // shapes: every table width, both modes
    for (uint32_t bits : {1u, 2u, 4u, 8u, 16u}) {
        for (uint32_t mode : {HSW_MODE_DEFAULT, HSW_MODE_HALO2_INTERNALS}) {
            hsw_shape s;
            CHECK(hsw_shape_query_ex(bits, 3, mode, &s) == HSW_OK);
            size_t n = 0;
            CHECK(hsw_gate_tape(&s, nullptr, 0, &n) == HSW_OK && n == s.gate_calls_per_block);
            std::vector<uint8_t> lens(n);
            CHECK(hsw_gate_tape(&s, lens.data(), n, nullptr) == HSW_OK);
            CHECK(hsw_gate_tape(&s, lens.data(), n - 1, nullptr) == HSW_ERR_INVALID_ARG);
            size_t cells = 0;
            for (uint8_t l : lens) cells += l;
            CHECK(cells == s.gate_cells_per_block);
            hsw_pack_plan plan;
            CHECK(hsw_pack_plan_query(&s, 3, 17, (uint64_t)s.gate_cells_per_block / 2 + 1000, &plan) == HSW_OK);
            CHECK(plan.n_breaks >= 5 && plan.n_breaks <= HSW_MAX_BREAKS);
            CHECK(hsw_pack_plan_query(&s, 100, 0, 1000, &plan) == HSW_ERR_TOO_LARGE);
            CHECK(hsw_chip_rows(&s, 5, 7) == (5 % 3 + (uint64_t)s.limb_calls_per_block * 7 + 2) / 3);
        }
    }
    {   // digest frame arithmetic: counts and tapes with exact-size buffers
        hsw_shape si, sd;
        CHECK(hsw_shape_query_ex(8, 2, HSW_MODE_HALO2_INTERNALS, &si) == HSW_OK);
        CHECK(hsw_shape_query_ex(8, 2, HSW_MODE_DEFAULT, &sd) == HSW_OK);
        hsw_frame_shape fs;
        CHECK(hsw_frame_query(&sd, 64, 0, &fs) == HSW_ERR_INVALID_ARG);      // frames are halo2-base internals
        CHECK(hsw_frame_query(&si, 100, 0, &fs) == HSW_ERR_SHAPE);
        CHECK(hsw_frame_query(&si, ((size_t)1 << 32) + 64, 0, &fs) == HSW_ERR_TOO_LARGE);   // a frame counts its blocks in 32 bits
        CHECK(hsw_frame_query(&si, (size_t)1 << 32, 1, &fs) == HSW_OK && fs.n_blocks == ((uint64_t)1 << 26));
        for (size_t maxb : {64u, 128u, 1024u}) {
            for (int rc = 0; rc < 2; rc++) {
                CHECK(hsw_frame_query(&si, maxb, rc, &fs) == HSW_OK);
                CHECK(fs.prologue_cells == 46 + maxb * (rc ? 5 : 1) && fs.epilogue_cells == 76 * (maxb / 64 + 1) + 288);
                for (int section = 0; section < 2; section++) {
                    size_t n = 0;
                    CHECK(hsw_frame_tape(&si, maxb, rc, section, nullptr, 0, &n) == HSW_OK);
                    CHECK(n == (section ? fs.epilogue_calls : fs.prologue_calls));
                    std::vector<uint8_t> lens(n);
                    CHECK(hsw_frame_tape(&si, maxb, rc, section, lens.data(), n, nullptr) == HSW_OK);
                    CHECK(hsw_frame_tape(&si, maxb, rc, section, lens.data(), n - 1, nullptr) == HSW_ERR_INVALID_ARG);
                    size_t cells = 0;
                    for (uint8_t l : lens) cells += l;
                    CHECK(cells == (section ? fs.epilogue_cells : fs.prologue_cells));
                }
                CHECK(hsw_frame_tape(&si, maxb, rc, 2, nullptr, 0, nullptr) == HSW_ERR_INVALID_ARG);
            }
        }
    }
    {   // constraint structure builders with exact-size buffers
        for (uint32_t mode : {HSW_MODE_DEFAULT, HSW_MODE_HALO2_INTERNALS}) {
            hsw_shape s;
            CHECK(hsw_shape_query_ex(8, 2, mode, &s) == HSW_OK);
            hsw_structure_counts c;
            CHECK(hsw_block_structure(&s, &c, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr) == HSW_OK);
            CHECK(c.gate_cells == s.gate_cells_per_block && c.limb_calls == s.limb_calls_per_block && c.assert_eq >= 3850);
            std::vector<uint8_t> kind(c.gate_cells);
            std::vector<int64_t> ref(c.gate_cells), aeq(2 * c.assert_eq), rng(2 * c.ranges), lk(c.lookups), chip(2 * c.limb_calls);
            std::vector<uint32_t> rows(c.gate_rows);
            int64_t ns[8];
            CHECK(hsw_block_structure(&s, nullptr, kind.data(), ref.data(), rows.data(), aeq.data(), rng.data(), lk.data(),
                                      chip.data(), ns) == HSW_OK);
            for (size_t i = 0; i < kind.size(); i++)
                CHECK(kind[i] <= HSW_KIND_EXISTING && (kind[i] != HSW_KIND_EXISTING || ref[i] < (int64_t)i));   // copies point backwards
            for (uint32_t r : rows) CHECK(r + 3 < c.gate_cells);
        }
        hsw_shape si;
        CHECK(hsw_shape_query_ex(8, 2, HSW_MODE_HALO2_INTERNALS, &si) == HSW_OK);
        for (int section = 0; section < 2; section++) {
            hsw_frame_structure_counts fc;
            CHECK(hsw_frame_structure(&si, 192, 1, section, &fc, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr) == HSW_OK);
            std::vector<uint8_t> kind(fc.cells);
            std::vector<int64_t> ref(fc.cells), aeq(2 * fc.assert_eq), ac(2 * fc.assert_const + 2), rng(2 * fc.ranges), lk(fc.lookups);
            std::vector<uint32_t> rows(fc.gate_rows);
            CHECK(hsw_frame_structure(&si, 192, 1, section, nullptr, kind.data(), ref.data(), rows.data(), aeq.data(), ac.data(),
                                      rng.data(), lk.data()) == HSW_OK);
            for (uint32_t r : rows) CHECK(r + 3 < fc.cells);
        }
        CHECK(hsw_frame_structure(&si, 192, 1, 2, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr) == HSW_ERR_INVALID_ARG);
    }
    hsw_shape bad;
    CHECK(hsw_shape_query(3, 2, &bad) == HSW_ERR_SHAPE);
    CHECK(hsw_shape_query(8, 0, &bad) == HSW_ERR_SHAPE);
    CHECK(hsw_shape_query_ex(8, 2, 7, &bad) == HSW_ERR_INVALID_ARG);

    // digest_prepare: every length around the padding boundaries, exact-size output buffers
    for (size_t maxb : {64u, 128u, 1024u}) {
        for (size_t len = 0; len + 9 <= maxb; len++) {
            std::vector<uint8_t> msg(len ? len : 1, 0xA5), blocks(maxb);
            uint32_t init[8];
            hsw_digest_info info;
            CHECK(hsw_digest_prepare(len ? msg.data() : nullptr, len, 0, maxb, blocks.data(), init, &info) == HSW_OK);
            CHECK(info.n_blocks == maxb / 64 && info.num_round == (len + 9 + 63) / 64);
            CHECK(blocks[len] == 0x80 && blocks[info.num_round * 64 - 1] == (uint8_t)(8 * len));
            for (size_t i = info.num_round * 64; i < maxb; i++) CHECK(blocks[i] == 0);
        }
        std::vector<uint8_t> big(maxb, 1);
        CHECK(hsw_digest_prepare(big.data(), maxb - 8, 0, maxb, nullptr, nullptr, nullptr) == HSW_ERR_TOO_LARGE);
    }
    {   // precomputed prefix: 192-byte message, 128 precomputed (lib.rs:587-611)
        std::vector<uint8_t> msg(192, 7), blocks(128);
        uint32_t init[8];
        hsw_digest_info info;
        CHECK(hsw_digest_prepare(msg.data(), 192, 128, 128, blocks.data(), init, &info) == HSW_OK);
        CHECK(info.precomputed_round == 2 && info.target_round == 2 && info.num_round == 4);
        CHECK(hsw_digest_prepare(msg.data(), 192, 100, 128, nullptr, nullptr, nullptr) == HSW_ERR_SHAPE);
    }
    // engine creation without a device must fail cleanly (or succeed on a GPU box)
    hsw_engine *e = nullptr;
    int rc = hsw_engine_create(0, nullptr, 8, 2, &e);
    CHECK(rc == HSW_OK || rc == HSW_ERR_NO_DEVICE);
    if (e) hsw_engine_destroy(e);
    CHECK(std::strlen(hsw_strerror(HSW_ERR_TOO_LARGE)) > 0 && hsw_last_error(nullptr)[0] == 0);
    std::puts("host sanity ok");
    return 0;
}
