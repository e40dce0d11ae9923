// hsw_api_region.cpp -- the C ABI beyond the block loop: digest frames, the constraint
// structure for replayers, and the on-device verification of blocks and frames.  See include/hsw.h.
#include <cstring>
#include <vector>

#include "hsw_engine.hpp"
#include "hsw_fr.hpp"
#include "hsw_frame.hpp"
#include "hsw_nounwind.hpp"
#include "hsw_kernels.h"
#include "hsw_structure.hpp"

extern "C" {

int hsw_block_structure(const hsw_shape *shape, hsw_structure_counts *counts, uint8_t *cell_kind,
                        int64_t *cell_ref, uint32_t *gate_rows, int64_t *assert_eq, int64_t *range,
                        int64_t *lookup_src, int64_t *chip, int64_t *next_state) try {
    if (!shape || shape->limbs_per_spread == 0 || 16 % shape->limbs_per_spread != 0) return HSW_ERR_INVALID_ARG;
    const hsw::BlockStructure st =
        hsw::StructureBuilder((int)shape->limbs_per_spread, shape->mode == HSW_MODE_HALO2_INTERNALS).block();
    if (st.kind.size() != shape->gate_cells_per_block || st.chip.size() != 2u * shape->limb_calls_per_block ||
        st.lookup_src.size() != shape->lookup_cells_per_block)
        return HSW_ERR_INVALID_ARG;                      // the builder and the layout arithmetic must agree
    if (counts) {
        counts->gate_cells = st.kind.size();
        counts->gate_rows = st.gate_rows.size();
        counts->assert_eq = st.assert_eq.size() / 2;
        counts->ranges = st.range.size() / 2;
        counts->lookups = st.lookup_src.size();
        counts->limb_calls = st.chip.size() / 2;
    }
    if (cell_kind && !st.kind.empty()) std::memcpy(cell_kind, st.kind.data(), st.kind.size());
    if (cell_ref && !st.ref.empty()) std::memcpy(cell_ref, st.ref.data(), st.ref.size() * sizeof(int64_t));
    if (gate_rows && !st.gate_rows.empty()) std::memcpy(gate_rows, st.gate_rows.data(), st.gate_rows.size() * sizeof(uint32_t));
    if (assert_eq && !st.assert_eq.empty()) std::memcpy(assert_eq, st.assert_eq.data(), st.assert_eq.size() * sizeof(int64_t));
    if (range && !st.range.empty()) std::memcpy(range, st.range.data(), st.range.size() * sizeof(int64_t));
    if (lookup_src && !st.lookup_src.empty()) std::memcpy(lookup_src, st.lookup_src.data(), st.lookup_src.size() * sizeof(int64_t));
    if (chip && !st.chip.empty()) std::memcpy(chip, st.chip.data(), st.chip.size() * sizeof(int64_t));
    if (next_state) std::memcpy(next_state, st.next_state, sizeof st.next_state);
    return HSW_OK;
} HSW_NO_UNWIND

// ------------------------------------------------------------ digest frames
int hsw_frame_query(const hsw_shape *shape, size_t max_variable_byte_size, int is_input_range_check,
                    hsw_frame_shape *out) try {
    if (!shape || !out) return HSW_ERR_INVALID_ARG;
    if (shape->mode != HSW_MODE_HALO2_INTERNALS) return HSW_ERR_INVALID_ARG;   // a frame is halo2-base internals
    if (max_variable_byte_size % 64 != 0) return HSW_ERR_SHAPE;                 // lib.rs:57-59
    if ((uint64_t)max_variable_byte_size > (1ull << 32)) return HSW_ERR_TOO_LARGE;   // FrameDesc::n_blocks is 32 bits
    const bool rc = is_input_range_check != 0;
    const uint64_t nb = max_variable_byte_size / 64;
    out->n_blocks = nb;
    out->prologue_cells = hsw::frame::prologue_cells(max_variable_byte_size, rc);
    out->epilogue_cells = hsw::frame::epilogue_cells(nb);
    out->prologue_lookups = hsw::frame::prologue_lookups(max_variable_byte_size, rc);
    out->epilogue_lookups = hsw::frame::E_LOOKUPS;
    out->prologue_calls = hsw::frame::prologue_calls(max_variable_byte_size, rc);
    out->epilogue_calls = hsw::frame::epilogue_calls(nb);
    out->digest_cells = out->prologue_cells + nb * shape->gate_cells_per_block + out->epilogue_cells;
    out->digest_lookups = out->prologue_lookups + nb * shape->lookup_cells_per_block + out->epilogue_lookups;
    return HSW_OK;
} HSW_NO_UNWIND

int hsw_frame_tape(const hsw_shape *shape, size_t max_variable_byte_size, int is_input_range_check,
                   int section, uint8_t *lens_out, size_t cap, size_t *n_calls) try {
    hsw_frame_shape fs;
    const int rc = hsw_frame_query(shape, max_variable_byte_size, is_input_range_check, &fs);
    if (rc != HSW_OK) return rc;
    if (section != 0 && section != 1) return HSW_ERR_INVALID_ARG;
    std::vector<uint8_t> lens;
    lens.reserve((size_t)(section == 0 ? fs.prologue_calls : fs.epilogue_calls));
    if (section == 0) {
        // lib.rs:124-165: lw, lw, mul, add, sub, is_less_than (7), its range_check (4), is_zero (8), lw, sub, 8 x lw
        static const uint8_t fixed[] = {1, 1, 4, 4, 4, 7, 4, 8, 1, 4, 1, 1, 1, 1, 1, 1, 1, 1};
        lens.assign(fixed, fixed + sizeof fixed);
        lens.insert(lens.end(), max_variable_byte_size, 1);                          // :170-173
        if (is_input_range_check) lens.insert(lens.end(), max_variable_byte_size, 4);   // :174-178
    } else {
        for (uint64_t n = 0; n <= fs.n_blocks; n++) {                                // :296-310
            lens.push_back(4); lens.push_back(8);                                    // is_equal = sub row + is_zero
            lens.insert(lens.end(), 8, 8);                                           // 8 x select
        }
        for (int w = 0; w < 8; w++) {                                                // :311-341
            for (int i = 0; i < 4; i++) { lens.push_back(1); lens.push_back(4); }    // load_witness + range_check 8
            lens.insert(lens.end(), 4, 4);                                           // 4 x mul_add
        }
    }
    if (n_calls) *n_calls = lens.size();
    if (lens_out) {
        if (cap < lens.size()) return HSW_ERR_INVALID_ARG;
        std::memcpy(lens_out, lens.data(), lens.size());
    }
    return HSW_OK;
} HSW_NO_UNWIND

// ------------------------------------------------------------ on-device verification
static int ensure_structure(hsw_engine *e) {
    if (e->d_structure) return HSW_OK;
    const hsw::BlockStructure st = hsw::StructureBuilder(e->limbs, e->mode == HSW_MODE_HALO2_INTERNALS).block();
    // the kernel checks constants / copies while it walks the gate rows: every such cell must sit in one
    std::vector<uint8_t> in_row(st.kind.size(), 0);
    for (uint32_t r : st.gate_rows) for (int j = 0; j < 4; j++) in_row[r + j] = 1;
    uint64_t fixed = 0;
    for (size_t c = 0; c < st.kind.size(); c++) {
        if (st.kind[c] != 0 && !in_row[c])
            return set_err(e, HSW_ERR_UNSUPPORTED, "structure has a fixed / copied cell outside every gate row");
        fixed += st.kind[c] != 0;
    }
    // one device allocation: [ref | assert_eq | range | chip | lookup_src | next_state | gate_rows | kind]
    const size_t n_i64 = st.ref.size() + st.assert_eq.size() + st.range.size() + st.chip.size() + st.lookup_src.size() + 8;
    const size_t bytes = n_i64 * 8 + st.gate_rows.size() * 4 + st.kind.size();
    std::vector<uint8_t> h(bytes);
    size_t at = 0;
    auto put = [&](const void *src, size_t n) { if (n) std::memcpy(h.data() + at, src, n); const size_t was = at; at += n; return was; };
    const size_t o_ref = put(st.ref.data(), st.ref.size() * 8), o_aeq = put(st.assert_eq.data(), st.assert_eq.size() * 8);
    const size_t o_rng = put(st.range.data(), st.range.size() * 8), o_chip = put(st.chip.data(), st.chip.size() * 8);
    const size_t o_lk = put(st.lookup_src.data(), st.lookup_src.size() * 8), o_ns = put(st.next_state, 64);
    const size_t o_rows = put(st.gate_rows.data(), st.gate_rows.size() * 4), o_kind = put(st.kind.data(), st.kind.size());
    void *d = nullptr;
    hipError_t he = hipMalloc(&d, bytes);
    if (he == hipSuccess) he = hipMemcpy(d, h.data(), bytes, hipMemcpyHostToDevice);
    if (he == hipSuccess && !e->d_report) he = hipMalloc((void **)&e->d_report, sizeof(hsw::VerifyReport));   // (hsw_verify_frames may have made it)
    if (he != hipSuccess) { if (d) (void)hipFree(d); return set_err(e, he == hipErrorOutOfMemory ? HSW_ERR_NOMEM : HSW_ERR_HIP, "structure upload", he); }
    e->d_structure = d;
    const uint8_t *b = static_cast<const uint8_t *>(d);
    hsw::VerifyParams &p = e->verify_tpl;
    p.ref = reinterpret_cast<const int64_t *>(b + o_ref);
    p.assert_eq = reinterpret_cast<const int64_t *>(b + o_aeq);
    p.range = reinterpret_cast<const int64_t *>(b + o_rng);
    p.chip = reinterpret_cast<const int64_t *>(b + o_chip);
    p.lookup_src = reinterpret_cast<const int64_t *>(b + o_lk);
    p.next_state_cells = reinterpret_cast<const int64_t *>(b + o_ns);
    p.gate_rows = reinterpret_cast<const uint32_t *>(b + o_rows);
    p.kind = b + o_kind;
    p.gate_cells = (uint32_t)st.kind.size();
    p.n_rows = (uint32_t)st.gate_rows.size();
    p.n_assert_eq = (uint32_t)(st.assert_eq.size() / 2);
    p.n_range = (uint32_t)(st.range.size() / 2);
    p.limb_calls = (uint32_t)(st.chip.size() / 2);
    p.lookup_cells = (uint32_t)st.lookup_src.size();
    e->verify_checks_per_block = fixed + p.n_rows + p.n_assert_eq + p.n_range;
    return HSW_OK;
}

int hsw_verify_blocks(hsw_engine *e, const hsw_witness_args *args, hsw_verify_report *report) try {
    if (!e || !args || !report) return HSW_ERR_INVALID_ARG;
    std::memset(report, 0, sizeof *report);
    if (args->n_blocks == 0) return HSW_OK;
    if (!args->d_gate || !args->d_blocks || !args->d_pre_states) return set_err(e, HSW_ERR_INVALID_ARG, "null pointer");
    if (args->flags & HSW_REPR_COMPACT64) return set_err(e, HSW_ERR_UNSUPPORTED, "hsw_verify_blocks checks 32-byte cells (canonical or Montgomery)");
    if (args->pack && args->pack->n_breaks > HSW_MAX_BREAKS) return set_err(e, HSW_ERR_INVALID_ARG, "too many column breaks");
    if (args->frame_every && e->mode != HSW_MODE_HALO2_INTERNALS)
        return set_err(e, HSW_ERR_INVALID_ARG, "digest frames need an engine created with HSW_MODE_HALO2_INTERNALS");
    if ((args->d_chip_dense == nullptr) != (args->d_chip_spread == nullptr)) return set_err(e, HSW_ERR_INVALID_ARG, "both chip families or none");
    if (args->d_lookup && e->mode != HSW_MODE_HALO2_INTERNALS) return set_err(e, HSW_ERR_INVALID_ARG, "d_lookup needs HSW_MODE_HALO2_INTERNALS");
    // the kernel reads cells as 16-byte pieces at (N % ncols) * chip_col_stride + row: the same alignment and
    // stride rules as for hsw_witness_blocks_ex, or a bad argument would become a device fault
    if (((uintptr_t)args->d_gate & 15u) || ((uintptr_t)args->d_chip_dense & 15u) || ((uintptr_t)args->d_chip_spread & 15u) ||
        ((uintptr_t)args->d_lookup & 15u))
        return set_err(e, HSW_ERR_INVALID_ARG, "stream buffers must be 16-byte aligned");
    if (((uintptr_t)args->d_blocks & 3u) || ((uintptr_t)args->d_pre_states & 3u) || ((uintptr_t)args->d_next_states & 3u))
        return set_err(e, HSW_ERR_INVALID_ARG, "inputs must be 4-byte aligned");
    if (args->d_chip_dense && args->chip_col_stride < hsw_chip_rows(&e->shape, args->spread_cursor0, args->n_blocks))
        return set_err(e, HSW_ERR_INVALID_ARG, "chip_col_stride smaller than hsw_chip_rows()");
    DeviceScope ds(e->device);
    if (!ds.ok) return set_err(e, HSW_ERR_NO_DEVICE, "hipSetDevice failed");
    const int rc = ensure_structure(e);
    if (rc != HSW_OK) return rc;
    hsw::VerifyParams p = e->verify_tpl;
    p.gate = args->d_gate; p.chip_dense = args->d_chip_dense; p.chip_spread = args->d_chip_spread; p.lookup = args->d_lookup;
    p.blocks = args->d_blocks; p.pre_states = args->d_pre_states; p.next_states = args->d_next_states;
    p.cursor0 = args->spread_cursor0; p.chip_col_stride = args->chip_col_stride;
    p.ncols = e->shape.num_advice_columns; p.num_bits_lookup = e->shape.num_bits_lookup;
    p.montgomery = (args->flags & HSW_REPR_MONTGOMERY) ? 1u : 0u;
    p.gate_cell0 = p.lookup_cell0 = 0;
    p.frame_every = args->frame_every; p.frame_cells = args->frame_cells; p.frame_lookups = args->frame_lookups;
    p.n_breaks = args->pack ? args->pack->n_breaks : 0;
    for (uint32_t k = 0; k < p.n_breaks; k++) { p.break_cell[k] = args->pack->break_cell[k]; p.break_gap[k] = args->pack->break_gap[k]; }
    p.report = e->d_report;
    // four workgroups per block for large batches (measured 3-7 % better than one, tools/verify_slices.py);
    // smaller batches are sliced so that ~1,024 workgroups run
    p.slices = e->verify_slices > 0 ? (uint32_t)e->verify_slices
               : (uint32_t)(args->n_blocks >= 1024 ? 4 : (1024 / args->n_blocks > 64 ? 64 : 1024 / args->n_blocks));
    const hsw::VerifyReport zero{0, ~0ull, 0};
    hipError_t he = hipMemcpyAsync(e->d_report, &zero, sizeof zero, hipMemcpyHostToDevice, e->stream);
    if (he == hipSuccess) he = hipEventRecord(e->ev0, e->stream);
    if (he == hipSuccess) he = hsw::launch_verify(p, args->n_blocks, e->stream);
    if (he == hipSuccess) he = hipEventRecord(e->ev1, e->stream);
    hsw::VerifyReport got{};
    if (he == hipSuccess) he = hipMemcpyAsync(&got, e->d_report, sizeof got, hipMemcpyDeviceToHost, e->stream);
    if (he == hipSuccess) he = hipStreamSynchronize(e->stream);
    if (he == hipSuccess) he = hipEventElapsedTime(&report->kernel_ms, e->ev0, e->ev1);
    e->timed = false;
    if (he != hipSuccess) return set_err(e, HSW_ERR_HIP, "hsw_verify_blocks", he);
    report->violations = got.violations;
    uint64_t per_block = e->verify_checks_per_block;
    if (args->d_chip_dense) per_block += 4ull * p.limb_calls;
    if (args->d_lookup) per_block += 2ull * p.lookup_cells;
    if (args->d_next_states) per_block += 8;
    report->checks = per_block * args->n_blocks;
    if (got.violations) {
        report->first_block = got.first_key >> 36;
        report->first_cell = (int64_t)((got.first_key >> 4) & 0xffffffffu);
        report->first_class = (uint32_t)(got.first_key & 15u);
    }
    return HSW_OK;
} HSW_NO_UNWIND

int hsw_verify_frames(hsw_engine *e, const hsw_frame_desc *descs, size_t n, const uint8_t *d_blocks,
                      const uint32_t *d_pre_states, const uint32_t *d_next_states, const void *d_gate,
                      const void *d_lookup, const hsw_pack_plan *pack, uint32_t flags, hsw_verify_report *report) try {
    if (!e || !report) return HSW_ERR_INVALID_ARG;
    std::memset(report, 0, sizeof *report);
    if (n == 0) return HSW_OK;
    if (!descs || !d_blocks || !d_pre_states || !d_next_states || !d_gate) return set_err(e, HSW_ERR_INVALID_ARG, "null pointer");
    if (e->mode != HSW_MODE_HALO2_INTERNALS)
        return set_err(e, HSW_ERR_INVALID_ARG, "digest frames need an engine created with HSW_MODE_HALO2_INTERNALS");
    if (flags & HSW_REPR_COMPACT64) return set_err(e, HSW_ERR_UNSUPPORTED, "hsw_verify_frames checks 32-byte cells (canonical or Montgomery)");
    if (pack && pack->n_breaks > HSW_MAX_BREAKS) return set_err(e, HSW_ERR_INVALID_ARG, "too many column breaks");
    if (((uintptr_t)d_gate & 15u) || ((uintptr_t)d_lookup & 15u)) return set_err(e, HSW_ERR_INVALID_ARG, "gate / lookup buffer not 16-byte aligned");
    if (((uintptr_t)d_blocks & 3u) || ((uintptr_t)d_pre_states & 3u) || ((uintptr_t)d_next_states & 3u))
        return set_err(e, HSW_ERR_INVALID_ARG, "inputs must be 4-byte aligned");
    for (size_t i = 0; i < n; i++) {
        if (descs[i].n_blocks == 0 || descs[i].n_blocks != descs[0].n_blocks ||
            (descs[i].is_input_range_check != 0) != (descs[0].is_input_range_check != 0))
            return set_err(e, HSW_ERR_INVALID_ARG, "one call verifies equally shaped digests (same n_blocks, same range-check setting)");
        if ((uint64_t)descs[i].num_round != (descs[i].input_len + 9 + 63) / 64 || descs[i].precomputed_round > descs[i].num_round ||
            descs[i].num_round - descs[i].precomputed_round > descs[i].n_blocks)       // lib.rs:80-90
            return set_err(e, HSW_ERR_INVALID_ARG, "inconsistent digest descriptor");
    }
    DeviceScope ds(e->device);
    if (!ds.ok) return set_err(e, HSW_ERR_NO_DEVICE, "hipSetDevice failed");
    if (!e->d_report) {
        hipError_t h0 = hipMalloc((void **)&e->d_report, sizeof(hsw::VerifyReport));
        if (h0 != hipSuccess) return set_err(e, HSW_ERR_NOMEM, "hipMalloc", h0);
    }
    // structures of this shape + the descriptors, in one temporary device buffer
    hsw::FrameStructureBuilder fb;
    const hsw::FrameStructure st[2] = {fb.prologue((uint64_t)descs[0].n_blocks * 64, descs[0].is_input_range_check != 0),
                                       fb.epilogue(descs[0].n_blocks)};
    std::vector<hsw::FrameDesc> hd(n);
    for (size_t i = 0; i < n; i++) {
        hsw::FrameDesc &o = hd[i];
        const hsw_frame_desc &d = descs[i];
        o.input_len = d.input_len; o.first_block = d.first_block; o.prologue_cell = d.prologue_cell; o.epilogue_cell = d.epilogue_cell;
        o.prologue_lookup = d.prologue_lookup; o.epilogue_lookup = d.epilogue_lookup; o.zero_cell = d.zero_cell;
        o.n_blocks = d.n_blocks; o.num_round = d.num_round; o.precomputed_round = d.precomputed_round;
        o.range_check_inputs = d.is_input_range_check ? 1u : 0u;
    }
    std::vector<uint8_t> h;
    auto put = [&](const void *src, size_t bytes) { const size_t at = (h.size() + 7) & ~(size_t)7; h.resize(at + bytes); if (bytes) std::memcpy(h.data() + at, src, bytes); return at; };
    size_t off[2][7];
    for (int s2 = 0; s2 < 2; s2++) {
        off[s2][0] = put(st[s2].kind.data(), st[s2].kind.size());
        off[s2][1] = put(st[s2].ref.data(), st[s2].ref.size() * 8);
        off[s2][2] = put(st[s2].gate_rows.data(), st[s2].gate_rows.size() * 4);
        off[s2][3] = put(st[s2].assert_eq.data(), st[s2].assert_eq.size() * 8);
        off[s2][4] = put(st[s2].assert_const.data(), st[s2].assert_const.size() * 8);
        off[s2][5] = put(st[s2].range.data(), st[s2].range.size() * 8);
        off[s2][6] = put(st[s2].lookup_src.data(), st[s2].lookup_src.size() * 8);
    }
    const size_t o_desc = put(hd.data(), hd.size() * sizeof(hsw::FrameDesc));
    uint8_t *dbuf = nullptr;
    hipError_t he = hipMalloc((void **)&dbuf, h.size());
    if (he != hipSuccess) return set_err(e, he == hipErrorOutOfMemory ? HSW_ERR_NOMEM : HSW_ERR_HIP, "hipMalloc", he);
    hsw::FrameVerifyParams p{};
    p.descs = reinterpret_cast<const hsw::FrameDesc *>(dbuf + o_desc);
    p.gate = d_gate; p.lookup = d_lookup; p.blocks = d_blocks; p.pre_states = d_pre_states; p.next_states = d_next_states;
    p.n_breaks = pack ? pack->n_breaks : 0;
    p.montgomery = (flags & HSW_REPR_MONTGOMERY) ? 1u : 0u;
    for (uint32_t k = 0; k < p.n_breaks; k++) { p.break_cell[k] = pack->break_cell[k]; p.break_gap[k] = pack->break_gap[k]; }
    uint64_t checks = 0;
    for (int s2 = 0; s2 < 2; s2++) {
        hsw::FrameVerifyParams::Section &S = s2 ? p.epi : p.pro;
        S.cells = (uint32_t)st[s2].kind.size(); S.n_rows = (uint32_t)st[s2].gate_rows.size();
        S.n_assert_eq = (uint32_t)(st[s2].assert_eq.size() / 2); S.n_assert_const = (uint32_t)(st[s2].assert_const.size() / 2);
        S.n_range = (uint32_t)(st[s2].range.size() / 2); S.n_lookup = (uint32_t)st[s2].lookup_src.size();
        S.kind = dbuf + off[s2][0];
        S.ref = reinterpret_cast<const int64_t *>(dbuf + off[s2][1]);
        S.gate_rows = reinterpret_cast<const uint32_t *>(dbuf + off[s2][2]);
        S.assert_eq = reinterpret_cast<const int64_t *>(dbuf + off[s2][3]);
        S.assert_const = reinterpret_cast<const int64_t *>(dbuf + off[s2][4]);
        S.range = reinterpret_cast<const int64_t *>(dbuf + off[s2][5]);
        S.lookup_src = reinterpret_cast<const int64_t *>(dbuf + off[s2][6]);
        uint64_t fixed = 0;
        for (uint8_t k : st[s2].kind) fixed += k != 0;
        checks += fixed + S.n_rows + S.n_assert_eq + S.n_assert_const + S.n_range + (d_lookup ? 2ull * S.n_lookup : 0);
    }
    checks += 2 + 8 + 64ull * descs[0].n_blocks + 8ull * (descs[0].n_blocks - 1);      // facts and links
    p.report = e->d_report;
    const hsw::VerifyReport zero{0, ~0ull, 0};
    he = hipMemcpyAsync(dbuf, h.data(), h.size(), hipMemcpyHostToDevice, e->stream);
    if (he == hipSuccess) he = hipMemcpyAsync(e->d_report, &zero, sizeof zero, hipMemcpyHostToDevice, e->stream);
    if (he == hipSuccess) he = hipEventRecord(e->ev0, e->stream);
    if (he == hipSuccess) he = hsw::launch_verify_frames(p, n, e->stream);
    if (he == hipSuccess) he = hipEventRecord(e->ev1, e->stream);
    hsw::VerifyReport got{};
    if (he == hipSuccess) he = hipMemcpyAsync(&got, e->d_report, sizeof got, hipMemcpyDeviceToHost, e->stream);
    if (he == hipSuccess) he = hipStreamSynchronize(e->stream);
    if (he == hipSuccess) he = hipEventElapsedTime(&report->kernel_ms, e->ev0, e->ev1);
    e->timed = false;
    (void)hipFree(dbuf);
    if (he != hipSuccess) return set_err(e, HSW_ERR_HIP, "hsw_verify_frames", he);
    report->violations = got.violations;
    report->checks = checks * n;
    if (got.violations) {
        report->first_block = got.first_key >> 36;
        report->first_cell = (int64_t)((got.first_key >> 4) & 0xffffffffu);
        report->first_class = (uint32_t)(got.first_key & 15u);
    }
    return HSW_OK;
} HSW_NO_UNWIND

int hsw_frame_structure(const hsw_shape *shape, size_t max_variable_byte_size, int is_input_range_check,
                        int section, hsw_frame_structure_counts *counts, uint8_t *cell_kind, int64_t *cell_ref,
                        uint32_t *gate_rows, int64_t *assert_eq, int64_t *assert_const, int64_t *range,
                        int64_t *lookup_src) try {
    hsw_frame_shape fs;
    const int rc = hsw_frame_query(shape, max_variable_byte_size, is_input_range_check, &fs);
    if (rc != HSW_OK) return rc;
    if (section != 0 && section != 1) return HSW_ERR_INVALID_ARG;
    hsw::FrameStructureBuilder b;
    const hsw::FrameStructure st = section == 0 ? b.prologue(max_variable_byte_size, is_input_range_check != 0)
                                                : b.epilogue(fs.n_blocks);
    if (st.kind.size() != (section ? fs.epilogue_cells : fs.prologue_cells) ||
        st.lookup_src.size() != (section ? fs.epilogue_lookups : fs.prologue_lookups))
        return HSW_ERR_INVALID_ARG;                      // the builder and the layout arithmetic must agree
    if (counts) {
        counts->cells = st.kind.size();
        counts->gate_rows = st.gate_rows.size();
        counts->assert_eq = st.assert_eq.size() / 2;
        counts->assert_const = st.assert_const.size() / 2;
        counts->ranges = st.range.size() / 2;
        counts->lookups = st.lookup_src.size();
    }
    if (cell_kind && !st.kind.empty()) std::memcpy(cell_kind, st.kind.data(), st.kind.size());
    if (cell_ref && !st.ref.empty()) std::memcpy(cell_ref, st.ref.data(), st.ref.size() * sizeof(int64_t));
    if (gate_rows && !st.gate_rows.empty()) std::memcpy(gate_rows, st.gate_rows.data(), st.gate_rows.size() * sizeof(uint32_t));
    if (assert_eq && !st.assert_eq.empty()) std::memcpy(assert_eq, st.assert_eq.data(), st.assert_eq.size() * sizeof(int64_t));
    if (assert_const && !st.assert_const.empty()) std::memcpy(assert_const, st.assert_const.data(), st.assert_const.size() * sizeof(int64_t));
    if (range && !st.range.empty()) std::memcpy(range, st.range.data(), st.range.size() * sizeof(int64_t));
    if (lookup_src && !st.lookup_src.empty()) std::memcpy(lookup_src, st.lookup_src.data(), st.lookup_src.size() * sizeof(int64_t));
    return HSW_OK;
} HSW_NO_UNWIND

// k^-1 mod p for k < n, both representations, on the device (k = 0 -> 0, never read)
static int ensure_inv_table(hsw_engine *e, size_t n) {
    if (n <= e->inv_n) return HSW_OK;
    size_t cap = e->inv_n ? e->inv_n : 64;
    while (cap < n) cap *= 2;
    std::vector<uint64_t> canon(4 * cap, 0), mont(4 * cap, 0);
    for (size_t k = 1; k < cap; k++) {
        const hsw::fr::Fe im = hsw::fr::inv_mont((uint64_t)k);
        const hsw::fr::Fe ic = hsw::fr::from_mont(im);
        std::memcpy(&mont[4 * k], im.l, 32);
        std::memcpy(&canon[4 * k], ic.l, 32);
    }
    uint64_t *d[2] = {nullptr, nullptr};
    hipError_t he = hipMalloc((void **)&d[0], cap * 32);
    if (he == hipSuccess) he = hipMalloc((void **)&d[1], cap * 32);
    if (he == hipSuccess) he = hipMemcpy(d[0], canon.data(), cap * 32, hipMemcpyHostToDevice);
    if (he == hipSuccess) he = hipMemcpy(d[1], mont.data(), cap * 32, hipMemcpyHostToDevice);
    if (he != hipSuccess) {
        if (d[0]) (void)hipFree(d[0]);
        if (d[1]) (void)hipFree(d[1]);
        return set_err(e, he == hipErrorOutOfMemory ? HSW_ERR_NOMEM : HSW_ERR_HIP, "inverse table", he);
    }
    // earlier launches may still read the old table: drain the stream before freeing it
    if (e->d_inv_tbl[0]) { (void)hipStreamSynchronize(e->stream); (void)hipFree(e->d_inv_tbl[0]); (void)hipFree(e->d_inv_tbl[1]); }
    e->d_inv_tbl[0] = d[0];
    e->d_inv_tbl[1] = d[1];
    e->inv_n = cap;
    return HSW_OK;
}

// One public descriptor checked (lib.rs:80-90) and turned into the device's form.
static int convert_frame_desc(hsw_engine *e, const hsw_frame_desc &d, hsw::FrameDesc *out) {
    if (d.n_blocks == 0) return set_err(e, HSW_ERR_UNSUPPORTED, "a digest frame needs max_variable_byte_size >= 64");
    if ((uint64_t)d.num_round != (d.input_len + 9 + 63) / 64)
        return set_err(e, HSW_ERR_INVALID_ARG, "num_round is not ceil((input_len + 9) / 64) (lib.rs:80-84)");
    if (d.precomputed_round > d.num_round || d.num_round - d.precomputed_round > d.n_blocks)
        return set_err(e, HSW_ERR_TOO_LARGE, "padded message does not fit max_variable_byte_size (lib.rs:90)");
    hsw::FrameDesc &o = *out;
    o.input_len = d.input_len; o.first_block = d.first_block;
    o.prologue_cell = d.prologue_cell; o.epilogue_cell = d.epilogue_cell;
    o.prologue_lookup = d.prologue_lookup; o.epilogue_lookup = d.epilogue_lookup;
    o.zero_cell = d.zero_cell; o.n_blocks = d.n_blocks; o.num_round = d.num_round;
    o.precomputed_round = d.precomputed_round; o.range_check_inputs = d.is_input_range_check ? 1u : 0u;
    return HSW_OK;
}

// Checks n frame descriptors and stages them in one of the engine's pinned, device-mapped descriptor
// buffers (the kernel reads them in place: no H2D copy).  The caller records slot->done after its launch.
static int stage_frame_descs(hsw_engine *e, const hsw_frame_desc *descs, size_t n, hsw::FrameDesc **d_descs,
                             size_t *max_blocks_out, hsw_engine::FrameSlot **slot_out) {
    hipError_t he;
    hsw_engine::FrameSlot &slot = e->frame_slot[e->frame_next++ & 3u];
    if (slot.inflight) {                      // the launch that last used this slot must have read it
        he = hipEventSynchronize(slot.done);
        if (he != hipSuccess) return set_err(e, HSW_ERR_HIP, "hipEventSynchronize", he);
        slot.inflight = false;
    }
    if (slot.cap < n) {
        if (slot.h) (void)hipHostFree(slot.h);
        slot.h = nullptr; slot.cap = 0;
        size_t cap = 16;
        while (cap < n) cap *= 2;
        he = hipHostMalloc((void **)&slot.h, cap * sizeof(hsw::FrameDesc), hipHostMallocMapped);
        if (he != hipSuccess) return set_err(e, he == hipErrorOutOfMemory ? HSW_ERR_NOMEM : HSW_ERR_HIP, "hipHostMalloc", he);
        he = hipHostGetDevicePointer((void **)&slot.d, slot.h, 0);
        if (he != hipSuccess) return set_err(e, HSW_ERR_HIP, "hipHostGetDevicePointer", he);
        slot.cap = cap;
    }
    if (!slot.done) {
        he = hipEventCreateWithFlags(&slot.done, hipEventDisableTiming);
        if (he != hipSuccess) return set_err(e, HSW_ERR_HIP, "hipEventCreate", he);
    }
    size_t max_blocks = 0;
    for (size_t i = 0; i < n; i++) {
        const int rc1 = convert_frame_desc(e, descs[i], &slot.h[i]);
        if (rc1 != HSW_OK) return rc1;
        if (descs[i].n_blocks > max_blocks) max_blocks = descs[i].n_blocks;
    }
    int rc = ensure_inv_table(e, max_blocks + 1);
    if (rc != HSW_OK) return rc;
    *d_descs = slot.d;
    *max_blocks_out = max_blocks;
    *slot_out = &slot;
    return HSW_OK;
}

static int check_frame_args(hsw_engine *e, const hsw_frame_desc *descs, const uint8_t *d_blocks,
                            const uint32_t *d_pre_states, void *d_gate, void *d_lookup, const hsw_pack_plan *pack,
                            uint32_t flags, hsw::FrameBreaks *brk) {
    if (!descs || !d_blocks || !d_pre_states || !d_gate) return set_err(e, HSW_ERR_INVALID_ARG, "null pointer");
    if (e->mode != HSW_MODE_HALO2_INTERNALS)
        return set_err(e, HSW_ERR_INVALID_ARG, "digest frames need an engine created with HSW_MODE_HALO2_INTERNALS");
    if (flags & ~HSW_REPR_MASK) return set_err(e, HSW_ERR_INVALID_ARG, "unknown flag bits");
    if (flags & HSW_REPR_COMPACT64)
        return set_err(e, HSW_ERR_UNSUPPORTED, "digest frames hold full-width cells: no HSW_REPR_COMPACT64");
    if (((uintptr_t)d_gate & 15u) || ((uintptr_t)d_lookup & 15u))
        return set_err(e, HSW_ERR_INVALID_ARG, "gate / lookup buffer not 16-byte aligned");
    if (pack && pack->n_breaks > HSW_MAX_BREAKS) return set_err(e, HSW_ERR_INVALID_ARG, "too many column breaks");
    *brk = hsw::FrameBreaks{};
    for (uint32_t k = 0; k < HSW_MAX_BREAKS; k++) brk->cell[k] = ~0ull;      // unused entries: beyond every stream cell
    if (pack) {
        brk->n = pack->n_breaks;
        for (uint32_t k = 0; k < pack->n_breaks; k++) { brk->cell[k] = pack->break_cell[k]; brk->gap[k] = pack->break_gap[k]; }
    }
    return HSW_OK;
}

int hsw_witness_frames(hsw_engine *e, const hsw_frame_desc *descs, size_t n, const uint8_t *d_blocks,
                       const uint32_t *d_pre_states, const uint32_t *d_next_states, void *d_gate,
                       void *d_lookup, const hsw_pack_plan *pack, uint32_t flags) try {
    if (!e) return HSW_ERR_INVALID_ARG;
    if (n == 0) return HSW_OK;
    if (!d_next_states) return set_err(e, HSW_ERR_INVALID_ARG, "null pointer");
    hsw::FrameBreaks brk{};
    int rc = check_frame_args(e, descs, d_blocks, d_pre_states, d_gate, d_lookup, pack, flags, &brk);
    if (rc != HSW_OK) return rc;
    DeviceScope ds(e->device);
    if (!ds.ok) return set_err(e, HSW_ERR_NO_DEVICE, "hipSetDevice failed");
    hsw::FrameDesc *d_descs = nullptr;
    size_t max_blocks = 0;
    hsw_engine::FrameSlot *slot = nullptr;
    rc = stage_frame_descs(e, descs, n, &d_descs, &max_blocks, &slot);
    if (rc != HSW_OK) return rc;
    const bool mont = (flags & HSW_REPR_MONTGOMERY) != 0;
    hipError_t he = hsw::launch_frames(d_descs, n, d_blocks, d_pre_states, d_next_states, e->d_inv_tbl[mont ? 1 : 0],
                            d_gate, d_lookup, brk,
                            /* workgroups per digest: one per 4 blocks (256 input bytes each), at most 64 */
                            (unsigned)(max_blocks / 4 < 1 ? 1 : (max_blocks / 4 > 64 ? 64 : max_blocks / 4)), mont, e->stream);
    if (he != hipSuccess) return set_err(e, HSW_ERR_HIP, "launch hsw_frame_kernel", he);
    he = hipEventRecord(slot->done, e->stream);
    if (he != hipSuccess) return set_err(e, HSW_ERR_HIP, "hipEventRecord", he);
    slot->inflight = true;
    return HSW_OK;
} HSW_NO_UNWIND

}  // extern "C"

// dev_next_states: the DEVICE address of a->host_next_states when the caller already holds it (the gadget: its
// own pinned staging, mapped once at creation), else NULL = ask the runtime.  The public entry point always asks:
// round 2 cached one (host, device) pair per engine and translated every later pointer within 64 KiB above it by
// offset -- a pointer from another (or from no) pinned allocation in that window skipped the check and the kernel
// stored the next states through a stale or unmapped address.
int hsw_witness_digests_impl(hsw_engine *e, const hsw_digests_args *a, uint32_t *dev_next_states) {
    if (!e || !a) return HSW_ERR_INVALID_ARG;
    const hsw_witness_args &b = a->blocks;
    if (a->n_digests == 0 || b.n_blocks == 0) return HSW_OK;
    if (!a->d_next_states0) return set_err(e, HSW_ERR_INVALID_ARG, "null pointer");
    hsw::FrameBreaks brk{};
    int rc = check_frame_args(e, a->descs, a->d_blocks0, a->d_pre_states0, a->d_gate0, a->d_lookup0, a->frame_pack,
                              b.flags & HSW_REPR_MASK, &brk);
    if (rc != HSW_OK) return rc;
    // the block streams of the call must be exactly the blocks of its digests, in order, with the frames
    // of equally sized digests in between (what frame_every describes)
    size_t sum = 0;
    for (size_t i = 0; i < a->n_digests; i++) {
        if (a->descs[i].n_blocks != a->descs[0].n_blocks)
            return set_err(e, HSW_ERR_INVALID_ARG, "one call = equally sized digests");
        sum += a->descs[i].n_blocks;
    }
    if (sum != b.n_blocks || (a->n_digests > 1 && b.frame_every != a->descs[0].n_blocks))
        return set_err(e, HSW_ERR_INVALID_ARG, "blocks.n_blocks / frame_every do not match the digests");
    DeviceScope ds(e->device);
    if (!ds.ok) return set_err(e, HSW_ERR_NO_DEVICE, "hipSetDevice failed");
    if (!hsw_small_eligible(e, b.n_blocks) || a->descs[0].n_blocks > hsw::SMALL_FRAME_MAX_BLOCKS) {   // (the frame waves stage the candidate states in LDS)
        // two launches: the expansion, then the frames from the next states it left in HBM
        rc = hsw_witness_blocks_ex(e, &b);
        if (rc == HSW_OK)
            rc = hsw_witness_frames(e, a->descs, a->n_digests, a->d_blocks0, a->d_pre_states0, a->d_next_states0,
                                    a->d_gate0, a->d_lookup0, a->frame_pack, b.flags & HSW_REPR_MASK);
        if (rc == HSW_OK && a->host_next_states && b.d_next_states) {
            hipError_t he = hipMemcpyAsync(a->host_next_states, b.d_next_states, b.n_blocks * 32, hipMemcpyDeviceToHost, e->stream);
            if (he != hipSuccess) return set_err(e, HSW_ERR_HIP, "D2H next states", he);
        }
        return rc;
    }
    // ONE launch: frame waves ride on the small-batch kernel's grid.  The first descriptor travels by value in
    // the kernel arguments; only further ones are staged in pinned memory (a single digest -- the reference's
    // bench circuit -- then needs no staging slot, no event and no read of host memory to get going).
    hsw::SmallFrames fr{};
    hsw::FrameDesc *d_descs = nullptr;
    size_t max_blocks = a->descs[0].n_blocks;
    hsw_engine::FrameSlot *slot = nullptr;
    if (a->n_digests == 1) {
        rc = convert_frame_desc(e, a->descs[0], &fr.d0);
        if (rc == HSW_OK) rc = ensure_inv_table(e, max_blocks + 1);
        if (rc != HSW_OK) return rc;
    } else {
        rc = stage_frame_descs(e, a->descs, a->n_digests, &d_descs, &max_blocks, &slot);
        if (rc != HSW_OK) return rc;
        fr.d0 = *slot->h;
    }
    fr.descs = d_descs;
    fr.inv_tbl = e->d_inv_tbl[(b.flags & HSW_REPR_MONTGOMERY) ? 1 : 0];
    fr.blocks0 = a->d_blocks0;
    fr.pre0 = a->d_pre_states0;
    fr.gate0 = a->d_gate0;
    fr.lookup0 = a->d_lookup0;
    fr.n_frames = (uint32_t)a->n_digests;
    fr.max_frame_blocks = (uint32_t)max_blocks;
    fr.byte_waves = (uint32_t)(max_blocks > 32 ? 32 : max_blocks);                  // one wave per 64 input bytes
    // state waves: one frame cell per work item (hsw_frame_body.hpp), about two per thread -- each wave repeats the
    // last block's 64-round recurrence first, so more waves cost SIMDs, not time
    const size_t state_items = 192 + (size_t)hsw::frame::E_STATE * (max_blocks + 1) + 32 * 11;
    fr.state_waves = (uint32_t)((state_items + 127) / 128 < 16 ? (state_items + 127) / 128 : 16);
    fr.brk = brk;
    uint32_t *host_next = dev_next_states;
    if (a->host_next_states && !host_next) {
        hipError_t he = hipHostGetDevicePointer((void **)&host_next, a->host_next_states, 0);
        if (he != hipSuccess || !host_next) {
            (void)hipGetLastError();
            return set_err(e, HSW_ERR_INVALID_ARG, "host_next_states is not pinned, device-mapped host memory (hsw_host_alloc)", he);
        }
    }
    rc = hsw_witness_blocks_impl(e, &b, &fr, host_next);
    if (rc != HSW_OK) return rc;
    if (slot) {
        hipError_t he = hipEventRecord(slot->done, e->stream);
        if (he != hipSuccess) return set_err(e, HSW_ERR_HIP, "hipEventRecord", he);
        slot->inflight = true;
    }
    return HSW_OK;
}

extern "C" int hsw_witness_digests(hsw_engine *e, const hsw_digests_args *a) try {
    return hsw_witness_digests_impl(e, a, nullptr);
} HSW_NO_UNWIND
