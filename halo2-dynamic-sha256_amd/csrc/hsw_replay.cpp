// hsw_replay.cpp -- distinct-value delivery of a whole region (include/hsw.h "distinct-value delivery").
//
// A CPU prover wants the advice columns in host memory, and for a whole region that is PCIe-bound: 32 bytes for
// every cell (0.79 ms for the reference's bench circuit, of which 0.03 ms is the synthesis).  But ~60 % of a
// region's cells are copies of earlier cells or gate constants, at positions that do not depend on the input --
// the same constraint structure the on-device verifier walks (hsw_structure.hpp: Witness / Constant / Existing per
// cell, lookup sources, chip ties; linked across prologue, zero cell, blocks and epilogue exactly as
// tests/test_structure.py links it against the oracle's recorder).  So:
//   * the TAPE (input independent, built once per gadget layout on the host): for every cell of the gate, lookup
//     and chip streams either "constant #k" or "distinct value #w" -- copy chains resolved to their root, so a
//     replay never reads a cell it wrote and any number of threads can rebuild any part of the image;
//   * the DEVICE packs the new witnesses only (one gather launch) and they alone cross PCIe, in the gadget's cell
//     representation (Montgomery for a halo2 prover);
//   * hsw_gadget_replay_region rebuilds the column image + lookup column + chip columns on the host -- or the
//     consumer reads distinct[code[i]] straight through the tape and never materialises the image.
#include <hip/hip_runtime.h>

#include <cstring>
#include <map>
#include <new>
#include <thread>
#include <vector>

#include "hsw_fr.hpp"
#include "hsw_frame.hpp"
#include "hsw_gadget.hpp"
#include "hsw_kernels.h"
#include "hsw_nounwind.hpp"
#include "hsw_structure.hpp"

namespace hsw {

struct RegionTape {
    std::vector<uint32_t> gate_code, lookup_code, chip_dense_code, chip_spread_code;
    std::vector<uint32_t> wit_cell;                       // gate-stream cell of distinct value #w
    // after digests 0 .. h: gate cells, distinct values, lookup entries, limb calls
    std::vector<uint64_t> end_cell, end_wit, end_lookup, end_limb;
    std::vector<int64_t> const_key;                       // k, or -k for p - k
    std::vector<fr::Fe> const_canon, const_mont;
    // device side of the gather: image position of every witness cell, staging of the packed values
    uint32_t *d_wit_pos = nullptr;
    void *d_distinct = nullptr;
    int device = 0;
};

void free_region_tape(RegionTape *t) {
    if (!t) return;
    if (t->d_wit_pos || t->d_distinct) {
        int prev = -1;
        (void)hipGetDevice(&prev);
        if (prev != t->device) (void)hipSetDevice(t->device);
        (void)hipFree(t->d_wit_pos);
        (void)hipFree(t->d_distinct);
        if (prev >= 0 && prev != t->device) (void)hipSetDevice(prev);
    }
    delete t;
}

// The layout moved (column height, origin row) but the stream did not: the codes stay, the image positions of
// the witnesses are worked out again at the next delivery.
void drop_region_tape_positions(RegionTape *t) {
    if (!t || (!t->d_wit_pos && !t->d_distinct)) return;
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (prev != t->device) (void)hipSetDevice(t->device);
    (void)hipFree(t->d_wit_pos);
    (void)hipFree(t->d_distinct);
    t->d_wit_pos = nullptr;
    t->d_distinct = nullptr;
    if (prev >= 0 && prev != t->device) (void)hipSetDevice(prev);
}

namespace {

constexpr uint32_t TAPE_CONST = 0x80000000u;

struct Builder {
    RegionTape &t;
    std::map<int64_t, uint32_t> const_index;
    explicit Builder(RegionTape &tape) : t(tape) {}
    uint32_t constant(int64_t k) {
        auto it = const_index.find(k);
        if (it != const_index.end()) return TAPE_CONST | it->second;
        const uint32_t idx = (uint32_t)t.const_key.size();
        fr::Fe c = {{(uint64_t)(k < 0 ? -k : k), 0, 0, 0}};
        if (k < 0) c = fr::sub_raw(fr::P, c);                       // p - |k|
        t.const_key.push_back(k);
        t.const_canon.push_back(c);
        t.const_mont.push_back(fr::to_mont(c));
        const_index[k] = idx;
        return TAPE_CONST | idx;
    }
    // one section (prologue / block / epilogue) whose cell 0 is stream cell `base`; ext resolves a reference
    template <class ST, class EXT>
    bool section(const ST &st, uint64_t base, EXT ext) {
        for (size_t c = 0; c < st.kind.size(); c++) {
            uint32_t code;
            if (st.kind[c] == BlockStructure::WITNESS) {
                code = (uint32_t)t.wit_cell.size();
                t.wit_cell.push_back((uint32_t)(base + c));
            } else if (st.kind[c] == BlockStructure::CONSTANT) {
                code = constant(st.ref[c]);
            } else {
                const int64_t src = ext(st.ref[c]);
                if (src == INT64_MIN) code = constant(0);           // the Context's zero cell, assigned before the gadget took over
                else if (src < 0 || (uint64_t)src >= base + c) return false;   // every copy points backwards
                else code = t.gate_code[(size_t)src];                // the root of the chain: already resolved
            }
            t.gate_code[(size_t)(base + c)] = code;
        }
        return true;
    }
    uint32_t code_of(int64_t src) { return src == INT64_MIN ? constant(0) : t.gate_code[(size_t)src]; }
};

}  // namespace

// Builds the tape of EVERY digest of the gadget (the layout follows from max_variable_byte_sizes alone).
static int build_region_tape(const hsw_gadget *g, RegionTape **out) {
    const Context &c = *g->ctx;
    if (!c.whole) return HSW_ERR_INVALID_ARG;
    if (c.gate_capacity >= (1ull << 30)) return HSW_ERR_TOO_LARGE;          // codes are 31-bit indices
    RegionTape *t = new (std::nothrow) RegionTape();
    if (!t) return HSW_ERR_NOMEM;
    struct Guard { RegionTape *p; ~Guard() { free_region_tape(p); } } guard{t};
    const uint64_t G = c.shape.gate_cells_per_block, LK = c.shape.lookup_cells_per_block, LC = c.shape.limb_calls_per_block;
    const BlockStructure blk = StructureBuilder((int)c.shape.limbs_per_spread, true).block();
    if (blk.kind.size() != G || blk.lookup_src.size() != LK || blk.chip.size() != 2 * LC) return HSW_ERR_INVALID_ARG;
    t->gate_code.assign((size_t)c.gate_capacity, 0);
    Builder b(*t);
    const bool rc_in = g->cfg.is_input_range_check;
    uint64_t gc = 0;
    int64_t zero_abs = INT64_MIN;                       // INT64_MIN: a zero cell outside the stream (origin_zero_loaded)
    bool zero_seen = c.origin_zero_loaded;
    FrameStructureBuilder fb;
    for (size_t h = 0; h < g->cfg.max_variable_byte_sizes.size(); h++) {
        const uint64_t mx = g->cfg.max_variable_byte_sizes[h], nb = mx / 64;
        const FrameStructure pro = fb.prologue(mx, rc_in), epi = fb.epilogue(nb);
        const uint64_t g0 = gc, P = pro.kind.size();
        auto pro_ext = [&](int64_t r) -> int64_t { return r >= 0 ? (int64_t)(g0 + (uint64_t)r) : -1; };
        if (!b.section(pro, g0, pro_ext)) return HSW_ERR_INVALID_ARG;
        for (int64_t src : pro.lookup_src) t->lookup_code.push_back(b.code_of(pro_ext(src)));
        gc += P;
        if (!zero_seen || c.independent) {                // this digest's Context loads its zero cell (A4-iii)
            zero_abs = (int64_t)gc;
            t->gate_code[(size_t)gc] = b.constant(0);
            gc += 1;
            zero_seen = true;
        }
        const uint64_t B0 = gc;
        auto state_cell = [&](uint64_t n, uint64_t i) -> int64_t {
            return n == 0 ? (int64_t)(g0 + frame::P_STATE + i) : (int64_t)(B0 + (n - 1) * G + (uint64_t)blk.next_state[i]);
        };
        for (uint64_t k = 0; k < nb; k++) {
            const uint64_t Bk = B0 + k * G;
            auto blk_ext = [&](int64_t r) -> int64_t {
                if (r >= 0) return (int64_t)(Bk + (uint64_t)r);
                if (r <= BlockStructure::INPUT_BYTE0 && r > BlockStructure::INPUT_BYTE0 - 64)
                    return (int64_t)(g0 + frame::P_BYTES + 64 * k + (uint64_t)(BlockStructure::INPUT_BYTE0 - r));
                if (r <= BlockStructure::PRE_STATE0 && r > BlockStructure::PRE_STATE0 - 8)
                    return state_cell(k, (uint64_t)(BlockStructure::PRE_STATE0 - r));
                if (r == BlockStructure::ZERO) return zero_abs;
                return -1;                                // HIDDEN: not in an internals-mode stream
            };
            if (!b.section(blk, Bk, blk_ext)) return HSW_ERR_INVALID_ARG;
            for (int64_t src : blk.lookup_src) t->lookup_code.push_back(b.code_of(blk_ext(src)));
            for (size_t n = 0; n < (size_t)LC; n++) {
                t->chip_dense_code.push_back(b.code_of(blk_ext(blk.chip[2 * n])));
                t->chip_spread_code.push_back(b.code_of(blk_ext(blk.chip[2 * n + 1])));
            }
        }
        gc += nb * G;
        const uint64_t E = gc;
        auto epi_ext = [&](int64_t r) -> int64_t {
            if (r >= 0) return (int64_t)(E + (uint64_t)r);
            if (r == FrameStructure::ZERO) return zero_abs;
            if (r == FrameStructure::TARGET) return (int64_t)(g0 + 34);          // assigned_target_round: prologue cell 34
            const uint64_t q = (uint64_t)(FrameStructure::STATE0 - r);
            return state_cell(q / 8, q % 8);
        };
        if (!b.section(epi, E, epi_ext)) return HSW_ERR_INVALID_ARG;
        for (int64_t src : epi.lookup_src) t->lookup_code.push_back(b.code_of(epi_ext(src)));
        gc += epi.kind.size();
        t->end_cell.push_back(gc);
        t->end_wit.push_back(t->wit_cell.size());
        t->end_lookup.push_back(t->lookup_code.size());
        t->end_limb.push_back(t->chip_dense_code.size());
    }
    // (a Context that came with its zero cell leaves the one cell reserved for it unused)
    const uint64_t unused = (c.origin_zero_loaded && !c.independent) ? 1 : 0;
    if (gc + unused != c.gate_capacity || t->lookup_code.size() != c.own_lookup_capacity) return HSW_ERR_INVALID_ARG;
    guard.p = nullptr;
    *out = t;
    return HSW_OK;
}

// stream cell -> image cell (Context::position as one index; the image's column 0 is FlexGate column origin_column)
static inline uint64_t image_cell(const Context &c, uint64_t cell) {
    uint64_t at = cell + (c.max_rows ? c.origin_row : 0);
    for (size_t k = 0; k < c.break_cell.size(); k++)
        if (c.break_cell[k] <= cell) at += c.break_gap[k];
    return at;
}

static int ensure_tape(hsw_gadget *g) {
    if (g->tape) return HSW_OK;
    return build_region_tape(g, &g->tape);
}

// what the digests assigned so far cover
struct Extent { uint64_t cells, wit, lookups, limbs; };
static Extent extent_so_far(const hsw_gadget *g) {
    const size_t h = g->cfg.cur_hash_idx;
    if (h == 0) return Extent{0, 0, 0, 0};
    const RegionTape &t = *g->tape;
    return Extent{t.end_cell[h - 1], t.end_wit[h - 1], t.end_lookup[h - 1], t.end_limb[h - 1]};
}

}  // namespace hsw

using namespace hsw;

extern "C" {

int hsw_gadget_region_tape(hsw_gadget *g, hsw_region_tape *out) try {
    if (!g || !out) return HSW_ERR_INVALID_ARG;
    if (g->ctx->repr_flags & HSW_REPR_COMPACT64) return HSW_ERR_UNSUPPORTED;
    const int rc = ensure_tape(g);
    if (rc != HSW_OK) return rc;
    const RegionTape &t = *g->tape;
    const Extent e = extent_so_far(g);
    out->n_distinct = e.wit;
    out->gate_cells = e.cells;
    out->lookup_cells = e.lookups;
    out->limb_calls = e.limbs;
    out->gate_code = t.gate_code.data();
    out->lookup_code = t.lookup_code.data();
    out->chip_dense_code = t.chip_dense_code.data();
    out->chip_spread_code = t.chip_spread_code.data();
    out->n_consts = t.const_key.size();
    out->consts = (g->ctx->repr_flags & HSW_REPR_MONTGOMERY) ? (const void *)t.const_mont.data() : (const void *)t.const_canon.data();
    out->distinct_capacity = t.wit_cell.size();
    return HSW_OK;
} HSW_NO_UNWIND

int hsw_gadget_download_region_distinct(hsw_gadget *g, void *distinct, size_t cap_cells, size_t *n_cells) try {
    if (!g || (!distinct && cap_cells)) return HSW_ERR_INVALID_ARG;
    Context &c = *g->ctx;
    if (c.repr_flags & HSW_REPR_COMPACT64) return HSW_ERR_UNSUPPORTED;
    int rc = ensure_tape(g);
    if (rc != HSW_OK) return rc;
    RegionTape &t = *g->tape;
    const Extent e = extent_so_far(g);
    if (n_cells) *n_cells = (size_t)e.wit;
    if (e.wit > cap_cells) return HSW_ERR_TOO_LARGE;
    if (e.wit == 0) return HSW_OK;
    hipStream_t stream = nullptr;
    int device = 0;
    hsw_engine_stream(c.engine, reinterpret_cast<void **>(&stream), &device);
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (prev != device && hipSetDevice(device) != hipSuccess) return HSW_ERR_NO_DEVICE;
    hipError_t he = hipSuccess;
    if (!t.d_wit_pos) {                                   // first delivery with this layout: where every witness sits in the image
        std::vector<uint32_t> pos(t.wit_cell.size());
        for (size_t w = 0; w < pos.size(); w++) pos[w] = (uint32_t)image_cell(c, t.wit_cell[w]);
        t.device = device;
        he = hipMalloc((void **)&t.d_wit_pos, pos.size() * sizeof(uint32_t));
        if (he == hipSuccess) he = hipMalloc(&t.d_distinct, pos.size() * (size_t)HSW_CELL_BYTES);
        if (he == hipSuccess) he = hipMemcpy(t.d_wit_pos, pos.data(), pos.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
        if (he != hipSuccess) {
            (void)hipFree(t.d_wit_pos); (void)hipFree(t.d_distinct);
            t.d_wit_pos = nullptr; t.d_distinct = nullptr;
        }
    }
    if (he == hipSuccess) he = launch_gather32(c.d_gate, t.d_wit_pos, t.d_distinct, (size_t)e.wit, stream);
    if (he == hipSuccess) he = hipMemcpyAsync(distinct, t.d_distinct, (size_t)e.wit * HSW_CELL_BYTES, hipMemcpyDeviceToHost, stream);
    if (he == hipSuccess) he = hipStreamSynchronize(stream);
    if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
    return he == hipSuccess ? HSW_OK : (he == hipErrorOutOfMemory ? HSW_ERR_NOMEM : HSW_ERR_HIP);
} HSW_NO_UNWIND

int hsw_gadget_replay_region(hsw_gadget *g, const void *distinct, const hsw_region_host *dst, unsigned threads) try {
    if (!g || !distinct || !dst) return HSW_ERR_INVALID_ARG;
    Context &c = *g->ctx;
    if (c.repr_flags & HSW_REPR_COMPACT64) return HSW_ERR_UNSUPPORTED;
    const int rc = ensure_tape(g);
    if (rc != HSW_OK) return rc;
    const RegionTape &t = *g->tape;
    const Extent e = extent_so_far(g);
    struct Cell { uint64_t l[4]; };
    const Cell *val = static_cast<const Cell *>(distinct);
    const Cell *consts = reinterpret_cast<const Cell *>((c.repr_flags & HSW_REPR_MONTGOMERY) ? t.const_mont.data() : t.const_canon.data());
    auto value = [&](uint32_t code) -> const Cell & { return (code & TAPE_CONST) ? consts[code & ~TAPE_CONST] : val[code]; };
    const uint32_t ncols = c.shape.num_advice_columns;
    if (threads == 0) threads = 1;
    if (threads > 64) threads = 64;
    // gate stream -> image: thread k takes cells [lo, hi); the breaks it passes are walked once
    auto gate_part = [&](uint64_t lo, uint64_t hi) {
        if (!dst->gate) return;
        Cell *img = static_cast<Cell *>(dst->gate);
        size_t nb = 0;
        uint64_t shift = c.max_rows ? c.origin_row : 0;
        while (nb < c.break_cell.size() && c.break_cell[nb] <= lo) shift += c.break_gap[nb++];
        for (uint64_t i = lo; i < hi; i++) {
            while (nb < c.break_cell.size() && c.break_cell[nb] <= i) shift += c.break_gap[nb++];
            img[i + shift] = value(t.gate_code[i]);
        }
    };
    auto lookup_part = [&](uint64_t lo, uint64_t hi) {
        if (!dst->lookup) return;
        Cell *lk = static_cast<Cell *>(dst->lookup) + c.origin_lookups;
        for (uint64_t j = lo; j < hi; j++) lk[j] = value(t.lookup_code[j]);
    };
    auto chip_part = [&](uint64_t lo, uint64_t hi) {              // limb call n: column n % ncols, row n / ncols
        Cell *cd = static_cast<Cell *>(dst->chip_dense), *cs = static_cast<Cell *>(dst->chip_spread);
        for (uint64_t n = lo; n < hi; n++) {
            const size_t at = (size_t)(n % ncols) * c.chip_col_stride + (size_t)(n / ncols);
            if (cd) cd[at] = value(t.chip_dense_code[n]);
            if (cs) cs[at] = value(t.chip_spread_code[n]);
        }
    };
    auto work = [&](unsigned k) {
        gate_part(e.cells * k / threads, e.cells * (k + 1) / threads);
        lookup_part(e.lookups * k / threads, e.lookups * (k + 1) / threads);
        chip_part(e.limbs * k / threads, e.limbs * (k + 1) / threads);
    };
    if (threads == 1) { work(0); return HSW_OK; }
    std::vector<std::thread> pool;
    pool.reserve(threads - 1);
    for (unsigned k = 1; k < threads; k++) pool.emplace_back(work, k);
    work(0);
    for (std::thread &th : pool) th.join();
    return HSW_OK;
} HSW_NO_UNWIND

}  // extern "C"
