// hsw_api.cpp -- the C ABI of include/hsw.h over the gfx950 kernels.
//
// Boundary it replaces (reference has no FFI; see include/hsw.h header):
//   src/lib.rs:180-189 block loop -> src/compression.rs:19-25 sha256_compression.
// There is deliberately NO CPU fallback here: without a HIP device every
// compute entry point fails with HSW_ERR_NO_DEVICE / HSW_ERR_HIP.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/hsw.h"
#include "hsw_fr.hpp"
#include "hsw_frame.hpp"
#include "hsw_nounwind.hpp"
#include "hsw_kernels.h"
#include "hsw_layout.h"
#include "hsw_structure.hpp"
#include "hsw_tape.hpp"
#include "hsw_verify.h"

#include "hsw_engine.hpp"


namespace {

template <int L, bool RC>
void fill_shape(hsw_shape *s) {
    using LY = hsw::Lay<L, RC>;
    s->lookup_cells_per_block = LY::LOOKUP_CELLS;
    s->mode = RC ? HSW_MODE_HALO2_INTERNALS : HSW_MODE_DEFAULT;
    s->gate_calls_per_block = (uint32_t)hsw::TapeBuilder(L, RC).block().size();
    s->limbs_per_spread = LY::LIMBS;
    s->cells_per_spread = LY::S;
    s->cells_per_state_spread = LY::S2S;
    s->cells_per_sigma = LY::SIGMA;
    s->cells_per_ch = LY::CH;
    s->cells_per_maj = LY::MAJ;
    s->cells_per_sched_step = LY::SCHED;
    s->cells_per_round = LY::ROUND;
    s->off_words = LY::OFF_WORDS;
    s->off_msg_spread = LY::OFF_MSG;
    s->off_sched = LY::OFF_SCHED;
    s->off_state_spread = LY::OFF_STATE;
    s->off_rounds = LY::OFF_ROUNDS;
    s->off_feed = LY::OFF_FEED;
    s->gate_cells_per_block = LY::GATE_CELLS;
    s->spread_calls_per_block = LY::SPREAD_CALLS;
    s->limb_calls_per_block = LY::LIMB_CALLS;
    s->chip_cells_per_block = LY::CHIP_CELLS;
    s->algorithmic_bytes_per_block =
        ((uint64_t)LY::GATE_CELLS + (uint64_t)LY::CHIP_CELLS) * HSW_CELL_BYTES + 64 + 32 + 32;
}


// Tile shape and waves per block (tuning only; results never change).
//  * tile: cells per contiguous run of one unit.  Measured on MI355X (tools/ab.py,
//    interleaved in one process, 4,096 blocks): canonical output is ~3 % faster
//    with [32 rows][64 cells] tiles and 4 waves per block than with [64][32] and
//    one wave; Montgomery output ~5 % faster with [16][128] tiles and 4 waves;
//    the 8-byte compact form (instruction-bound) ~8 % faster with [16][64] and 8.
//  * parts: a block is 64 + 48 + ... independent units; one wave can expand all
//    of them (lane = unit) or they can be dealt to 2..16 waves.  Small batches
//    (e.g. the 16-block message of BASELINE configs[1]) need the split to
//    occupy 256 CUs.
// The byte tables of the emit-time Montgomery kernels (hsw_expand.hpp Em::M32): i, spread(i) and i << 8 for
// i < 256 in Montgomery form, 3 x 256 cells.  x * R mod p is additive in x, so any 16-bit value is two entries and
// one field addition.  Built once per engine on the host (hsw_fr.hpp), 24 KiB, read through L1 / L2.
int ensure_mont_tab(hsw_engine *e) {
    if (e->d_mont_tab) return HSW_OK;
    std::vector<uint64_t> h(3 * 256 * 4);
    for (uint64_t i = 0; i < 256; i++) {
        uint64_t sp = 0;
        for (int b = 0; b < 8; b++) sp |= ((i >> b) & 1ull) << (2 * b);
        const uint64_t vals[3] = {i, sp, i << 8};
        for (int t = 0; t < 3; t++) {
            const hsw::fr::Fe m = hsw::fr::to_mont(hsw::fr::Fe{{vals[t], 0, 0, 0}});
            std::memcpy(&h[(size_t)(t * 256 + i) * 4], m.l, 32);
        }
    }
    void *d = nullptr;
    hipError_t he = hipMalloc(&d, h.size() * 8);
    if (he == hipSuccess) he = hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    if (he != hipSuccess) { if (d) (void)hipFree(d); return set_err(e, he == hipErrorOutOfMemory ? HSW_ERR_NOMEM : HSW_ERR_HIP, "Montgomery byte tables", he); }
    e->d_mont_tab = d;
    return HSW_OK;
}

int choose_tile(const hsw_engine *e, uint32_t flags) {
    if (e->limbs != 2) return 32;                 // other tile shapes are built for the 8-bit table only
    if (e->tile > 0) return e->tile;
    if (flags & HSW_REPR_COMPACT64)               // [16 rows][64 cells]: not HBM-bound, wants many small waves
        return e->mode == HSW_MODE_HALO2_INTERNALS ? 32 : 6416;
    return (flags & HSW_REPR_MONTGOMERY) ? 128 : 64;
}
int choose_parts(const hsw_engine *e, size_t n_blocks, int tile, uint32_t flags = 0) {
    // a T-cell tile has 64*32/T rows: [64][32] [32][64] [16][128]; experimental codes TTRR: [32][32] [16][32] [16][64]
    const int min_parts = tile == 6416 ? 4 : tile / 32;
    if (e->parts > 0) return e->parts < min_parts ? min_parts : e->parts;
    int parts = (flags & HSW_REPR_COMPACT64) ? 8 : ((tile >= 64 && tile < 1000) ? 4 : min_parts);
    if (parts < min_parts) parts = min_parts;
    while (parts < 16 && n_blocks * (size_t)parts < 2048) parts *= 2;
    return parts < min_parts ? min_parts : parts;
}


}  // namespace

extern "C" {

uint32_t hsw_abi_version(void) { return HSW_ABI_VERSION; }

const char *hsw_strerror(int status) {
    switch (status) {
        case HSW_OK: return "ok";
        case HSW_ERR_INVALID_ARG: return "invalid argument";
        case HSW_ERR_SHAPE: return "invalid shape (16 % num_bits_lookup != 0, zero columns, or size not a multiple of 64)";
        case HSW_ERR_NO_DEVICE: return "no usable HIP device";
        case HSW_ERR_HIP: return "HIP runtime error";
        case HSW_ERR_UNSUPPORTED: return "not supported by this build";
        case HSW_ERR_TOO_LARGE: return "message does not fit max_variable_byte_size";
        case HSW_ERR_NOMEM: return "out of memory";
        default: return "unknown status";
    }
}

const char *hsw_last_error(const hsw_engine *e) { return e ? e->err.c_str() : ""; }

int hsw_shape_query(uint32_t num_bits_lookup, uint32_t num_advice_columns, hsw_shape *out) try {
    return hsw_shape_query_ex(num_bits_lookup, num_advice_columns, HSW_MODE_DEFAULT, out);
} HSW_NO_UNWIND

int hsw_shape_query_ex(uint32_t num_bits_lookup, uint32_t num_advice_columns, uint32_t mode, hsw_shape *out) try {
    if (!out) return HSW_ERR_INVALID_ARG;
    if (mode != HSW_MODE_DEFAULT && mode != HSW_MODE_HALO2_INTERNALS) return HSW_ERR_INVALID_ARG;
    const bool rc = mode == HSW_MODE_HALO2_INTERNALS;
    // spread.rs:37 debug_assert_eq!(16 % num_bits_lookup, 0)
    if (num_bits_lookup == 0 || num_bits_lookup > 16 || 16 % num_bits_lookup != 0) return HSW_ERR_SHAPE;
    if (num_advice_columns == 0) return HSW_ERR_SHAPE;
    std::memset(out, 0, sizeof *out);
    out->num_bits_lookup = num_bits_lookup;
    out->num_advice_columns = num_advice_columns;
    switch (16 / num_bits_lookup) {
        case 1: rc ? fill_shape<1, true>(out) : fill_shape<1, false>(out); break;
        case 2: rc ? fill_shape<2, true>(out) : fill_shape<2, false>(out); break;
        case 4: rc ? fill_shape<4, true>(out) : fill_shape<4, false>(out); break;
        case 8: rc ? fill_shape<8, true>(out) : fill_shape<8, false>(out); break;
        case 16: rc ? fill_shape<16, true>(out) : fill_shape<16, false>(out); break;
        default: return HSW_ERR_SHAPE;
    }
    return HSW_OK;
} HSW_NO_UNWIND

int hsw_spread_table(uint32_t num_bits_lookup, uint64_t *dense_out, uint64_t *spread_out) try {
    if (num_bits_lookup == 0 || num_bits_lookup > 16 || 16 % num_bits_lookup != 0) return HSW_ERR_SHAPE;
    for (uint64_t idx = 0; idx < (1ull << num_bits_lookup); idx++) {              // spread.rs:169-189
        uint64_t sp = 0;
        for (int b = 0; b < 16; b++) sp |= ((idx >> b) & 1ull) << (2 * b);        // bit i -> bit 2i (:172-175)
        if (dense_out) dense_out[idx] = idx;
        if (spread_out) spread_out[idx] = sp;
    }
    return HSW_OK;
} HSW_NO_UNWIND

uint32_t hsw_cell_bytes(uint32_t flags) { return (flags & HSW_REPR_COMPACT64) ? 8u : HSW_CELL_BYTES; }

int hsw_neg_cells(const hsw_shape *s, uint32_t *out, size_t cap, size_t *n) try {
    if (!s || s->cells_per_round == 0) return HSW_ERR_INVALID_ARG;
    if (n) *n = 256;
    if (out) {
        if (cap < 256) return HSW_ERR_INVALID_ARG;
        // inside ch: 2 add rows (8 cells), neg rows [a,-a,1,0] x2, then add(M,-x) / add(.,z) twice
        static const uint32_t in_ch[4] = {9, 13, 17, 25};
        for (uint32_t r = 0; r < 64; r++)
            for (int k = 0; k < 4; k++)
                out[4 * r + k] = s->off_rounds + r * s->cells_per_round + s->cells_per_sigma + in_ch[k];
    }
    return HSW_OK;
} HSW_NO_UNWIND

uint64_t hsw_chip_rows(const hsw_shape *s, uint64_t cursor0, uint64_t n_blocks) {
    if (!s || s->num_advice_columns == 0 || s->limb_calls_per_block == 0) return 0;
    const uint64_t nc = s->num_advice_columns;
    return (cursor0 % nc + (uint64_t)s->limb_calls_per_block * n_blocks + nc - 1) / nc;
}

int hsw_engine_create(int device, void *hip_stream, uint32_t num_bits_lookup,
                      uint32_t num_advice_columns, hsw_engine **out) try {
    return hsw_engine_create_ex(device, hip_stream, num_bits_lookup, num_advice_columns, HSW_MODE_DEFAULT, out);
} HSW_NO_UNWIND

int hsw_engine_create_ex(int device, void *hip_stream, uint32_t num_bits_lookup,
                         uint32_t num_advice_columns, uint32_t mode, hsw_engine **out) try {
    if (!out) return HSW_ERR_INVALID_ARG;
    *out = nullptr;
    hsw_shape shape;
    int rc = hsw_shape_query_ex(num_bits_lookup, num_advice_columns, mode, &shape);
    if (rc != HSW_OK) return rc;

    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return HSW_ERR_NO_DEVICE;
    if (device < 0 || device >= count) return HSW_ERR_NO_DEVICE;
    hsw_engine *e = new (std::nothrow) hsw_engine();
    if (!e) return HSW_ERR_NOMEM;
    e->device = device;
    e->stream = static_cast<hipStream_t>(hip_stream);
    e->shape = shape;
    e->mode = mode;
    e->limbs = (int)shape.limbs_per_spread;
    DeviceScope ds(device);
    if (!ds.ok) { delete e; return HSW_ERR_NO_DEVICE; }
    if (hipEventCreate(&e->ev0) != hipSuccess || hipEventCreate(&e->ev1) != hipSuccess) {
        if (e->ev0) (void)hipEventDestroy(e->ev0);
        delete e;
        return HSW_ERR_HIP;
    }
    *out = e;
    return HSW_OK;
} HSW_NO_UNWIND

static void free_pipeline(hsw_engine *e) {
    for (auto &s : e->slot) {
        (void)hipFree(s.gate); (void)hipFree(s.cd); (void)hipFree(s.cs);
        if (s.kernel_done) (void)hipEventDestroy(s.kernel_done);
        if (s.copy_done) (void)hipEventDestroy(s.copy_done);
        s = hsw_engine::Slot();
    }
    if (e->copy_stream) (void)hipStreamDestroy(e->copy_stream);
    e->copy_stream = nullptr;
    e->slot_blocks = e->slot_rows = 0;
}

void hsw_engine_destroy(hsw_engine *e) {
    if (!e) return;
    {
        DeviceScope ds(e->device);
        (void)hipStreamSynchronize(e->stream);      // launches of this engine may still read the buffers freed below
        if (e->copy_stream) (void)hipStreamSynchronize(e->copy_stream);
        free_pipeline(e);
        for (auto &fs : e->frame_slot) {
            if (fs.h) (void)hipHostFree(fs.h);
            if (fs.done) (void)hipEventDestroy(fs.done);
        }
        if (e->d_structure) (void)hipFree(e->d_structure);
        if (e->d_mont_tab) (void)hipFree(e->d_mont_tab);
        if (e->d_report) (void)hipFree(e->d_report);
        if (e->d_inv_tbl[0]) (void)hipFree(e->d_inv_tbl[0]);
        if (e->d_inv_tbl[1]) (void)hipFree(e->d_inv_tbl[1]);
        if (e->ev0) (void)hipEventDestroy(e->ev0);
        if (e->ev1) (void)hipEventDestroy(e->ev1);
    }
    delete e;
}

int hsw_engine_shape(const hsw_engine *e, hsw_shape *out) try {
    if (!e || !out) return HSW_ERR_INVALID_ARG;
    *out = e->shape;
    return HSW_OK;
} HSW_NO_UNWIND

int hsw_engine_stream(const hsw_engine *e, void **hip_stream, int *device) try {
    if (!e) return HSW_ERR_INVALID_ARG;
    if (hip_stream) *hip_stream = e->stream;
    if (device) *device = e->device;
    return HSW_OK;
} HSW_NO_UNWIND

int hsw_engine_synchronize(hsw_engine *e) try {
    if (!e) return HSW_ERR_INVALID_ARG;
    DeviceScope ds(e->device);
    hipError_t he = hipStreamSynchronize(e->stream);
    if (he != hipSuccess) return set_err(e, HSW_ERR_HIP, "hipStreamSynchronize", he);
    return HSW_OK;
} HSW_NO_UNWIND

int hsw_engine_set_option(hsw_engine *e, const char *name, int64_t value) try {
    if (!e || !name) return HSW_ERR_INVALID_ARG;
    if (std::strcmp(name, "parts") == 0) {
        if (value != 0 && value != 1 && value != 2 && value != 4 && value != 8 && value != 16 && value != 32)
            return set_err(e, HSW_ERR_INVALID_ARG, "parts must be 0 (auto), 1, 2, 4, 8, 16 or 32");
        e->parts = (int)value;
        return HSW_OK;
    }
    if (std::strcmp(name, "verify_slices") == 0) {
        if (value < 0 || value > 256) return set_err(e, HSW_ERR_INVALID_ARG, "verify_slices must be 0 (default) .. 256");
        e->verify_slices = (int)value;
        return HSW_OK;
    }
    if (std::strcmp(name, "chunk_blocks") == 0) {      // blocks per launch of a long batch; tests shrink it to reach that loop
        if (value < 1 || value > (1 << 20)) return set_err(e, HSW_ERR_INVALID_ARG, "chunk_blocks must be 1 .. 2^20");
        e->chunk_blocks = (size_t)value;
        return HSW_OK;
    }
    if (std::strcmp(name, "split") == 0) {
        if (value < -1 || value > 2) return set_err(e, HSW_ERR_INVALID_ARG, "split must be -1 (auto), 0, 1 or 2");
        e->split = (int)value;
        return HSW_OK;
    }
    if (std::strcmp(name, "helpers") == 0) {
        if (value < 0 || value > HSW_SMALL_MAX_HELPERS) return set_err(e, HSW_ERR_INVALID_ARG, "helpers must be 0 (auto) .. 4");
        e->helpers = (int)value;
        return HSW_OK;
    }
    if (std::strcmp(name, "mont_emit") == 0) {
        if (value < 0 || value > 2) return set_err(e, HSW_ERR_INVALID_ARG, "mont_emit must be 0, 1 or 2");
        e->mont_emit = (int)value;
        return HSW_OK;
    }
    if (std::strcmp(name, "tile") == 0) {
        if (value != 0 && value != 32 && value != 64 && value != 128 && value != 6416)
            return set_err(e, HSW_ERR_INVALID_ARG, "tile must be 0 (auto), 32, 64 or 128");
        e->tile = (int)value;
        return HSW_OK;
    }
    return set_err(e, HSW_ERR_INVALID_ARG, "unknown option");
} HSW_NO_UNWIND

int hsw_last_launch(const hsw_engine *e, hsw_launch_info *out) try {
    if (!e || !out) return HSW_ERR_INVALID_ARG;
    if (e->last_launch.grid == 0) return HSW_ERR_INVALID_ARG;      // nothing launched yet
    *out = e->last_launch;
    return HSW_OK;
} HSW_NO_UNWIND

int hsw_set_timing(hsw_engine *e, int enabled) try {
    if (!e) return HSW_ERR_INVALID_ARG;
    e->timing = enabled != 0;
    e->timed = false;
    return HSW_OK;
} HSW_NO_UNWIND

int hsw_last_kernel_ms(hsw_engine *e, float *ms) try {
    if (!e || !ms) return HSW_ERR_INVALID_ARG;
    if (!e->timed) return set_err(e, HSW_ERR_INVALID_ARG, "no timed launch (call hsw_set_timing(e, 1) first)");
    DeviceScope ds(e->device);
    hipError_t he = hipEventSynchronize(e->ev1);
    if (he != hipSuccess) return set_err(e, HSW_ERR_HIP, "hipEventSynchronize", he);
    he = hipEventElapsedTime(ms, e->ev0, e->ev1);
    if (he != hipSuccess) return set_err(e, HSW_ERR_HIP, "hipEventElapsedTime", he);
    return HSW_OK;
} HSW_NO_UNWIND

int hsw_witness_blocks(hsw_engine *e, const uint8_t *d_blocks, const uint32_t *d_pre_states,
                       size_t n_blocks, uint64_t spread_cursor0, void *d_gate, void *d_chip_dense,
                       void *d_chip_spread, size_t chip_col_stride, uint32_t *d_next_states,
                       uint32_t flags) try {
    hsw_witness_args a{};
    a.d_blocks = d_blocks; a.d_pre_states = d_pre_states; a.n_blocks = n_blocks;
    a.spread_cursor0 = spread_cursor0; a.d_gate = d_gate; a.d_chip_dense = d_chip_dense;
    a.d_chip_spread = d_chip_spread; a.chip_col_stride = chip_col_stride;
    a.d_next_states = d_next_states; a.flags = flags;
    return hsw_witness_blocks_ex(e, &a);
} HSW_NO_UNWIND

int hsw_witness_blocks_ex(hsw_engine *e, const hsw_witness_args *args) try {
    return hsw_witness_blocks_impl(e, args, nullptr, nullptr);
} HSW_NO_UNWIND

}  // extern "C"

// The small-batch kernel (hsw_small.hpp) takes a launch when the table is the reference's 8-bit one and the
// batch is tiny (or the "split" option asks for it); everything else goes to hsw_expand_kernel.
bool hsw_small_eligible(const hsw_engine *e, size_t n_blocks) {
    if (e->limbs != 2) return false;
    if (e->split == 2) return true;
    return e->split < 0 && e->parts == 0 && e->tile == 0 && n_blocks <= (size_t)hsw::HSW_SMALL_AUTO_BLOCKS;
}

// hsw_witness_blocks_ex, plus (small-batch launches only) the digest frames written by waves of the same
// launch and a second copy of the next states in pinned host memory.
int hsw_witness_blocks_impl(hsw_engine *e, const hsw_witness_args *args, const hsw::SmallFrames *frames,
                            uint32_t *host_next_states) {
    if (!e || !args) return HSW_ERR_INVALID_ARG;
    const uint8_t *d_blocks = args->d_blocks;
    const uint32_t *d_pre_states = args->d_pre_states;
    const size_t n_blocks = args->n_blocks, chip_col_stride = args->chip_col_stride;
    const uint64_t spread_cursor0 = args->spread_cursor0;
    void *d_gate = args->d_gate, *d_chip_dense = args->d_chip_dense, *d_chip_spread = args->d_chip_spread;
    uint32_t *d_next_states = args->d_next_states;
    const uint32_t flags = args->flags;
    if (n_blocks == 0) return HSW_OK;
    if (flags & ~(HSW_REPR_MASK | HSW_SKIP_GATE | HSW_SKIP_CHIP | HSW_CHAINED))
        return set_err(e, HSW_ERR_INVALID_ARG, "unknown flag bits");
    if ((flags & HSW_CHAINED) && (!hsw_small_eligible(e, n_blocks) || n_blocks > 32 || n_blocks > e->chunk_blocks || frames))
        return set_err(e, HSW_ERR_UNSUPPORTED, "HSW_CHAINED: small-batch launches only (<= 32 blocks, 8-bit table); run hsw_sha256_chain first");
    if ((flags & HSW_REPR_MASK) == HSW_REPR_MASK)
        return set_err(e, HSW_ERR_INVALID_ARG, "HSW_REPR_MONTGOMERY and HSW_REPR_COMPACT64 are exclusive");
    const size_t cb = hsw_cell_bytes(flags);
    if (!d_blocks || !d_pre_states) return set_err(e, HSW_ERR_INVALID_ARG, "null input pointer");
    if (((uintptr_t)d_blocks & 3u) || ((uintptr_t)d_pre_states & 3u))
        return set_err(e, HSW_ERR_INVALID_ARG, "inputs must be 4-byte aligned");
    const bool want_gate = !(flags & HSW_SKIP_GATE), want_chip = !(flags & HSW_SKIP_CHIP);
    if (want_gate && (!d_gate || ((uintptr_t)d_gate & 15u)))
        return set_err(e, HSW_ERR_INVALID_ARG, "gate buffer null or not 16-byte aligned");
    if (want_chip) {
        if (!d_chip_dense || !d_chip_spread || ((uintptr_t)d_chip_dense & 15u) ||
            ((uintptr_t)d_chip_spread & 15u))
            return set_err(e, HSW_ERR_INVALID_ARG, "chip buffers null or not 16-byte aligned");
        if (chip_col_stride < hsw_chip_rows(&e->shape, spread_cursor0, n_blocks))
            return set_err(e, HSW_ERR_INVALID_ARG, "chip_col_stride smaller than hsw_chip_rows()");
    }
    if (args->d_lookup && e->mode != HSW_MODE_HALO2_INTERNALS)
        return set_err(e, HSW_ERR_INVALID_ARG, "d_lookup needs an engine created with HSW_MODE_HALO2_INTERNALS");
    if (e->mode == HSW_MODE_HALO2_INTERNALS && e->limbs > 4 && (flags & HSW_REPR_COMPACT64))
        return set_err(e, HSW_ERR_UNSUPPORTED, "internals mode with a 2- or 1-bit spread table: 32-byte cells only (no HSW_REPR_COMPACT64)");
    if (args->d_lookup && ((uintptr_t)args->d_lookup & 15u))
        return set_err(e, HSW_ERR_INVALID_ARG, "lookup buffer not 16-byte aligned");
    if (args->pack && args->pack->n_breaks > HSW_MAX_BREAKS)
        return set_err(e, HSW_ERR_INVALID_ARG, "too many column breaks");
    if (args->pack)      // the kernel applies at most two breaks inside one block
        for (uint32_t k = 0; k + 2 < args->pack->n_breaks; k++)
            if (args->pack->break_cell[k] != 0 &&       // breaks at cell 0 are plain offsets of the whole call
                args->pack->break_cell[k + 2] - args->pack->break_cell[k] < e->shape.gate_cells_per_block)
                return set_err(e, HSW_ERR_UNSUPPORTED, "more than two column breaks inside one block (max_rows too small)");
    if (args->frame_every) {
        if (e->mode != HSW_MODE_HALO2_INTERNALS)
            return set_err(e, HSW_ERR_INVALID_ARG, "digest frames need an engine created with HSW_MODE_HALO2_INTERNALS");
        if (flags & HSW_REPR_COMPACT64)
            return set_err(e, HSW_ERR_UNSUPPORTED, "digest frames hold full-width cells: no HSW_REPR_COMPACT64");
        if (n_blocks > ((size_t)1 << 20))
            return set_err(e, HSW_ERR_UNSUPPORTED, "more than 2^20 blocks in one framed call");
    }
    if ((frames || host_next_states) && !hsw_small_eligible(e, n_blocks))
        return set_err(e, HSW_ERR_INVALID_ARG, "frames / host next states ride on small-batch launches only");
    DeviceScope ds(e->device);
    if (!ds.ok) return set_err(e, HSW_ERR_NO_DEVICE, "hipSetDevice failed");

    // One launch covers up to 2^20 blocks (2.5 TB of cells would be far past
    // HBM anyway); longer batches are issued as consecutive launches.
    // (a framed call is always one launch: n_blocks <= 2^20 was checked above; the option is a test knob)
    const size_t CHUNK = args->frame_every ? ((size_t)1 << 20) : e->chunk_blocks;
    const size_t G = e->shape.gate_cells_per_block;
    hipError_t he;
    if (e->timing) {
        he = hipEventRecord(e->ev0, e->stream);
        if (he != hipSuccess) return set_err(e, HSW_ERR_HIP, "hipEventRecord", he);
    }
    for (size_t done = 0; done < n_blocks; done += CHUNK) {
        const size_t n = n_blocks - done < CHUNK ? n_blocks - done : CHUNK;
        hsw::ExpandParams p{};
        p.blocks = d_blocks + 64 * done;
        p.pre_states = d_pre_states + 8 * done;
        p.gate = want_gate ? static_cast<uint8_t *>(d_gate) + cb * G * done : nullptr;
        p.chip_dense = d_chip_dense;
        p.chip_spread = d_chip_spread;
        p.next_states = d_next_states ? d_next_states + 8 * done : nullptr;
        p.lookup = args->d_lookup ? static_cast<uint8_t *>(args->d_lookup) +
                                        cb * e->shape.lookup_cells_per_block * done
                                  : nullptr;
        p.n_blocks = n;
        p.chip_col_stride = chip_col_stride;
        p.cursor0 = spread_cursor0;
        p.ncols = e->shape.num_advice_columns;
        p.flags = (want_gate ? 0u : hsw::HSW_K_SKIP_GATE) | (want_chip ? 0u : hsw::HSW_K_SKIP_CHIP) |
                  ((flags & HSW_REPR_MONTGOMERY) ? hsw::HSW_K_MONTGOMERY : 0u) |
                  ((flags & HSW_REPR_COMPACT64) ? hsw::HSW_K_COMPACT : 0u) |
                  (e->mode == HSW_MODE_HALO2_INTERNALS ? hsw::HSW_K_INTERNALS : 0u) |
                  ((flags & HSW_CHAINED) ? hsw::HSW_K_CHAINED : 0u);
        p.frame_every = args->frame_every;
        p.frame_cells = args->frame_cells;
        p.frame_lookups = args->frame_lookups;
        const int tile = choose_tile(e, flags);
        p.parts = (uint32_t)choose_parts(e, n_blocks, tile, flags);
        // tiny batches are latency-bound: the small-batch kernel (37 waves per block, one sub-unit program
        // each; hsw_small.hpp).  "split" = 1 keeps the older one-phase-per-wave mode of hsw_expand_kernel.
        const bool small = hsw_small_eligible(e, n_blocks);
        if (!small && e->limbs == 2 && e->split == 1) {
            p.parts = 32;
            p.flags |= hsw::HSW_K_SPLIT;
        }
        // Montgomery cells, streaming kernel, the reference's 8-bit table: converted at emit time, [64][16] tiles of
        // finished 32-byte cells, one wave per block (hsw_expand.hpp Em::M32)
        // (default mode only unless forced: with the internals-mode tile -- realignment columns, lookup staging -- it
        //  drops to 4 waves per CU and loses to the write-out conversion, 2.35 vs 1.96 ms)
        // (and only for launches that fill the chip with one wave per block: below ~1,500 blocks the write-out
        //  conversion, which spreads a block over up to 16 waves, is faster -- 160 blocks 0.15 against 0.30 ms)
        const bool m32 = !small && e->limbs == 2 && (flags & HSW_REPR_MONTGOMERY) && !(p.flags & hsw::HSW_K_SPLIT) &&
                         (e->mont_emit == 2 || (e->mont_emit == 1 && e->mode != HSW_MODE_HALO2_INTERNALS && n >= 1536));
        if (m32) {
            const int rc = ensure_mont_tab(e);
            if (rc != HSW_OK) return rc;
            p.flags |= hsw::HSW_K_M32;
            p.parts = 1;
            p.mont_tab = e->d_mont_tab;
        }
        if (args->pack) {
            // breaks are given in call-relative stream indices; this launch starts at cell done*G
            for (uint32_t k = 0; k < args->pack->n_breaks; k++) {
                const uint64_t bc = args->pack->break_cell[k], first = (uint64_t)done * G;
                p.break_cell[p.n_breaks] = bc > first ? bc - first : 0;
                p.break_gap[p.n_breaks] = args->pack->break_gap[k];
                p.n_breaks++;
            }
        }
        if (done != 0) {
            // later chunks: keep buffer row 0 fixed by pre-offsetting the column
            // base instead of the cursor origin
            const uint64_t nc = p.ncols;
            const uint64_t c1 = spread_cursor0 + (uint64_t)done * e->shape.limb_calls_per_block;
            const uint64_t row_shift = c1 / nc - spread_cursor0 / nc;
            p.cursor0 = c1;
            if (want_chip) {
                p.chip_dense = static_cast<uint8_t *>(d_chip_dense) + (size_t)row_shift * cb;
                p.chip_spread = static_cast<uint8_t *>(d_chip_spread) + (size_t)row_shift * cb;
            }
        }
        if (small) {
            // waves per workgroup: they share every flush (hsw_expand.hpp Em::HELPERS)
            // (measured, tools/region_latency.c sweep and tools/small_n.py: the Montgomery conversion always wants
            //  4; the plain write-out 4 up to 16 blocks, 2 up to 64 and 1 beyond -- enough waves there already)
            const uint32_t helpers = e->helpers ? (uint32_t)e->helpers
                                     : ((flags & HSW_REPR_MONTGOMERY) || n <= 16) ? 4u : n <= 64 ? 2u : 1u;
            p.parts = helpers;
            if (n <= 16) p.flags |= hsw::HSW_K_ROLE_MAJOR;
            else p.flags &= ~(uint32_t)hsw::HSW_K_ROLE_MAJOR;
            p.next_states_host = host_next_states ? host_next_states + 8 * done : nullptr;
            he = hsw::launch_small(p, done == 0 ? frames : nullptr, e->limbs, e->stream);
            if (he != hipSuccess) return set_err(e, HSW_ERR_HIP, "launch hsw_small_kernel", he);
            hsw_launch_info &li = e->last_launch;
            li.limbs = 2; li.tile_cells = 128; li.tile_rows = 16;
            li.repr = (flags & HSW_REPR_MONTGOMERY) ? 1u : (flags & HSW_REPR_COMPACT64) ? 2u : 0u;
            li.internals = e->mode == HSW_MODE_HALO2_INTERNALS ? 1u : 0u;
            li.parts = hsw::HSW_SMALL_WAVES_PER_BLOCK * helpers; li.split = 2; li.n_blocks = n;
            li.grid = (uint64_t)n * hsw::HSW_SMALL_WAVES_PER_BLOCK + ((done == 0 && frames) ? (uint64_t)frames->n_frames * (frames->state_waves + frames->byte_waves) : 0);
            continue;
        }
        he = hsw::launch_expand(p, e->limbs, tile, e->stream);
        if (he != hipSuccess) return set_err(e, HSW_ERR_HIP, "launch hsw_expand_kernel", he);
        {
            hsw_launch_info &li = e->last_launch;
            const bool wide_internals = e->mode == HSW_MODE_HALO2_INTERNALS && e->limbs > 2;   // [64][32] tiles only
            li.limbs = (uint32_t)e->limbs;
            li.tile_cells = m32 ? (uint32_t)hsw::HSW_M32_TILE : wide_internals ? 32u : tile == 6416 ? 64u : (uint32_t)tile;
            li.tile_rows = m32 ? 64u : wide_internals ? 64u : tile == 6416 ? 16u : 2048u / (uint32_t)tile;
            li.repr = m32 ? 3u : (flags & HSW_REPR_MONTGOMERY) ? 1u : (flags & HSW_REPR_COMPACT64) ? 2u : 0u;
            li.internals = e->mode == HSW_MODE_HALO2_INTERNALS ? 1u : 0u;
            li.parts = p.parts;
            li.split = (p.flags & hsw::HSW_K_SPLIT) ? 1u : 0u;
            li.n_blocks = n;
            li.grid = (uint64_t)n * p.parts;
        }
    }
    if (e->timing) {
        he = hipEventRecord(e->ev1, e->stream);
        if (he != hipSuccess) return set_err(e, HSW_ERR_HIP, "hipEventRecord", he);
        e->timed = true;
    }
    return HSW_OK;
}

extern "C" {

int hsw_gate_tape(const hsw_shape *shape, uint8_t *lens_out, size_t cap, size_t *n_calls) try {
    if (!shape || shape->limbs_per_spread == 0) return HSW_ERR_INVALID_ARG;
    const std::vector<uint8_t> lens =
        hsw::TapeBuilder((int)shape->limbs_per_spread, shape->mode == HSW_MODE_HALO2_INTERNALS).block();
    if (n_calls) *n_calls = lens.size();
    if (lens_out) {
        if (cap < lens.size()) return HSW_ERR_INVALID_ARG;
        std::memcpy(lens_out, lens.data(), lens.size());
    }
    return HSW_OK;
} HSW_NO_UNWIND

int hsw_pack_plan_query(const hsw_shape *shape, size_t n_blocks, uint64_t start_row, uint64_t max_rows,
                        hsw_pack_plan *out) try {
    if (!shape || !out || shape->limbs_per_spread == 0) return HSW_ERR_INVALID_ARG;
    if (max_rows < 8 || start_row >= max_rows) return HSW_ERR_INVALID_ARG;
    std::memset(out, 0, sizeof *out);
    const std::vector<uint8_t> lens =
        hsw::TapeBuilder((int)shape->limbs_per_spread, shape->mode == HSW_MODE_HALO2_INTERNALS).block();
    const uint64_t G = shape->gate_cells_per_block;
    uint64_t row = start_row, cell = 0, gaps = 0;
    for (size_t b = 0; b < n_blocks; b++) {
        if (row + G + 4 < max_rows) {          // no call of this block can reach the end of the column
            row += G;
            cell += G;
            continue;
        }
        for (uint8_t len : lens) {
            if (row + len >= max_rows) {       // halo2-lib v0.2.x assign_region: move to the next column (A3)
                if (out->n_breaks == HSW_MAX_BREAKS) return HSW_ERR_TOO_LARGE;
                out->break_cell[out->n_breaks] = cell;
                out->break_gap[out->n_breaks] = max_rows - row;
                out->n_breaks++;
                gaps += max_rows - row;
                row = 0;
            }
            row += len;
            cell += len;
        }
    }
    out->columns_touched = out->n_breaks + 1;
    out->span_cells = cell + gaps;
    out->end_row = row;
    return HSW_OK;
} HSW_NO_UNWIND

int hsw_download(hsw_engine *e, void *host_dst, const void *d_src, size_t bytes) try {
    if (!e || (bytes && (!host_dst || !d_src))) return HSW_ERR_INVALID_ARG;
    if (bytes == 0) return HSW_OK;
    DeviceScope ds(e->device);
    if (!ds.ok) return set_err(e, HSW_ERR_NO_DEVICE, "hipSetDevice failed");
    hipError_t he = hipMemcpyAsync(host_dst, d_src, bytes, hipMemcpyDeviceToHost, e->stream);
    if (he == hipSuccess) he = hipStreamSynchronize(e->stream);
    if (he != hipSuccess) return set_err(e, HSW_ERR_HIP, "hsw_download", he);
    return HSW_OK;
} HSW_NO_UNWIND

int hsw_fill_calibrate(hsw_engine *e, void *d_buf, size_t bytes, float *ms) try {
    if (!e || !d_buf || !ms || ((uintptr_t)d_buf & 15u)) return HSW_ERR_INVALID_ARG;
    DeviceScope ds(e->device);
    if (!ds.ok) return set_err(e, HSW_ERR_NO_DEVICE, "hipSetDevice failed");
    hipError_t he = hipEventRecord(e->ev0, e->stream);
    if (he == hipSuccess) he = hsw::launch_fill(d_buf, bytes, e->stream);
    if (he == hipSuccess) he = hipEventRecord(e->ev1, e->stream);
    if (he == hipSuccess) he = hipEventSynchronize(e->ev1);
    if (he == hipSuccess) he = hipEventElapsedTime(ms, e->ev0, e->ev1);
    e->timed = false;
    if (he != hipSuccess) return set_err(e, HSW_ERR_HIP, "hsw_fill_calibrate", he);
    return HSW_OK;
} HSW_NO_UNWIND

int hsw_sha256_chain(hsw_engine *e, const uint8_t *d_blocks, size_t n_messages,
                     size_t blocks_per_message, const uint32_t *d_init_states,
                     uint32_t *d_pre_states) try {
    if (!e) return HSW_ERR_INVALID_ARG;
    if (n_messages == 0 || blocks_per_message == 0) return HSW_OK;
    if (!d_blocks || !d_pre_states) return set_err(e, HSW_ERR_INVALID_ARG, "null pointer");
    if (((uintptr_t)d_blocks & 3u) || ((uintptr_t)d_pre_states & 3u) || ((uintptr_t)d_init_states & 3u))
        return set_err(e, HSW_ERR_INVALID_ARG, "pointers must be 4-byte aligned");
    DeviceScope ds(e->device);
    if (!ds.ok) return set_err(e, HSW_ERR_NO_DEVICE, "hipSetDevice failed");
    hipError_t he = hsw::launch_chain(d_blocks, n_messages, blocks_per_message, d_init_states,
                                      d_pre_states, e->stream);
    if (he != hipSuccess) return set_err(e, HSW_ERR_HIP, "launch hsw_chain_kernel", he);
    return HSW_OK;
} HSW_NO_UNWIND

// Host delivery, pipelined: chunks of blocks are expanded on the engine's stream
// into one of two device staging slots while the previous slot drains to host
// memory on a second stream (kernel || D2H overlap; the path is PCIe-bound:
// 2.39 MB per block against ~60 GB/s).
static int pipelined_to_host(hsw_engine *e, const uint8_t *blocks, const uint32_t *pre_states,
                             size_t n_blocks, uint64_t cursor0, void *gate, void *chip_dense,
                             void *chip_spread, size_t chip_col_stride, uint32_t *next_states,
                             uint32_t flags, bool pin) {
    const size_t G = e->shape.gate_cells_per_block, LC = e->shape.limb_calls_per_block;
    const size_t ncols = e->shape.num_advice_columns;
    const bool want_gate = !(flags & HSW_SKIP_GATE), want_chip = !(flags & HSW_SKIP_CHIP);
    const size_t cb = hsw_cell_bytes(flags);     // staging slots are sized for 32-byte cells either way
    size_t CH = 128;
    CH = ncols <= 64 ? CH - CH % ncols : ncols;            // chunk * LC must be a multiple of ncols
    if (CH > n_blocks) CH = ((n_blocks + ncols - 1) / ncols) * ncols;
    const size_t ch_rows = CH * LC / ncols;
    hipError_t he = hipSuccess;
    int rc = HSW_OK;
    auto fail = [&](const char *what) {
        rc = set_err(e, he == hipErrorOutOfMemory ? HSW_ERR_NOMEM : HSW_ERR_HIP, what, he);
    };
    // (re)build the staging slots
    if (e->slot_blocks < CH || e->slot_rows < ch_rows) {
        free_pipeline(e);
        if ((he = hipStreamCreateWithFlags(&e->copy_stream, hipStreamNonBlocking)) != hipSuccess) { fail("hipStreamCreate"); return rc; }
        for (auto &s : e->slot) {
            if ((he = hipMalloc(&s.gate, CH * G * HSW_CELL_BYTES)) != hipSuccess ||
                (he = hipMalloc(&s.cd, ncols * ch_rows * HSW_CELL_BYTES)) != hipSuccess ||
                (he = hipMalloc(&s.cs, ncols * ch_rows * HSW_CELL_BYTES)) != hipSuccess ||
                (he = hipEventCreateWithFlags(&s.kernel_done, hipEventDisableTiming)) != hipSuccess ||
                (he = hipEventCreateWithFlags(&s.copy_done, hipEventDisableTiming)) != hipSuccess) {
                fail("pipeline staging allocation");
                free_pipeline(e);
                return rc;
            }
        }
        e->slot_blocks = CH;
        e->slot_rows = ch_rows;
    }
    const size_t rows_total = (size_t)hsw_chip_rows(&e->shape, cursor0, n_blocks);
    // HSW_HOST_REGISTER is accepted and ignored (`pin`): the library does not pin memory it does not own.
    // Rounds 1 and 2 both tried hipHostRegister on the caller's buffers -- whole pages first, then only the page
    // interior with every copy cut at the registration boundary -- and both ended in "Memory access fault by
    // GPU ... on address <host heap>" under the randomised differential run (profiles/r01_fuzz_parity.json,
    // profiles/r02_fuzz_parity.json): a user-pointer registration inside an allocator's heap does not survive
    // the allocator trimming and re-growing that heap between calls.  Pageable destinations take the runtime's
    // own staged copies; callers that want the PCIe rate allocate with hsw_host_alloc.
    (void)pin;
    (void)rows_total;
    auto d2h = [&](void *dst, const void *src, size_t bytes, hipStream_t st) -> hipError_t {
        return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st);
    };
    uint8_t *d_blocks = nullptr; uint32_t *d_pre = nullptr, *d_next = nullptr;
    do {
        if ((he = hipMalloc((void **)&d_blocks, n_blocks * 64)) != hipSuccess) { fail("hipMalloc blocks"); break; }
        if ((he = hipMalloc((void **)&d_pre, n_blocks * 32)) != hipSuccess) { fail("hipMalloc pre_states"); break; }
        if ((he = hipMalloc((void **)&d_next, n_blocks * 32)) != hipSuccess) { fail("hipMalloc next_states"); break; }
        if ((he = hipMemcpyAsync(d_blocks, blocks, n_blocks * 64, hipMemcpyHostToDevice, e->stream)) != hipSuccess) { fail("H2D blocks"); break; }
        if ((he = hipMemcpyAsync(d_pre, pre_states, n_blocks * 32, hipMemcpyHostToDevice, e->stream)) != hipSuccess) { fail("H2D pre_states"); break; }
        size_t chunk_idx = 0;
        for (size_t done = 0; done < n_blocks && rc == HSW_OK; done += CH, chunk_idx++) {
            const size_t nb = n_blocks - done < CH ? n_blocks - done : CH;
            hsw_engine::Slot &s = e->slot[chunk_idx & 1];
            const uint64_t cur = cursor0 + (uint64_t)done * LC;
            const size_t rows = (size_t)hsw_chip_rows(&e->shape, cur, nb);
            const size_t row_off = (size_t)(cur / ncols - cursor0 / ncols);
            if (chunk_idx >= 2 && (he = hipStreamWaitEvent(e->stream, s.copy_done, 0)) != hipSuccess) { fail("wait copy_done"); break; }
            rc = hsw_witness_blocks(e, d_blocks + 64 * done, d_pre + 8 * done, nb, cur, s.gate, s.cd, s.cs,
                                    e->slot_rows, d_next + 8 * done, flags);
            if (rc != HSW_OK) break;
            if ((he = hipEventRecord(s.kernel_done, e->stream)) != hipSuccess) { fail("record kernel_done"); break; }
            if ((he = hipStreamWaitEvent(e->copy_stream, s.kernel_done, 0)) != hipSuccess) { fail("wait kernel_done"); break; }
            if (want_gate && (he = d2h(static_cast<uint8_t *>(gate) + done * G * cb, s.gate,
                                       nb * G * cb, e->copy_stream)) != hipSuccess) { fail("D2H gate"); break; }
            if (want_chip) {
                // only the last chunk can end inside a row: the cells of that row past the call's last limb
                // belong to the next call and must keep what the caller's buffer holds
                const size_t tail = (size_t)((cur + (uint64_t)nb * LC) % ncols);
                for (size_t c = 0; c < ncols && he == hipSuccess; c++) {
                    const size_t dst = (c * chip_col_stride + row_off) * cb, src = c * e->slot_rows * cb;
                    const size_t own = rows - ((tail != 0 && c >= tail) ? 1 : 0);
                    if (own == 0) continue;
                    he = d2h(static_cast<uint8_t *>(chip_dense) + dst, static_cast<uint8_t *>(s.cd) + src,
                             own * cb, e->copy_stream);
                    if (he == hipSuccess)
                        he = d2h(static_cast<uint8_t *>(chip_spread) + dst, static_cast<uint8_t *>(s.cs) + src,
                                 own * cb, e->copy_stream);
                }
                if (he != hipSuccess) { fail("D2H chip columns"); break; }
            }
            if ((he = hipEventRecord(s.copy_done, e->copy_stream)) != hipSuccess) { fail("record copy_done"); break; }
        }
        if (rc != HSW_OK) break;
        if (next_states && (he = hipMemcpyAsync(next_states, d_next, n_blocks * 32, hipMemcpyDeviceToHost, e->stream)) != hipSuccess) { fail("D2H next_states"); break; }
        if ((he = hipStreamSynchronize(e->stream)) != hipSuccess) { fail("sync kernel stream"); break; }
        if ((he = hipStreamSynchronize(e->copy_stream)) != hipSuccess) { fail("sync copy stream"); break; }
    } while (0);
    if (rc != HSW_OK) { (void)hipStreamSynchronize(e->stream); (void)hipStreamSynchronize(e->copy_stream); }
    (void)hipFree(d_blocks); (void)hipFree(d_pre); (void)hipFree(d_next);
    return rc;
}

int hsw_host_alloc(size_t bytes, void **out) try {
    if (!out) return HSW_ERR_INVALID_ARG;
    *out = nullptr;
    hipError_t he = hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault);
    if (he != hipSuccess) return he == hipErrorOutOfMemory ? HSW_ERR_NOMEM : HSW_ERR_HIP;
    return HSW_OK;
} HSW_NO_UNWIND
void hsw_host_free(void *p) { if (p) (void)hipHostFree(p); }

int hsw_witness_blocks_host(hsw_engine *e, const uint8_t *blocks, const uint32_t *pre_states,
                            size_t n_blocks, uint64_t spread_cursor0, void *gate, void *chip_dense,
                            void *chip_spread, size_t chip_col_stride, uint32_t *next_states,
                            uint32_t flags) try {
    if (!e) return HSW_ERR_INVALID_ARG;
    if (n_blocks == 0) return HSW_OK;
    if (!blocks || !pre_states) return set_err(e, HSW_ERR_INVALID_ARG, "null input pointer");
    if (flags & HSW_CHAINED) return set_err(e, HSW_ERR_UNSUPPORTED, "HSW_CHAINED is for device-resident launches");
    if (!gate) flags |= HSW_SKIP_GATE;
    if (!chip_dense || !chip_spread) flags |= HSW_SKIP_CHIP;
    DeviceScope ds(e->device);
    if (!ds.ok) return set_err(e, HSW_ERR_NO_DEVICE, "hipSetDevice failed");

    const size_t G = e->shape.gate_cells_per_block;
    const size_t ncols = e->shape.num_advice_columns;
    const size_t rows = (size_t)hsw_chip_rows(&e->shape, spread_cursor0, n_blocks);
    if (!(flags & HSW_SKIP_CHIP) && chip_col_stride < rows)
        return set_err(e, HSW_ERR_INVALID_ARG, "chip_col_stride smaller than hsw_chip_rows()");
    const size_t cb = hsw_cell_bytes(flags);
    const bool pin = (flags & HSW_HOST_REGISTER) != 0;
    flags &= ~HSW_HOST_REGISTER;
    // Chip rows of consecutive chunks do not share a row when the cursor is a
    // multiple of ncols: then chunks can be produced and copied out independently.
    if (spread_cursor0 % ncols == 0)
        return pipelined_to_host(e, blocks, pre_states, n_blocks, spread_cursor0, gate, chip_dense, chip_spread,
                                 chip_col_stride, next_states, flags, pin);
    const size_t gate_bytes = (flags & HSW_SKIP_GATE) ? 0 : n_blocks * G * cb;
    const size_t col_bytes = (flags & HSW_SKIP_CHIP) ? 0 : ncols * rows * cb;

    uint8_t *d_blocks = nullptr; uint32_t *d_pre = nullptr, *d_next = nullptr;
    void *d_gate = nullptr, *d_cd = nullptr, *d_cs = nullptr;
    int rc = HSW_OK;
    hipError_t he = hipSuccess;
    auto fail = [&](const char *what) { rc = set_err(e, he == hipErrorOutOfMemory ? HSW_ERR_NOMEM : HSW_ERR_HIP, what, he); };
    do {
        if ((he = hipMalloc((void **)&d_blocks, n_blocks * 64)) != hipSuccess) { fail("hipMalloc blocks"); break; }
        if ((he = hipMalloc((void **)&d_pre, n_blocks * 32)) != hipSuccess) { fail("hipMalloc pre_states"); break; }
        if ((he = hipMalloc((void **)&d_next, n_blocks * 32)) != hipSuccess) { fail("hipMalloc next_states"); break; }
        if (gate_bytes && (he = hipMalloc(&d_gate, gate_bytes)) != hipSuccess) { fail("hipMalloc gate"); break; }
        if (col_bytes && (he = hipMalloc(&d_cd, col_bytes)) != hipSuccess) { fail("hipMalloc chip dense"); break; }
        if (col_bytes && (he = hipMalloc(&d_cs, col_bytes)) != hipSuccess) { fail("hipMalloc chip spread"); break; }
        if ((he = hipMemcpyAsync(d_blocks, blocks, n_blocks * 64, hipMemcpyHostToDevice, e->stream)) != hipSuccess) { fail("H2D blocks"); break; }
        if ((he = hipMemcpyAsync(d_pre, pre_states, n_blocks * 32, hipMemcpyHostToDevice, e->stream)) != hipSuccess) { fail("H2D pre_states"); break; }
        if (col_bytes) {
            // cells of the first / last row owned by neighbouring calls must survive the round trip
            for (size_t c = 0; c < ncols && he == hipSuccess; c++) {
                he = hipMemcpyAsync((uint8_t *)d_cd + c * rows * cb,
                                    (const uint8_t *)chip_dense + c * chip_col_stride * cb,
                                    rows * cb, hipMemcpyHostToDevice, e->stream);
                if (he == hipSuccess)
                    he = hipMemcpyAsync((uint8_t *)d_cs + c * rows * cb,
                                        (const uint8_t *)chip_spread + c * chip_col_stride * cb,
                                        rows * cb, hipMemcpyHostToDevice, e->stream);
            }
            if (he != hipSuccess) { fail("H2D chip columns"); break; }
        }
        rc = hsw_witness_blocks(e, d_blocks, d_pre, n_blocks, spread_cursor0, d_gate, d_cd, d_cs, rows, d_next, flags);
        if (rc != HSW_OK) break;
        if (gate_bytes && (he = hipMemcpyAsync(gate, d_gate, gate_bytes, hipMemcpyDeviceToHost, e->stream)) != hipSuccess) { fail("D2H gate"); break; }
        if (col_bytes) {
            for (size_t c = 0; c < ncols && he == hipSuccess; c++) {
                he = hipMemcpyAsync((uint8_t *)chip_dense + c * chip_col_stride * cb,
                                    (uint8_t *)d_cd + c * rows * cb, rows * cb,
                                    hipMemcpyDeviceToHost, e->stream);
                if (he == hipSuccess)
                    he = hipMemcpyAsync((uint8_t *)chip_spread + c * chip_col_stride * cb,
                                        (uint8_t *)d_cs + c * rows * cb, rows * cb,
                                        hipMemcpyDeviceToHost, e->stream);
            }
            if (he != hipSuccess) { fail("D2H chip columns"); break; }
        }
        if (next_states && (he = hipMemcpyAsync(next_states, d_next, n_blocks * 32, hipMemcpyDeviceToHost, e->stream)) != hipSuccess) { fail("D2H next_states"); break; }
        if ((he = hipStreamSynchronize(e->stream)) != hipSuccess) { fail("hipStreamSynchronize"); break; }
    } while (0);
    if (rc != HSW_OK) (void)hipStreamSynchronize(e->stream);
    (void)hipFree(d_blocks); (void)hipFree(d_pre); (void)hipFree(d_next);
    (void)hipFree(d_gate); (void)hipFree(d_cd); (void)hipFree(d_cs);
    return rc;
} HSW_NO_UNWIND

}  // extern "C"
