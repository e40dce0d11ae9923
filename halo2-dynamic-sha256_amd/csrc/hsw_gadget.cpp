// hsw_gadget.cpp -- Sha256DynamicConfig / Context mirror (see hsw_gadget.hpp)
// and its C ABI (include/hsw.h, "gadget front-end").
#include "hsw_gadget.hpp"

#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdlib>
#include <vector>
#include <cstring>
#include <new>
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#include <immintrin.h>
#endif

#include "hsw_nounwind.hpp"
#include "hsw_frame.hpp"
#include "hsw_kernels.h"

// library-internal entry points of hsw_api.cpp (hsw_engine.hpp)
bool hsw_small_eligible(const hsw_engine *e, size_t n_blocks);
int hsw_witness_blocks_impl(hsw_engine *e, const hsw_witness_args *args, const hsw::SmallFrames *frames,
                            uint32_t *host_next_states);
int hsw_witness_digests_impl(hsw_engine *e, const hsw_digests_args *args, uint32_t *dev_next_states);

namespace hsw {

namespace {

const uint32_t K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
    0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
    0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
    0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
    0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
const uint32_t INIT_STATE[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a,      // compression.rs:1003-1012
                                0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};

inline uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

// What sha2::compress256 does for the precomputed prefix (lib.rs:160).  The
// prefix is by definition NOT part of the circuit, so the reference hashes it
// on the CPU too; this is not a fallback of the witness path.
void plain_compress_scalar(uint32_t st[8], const uint8_t *block) {
    uint32_t w[64];
    for (int i = 0; i < 16; i++)
        w[i] = ((uint32_t)block[4 * i] << 24) | ((uint32_t)block[4 * i + 1] << 16) |
               ((uint32_t)block[4 * i + 2] << 8) | block[4 * i + 3];
    for (int i = 16; i < 64; i++) {
        const uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3);
        const uint32_t s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
        w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
    for (int i = 0; i < 64; i++) {
        const uint32_t t1 = h + (rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25)) + ((e & f) ^ (~e & g)) + K[i] + w[i];
        const uint32_t t2 = (rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
        h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
}

#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
// The same with the x86 SHA extensions (sha2 0.10.6 itself dispatches to them at run time): the plain
// chain of a long digest is the only serial part of the path, 0.4 us per block in scalar code.
// State lives as ABEF / CDGH, the operand order of sha256rnds2; a group of four rounds takes the four
// message words + K in one register, and sha256msg1 / sha256msg2 compute the next four schedule words.
__attribute__((target("sha,sse4.1,ssse3")))
void plain_compress_shani(uint32_t st[8], const uint8_t *block) {
    const __m128i bswap = _mm_set_epi64x(0x0c0d0e0f08090a0bULL, 0x0405060700010203ULL);
    __m128i tmp = _mm_loadu_si128(reinterpret_cast<const __m128i *>(&st[0]));        // d c b a (high .. low lane)
    __m128i s1 = _mm_loadu_si128(reinterpret_cast<const __m128i *>(&st[4]));         // h g f e
    tmp = _mm_shuffle_epi32(tmp, 0xB1);                                              // c d a b
    s1 = _mm_shuffle_epi32(s1, 0x1B);                                                // e f g h
    __m128i s0 = _mm_alignr_epi8(tmp, s1, 8);                                        // a b e f
    s1 = _mm_blend_epi16(s1, tmp, 0xF0);                                             // c d g h
    const __m128i abef_save = s0, cdgh_save = s1;
    __m128i m[4];
    for (int i = 0; i < 16; i++) {
        if (i < 4) {
            m[i] = _mm_shuffle_epi8(_mm_loadu_si128(reinterpret_cast<const __m128i *>(block + 16 * i)), bswap);
        } else {
            // W[4i..4i+3] from W[4i-16..], W[4i-12..], W[4i-8..], W[4i-4..]
            __m128i x = _mm_sha256msg1_epu32(m[i & 3], m[(i + 1) & 3]);              // W[t-16] + sigma0(W[t-15])
            x = _mm_add_epi32(x, _mm_alignr_epi8(m[(i + 3) & 3], m[(i + 2) & 3], 4));  // + W[t-7]
            m[i & 3] = _mm_sha256msg2_epu32(x, m[(i + 3) & 3]);                      // + sigma1(W[t-2])
        }
        __m128i wk = _mm_add_epi32(m[i & 3], _mm_loadu_si128(reinterpret_cast<const __m128i *>(&K[4 * i])));
        s1 = _mm_sha256rnds2_epu32(s1, s0, wk);
        wk = _mm_shuffle_epi32(wk, 0x0E);
        s0 = _mm_sha256rnds2_epu32(s0, s1, wk);
    }
    s0 = _mm_add_epi32(s0, abef_save);
    s1 = _mm_add_epi32(s1, cdgh_save);
    tmp = _mm_shuffle_epi32(s0, 0x1B);                                               // f e b a
    s1 = _mm_shuffle_epi32(s1, 0xB1);                                                // d c h g
    s0 = _mm_blend_epi16(tmp, s1, 0xF0);                                             // d c b a
    s1 = _mm_alignr_epi8(s1, tmp, 8);                                                // h g f e
    _mm_storeu_si128(reinterpret_cast<__m128i *>(&st[0]), s0);
    _mm_storeu_si128(reinterpret_cast<__m128i *>(&st[4]), s1);
}
bool have_shani() {
    static const bool ok = [] {
        if (std::getenv("HSW_NO_SHANI")) return false;      // tests: force the scalar code
        __builtin_cpu_init();
        return __builtin_cpu_supports("sha") != 0;
    }();
    return ok;
}
void plain_compress(uint32_t st[8], const uint8_t *block) {
    if (have_shani()) plain_compress_shani(st, block);
    else plain_compress_scalar(st, block);
}
bool host_sha_is_fast() { return have_shani(); }
#else
void plain_compress(uint32_t st[8], const uint8_t *block) { plain_compress_scalar(st, block); }
bool host_sha_is_fast() { return false; }
#endif

struct DeviceScope {
    int prev = -1;
    bool ok = false;
    explicit DeviceScope(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev == dev) { ok = true; prev = -1; }          // already current: nothing to set, nothing to restore
        else ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceScope() { if (prev >= 0) (void)hipSetDevice(prev); }
};

}  // namespace

struct DeviceScopeG : DeviceScope { using DeviceScope::DeviceScope; };   // for the C ABI functions below

int digest_prepare(const uint8_t *input, size_t input_byte_size, size_t precomputed_input_len,
                   size_t max_variable_byte_size, DigestPlan *plan) {
    if (!plan || (!input && input_byte_size)) return HSW_ERR_INVALID_ARG;
    const size_t one_round_size = 64;                                         // lib.rs:48
    if (max_variable_byte_size % one_round_size != 0) return HSW_ERR_SHAPE;   // lib.rs:57-59
    const size_t input_byte_size_with_9 = input_byte_size + 9;                // lib.rs:78
    const size_t num_round = (input_byte_size_with_9 + one_round_size - 1) / one_round_size;   // lib.rs:80-84
    const size_t padded_size = one_round_size * num_round;                    // lib.rs:85
    if (precomputed_input_len % one_round_size != 0) return HSW_ERR_SHAPE;    // lib.rs:89
    if (precomputed_input_len > padded_size ||
        padded_size - precomputed_input_len > max_variable_byte_size)
        return HSW_ERR_TOO_LARGE;                                             // lib.rs:90
    const size_t zero_padding_byte_size = padded_size - input_byte_size_with_9;               // lib.rs:91
    const size_t remaining_byte_size = max_variable_byte_size + precomputed_input_len - padded_size;   // lib.rs:92
    const size_t precomputed_round = precomputed_input_len / one_round_size;  // lib.rs:93
    const size_t total = max_variable_byte_size + precomputed_input_len;

    std::memcpy(plan->init_state, INIT_STATE, sizeof INIT_STATE);             // lib.rs:155
    const uint64_t bitlen = 8ull * (uint64_t)input_byte_size;                 // lib.rs:103-108 (big-endian)
    if (precomputed_input_len == 0) {
        // the common case: no prefix -- pad straight into the bytes fed to the circuit (lib.rs:98-117,170)
        plan->blocks.assign(max_variable_byte_size, 0);
        if (input_byte_size) std::memcpy(plan->blocks.data(), input, input_byte_size);
        size_t n = input_byte_size;
        plan->blocks[n++] = 0x80;                                             // lib.rs:99
        n += zero_padding_byte_size;                                          // lib.rs:100-102
        for (int i = 7; i >= 0; i--) plan->blocks[n++] = (uint8_t)(bitlen >> (8 * i));
        if (n != num_round * one_round_size) return HSW_ERR_INVALID_ARG;      // lib.rs:110
        if (n + remaining_byte_size != total) return HSW_ERR_INVALID_ARG;     // lib.rs:111-117
    } else {
        std::vector<uint8_t> padded(total, 0);                                // lib.rs:98-117
        if (input_byte_size) std::memcpy(padded.data(), input, input_byte_size);
        size_t n = input_byte_size;
        padded[n++] = 0x80;                                                   // lib.rs:99
        n += zero_padding_byte_size;                                          // lib.rs:100-102
        for (int i = 7; i >= 0; i--) padded[n++] = (uint8_t)(bitlen >> (8 * i));
        if (n != num_round * one_round_size) return HSW_ERR_INVALID_ARG;      // lib.rs:110
        if (n + remaining_byte_size != total) return HSW_ERR_INVALID_ARG;     // lib.rs:111-117
        for (size_t r = 0; r < precomputed_round; r++)                        // lib.rs:156-160
            plain_compress(plan->init_state, padded.data() + r * one_round_size);
        plan->blocks.assign(padded.begin() + (ptrdiff_t)precomputed_input_len, padded.end());   // lib.rs:170
    }
    plan->num_round = num_round;
    plan->precomputed_round = precomputed_round;
    plan->target_round = num_round - precomputed_round;
    plan->max_variable_round = max_variable_byte_size / one_round_size;
    return HSW_OK;
}

int Sha256DynamicConfig::configure(const std::vector<size_t> &sizes, uint32_t num_bits_lookup,
                                   uint32_t num_advice_columns, bool is_input_range_check,
                                   Sha256DynamicConfig *out) {
    if (!out) return HSW_ERR_INVALID_ARG;
    for (size_t b : sizes)
        if (b % 64 != 0) return HSW_ERR_SHAPE;                                // lib.rs:57-59
    hsw_shape s;
    const int rc = hsw_shape_query(num_bits_lookup, num_advice_columns, &s);  // SpreadConfig::configure, spread.rs:37
    if (rc != HSW_OK) return rc;
    out->max_variable_byte_sizes = sizes;
    out->cur_hash_idx = 0;                                                    // lib.rs:66
    out->num_bits_lookup = num_bits_lookup;
    out->num_advice_columns = num_advice_columns;
    out->is_input_range_check = is_input_range_check;
    return HSW_OK;
}

std::vector<std::pair<uint64_t, uint64_t>> Sha256DynamicConfig::load() const {
    std::vector<std::pair<uint64_t, uint64_t>> rows;                          // spread.rs:169-189
    for (uint64_t idx = 0; idx < (1ull << num_bits_lookup); idx++) {
        uint64_t sp = 0;
        for (int b = 0; b < 32; b++) sp |= ((idx >> b) & 1ull) << (2 * b);
        rows.emplace_back(idx, sp);
    }
    return rows;
}

Context::~Context() {
    (void)hipFree(d_gate); (void)hipFree(d_chip_dense); (void)hipFree(d_chip_spread);
    (void)hipFree(d_next_states); (void)hipFree(d_blocks); (void)hipFree(d_pre_states);
    (void)hipFree(d_init_states); (void)hipFree(d_offsets); (void)hipFree(d_lookup);
    if (hp_blocks) (void)hipHostFree(hp_blocks);
    free_compact_staging();
}

void Context::free_compact_staging() {
    (void)hipFree(d_c_gate); (void)hipFree(d_c_lookup); (void)hipFree(d_c_dense); (void)hipFree(d_c_spread);
    (void)hipFree(d_wide); (void)hipFree(d_wide_count);
    if (hp_wide_count) (void)hipHostFree(hp_wide_count);
    d_c_gate = d_c_lookup = d_c_dense = d_c_spread = d_wide = nullptr;
    d_wide_count = hp_wide_count = nullptr;
    wide_cap = 0;
}

int Sha256DynamicConfig::new_context(hsw_engine *engine, Context **out, bool whole_digest, bool independent) const {
    if (!engine || !out) return HSW_ERR_INVALID_ARG;
    *out = nullptr;
    hsw_shape s;
    int rc = hsw_engine_shape(engine, &s);
    if (rc != HSW_OK) return rc;
    if (s.num_bits_lookup != num_bits_lookup || s.num_advice_columns != num_advice_columns)
        return HSW_ERR_SHAPE;
    int device = 0;
    hsw_engine_stream(engine, nullptr, &device);
    DeviceScope ds(device);                                   // the context's buffers live on the engine's GPU
    if (!ds.ok) return HSW_ERR_NO_DEVICE;
    Context *c = new (std::nothrow) Context();
    if (!c) return HSW_ERR_NOMEM;
    c->engine = engine;
    c->shape = s;
    size_t total = 0;
    for (size_t b : max_variable_byte_sizes) total += b / 64;
    c->capacity_blocks = total;
    c->chip_col_stride = (size_t)hsw_chip_rows(&s, 0, total);
    c->init_capacity = max_variable_byte_sizes.size();
    const size_t nb = total ? total : 1, nh = c->init_capacity ? c->init_capacity : 1;
    size_t gate_cells = nb * (size_t)s.gate_cells_per_block;
    if (whole_digest) {
        if (s.mode != HSW_MODE_HALO2_INTERNALS) { delete c; return HSW_ERR_INVALID_ARG; }
        c->whole = true;
        c->independent = independent;
        // the Context's zero cell: one, or one per digest when every digest is a Context of its own
        uint64_t cells = independent ? max_variable_byte_sizes.size() : 1, lookups = 0;
        for (size_t b : max_variable_byte_sizes) {
            if (independent && ((b / 64) * (uint64_t)s.limb_calls_per_block) % s.num_advice_columns != 0) {
                delete c;
                return HSW_ERR_UNSUPPORTED;                   // a context's chip rows must start on a row of their own
            }
            hsw_frame_shape fs;
            rc = hsw_frame_query(&s, b, is_input_range_check ? 1 : 0, &fs);
            if (rc == HSW_OK && fs.n_blocks == 0) rc = HSW_ERR_UNSUPPORTED;
            if (rc != HSW_OK) { delete c; return rc; }
            cells += fs.digest_cells;
            lookups += fs.digest_lookups;
        }
        c->gate_capacity = cells;
        c->lookup_capacity = c->own_lookup_capacity = lookups;
        gate_cells = (size_t)cells;
    }
    hipError_t he = hipMalloc(&c->d_gate, gate_cells * HSW_CELL_BYTES);
    // touch the stream buffers once: the first write into fresh device memory is several times slower
    // (measured: 16-block digests 266 us instead of 54 us while a context's buffer was still untouched)
    if (he == hipSuccess) he = hipMemset(c->d_gate, 0, gate_cells * HSW_CELL_BYTES);
    if (he == hipSuccess && whole_digest) {
        const size_t lbytes = (size_t)(c->lookup_capacity ? c->lookup_capacity : 1) * HSW_CELL_BYTES;
        he = hipMalloc(&c->d_lookup, lbytes);
        if (he == hipSuccess) he = hipMemset(c->d_lookup, 0, lbytes);
    }
    const size_t col_bytes = (size_t)s.num_advice_columns * (c->chip_col_stride ? c->chip_col_stride : 1) * HSW_CELL_BYTES;
    if (he == hipSuccess) he = hipMalloc(&c->d_chip_dense, col_bytes);
    if (he == hipSuccess) he = hipMalloc(&c->d_chip_spread, col_bytes);
    if (he == hipSuccess) he = hipMalloc((void **)&c->d_next_states, nb * 32);
    if (he == hipSuccess) he = hipMalloc((void **)&c->d_blocks, nb * 64);
    if (he == hipSuccess) he = hipMalloc((void **)&c->d_pre_states, nb * 32);
    if (he == hipSuccess) he = hipMalloc((void **)&c->d_init_states, nh * 32);
    if (he == hipSuccess) he = hipMalloc((void **)&c->d_offsets, (nh + 1) * sizeof(uint32_t));
    if (he == hipSuccess) {
        void *pin = nullptr, *dpin = nullptr;
        he = hipHostMalloc(&pin, nb * 128, hipHostMallocMapped);
        if (he == hipSuccess) {
            c->hp_blocks = static_cast<uint8_t *>(pin);
            he = hipHostGetDevicePointer(&dpin, pin, 0);
        }
        if (he == hipSuccess) {
            c->hp_pre = reinterpret_cast<uint32_t *>(c->hp_blocks + nb * 64);
            c->hp_next = reinterpret_cast<uint32_t *>(c->hp_blocks + nb * 96);
            c->dp_blocks = static_cast<uint8_t *>(dpin);
            c->dp_pre = reinterpret_cast<uint32_t *>(c->dp_blocks + nb * 64);
            c->dp_next = reinterpret_cast<uint32_t *>(c->dp_blocks + nb * 96);
        }
    }
    if (he == hipSuccess) he = hipMemset(c->d_chip_dense, 0, col_bytes);
    if (he == hipSuccess) he = hipMemset(c->d_chip_spread, 0, col_bytes);
    if (he != hipSuccess) {
        delete c;
        return he == hipErrorOutOfMemory ? HSW_ERR_NOMEM : HSW_ERR_HIP;
    }
    *out = c;
    return HSW_OK;
}

int Context::set_columns(const std::vector<size_t> &sizes, bool rc_inputs, uint64_t rows) {
    if (!whole || blocks_done != 0 || gate_cursor != 0) return HSW_ERR_INVALID_ARG;
    if (independent) return HSW_ERR_UNSUPPORTED;         // K regions in one stream: linear only
    const uint64_t G = shape.gate_cells_per_block;
    if (rows < G + 16) return HSW_ERR_INVALID_ARG;        // keeps a block inside <= 2 columns (kernel: <= 2 breaks per block)
    if (origin_row >= rows) return HSW_ERR_INVALID_ARG;   // the Context's next free row lies inside its column
    size_t n = 0;
    if (hsw_gate_tape(&shape, nullptr, 0, &n) != HSW_OK) return HSW_ERR_INVALID_ARG;
    std::vector<uint8_t> block_tape(n);
    hsw_gate_tape(&shape, block_tape.data(), n, nullptr);
    std::vector<uint64_t> bc, bg;
    uint64_t row = origin_row, cell = 0;                  // the Context's next free row (hsw_gadget_set_origin)
    auto walk = [&](const std::vector<uint8_t> &lens) {
        for (uint8_t len : lens) {
            if (row + len >= rows) {                      // halo2-lib v0.2.x assign_region: next column (A3-iii)
                bc.push_back(cell); bg.push_back(rows - row);
                row = 0;
            }
            row += len; cell += len;
        }
    };
    bool zero = origin_zero_loaded;                       // a Context that already caches its zero cell assigns none
    for (size_t b : sizes) {
        for (int section = 0; section < 2; section++) {
            if (section == 1) {
                if (!zero) { walk({1}); zero = true; }    // Context.zero_cell, first load_zero
                for (size_t k = 0; k < b / 64; k++) {
                    if (row + G + 8 < rows) { row += G; cell += G; }
                    else walk(block_tape);
                }
            }
            size_t m = 0;
            int rc = hsw_frame_tape(&shape, b, rc_inputs ? 1 : 0, section, nullptr, 0, &m);
            if (rc != HSW_OK) return rc;
            std::vector<uint8_t> t(m);
            hsw_frame_tape(&shape, b, rc_inputs ? 1 : 0, section, t.data(), m, nullptr);
            walk(t);
        }
    }
    if (bc.size() > HSW_MAX_BREAKS) return HSW_ERR_TOO_LARGE;
    const uint64_t cols = bc.size() + 1;
    int device = 0;
    hsw_engine_stream(engine, nullptr, &device);
    DeviceScope ds2(device);
    if (!ds2.ok) return HSW_ERR_NO_DEVICE;
    // (outstanding work on the old image: the callers -- hsw_gadget_set_columns / _set_origin -- run on a drained engine)
    void *img = nullptr;
    hipError_t he = hipMalloc(&img, (size_t)(cols * rows) * HSW_CELL_BYTES);
    if (he == hipSuccess) he = hipMemset(img, 0, (size_t)(cols * rows) * HSW_CELL_BYTES);   // unassigned advice cells are 0
    if (he != hipSuccess) { if (img) (void)hipFree(img); return he == hipErrorOutOfMemory ? HSW_ERR_NOMEM : HSW_ERR_HIP; }
    (void)hipFree(d_gate);
    d_gate = img;
    free_compact_staging();                               // sized for the old geometry
    max_rows = rows;
    columns = cols;
    break_cell.swap(bc);
    break_gap.swap(bg);
    return HSW_OK;
}

void Context::position(uint64_t cell, uint64_t *column, uint64_t *row) const {
    uint64_t at = cell + origin_row;
    for (size_t k = 0; k < break_cell.size(); k++)
        if (break_cell[k] <= cell) at += break_gap[k];
    if (max_rows) { if (column) *column = origin_column + at / max_rows; if (row) *row = at % max_rows; }
    else { if (column) *column = origin_column; if (row) *row = at; }
}

int Context::set_origin(uint64_t column, uint64_t row, bool zero_cell_loaded, uint64_t lookups_queued) {
    if (!whole || blocks_done != 0 || gate_cursor != 0 || lookup_cursor != origin_lookups) return HSW_ERR_INVALID_ARG;
    if (independent) return HSW_ERR_UNSUPPORTED;
    if (max_rows && row >= max_rows) return HSW_ERR_INVALID_ARG;
    if (lookups_queued != origin_lookups) {
        // the lookup-advice stream is indexed from the Context's first queued cell: [0, lookups_queued) are the caller's
        int device = 0;
        hsw_engine_stream(engine, nullptr, &device);
        DeviceScope ds(device);
        if (!ds.ok) return HSW_ERR_NO_DEVICE;
        const size_t lbytes = (size_t)(own_lookup_capacity + lookups_queued ? own_lookup_capacity + lookups_queued : 1) * HSW_CELL_BYTES;
        void *lk = nullptr;
        hipError_t he = hipMalloc(&lk, lbytes);
        if (he == hipSuccess) he = hipMemset(lk, 0, lbytes);
        if (he != hipSuccess) { if (lk) (void)hipFree(lk); return he == hipErrorOutOfMemory ? HSW_ERR_NOMEM : HSW_ERR_HIP; }
        (void)hipFree(d_lookup);
        d_lookup = lk;
        free_compact_staging();
    }
    origin_column = column; origin_row = row; origin_zero_loaded = zero_cell_loaded; origin_lookups = lookups_queued;
    lookup_capacity = own_lookup_capacity + lookups_queued;
    lookup_cursor = lookups_queued;
    zero_loaded = zero_cell_loaded;
    // without the zero cell the stream is one cell shorter
    return HSW_OK;
}

int Sha256DynamicConfig::digest(Context &ctx, const uint8_t *input, size_t input_len,
                                size_t precomputed_input_len, AssignedHashResult *result) {
    return digest_batch(ctx, 1, &input, &input_len, &precomputed_input_len, result);
}

int Sha256DynamicConfig::digest_batch(Context &ctx, size_t n, const uint8_t *const *inputs,
                                      const size_t *input_lens, const size_t *precomputed_input_lens,
                                      AssignedHashResult *results) {
    if (!results || !inputs || !input_lens) return HSW_ERR_INVALID_ARG;
    if (n == 0) return HSW_OK;
    // max_variable_byte_sizes[cur_hash_idx] must exist for every hash (lib.rs:86 would panic)
    if (cur_hash_idx + n > max_variable_byte_sizes.size()) return HSW_ERR_INVALID_ARG;

    // ---- host: lib.rs:77-160 for every message; nothing is committed on error ----
    std::vector<DigestPlan> plans(n);
    size_t batch_blocks = 0;
    for (size_t i = 0; i < n; i++) {
        const size_t max_sz = max_variable_byte_sizes[cur_hash_idx + i];
        const int rc = digest_prepare(inputs[i], input_lens[i],
                                      precomputed_input_lens ? precomputed_input_lens[i] : 0, max_sz, &plans[i]);
        if (rc != HSW_OK) return rc;
        batch_blocks += plans[i].max_variable_round;
    }
    if (ctx.blocks_done + batch_blocks > ctx.capacity_blocks || n > ctx.init_capacity) return HSW_ERR_INVALID_ARG;

    std::vector<uint8_t> h_blocks(batch_blocks * 64 ? batch_blocks * 64 : 1);
    std::vector<uint32_t> h_init(n * 8), h_offsets(n + 1);
    size_t off = 0;
    for (size_t i = 0; i < n; i++) {
        h_offsets[i] = (uint32_t)off;
        if (!plans[i].blocks.empty()) std::memcpy(h_blocks.data() + off * 64, plans[i].blocks.data(), plans[i].blocks.size());
        std::memcpy(&h_init[8 * i], plans[i].init_state, 32);
        off += plans[i].max_variable_round;
    }
    h_offsets[n] = (uint32_t)off;
    // The plain SHA chain (pre-state of every block, lib.rs:188,236) is the only serial part.  Chained
    // on the host it sits next to the prefix pre-hash the reference also does on the CPU (lib.rs:153-160)
    // and saves a dependent kernel launch; on the GPU (hsw_chain_var_kernel) every message has its own
    // lane.  Either way the witness cells -- and the next_states the digest is read from -- come from the
    // GPU.  Host-chained batches stage blocks and pre-states in pinned, device-mapped host memory.
    // Which side chains: the host walks all blocks at ~0.1 us each (x86 SHA extensions; 0.4 us scalar), the
    // GPU chains every message on its own wave (up to 2,048 messages: ~1.8 us per block) or lane (~3.6 us per
    // block) plus a dependent launch.  Many short messages -> GPU; few long ones -> host.
    size_t longest = 0;
    for (size_t i = 0; i < n; i++) longest = plans[i].max_variable_round > longest ? plans[i].max_variable_round : longest;
    const double t_host_us = (double)batch_blocks * (host_sha_is_fast() ? 0.1 : 0.4);
    const double t_gpu_us = 15.0 + (n <= (size_t)HSW_CHAIN_WAVE_MAX_MESSAGES ? 1.8 : 3.6) * (double)longest;   // a wave / a lane per message
    const bool host_chain = t_host_us <= t_gpu_us;
    const size_t b0 = ctx.blocks_done;
    if (host_chain && batch_blocks) {
        std::memcpy(ctx.hp_blocks + 64 * b0, h_blocks.data(), batch_blocks * 64);
        uint32_t *h_pre = ctx.hp_pre + 8 * b0;
        for (size_t i = 0; i < n; i++) {
            uint32_t st[8];
            std::memcpy(st, plans[i].init_state, 32);
            for (size_t j = 0; j < plans[i].max_variable_round; j++) {
                const size_t b = h_offsets[i] + j;
                std::memcpy(&h_pre[8 * b], st, 32);
                plain_compress(st, h_blocks.data() + 64 * b);
            }
        }
    }

    // ---- device: chain pre-pass + ONE expansion launch for the whole batch ----
    hipStream_t stream = nullptr;
    int device = 0;
    hsw_engine_stream(ctx.engine, reinterpret_cast<void **>(&stream), &device);
    DeviceScope ds(device);
    if (!ds.ok) return HSW_ERR_NO_DEVICE;
    // Small-batch launches (and any launch of up to 32 blocks) read their 96 input bytes per block straight from
    // the pinned staging (uncached PCIe reads: cheaper than two dependent copies while the waves are few).
    // Tiny batches (the reference's bench circuit is ONE 16-block digest) are latency-bound: they go to the
    // small-batch kernel, which for a whole-digest context also writes the frames -- ONE launch, inputs read
    // in place from the pinned staging, next states written straight into pinned memory, no copy launches.
    // (whole-digest contexts: one such launch per run of equally sized digests, each with its own frames)
    const bool small = hsw_small_eligible(ctx.engine, batch_blocks);
    const bool zero_copy = host_chain && (ctx.whole ? small : (small || batch_blocks <= 32));
    const uint8_t *in_blocks = zero_copy ? ctx.dp_blocks : ctx.d_blocks;        // bases, indexed by absolute block
    const uint32_t *in_pre = zero_copy ? ctx.dp_pre : ctx.d_pre_states;
    const uint8_t *d_blk = in_blocks + 64 * b0;
    const uint32_t *d_pre = in_pre + 8 * b0;
    uint32_t *d_next = ctx.d_next_states + 8 * b0;
    uint32_t *d_off = ctx.d_offsets;
    uint32_t *h_next = ctx.hp_next + 8 * b0;                                     // pinned: the D2H below is asynchronous
    hipError_t he = hipSuccess;
    int rc = HSW_OK;
    bool next_in_pinned = false;               // the kernel wrote the next states into hp_next itself
    std::vector<hsw_frame_desc> frames;
    uint64_t new_gate_cursor = ctx.gate_cursor, new_lookup_cursor = ctx.lookup_cursor;
    do {
        if (batch_blocks == 0) break;
        if (host_chain && !zero_copy) {      // from pinned memory: both copies are asynchronous DMA
            if ((he = hipMemcpyAsync(ctx.d_blocks + 64 * b0, ctx.hp_blocks + 64 * b0, batch_blocks * 64, hipMemcpyHostToDevice, stream)) != hipSuccess) break;
            if ((he = hipMemcpyAsync(ctx.d_pre_states + 8 * b0, ctx.hp_pre + 8 * b0, batch_blocks * 32, hipMemcpyHostToDevice, stream)) != hipSuccess) break;
        }
        if (!host_chain) {
            if ((he = hipMemcpyAsync(ctx.d_blocks + 64 * b0, h_blocks.data(), batch_blocks * 64, hipMemcpyHostToDevice, stream)) != hipSuccess) break;
            if ((he = hipMemcpyAsync(ctx.d_init_states, h_init.data(), n * 32, hipMemcpyHostToDevice, stream)) != hipSuccess) break;
            if ((he = hipMemcpyAsync(d_off, h_offsets.data(), (n + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, stream)) != hipSuccess) break;
            if ((he = launch_chain_var(d_blk, n, d_off, ctx.d_init_states, ctx.d_pre_states + 8 * b0, stream)) != hipSuccess) break;
        }
        const size_t G = ctx.shape.gate_cells_per_block;
        const size_t cb = hsw_cell_bytes(ctx.repr_flags);
        const uint32_t ncols = ctx.shape.num_advice_columns;
        if (!ctx.whole) {
            // one call covers every block of the batch; the chip cursor is the running num_limb_sum.
            // Column buffers are addressed from absolute row 0 (cursor origin of the context).
            const uint64_t row_shift = ctx.num_limb_sum / ncols;
            hsw_witness_args a{};
            a.d_blocks = d_blk; a.d_pre_states = d_pre; a.n_blocks = batch_blocks;
            a.spread_cursor0 = ctx.num_limb_sum;
            a.d_gate = static_cast<uint8_t *>(ctx.d_gate) + b0 * G * cb;
            a.d_chip_dense = static_cast<uint8_t *>(ctx.d_chip_dense) + (size_t)row_shift * cb;
            a.d_chip_spread = static_cast<uint8_t *>(ctx.d_chip_spread) + (size_t)row_shift * cb;
            a.chip_col_stride = ctx.chip_col_stride;
            a.d_next_states = d_next;
            a.flags = ctx.repr_flags;
            rc = hsw_witness_blocks_impl(ctx.engine, &a, nullptr, small ? ctx.dp_next + 8 * b0 : nullptr);
            next_in_pinned = small && rc == HSW_OK;
        } else {
            // whole-digest stream: prologue | [zero cell] | blocks | epilogue per digest (hsw_frame.hpp).
            // Consecutive digests of equal size are ONE expansion launch (the kernel skips the frame
            // between their block streams); all frames of the batch are one hsw_frame_kernel launch.
            const size_t LK = ctx.shape.lookup_cells_per_block;
            uint64_t gc = ctx.gate_cursor, lc = ctx.lookup_cursor;
            bool zero_loaded = ctx.zero_loaded;
            frames.resize(n);
            std::vector<hsw_frame_shape> fss(n);
            size_t ob = 0;
            for (size_t i = 0; i < n && rc == HSW_OK; i++) {
                rc = hsw_frame_query(&ctx.shape, max_variable_byte_sizes[cur_hash_idx + i], is_input_range_check ? 1 : 0, &fss[i]);
                if (rc != HSW_OK) break;
                hsw_frame_desc &d = frames[i];
                AssignedHashResult &r = results[i];
                d.input_len = input_lens[i];
                d.first_block = b0 + ob;
                d.n_blocks = (uint32_t)plans[i].max_variable_round;
                d.num_round = (uint32_t)plans[i].num_round;
                d.precomputed_round = (uint32_t)plans[i].precomputed_round;
                d.is_input_range_check = is_input_range_check ? 1u : 0u;
                r.prologue_cell = d.prologue_cell = gc;      gc += fss[i].prologue_cells;
                r.prologue_lookup = d.prologue_lookup = lc;  lc += fss[i].prologue_lookups;
                d.zero_cell = ~0ull;
                if (!zero_loaded || ctx.independent) { d.zero_cell = gc++; zero_loaded = true; }   // compression.rs:34 of the first block of a Context
                r.block_cell = gc;                           gc += (uint64_t)d.n_blocks * G;
                r.block_lookup = lc;                         lc += (uint64_t)d.n_blocks * LK;
                r.epilogue_cell = d.epilogue_cell = gc;      gc += fss[i].epilogue_cells;
                r.epilogue_lookup = d.epilogue_lookup = lc;  lc += fss[i].epilogue_lookups;
                r.end_cell = gc;
                ob += d.n_blocks;
            }
            if (rc == HSW_OK && (gc > ctx.gate_capacity || lc > ctx.lookup_capacity)) rc = HSW_ERR_INVALID_ARG;
            ob = 0;
            for (size_t i = 0; i < n && rc == HSW_OK;) {
                size_t j = i + 1;                            // run [i, j) of equally sized digests
                while (j < n && frames[j].n_blocks == frames[i].n_blocks) j++;
                const size_t nb = frames[i].n_blocks, run_blocks = nb * (j - i);
                const uint64_t cursor = ctx.num_limb_sum + (uint64_t)ob * ctx.shape.limb_calls_per_block;
                const uint64_t row_shift = cursor / ncols;
                hsw_witness_args a{};
                a.d_blocks = d_blk + 64 * ob;
                a.d_pre_states = d_pre + 8 * ob;
                a.n_blocks = run_blocks;
                a.spread_cursor0 = cursor;
                a.d_gate = static_cast<uint8_t *>(ctx.gate_stream()) + (size_t)results[i].block_cell * cb;
                a.d_chip_dense = static_cast<uint8_t *>(ctx.d_chip_dense) + (size_t)row_shift * cb;
                a.d_chip_spread = static_cast<uint8_t *>(ctx.d_chip_spread) + (size_t)row_shift * cb;
                a.chip_col_stride = ctx.chip_col_stride;
                a.d_next_states = d_next + 8 * ob;
                a.d_lookup = static_cast<uint8_t *>(ctx.d_lookup) + (size_t)results[i].block_lookup * cb;
                a.flags = ctx.repr_flags;
                a.frame_every = nb;
                // (between the block streams of two digests: one epilogue, the next prologue -- and the next Context's
                //  zero cell when every digest is a Context of its own)
                a.frame_cells = fss[i].epilogue_cells + fss[i].prologue_cells + (ctx.independent ? 1u : 0u);
                a.frame_lookups = fss[i].epilogue_lookups + fss[i].prologue_lookups;
                hsw_pack_plan plan{};
                if (ctx.max_rows) {
                    // column breaks relative to this launch's first cell; breaks before it are pure offsets
                    const uint64_t base = results[i].block_cell;
                    plan.n_breaks = (uint32_t)ctx.break_cell.size();
                    for (size_t k = 0; k < ctx.break_cell.size(); k++) {
                        plan.break_cell[k] = ctx.break_cell[k] > base ? ctx.break_cell[k] - base : 0;
                        plan.break_gap[k] = ctx.break_gap[k];
                    }
                    a.pack = &plan;
                }
                if (small) {
                    hsw_digests_args da{};
                    da.blocks = a;
                    da.descs = frames.data() + i; da.n_digests = j - i;      // this run's digests: frames in the same launch
                    da.d_blocks0 = in_blocks; da.d_pre_states0 = in_pre; da.d_next_states0 = ctx.d_next_states;
                    da.d_gate0 = ctx.gate_stream(); da.d_lookup0 = ctx.d_lookup;
                    hsw_pack_plan abs_plan{};
                    abs_plan.n_breaks = (uint32_t)ctx.break_cell.size();
                    for (size_t k = 0; k < ctx.break_cell.size(); k++) {
                        abs_plan.break_cell[k] = ctx.break_cell[k];
                        abs_plan.break_gap[k] = ctx.break_gap[k];
                    }
                    da.frame_pack = ctx.max_rows ? &abs_plan : nullptr;
                    da.host_next_states = h_next + 8 * ob;
                    // (the device alias of the context's own pinned staging: no runtime lookup per call)
                    rc = hsw_witness_digests_impl(ctx.engine, &da, ctx.dp_next + 8 * (b0 + ob));
                    next_in_pinned = rc == HSW_OK;
                } else {
                    rc = hsw_witness_blocks_ex(ctx.engine, &a);
                }
                ob += run_blocks;
                i = j;
            }
            if (rc == HSW_OK && !small) {
                hsw_pack_plan plan{};
                plan.n_breaks = (uint32_t)ctx.break_cell.size();
                for (size_t k = 0; k < ctx.break_cell.size(); k++) {
                    plan.break_cell[k] = ctx.break_cell[k];
                    plan.break_gap[k] = ctx.break_gap[k];
                }
                rc = hsw_witness_frames(ctx.engine, frames.data(), n, in_blocks, in_pre, ctx.d_next_states,
                                        ctx.gate_stream(), ctx.d_lookup, ctx.max_rows ? &plan : nullptr, ctx.repr_flags);
            }
            if (rc == HSW_OK) { new_gate_cursor = gc; new_lookup_cursor = lc; }
        }
        if (rc != HSW_OK) break;
        if (!next_in_pinned &&
            (he = hipMemcpyAsync(h_next, d_next, batch_blocks * 32, hipMemcpyDeviceToHost, stream)) != hipSuccess) break;
        he = hipStreamSynchronize(stream);
    } while (0);
    if (rc != HSW_OK) return rc;
    if (he != hipSuccess) return he == hipErrorOutOfMemory ? HSW_ERR_NOMEM : HSW_ERR_HIP;

    // ---- results: the "select state #target_round" rule (lib.rs:294-310) ----
    off = 0;
    for (size_t i = 0; i < n; i++) {
        AssignedHashResult &r = results[i];
        const DigestPlan &pl = plans[i];
        r.input_len = input_lens[i];
        r.input_bytes = std::move(plans[i].blocks);          // the plan is done with them (copied to the staging above)
        r.first_block = b0 + off;
        r.n_blocks = pl.max_variable_round;
        r.spread_cursor0 = ctx.num_limb_sum + (uint64_t)off * ctx.shape.limb_calls_per_block;
        r.num_round = pl.num_round;
        r.target_round = pl.target_round;
        uint32_t sel[8] = {0, 0, 0, 0, 0, 0, 0, 0};            // output_h_out starts as zero cells (lib.rs:294-295)
        if (pl.target_round == 0) std::memcpy(sel, pl.init_state, 32);                 // candidate 0
        else if (pl.target_round <= pl.max_variable_round)
            std::memcpy(sel, &h_next[8 * (off + pl.target_round - 1)], 32);            // candidate target_round
        for (int w = 0; w < 8; w++) {                           // lib.rs:311-341 big-endian bytes
            r.output_bytes[4 * w] = (uint8_t)(sel[w] >> 24);
            r.output_bytes[4 * w + 1] = (uint8_t)(sel[w] >> 16);
            r.output_bytes[4 * w + 2] = (uint8_t)(sel[w] >> 8);
            r.output_bytes[4 * w + 3] = (uint8_t)sel[w];
        }
        off += pl.max_variable_round;
    }
    ctx.batches.push_back(Context::BatchRecord{cur_hash_idx, n, b0, batch_blocks, zero_copy, ctx.repr_flags});
    ctx.blocks_done += batch_blocks;
    if (ctx.whole) {
        ctx.gate_cursor = new_gate_cursor;
        ctx.lookup_cursor = new_lookup_cursor;
        ctx.zero_loaded = ctx.zero_loaded || batch_blocks != 0;
    }
    ctx.num_limb_sum += (uint64_t)batch_blocks * ctx.shape.limb_calls_per_block;   // spread.rs:228
    cur_hash_idx += n;                                                             // lib.rs:347
    return HSW_OK;
}

}  // namespace hsw

// ------------------------------------------------------------------- C ABI

extern "C" {

int hsw_digest_prepare(const uint8_t *input, size_t input_len, size_t precomputed_input_len,
                       size_t max_variable_byte_size, uint8_t *blocks_out, uint32_t init_state_out[8],
                       hsw_digest_info *info) try {
    hsw::DigestPlan plan;
    const int rc = hsw::digest_prepare(input, input_len, precomputed_input_len, max_variable_byte_size, &plan);
    if (rc != HSW_OK) return rc;
    if (blocks_out && !plan.blocks.empty()) std::memcpy(blocks_out, plan.blocks.data(), plan.blocks.size());
    if (init_state_out) std::memcpy(init_state_out, plan.init_state, 32);
    if (info) {
        info->num_round = plan.num_round;
        info->precomputed_round = plan.precomputed_round;
        info->target_round = plan.target_round;
        info->n_blocks = plan.max_variable_round;
    }
    return HSW_OK;
} HSW_NO_UNWIND

int hsw_gadget_create(hsw_engine *e, const size_t *max_variable_byte_sizes, size_t n_hashes,
                      int is_input_range_check, hsw_gadget **out) try {
    return hsw_gadget_create_ex(e, max_variable_byte_sizes, n_hashes, is_input_range_check, 0, out);
} HSW_NO_UNWIND

int hsw_gadget_create_ex(hsw_engine *e, const size_t *max_variable_byte_sizes, size_t n_hashes,
                         int is_input_range_check, uint32_t flags, hsw_gadget **out) try {
    if (!e || !out || (!max_variable_byte_sizes && n_hashes)) return HSW_ERR_INVALID_ARG;
    if (flags & ~(HSW_GADGET_WHOLE_DIGEST | HSW_GADGET_INDEPENDENT)) return HSW_ERR_INVALID_ARG;
    if ((flags & HSW_GADGET_INDEPENDENT) && !(flags & HSW_GADGET_WHOLE_DIGEST)) return HSW_ERR_INVALID_ARG;
    *out = nullptr;
    hsw_shape s;
    int rc = hsw_engine_shape(e, &s);
    if (rc != HSW_OK) return rc;
    hsw_gadget *g = new (std::nothrow) hsw_gadget();
    if (!g) return HSW_ERR_NOMEM;
    std::vector<size_t> sizes(max_variable_byte_sizes, max_variable_byte_sizes + n_hashes);
    rc = hsw::Sha256DynamicConfig::configure(sizes, s.num_bits_lookup, s.num_advice_columns,
                                             is_input_range_check != 0, &g->cfg);
    if (rc == HSW_OK) rc = g->cfg.new_context(e, &g->ctx, (flags & HSW_GADGET_WHOLE_DIGEST) != 0, (flags & HSW_GADGET_INDEPENDENT) != 0);
    if (rc != HSW_OK) { delete g; return rc; }
    *out = g;
    return HSW_OK;
} HSW_NO_UNWIND

void hsw_gadget_destroy(hsw_gadget *g) {
    if (!g) return;
    delete g->ctx;
    delete g;
}

static void fill_result(const hsw::AssignedHashResult &r, hsw_hash_result *o) {
    o->input_len = r.input_len;
    o->first_block = r.first_block;
    o->n_blocks = r.n_blocks;
    o->spread_cursor0 = r.spread_cursor0;
    o->num_round = r.num_round;
    o->target_round = r.target_round;
    std::memcpy(o->output_bytes, r.output_bytes, 32);
    o->prologue_cell = r.prologue_cell; o->block_cell = r.block_cell;
    o->epilogue_cell = r.epilogue_cell; o->end_cell = r.end_cell;
    o->prologue_lookup = r.prologue_lookup; o->block_lookup = r.block_lookup;
    o->epilogue_lookup = r.epilogue_lookup;
}

int hsw_gadget_digest_batch(hsw_gadget *g, size_t n, const uint8_t *const *inputs, const size_t *input_lens,
                            const size_t *precomputed_input_lens, hsw_hash_result *results) try {
    if (!g || !results) return HSW_ERR_INVALID_ARG;
    std::vector<hsw::AssignedHashResult> rs(n);
    const int rc = g->cfg.digest_batch(*g->ctx, n, inputs, input_lens, precomputed_input_lens, rs.data());
    if (rc != HSW_OK) return rc;
    for (size_t i = 0; i < n; i++) {
        fill_result(rs[i], &results[i]);
        g->results.push_back(std::move(rs[i]));
    }
    return HSW_OK;
} HSW_NO_UNWIND

int hsw_gadget_digest(hsw_gadget *g, const uint8_t *input, size_t input_len, size_t precomputed_input_len,
                      hsw_hash_result *result) try {
    return hsw_gadget_digest_batch(g, 1, &input, &input_len, &precomputed_input_len, result);
} HSW_NO_UNWIND

int hsw_gadget_streams(hsw_gadget *g, hsw_gadget_view *view) try {
    if (!g || !view) return HSW_ERR_INVALID_ARG;
    view->d_gate = g->ctx->d_gate;
    view->d_chip_dense = g->ctx->d_chip_dense;
    view->d_chip_spread = g->ctx->d_chip_spread;
    view->d_next_states = g->ctx->d_next_states;
    view->chip_col_stride = g->ctx->chip_col_stride;
    view->blocks_done = g->ctx->blocks_done;
    view->capacity_blocks = g->ctx->capacity_blocks;
    view->num_limb_sum = g->ctx->num_limb_sum;
    view->cur_hash_idx = g->cfg.cur_hash_idx;
    view->gate_cells = g->ctx->gate_cursor;
    view->gate_capacity = g->ctx->gate_capacity;
    view->d_lookup = g->ctx->d_lookup;
    view->lookup_cells = g->ctx->lookup_cursor;
    view->lookup_capacity = g->ctx->lookup_capacity;
    view->max_rows = g->ctx->max_rows;
    view->columns = g->ctx->columns;
    view->origin_column = g->ctx->origin_column;
    view->origin_row = g->ctx->origin_row;
    view->origin_lookups = g->ctx->origin_lookups;
    view->origin_zero_loaded = g->ctx->origin_zero_loaded ? 1u : 0u;
    view->reserved_ = 0;
    return HSW_OK;
} HSW_NO_UNWIND

int hsw_gadget_input_bytes(hsw_gadget *g, size_t hash_idx, uint8_t *out, size_t cap, size_t *len) try {
    if (!g || hash_idx >= g->results.size()) return HSW_ERR_INVALID_ARG;
    const std::vector<uint8_t> &b = g->results[hash_idx].input_bytes;
    if (len) *len = b.size();
    if (out) {
        if (cap < b.size()) return HSW_ERR_INVALID_ARG;
        if (!b.empty()) std::memcpy(out, b.data(), b.size());
    }
    return HSW_OK;
} HSW_NO_UNWIND

int hsw_gadget_result_cells(const hsw_gadget *g, size_t hash_idx, hsw_result_cells *out) try {
    if (!g || !out || hash_idx >= g->results.size()) return HSW_ERR_INVALID_ARG;
    if (!g->ctx->whole) return HSW_ERR_INVALID_ARG;                  // block-stream contexts hold no frame cells
    const hsw::AssignedHashResult &r = g->results[hash_idx];
    std::memset(out, 0, sizeof *out);
    out->input_len_cell = r.prologue_cell + hsw::frame::P_LEN;
    out->input_bytes_cell0 = r.prologue_cell + hsw::frame::P_BYTES;
    out->n_input_bytes = (uint64_t)r.n_blocks * 64;
    g->ctx->position(out->input_len_cell, &out->input_len_pos[0], &out->input_len_pos[1]);
    g->ctx->position(out->input_bytes_cell0, &out->input_bytes_pos0[0], &out->input_bytes_pos0[1]);
    for (uint32_t w = 0; w < 8; w++)
        for (uint32_t i = 0; i < 4; i++) {
            const uint64_t cell = r.epilogue_cell + (uint64_t)hsw::frame::E_STATE * (r.n_blocks + 1) +
                                  (uint64_t)hsw::frame::E_WORD * w + 5u * i;
            out->output_byte_cells[4 * w + i] = cell;
            g->ctx->position(cell, &out->output_byte_pos[4 * w + i][0], &out->output_byte_pos[4 * w + i][1]);
        }
    return HSW_OK;
} HSW_NO_UNWIND

int hsw_gadget_set_columns(hsw_gadget *g, uint64_t max_rows, uint64_t *n_columns) try {
    if (!g) return HSW_ERR_INVALID_ARG;
    int rc = hsw_engine_synchronize(g->ctx->engine);          // the image is reallocated: nothing may still write the old one
    if (rc == HSW_OK) rc = g->ctx->set_columns(g->cfg.max_variable_byte_sizes, g->cfg.is_input_range_check, max_rows);
    if (rc == HSW_OK && n_columns) *n_columns = g->ctx->columns;
    if (rc == HSW_OK) hsw::drop_region_tape_positions(g->tape);     // the codes are per stream cell; image positions follow the layout
    return rc;
} HSW_NO_UNWIND

int hsw_gadget_set_origin(hsw_gadget *g, uint64_t column, uint64_t row, int zero_cell_loaded,
                          uint64_t lookups_already_queued) try {
    if (!g) return HSW_ERR_INVALID_ARG;
    hsw::Context &c = *g->ctx;
    if (!c.whole || g->cfg.cur_hash_idx != 0) return HSW_ERR_INVALID_ARG;   // before the first digest of a synthesis pass
    int rc = hsw_engine_synchronize(c.engine);
    if (rc != HSW_OK) return rc;
    const uint64_t old[4] = {c.origin_column, c.origin_row, c.origin_zero_loaded ? 1u : 0u, c.origin_lookups};
    rc = c.set_origin(column, row, zero_cell_loaded != 0, lookups_already_queued);
    if (rc != HSW_OK) return rc;
    // the region tape (hsw_replay.cpp) numbers stream cells: only a zero cell that comes or goes changes it; a new
    // origin row moves the witnesses' image positions; column and queued lookups are offsets applied at delivery.
    // A prover that synthesizes the same circuit pass after pass keeps its tape.
    if (old[2] != (zero_cell_loaded ? 1u : 0u)) { hsw::free_region_tape(g->tape); g->tape = nullptr; }
    else if (old[1] != row) hsw::drop_region_tape_positions(g->tape);
    if (c.max_rows && (old[1] != row || old[2] != (zero_cell_loaded ? 1u : 0u))) {
        // the column breaks follow from where the stream starts: lay the image out again
        rc = c.set_columns(g->cfg.max_variable_byte_sizes, g->cfg.is_input_range_check, c.max_rows);
        if (rc != HSW_OK) {                                      // e.g. one column too many now: keep the old layout
            (void)c.set_origin(old[0], old[1], old[2] != 0, old[3]);
            return rc;
        }
    }
    return HSW_OK;
} HSW_NO_UNWIND

int hsw_gadget_reset(hsw_gadget *g) try {
    if (!g) return HSW_ERR_INVALID_ARG;
    const int rc = hsw_engine_synchronize(g->ctx->engine);
    if (rc != HSW_OK) return rc;
    hsw::Context &c = *g->ctx;
    c.blocks_done = 0;
    c.num_limb_sum = 0;                 // spread.rs:70-71
    c.gate_cursor = 0;
    c.lookup_cursor = c.origin_lookups; // the Context as the caller hands it over (hsw_gadget_set_origin)
    c.zero_loaded = c.origin_zero_loaded;
    c.batches.clear();
    g->cfg.cur_hash_idx = 0;            // lib.rs:66
    g->results.clear();
    return HSW_OK;
} HSW_NO_UNWIND

// Which allocations the chip columns live in, relative to the gate stream, is worth up to 8 % of an HBM-bound
// batch on MI355X and nothing in user space predicts it: try `candidates` allocations, timing the
// gadget's own batch (every digest an empty message) on each, and keep the fastest.
int hsw_gadget_place(hsw_gadget *g, unsigned candidates, float *ms_each, unsigned *kept) try {
    if (!g || candidates == 0 || candidates > 16) return HSW_ERR_INVALID_ARG;
    hsw::Context &c = *g->ctx;
    if (g->cfg.cur_hash_idx != 0 || c.blocks_done != 0) return HSW_ERR_INVALID_ARG;      // a fresh or reset gadget
    const size_t n = g->cfg.max_variable_byte_sizes.size();
    if (n == 0) return HSW_ERR_INVALID_ARG;
    hipStream_t stream = nullptr;
    int device = 0;
    hsw_engine_stream(c.engine, reinterpret_cast<void **>(&stream), &device);
    hsw::DeviceScopeG ds(device);
    if (!ds.ok) return HSW_ERR_NO_DEVICE;
    hsw_shape s;
    int rc = hsw_engine_shape(c.engine, &s);
    if (rc != HSW_OK) return rc;
    const size_t col_bytes = (size_t)s.num_advice_columns * (c.chip_col_stride ? c.chip_col_stride : 1) * HSW_CELL_BYTES;
    const uint8_t nothing = 0;
    std::vector<const uint8_t *> in(n, &nothing);
    std::vector<size_t> lens(n, 0), pres(n, 0);
    std::vector<hsw_hash_result> res(n);
    struct Cand { void *dense, *spread; float ms; };
    std::vector<Cand> cands;
    auto restore = [&](size_t keep) {                        // install candidate `keep`, free the others
        for (size_t k = 0; k < cands.size(); k++)
            if (k != keep) { (void)hipFree(cands[k].dense); (void)hipFree(cands[k].spread); }
        c.d_chip_dense = cands[keep].dense;
        c.d_chip_spread = cands[keep].spread;
    };
    for (unsigned k = 0; k < candidates; k++) {
        Cand cd{c.d_chip_dense, c.d_chip_spread, 0.f};
        if (k > 0) {
            cd.dense = cd.spread = nullptr;
            hipError_t he = hipMalloc(&cd.dense, col_bytes);
            if (he == hipSuccess) he = hipMalloc(&cd.spread, col_bytes);
            if (he == hipSuccess) he = hipMemset(cd.dense, 0, col_bytes);
            if (he == hipSuccess) he = hipMemset(cd.spread, 0, col_bytes);
            if (he != hipSuccess) {                               // out of memory: judge the candidates there are
                (void)hipFree(cd.dense); (void)hipFree(cd.spread);
                (void)hipGetLastError();
                break;
            }
        }
        cands.push_back(cd);
        c.d_chip_dense = cd.dense;
        c.d_chip_spread = cd.spread;
        float best = 0.f;
        for (int rep = 0; rep < 3 && rc == HSW_OK; rep++) {
            const auto t0 = std::chrono::steady_clock::now();
            rc = hsw_gadget_digest_batch(g, n, in.data(), lens.data(), pres.data(), res.data());
            const float ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (rc == HSW_OK && (rep == 1 || (rep > 1 && ms < best))) best = ms;
            const int rr = hsw_gadget_reset(g);
            if (rc == HSW_OK) rc = rr;
        }
        if (rc != HSW_OK) { restore(0); return rc; }
        cands.back().ms = best;
    }
    size_t keep = 0;
    for (size_t k = 1; k < cands.size(); k++)
        if (cands[k].ms < cands[keep].ms) keep = k;
    for (size_t k = 0; k < cands.size() && ms_each; k++) ms_each[k] = cands[k].ms;
    for (size_t k = cands.size(); k < candidates && ms_each; k++) ms_each[k] = 0.f;
    if (kept) *kept = (unsigned)keep;
    restore(keep);
    hipError_t he = hipMemsetAsync(c.d_chip_dense, 0, col_bytes, stream);       // as a fresh gadget has them
    if (he == hipSuccess) he = hipMemsetAsync(c.d_chip_spread, 0, col_bytes, stream);
    if (he == hipSuccess) he = hipStreamSynchronize(stream);
    return he == hipSuccess ? HSW_OK : HSW_ERR_HIP;
} HSW_NO_UNWIND

int hsw_gadget_download_region(hsw_gadget *g, const hsw_region_host *dst) try {
    if (!g || !dst) return HSW_ERR_INVALID_ARG;
    hsw::Context &c = *g->ctx;
    hipStream_t stream = nullptr;
    int device = 0;
    hsw_engine_stream(c.engine, reinterpret_cast<void **>(&stream), &device);
    hsw::DeviceScopeG ds(device);
    if (!ds.ok) return HSW_ERR_NO_DEVICE;
    const size_t cb = hsw_cell_bytes(c.repr_flags);
    hipError_t he = hipSuccess;
    auto copy = [&](void *h, const void *d, size_t cell0, size_t cells) {
        if (he == hipSuccess && cells)
            he = hipMemcpyAsync(static_cast<uint8_t *>(h) + cell0 * cb, static_cast<const uint8_t *>(d) + cell0 * cb,
                                cells * cb, hipMemcpyDeviceToHost, stream);
    };
    if (dst->gate) {
        if (c.whole && c.max_rows) {
            // used rows of column k: up to its break (max_rows - gap), the last column up to the cursor
            uint64_t last_col = 0, last_row = 0;
            if (c.gate_cursor) { c.position(c.gate_cursor - 1, &last_col, &last_row); last_row += 1; }
            last_col -= c.origin_column;                              // image column
            for (uint64_t k = 0; k <= last_col && c.gate_cursor; k++) {
                const uint64_t used = k < last_col ? c.max_rows - c.break_gap[k] : last_row;
                const uint64_t first = k == 0 ? c.origin_row : 0;    // rows above the origin are the caller's
                if (used > first) copy(dst->gate, c.d_gate, (size_t)(k * c.max_rows + first), (size_t)(used - first));
            }
        } else {
            const size_t cells = c.whole ? (size_t)c.gate_cursor : c.blocks_done * (size_t)c.shape.gate_cells_per_block;
            copy(dst->gate, c.d_gate, 0, cells);
        }
    }
    if (dst->lookup && c.d_lookup)
        copy(dst->lookup, c.d_lookup, (size_t)c.origin_lookups, (size_t)(c.lookup_cursor - c.origin_lookups));
    const uint32_t ncols = c.shape.num_advice_columns;
    const size_t rows = (size_t)((c.num_limb_sum + ncols - 1) / ncols);
    for (uint32_t k = 0; k < ncols; k++) {
        if (dst->chip_dense) copy(dst->chip_dense, c.d_chip_dense, k * c.chip_col_stride, rows);
        if (dst->chip_spread) copy(dst->chip_spread, c.d_chip_spread, k * c.chip_col_stride, rows);
    }
    if (he == hipSuccess) he = hipStreamSynchronize(stream);
    return he == hipSuccess ? HSW_OK : HSW_ERR_HIP;
} HSW_NO_UNWIND

int hsw_gadget_download_region_compact(hsw_gadget *g, hsw_region_compact *dst) try {
    if (!g || !dst) return HSW_ERR_INVALID_ARG;
    hsw::Context &c = *g->ctx;
    if (c.repr_flags != HSW_REPR_CANONICAL) return HSW_ERR_UNSUPPORTED;      // packs canonical 32-byte cells
    if (!dst->wide && dst->wide_cap) return HSW_ERR_INVALID_ARG;
    hipStream_t stream = nullptr;
    int device = 0;
    hsw_engine_stream(c.engine, reinterpret_cast<void **>(&stream), &device);
    hsw::DeviceScopeG ds(device);
    if (!ds.ok) return HSW_ERR_NO_DEVICE;
    const uint32_t ncols = c.shape.num_advice_columns;
    const size_t chip_cells = (size_t)ncols * (c.chip_col_stride ? c.chip_col_stride : 1);
    const size_t gate_cells = c.whole ? (c.max_rows ? (size_t)(c.max_rows * (c.break_cell.size() + 1)) : (size_t)c.gate_capacity)
                                      : c.capacity_blocks * (size_t)c.shape.gate_cells_per_block;
    hipError_t he = hipSuccess;
    if (!c.d_wide) {        // first use (or the geometry changed: set_columns / set_origin drop the staging):
                            // the 8-byte staging of every stream, the side list and its counter
        // wide cells: 4 ch negations per round (256 per block) + a few dozen per digest frame
        c.wide_cap = c.capacity_blocks * 256 + 128 * (c.init_capacity + 1) + 4 * c.capacity_blocks + 64;
        he = hipMalloc(&c.d_c_gate, (gate_cells ? gate_cells : 1) * 8);
        if (he == hipSuccess && c.d_lookup) he = hipMalloc(&c.d_c_lookup, (size_t)(c.lookup_capacity ? c.lookup_capacity : 1) * 8);
        if (he == hipSuccess) he = hipMalloc(&c.d_c_dense, chip_cells * 8);
        if (he == hipSuccess) he = hipMalloc(&c.d_c_spread, chip_cells * 8);
        if (he == hipSuccess) he = hipMalloc((void **)&c.d_wide_count, sizeof(uint32_t));
        if (he == hipSuccess) he = hipHostMalloc((void **)&c.hp_wide_count, sizeof(uint32_t), hipHostMallocDefault);
        if (he == hipSuccess) he = hipMalloc(&c.d_wide, c.wide_cap * 48);
        if (he != hipSuccess) { c.free_compact_staging(); return he == hipErrorOutOfMemory ? HSW_ERR_NOMEM : HSW_ERR_HIP; }
    }
    he = hipMemsetAsync(c.d_wide_count, 0, sizeof(uint32_t), stream);
    auto pack = [&](uint64_t *h, void *d8, const void *d32, uint64_t sid, size_t cell0, size_t cells) {
        if (he != hipSuccess || !cells || !h) return;
        he = hsw::launch_pack64(static_cast<const uint8_t *>(d32) + cell0 * 32, static_cast<uint8_t *>(d8) + cell0 * 8, cells, sid,
                                cell0, c.d_wide, (uint32_t)c.wide_cap, c.d_wide_count, stream);
        if (he == hipSuccess)
            he = hipMemcpyAsync(h + cell0, static_cast<uint8_t *>(d8) + cell0 * 8, cells * 8, hipMemcpyDeviceToHost, stream);
    };
    if (c.whole && c.max_rows) {
        // ONE pass over the image from (column 0, row 0) to the last assigned cell: the few unassigned rows at
        // the end of every column are zero on the device and travel as zeros (a launch and a copy per column
        // would cost more than the bytes they save)
        uint64_t last_col = 0, last_row = 0;
        if (c.gate_cursor) { c.position(c.gate_cursor - 1, &last_col, &last_row); last_row += 1; last_col -= c.origin_column; }
        // (from the origin row on: the rows above it in image column 0 are the caller's cells)
        pack(dst->gate, c.d_c_gate, c.d_gate, HSW_STREAM_GATE, (size_t)c.origin_row,
             c.gate_cursor ? (size_t)(last_col * c.max_rows + last_row - c.origin_row) : 0);
    } else {
        pack(dst->gate, c.d_c_gate, c.d_gate, HSW_STREAM_GATE, 0,
             c.whole ? (size_t)c.gate_cursor : c.blocks_done * (size_t)c.shape.gate_cells_per_block);
    }
    if (c.d_lookup)
        pack(dst->lookup, c.d_c_lookup, c.d_lookup, HSW_STREAM_LOOKUP, (size_t)c.origin_lookups, (size_t)(c.lookup_cursor - c.origin_lookups));
    const size_t rows = (size_t)((c.num_limb_sum + ncols - 1) / ncols);
    if (rows == c.chip_col_stride) {         // every column full: one pass per family
        pack(dst->chip_dense, c.d_c_dense, c.d_chip_dense, HSW_STREAM_CHIP_DENSE, 0, rows * ncols);
        pack(dst->chip_spread, c.d_c_spread, c.d_chip_spread, HSW_STREAM_CHIP_SPREAD, 0, rows * ncols);
    } else {
        for (uint32_t k = 0; k < ncols; k++) {
            pack(dst->chip_dense, c.d_c_dense, c.d_chip_dense, HSW_STREAM_CHIP_DENSE, k * c.chip_col_stride, rows);
            pack(dst->chip_spread, c.d_c_spread, c.d_chip_spread, HSW_STREAM_CHIP_SPREAD, k * c.chip_col_stride, rows);
        }
    }
    // the side list: the counter and as many entries as the caller has room for, in one pass of copies
    const size_t take = dst->wide_cap < c.wide_cap ? dst->wide_cap : c.wide_cap;
    if (he == hipSuccess) he = hipMemcpyAsync(c.hp_wide_count, c.d_wide_count, sizeof(uint32_t), hipMemcpyDeviceToHost, stream);
    if (he == hipSuccess && take) he = hipMemcpyAsync(dst->wide, c.d_wide, take * 48, hipMemcpyDeviceToHost, stream);
    if (he == hipSuccess) he = hipStreamSynchronize(stream);
    if (he != hipSuccess) return HSW_ERR_HIP;
    dst->n_wide = *c.hp_wide_count;
    if (dst->n_wide > take) return HSW_ERR_TOO_LARGE;         // n_wide says how many entries the region has
    return HSW_OK;
} HSW_NO_UNWIND

int hsw_region_widen(const uint64_t *compact, size_t n_cells, uint64_t stream_id, const hsw_wide_cell *wide, size_t n_wide,
                     void *cells32) try {
    if ((!compact || !cells32) && n_cells) return HSW_ERR_INVALID_ARG;
    if (!wide && n_wide) return HSW_ERR_INVALID_ARG;
    uint64_t *out = static_cast<uint64_t *>(cells32);
    for (size_t i = 0; i < n_cells; i++) { out[4 * i] = compact[i]; out[4 * i + 1] = 0; out[4 * i + 2] = 0; out[4 * i + 3] = 0; }
    for (size_t k = 0; k < n_wide; k++) {
        if (wide[k].stream != stream_id) continue;
        if (wide[k].index >= n_cells) return HSW_ERR_INVALID_ARG;
        std::memcpy(out + 4 * wide[k].index, wide[k].value, 32);
    }
    return HSW_OK;
} HSW_NO_UNWIND

int hsw_gadget_seek(hsw_gadget *g, size_t hash_idx) try {
    if (!g || hash_idx > g->cfg.max_variable_byte_sizes.size()) return HSW_ERR_INVALID_ARG;
    int rc = hsw_engine_synchronize(g->ctx->engine);
    if (rc != HSW_OK) return rc;
    hsw::Context &c = *g->ctx;
    size_t blocks = 0;
    uint64_t gate = 0, lookup = c.origin_lookups;
    for (size_t h = 0; h < hash_idx; h++) {
        const size_t b = g->cfg.max_variable_byte_sizes[h];
        blocks += b / 64;
        if (c.whole) {
            hsw_frame_shape fs;
            rc = hsw_frame_query(&c.shape, b, g->cfg.is_input_range_check ? 1 : 0, &fs);
            if (rc != HSW_OK) return rc;
            gate += fs.digest_cells + (c.independent || (h == 0 && !c.origin_zero_loaded) ? 1 : 0);   // + the Context's zero cell, loaded by digest #0
            lookup += fs.digest_lookups;
        }
    }
    c.blocks_done = blocks;
    c.num_limb_sum = (uint64_t)blocks * c.shape.limb_calls_per_block;       // spread.rs:228-231
    c.gate_cursor = gate;
    c.lookup_cursor = lookup;
    c.zero_loaded = c.origin_zero_loaded || hash_idx > 0;
    g->cfg.cur_hash_idx = hash_idx;
    c.batches.clear();
    g->results.clear();
    g->results.resize(hash_idx);        // keeps hash_idx -> result indexing of hsw_gadget_input_bytes
    return HSW_OK;
} HSW_NO_UNWIND

int hsw_gadget_verify(hsw_gadget *g, hsw_verify_report *report) try {
    if (!g || !report) return HSW_ERR_INVALID_ARG;
    std::memset(report, 0, sizeof *report);
    hsw::Context &c = *g->ctx;
    if (c.repr_flags & HSW_REPR_COMPACT64) return HSW_ERR_UNSUPPORTED;         // 32-byte cells only
    const size_t G = c.shape.gate_cells_per_block, cb = HSW_CELL_BYTES;
    const uint32_t ncols = c.shape.num_advice_columns;
    auto merge = [&](const hsw_verify_report &r) {
        if (r.violations && !report->violations) {
            report->first_block = r.first_block; report->first_cell = r.first_cell; report->first_class = r.first_class;
        }
        report->violations += r.violations; report->checks += r.checks; report->kernel_ms += r.kernel_ms;
    };
    hsw_pack_plan abs_plan{};
    abs_plan.n_breaks = (uint32_t)c.break_cell.size();
    for (size_t k = 0; k < c.break_cell.size(); k++) { abs_plan.break_cell[k] = c.break_cell[k]; abs_plan.break_gap[k] = c.break_gap[k]; }
    for (const hsw::Context::BatchRecord &b : c.batches) {
        const uint8_t *in_blocks = b.inputs_in_pinned ? c.dp_blocks : c.d_blocks;
        const uint32_t *in_pre = b.inputs_in_pinned ? c.dp_pre : c.d_pre_states;
        if (!c.whole) {
            hsw_witness_args a{};
            a.d_blocks = in_blocks + 64 * b.first_block; a.d_pre_states = in_pre + 8 * b.first_block; a.n_blocks = b.n_blocks;
            a.spread_cursor0 = (uint64_t)b.first_block * c.shape.limb_calls_per_block;
            const uint64_t row_shift = a.spread_cursor0 / ncols;
            a.d_gate = static_cast<uint8_t *>(c.d_gate) + b.first_block * G * cb;
            a.d_chip_dense = static_cast<uint8_t *>(c.d_chip_dense) + (size_t)row_shift * cb;
            a.d_chip_spread = static_cast<uint8_t *>(c.d_chip_spread) + (size_t)row_shift * cb;
            a.chip_col_stride = c.chip_col_stride;
            a.d_next_states = c.d_next_states + 8 * b.first_block;
            a.flags = b.repr_flags;
            hsw_verify_report r;
            const int rc = hsw_verify_blocks(c.engine, &a, &r);
            if (rc != HSW_OK) return rc;
            merge(r);
            continue;
        }
        size_t ob = 0;
        for (size_t i = 0; i < b.n_digests;) {                       // runs of equally sized digests, as generated
            const hsw::AssignedHashResult &r0 = g->results[b.first_digest + i];
            size_t j = i + 1;
            while (j < b.n_digests && g->results[b.first_digest + j].n_blocks == r0.n_blocks) j++;
            const size_t nb = r0.n_blocks, run_blocks = nb * (j - i), fb = b.first_block + ob;
            hsw_frame_shape fs;
            int rc = hsw_frame_query(&c.shape, nb * 64, g->cfg.is_input_range_check ? 1 : 0, &fs);
            if (rc != HSW_OK) return rc;
            hsw_witness_args a{};
            a.d_blocks = in_blocks + 64 * fb; a.d_pre_states = in_pre + 8 * fb; a.n_blocks = run_blocks;
            a.spread_cursor0 = (uint64_t)fb * c.shape.limb_calls_per_block;
            const uint64_t row_shift = a.spread_cursor0 / ncols;
            a.d_gate = static_cast<uint8_t *>(c.gate_stream()) + (size_t)r0.block_cell * cb;
            a.d_chip_dense = static_cast<uint8_t *>(c.d_chip_dense) + (size_t)row_shift * cb;
            a.d_chip_spread = static_cast<uint8_t *>(c.d_chip_spread) + (size_t)row_shift * cb;
            a.chip_col_stride = c.chip_col_stride;
            a.d_next_states = c.d_next_states + 8 * fb;
            a.d_lookup = static_cast<uint8_t *>(c.d_lookup) + (size_t)r0.block_lookup * cb;
            a.frame_every = nb; a.frame_cells = fs.epilogue_cells + fs.prologue_cells + (c.independent ? 1u : 0u);
            a.frame_lookups = fs.epilogue_lookups + fs.prologue_lookups;
            a.flags = b.repr_flags;
            hsw_pack_plan rel{};
            if (c.max_rows) {
                rel.n_breaks = abs_plan.n_breaks;
                for (uint32_t k = 0; k < rel.n_breaks; k++) {
                    rel.break_cell[k] = abs_plan.break_cell[k] > r0.block_cell ? abs_plan.break_cell[k] - r0.block_cell : 0;
                    rel.break_gap[k] = abs_plan.break_gap[k];
                }
                a.pack = &rel;
            }
            hsw_verify_report r;
            rc = hsw_verify_blocks(c.engine, &a, &r);
            if (rc != HSW_OK) return rc;
            merge(r);
            std::vector<hsw_frame_desc> descs(j - i);
            size_t blk = fb;
            for (size_t k = i; k < j; k++) {
                const hsw::AssignedHashResult &rk = g->results[b.first_digest + k];
                hsw_frame_desc &d = descs[k - i];
                d.input_len = rk.input_len; d.first_block = blk; d.n_blocks = (uint32_t)rk.n_blocks;
                d.num_round = (uint32_t)rk.num_round; d.precomputed_round = (uint32_t)(rk.num_round - rk.target_round);
                d.is_input_range_check = g->cfg.is_input_range_check ? 1u : 0u;
                d.prologue_cell = rk.prologue_cell; d.epilogue_cell = rk.epilogue_cell;
                d.prologue_lookup = rk.prologue_lookup; d.epilogue_lookup = rk.epilogue_lookup;
                d.zero_cell = rk.block_cell == rk.prologue_cell + fs.prologue_cells + 1 ? rk.block_cell - 1 : ~0ull;
                blk += rk.n_blocks;
            }
            rc = hsw_verify_frames(c.engine, descs.data(), descs.size(), in_blocks, in_pre, c.d_next_states, c.gate_stream(),
                                   c.d_lookup, c.max_rows ? &abs_plan : nullptr, b.repr_flags, &r);
            if (rc != HSW_OK) return rc;
            merge(r);
            ob += run_blocks;
            i = j;
        }
    }
    return HSW_OK;
} HSW_NO_UNWIND

int hsw_gadget_cell_position(const hsw_gadget *g, uint64_t cell, uint64_t *column, uint64_t *row) try {
    if (!g) return HSW_ERR_INVALID_ARG;
    g->ctx->position(cell, column, row);
    return HSW_OK;
} HSW_NO_UNWIND

int hsw_gadget_set_repr(hsw_gadget *g, uint32_t repr) try {
    if (!g || (repr & ~HSW_REPR_MASK) || repr == HSW_REPR_MASK) return HSW_ERR_INVALID_ARG;
    if (g->ctx->whole && (repr & HSW_REPR_COMPACT64)) return HSW_ERR_UNSUPPORTED;   // frames hold full-width cells
    if (g->ctx->blocks_done != 0 && hsw_cell_bytes(repr) != hsw_cell_bytes(g->ctx->repr_flags))
        return HSW_ERR_INVALID_ARG;               // the cell size of a context's streams cannot change midway
    g->ctx->repr_flags = repr;
    return HSW_OK;
} HSW_NO_UNWIND

}  // extern "C"
