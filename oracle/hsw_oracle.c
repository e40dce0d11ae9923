/*
 * hsw_oracle.c -- CPU restatement of the halo2-dynamic-sha256 witness path.
 * TEST INFRASTRUCTURE ONLY (see hsw_oracle.h for scope, assumptions, pinning).
 *
 * Every function cites the reference lines it follows; statement order inside
 * each function is the reference's, because the order of gate calls *is* the
 * order of the gate-cell stream.
 */
#include "hsw_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;

/* ------------------------------------------------------------------ field */
/* BN254 scalar field modulus (halo2curves::bn256::Fr), little-endian limbs. */
static const ofe_t FR_P = {{0x43e1f593f0000001ULL, 0x2833e84879b97091ULL,
                            0xb85045b68181585dULL, 0x30644e72e131a029ULL}};

static inline ofe_t fe_u64(uint64_t v) { ofe_t r = {{v, 0, 0, 0}}; return r; }
static inline int fe_is_u64(const ofe_t *a) { return (a->l[1] | a->l[2] | a->l[3]) == 0; }
static inline int fe_eq(const ofe_t *a, const ofe_t *b) {
    return a->l[0] == b->l[0] && a->l[1] == b->l[1] && a->l[2] == b->l[2] && a->l[3] == b->l[3];
}
static inline int fe_is_zero(const ofe_t *a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; }
static inline int fe_geq(const ofe_t *a, const ofe_t *b) {
    for (int i = 3; i >= 0; i--) {
        if (a->l[i] > b->l[i]) return 1;
        if (a->l[i] < b->l[i]) return 0;
    }
    return 1;
}
static inline ofe_t fe_sub_raw(const ofe_t *a, const ofe_t *b) {
    ofe_t r; u128 borrow = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)a->l[i] - b->l[i] - borrow;
        r.l[i] = (uint64_t)d;
        borrow = (d >> 64) & 1;
    }
    return r;
}
static ofe_t fe_add(const ofe_t *a, const ofe_t *b) {
    ofe_t r; u128 carry = 0;
    for (int i = 0; i < 4; i++) {
        u128 s = (u128)a->l[i] + b->l[i] + carry;
        r.l[i] = (uint64_t)s;
        carry = s >> 64;
    }
    /* both inputs < p < 2^254, so no carry out of 256 bits */
    if (fe_geq(&r, &FR_P)) r = fe_sub_raw(&r, &FR_P);
    return r;
}
static ofe_t fe_neg(const ofe_t *a) {
    if (fe_is_zero(a)) return *a;
    return fe_sub_raw(&FR_P, a);
}
/* 512-bit by p reduction, bit-serial.  Only reached if an operand of a gate
 * multiplication does not fit 64 bits, which the gadget never does; kept so
 * the oracle is a field-arithmetic restatement, not a u64 one. */
static ofe_t fe_mul_slow(const ofe_t *a, const ofe_t *b) {
    uint64_t prod[8] = {0};
    for (int i = 0; i < 4; i++) {
        u128 carry = 0;
        for (int j = 0; j < 4; j++) {
            u128 t = (u128)a->l[i] * b->l[j] + prod[i + j] + carry;
            prod[i + j] = (uint64_t)t;
            carry = t >> 64;
        }
        prod[i + 4] = (uint64_t)carry;
    }
    ofe_t r = {{0, 0, 0, 0}};
    for (int bit = 511; bit >= 0; bit--) {
        /* r = 2r + bit  (mod p); r < p < 2^254 so 2r+1 < 2^256 */
        uint64_t top = 0;
        for (int i = 0; i < 4; i++) {
            uint64_t nt = r.l[i] >> 63;
            r.l[i] = (r.l[i] << 1) | top;
            top = nt;
        }
        r.l[0] |= (prod[bit / 64] >> (bit % 64)) & 1;
        if (fe_geq(&r, &FR_P)) r = fe_sub_raw(&r, &FR_P);
    }
    return r;
}
static ofe_t fe_mul(const ofe_t *a, const ofe_t *b) {
    if (fe_is_u64(a) && fe_is_u64(b)) {
        u128 t = (u128)a->l[0] * b->l[0];      /* < 2^128 < p: already reduced */
        ofe_t r = {{(uint64_t)t, (uint64_t)(t >> 64), 0, 0}};
        return r;
    }
    return fe_mul_slow(a, b);
}
static ofe_t fe_sub(const ofe_t *a, const ofe_t *b) {
    ofe_t nb = fe_neg(b);
    return fe_add(a, &nb);
}
/* a^(p-2): Field::invert for a != 0 (is_zero's witness, A4) */
static ofe_t fe_inv(const ofe_t *a) {
    ofe_t e = FR_P; e.l[0] -= 2;           /* p - 2, no borrow */
    ofe_t r = fe_u64(1), base = *a;
    for (int bit = 0; bit < 254; bit++) {
        if ((e.l[bit / 64] >> (bit % 64)) & 1) r = fe_mul(&r, &base);
        base = fe_mul(&base, &base);
    }
    return r;
}
/* halo2-base ScalarField::get_lower_32 / get_lower_64: low bits of the
 * canonical representation (call sites compression.rs:224,228,274,278,814,818). */
static inline uint32_t fe_lower_32(const ofe_t *a) { return (uint32_t)a->l[0]; }
static inline uint64_t fe_lower_64(const ofe_t *a) { return a->l[0]; }

/* ---------------------------------------------------------------- context */
struct oracle_ctx {
    int num_bits_lookup;          /* spread.rs:24 */
    int num_advice_columns;       /* spread.rs:25 */
    uint64_t num_limb_sum;        /* spread.rs:26 */
    uint64_t row_offset;          /* spread.rs:27 */
    int check;
    int internals;                /* emit halo2-base's own range_check cells into the gate stream (A3) */
    int lookup_bits;              /* RangeConfig lookup_bits (16 at every reference config) */
    ofe_t *lookup; size_t lookup_cap, lookup_len;   /* the lookup-advice column finalize() fills (A3) */
    ofe_t *gate; size_t gate_cap, gate_len;
    uint8_t *kinds; size_t kinds_cap; int cur_kind;   /* optional per-cell tag stream */
    ofe_t *dense, *spread; size_t col_stride; uint64_t row_base;
    oracle_stats_t st;
    int failed; char msg[256];
    size_t block_start;            /* gate_len at the start of the current block */
    oracle_constraints_t *rec;     /* constraint-structure recorder (next block only), or NULL */
    /* whole-digest mode (oracle_digest_cells): cell ids are absolute stream indices, the
     * Context's zero cell is a stream cell, and the recorder stays attached across blocks */
    int whole;
    int zero_loaded; int64_t zero_cell;       /* Context.zero_cell (A4-iii) */
    uint8_t *call_lens; size_t call_cap, n_calls;     /* assign_region call lengths, in order */
    uint64_t *gate_rows; size_t rows_cap, n_rows;     /* stream index of every enabled gate row */
};

static void ofail(oracle_ctx *c, const char *what, uint64_t a, uint64_t b) {
    if (!c->failed) {
        c->failed = 1;
        snprintf(c->msg, sizeof c->msg, "%s (0x%llx vs 0x%llx) at gate cell %zu", what,
                 (unsigned long long)a, (unsigned long long)b, c->gate_len);
    }
}

oracle_ctx *oracle_create(int num_bits_lookup, int num_advice_columns, int check) {
    /* spread.rs:37  debug_assert_eq!(16 % num_bits_lookup, 0) */
    if (num_bits_lookup <= 0 || num_bits_lookup > 16 || 16 % num_bits_lookup != 0) return NULL;
    if (num_advice_columns <= 0) return NULL;
    oracle_ctx *c = (oracle_ctx *)calloc(1, sizeof *c);
    if (!c) return NULL;
    c->num_bits_lookup = num_bits_lookup;
    c->num_advice_columns = num_advice_columns;
    c->check = check;
    c->lookup_bits = 16;          /* lib.rs:493, benches/digest.rs:108 LOOKUP_BITS */
    return c;
}
void oracle_set_internals(oracle_ctx *c, int on) { c->internals = on; }
void oracle_set_lookup_output(oracle_ctx *c, ofe_t *lookup, size_t cap) {
    c->lookup = lookup; c->lookup_cap = cap; c->lookup_len = 0;
}
size_t oracle_lookup_len(const oracle_ctx *c) { return c->lookup_len; }
void oracle_destroy(oracle_ctx *c) { free(c); }
void oracle_set_outputs(oracle_ctx *c, ofe_t *gate, size_t gate_cap, ofe_t *dense,
                        ofe_t *spread, size_t col_stride, uint64_t row_base) {
    c->gate = gate; c->gate_cap = gate_cap; c->gate_len = 0;
    c->dense = dense; c->spread = spread; c->col_stride = col_stride; c->row_base = row_base;
}
void oracle_set_kinds(oracle_ctx *c, uint8_t *kinds, size_t cap) { c->kinds = kinds; c->kinds_cap = cap; }
void oracle_record_constraints(oracle_ctx *c, oracle_constraints_t *rec) {
    if (rec) { rec->n_eq = rec->n_const = rec->n_range = rec->n_chip = rec->n_lookup = 0; }
    c->rec = rec;
}
void oracle_set_cursor(oracle_ctx *c, uint64_t n) {
    c->num_limb_sum = n;
    c->row_offset = n / (uint64_t)c->num_advice_columns;   /* spread.rs:228-231 closed form */
}
uint64_t oracle_get_cursor(const oracle_ctx *c) { return c->num_limb_sum; }
size_t oracle_gate_len(const oracle_ctx *c) { return c->gate_len; }
void oracle_get_stats(const oracle_ctx *c, oracle_stats_t *out) { *out = c->st; }
int oracle_failed(const oracle_ctx *c, const char **msg) {
    if (msg) *msg = c->msg;
    return c->failed;
}

/* ------------------------------------------------- halo2-base gate mirror */
/* An AssignedValue: its value and the cell it lives in.  Cells of the gate
 * stream are numbered block-relative (0..G-1); cells outside the stream have
 * negative ids (ORACLE_CELL_*).  A QuantumCell::Constant is an av_t whose cell is
 * CELL_CONST.  The cell ids are what lets the oracle record the reference's
 * constraint *structure* (which cells halo2 copy-constrains / fixes), next to
 * the values. */
typedef struct { ofe_t v; int64_t cell; } av_t;
#define CELL_CONST   INT64_MIN
static inline av_t K(uint64_t x) { av_t r; r.v = fe_u64(x); r.cell = CELL_CONST; return r; }
static inline av_t ext_cell(uint64_t x, int64_t id) { av_t r; r.v = fe_u64(x); r.cell = id; return r; }

static void rec_eq(oracle_ctx *c, int64_t a, int64_t b) {
    if (!c->rec) return;
    oracle_constraints_t *r = c->rec;
    if (r->eq && r->n_eq < r->eq_cap) { r->eq[2 * r->n_eq] = a; r->eq[2 * r->n_eq + 1] = b; }
    r->n_eq++;
}
static void rec_const(oracle_ctx *c, int64_t cell_id, uint64_t value) {
    if (!c->rec) return;
    oracle_constraints_t *r = c->rec;
    if (r->konst && r->n_const < r->const_cap) { r->konst[2 * r->n_const] = cell_id; r->konst[2 * r->n_const + 1] = (int64_t)value; }
    r->n_const++;
}
static void rec_range(oracle_ctx *c, int64_t cell_id, int bits) {
    if (!c->rec) return;
    oracle_constraints_t *r = c->rec;
    if (r->range && r->n_range < r->range_cap) { r->range[2 * r->n_range] = cell_id; r->range[2 * r->n_range + 1] = bits; }
    r->n_range++;
}

/* One FlexGateConfig::assign_region call of `len` cells starting at the current
 * stream position, with the gate selector enabled at the given offsets
 * (halo2-base `gate_offsets`; -1 = unused).  Feeds the call-length tape and the
 * list of gate rows when they are attached. */
static void region(oracle_ctx *c, int len, int g0, int g1) {
    if (c->call_lens && c->n_calls < c->call_cap) c->call_lens[c->n_calls] = (uint8_t)len;
    c->n_calls++;
    const int g[2] = {g0, g1};
    for (int i = 0; i < 2; i++) {
        if (g[i] < 0) continue;
        if (c->gate_rows && c->n_rows < c->rows_cap) c->gate_rows[c->n_rows] = (uint64_t)c->gate_len + (uint64_t)g[i];
        c->n_rows++;
    }
}
static void region_row(oracle_ctx *c, size_t at) {      /* a further gate row of the region just opened */
    if (c->gate_rows && c->n_rows < c->rows_cap) c->gate_rows[c->n_rows] = (uint64_t)at;
    c->n_rows++;
}

/* One advice cell of the gate stream holding QuantumCell q.  Existing(x): the new
 * cell is copy-constrained to x's cell; Constant(k): fixed to k; Witness: free.
 * Returns the new cell's block-relative index. */
static inline int64_t cell(oracle_ctx *c, const av_t *q, int is_witness) {
    const int64_t idx = (int64_t)(c->gate_len - c->block_start);
    if (c->kinds && c->gate_len < c->kinds_cap) c->kinds[c->gate_len] = (uint8_t)c->cur_kind;
    if (c->cur_kind) c->cur_kind++;       /* 1..4 = position inside a 4-cell gate row */
    if (c->gate) {
        if (c->gate_len < c->gate_cap) c->gate[c->gate_len] = q->v;
        else ofail(c, "gate buffer overflow", c->gate_len, c->gate_cap);
    }
    c->gate_len++;
    c->st.gate_cells++;
    if (!is_witness) {
        if (q->cell == CELL_CONST) {
            if (fe_is_u64(&q->v)) rec_const(c, idx, q->v.l[0]);
            else { ofe_t m = fe_neg(&q->v); rec_const(c, idx, (uint64_t)(-(int64_t)m.l[0])); }   /* p - k, k small: stored as -k */
        } else rec_eq(c, idx, q->cell);
    }
    return idx;
}
static inline av_t witness_cell(oracle_ctx *c, ofe_t v) {
    av_t r; r.v = v; r.cell = 0;
    r.cell = cell(c, &r, 1);
    return r;
}
/* GateInstructions::load_witness -> [v] */
static av_t g_load_witness(oracle_ctx *c, ofe_t v) {
    c->st.load_witness++;
    c->cur_kind = 0;
    region(c, 1, -1, -1);
    return witness_cell(c, v);
}
/* GateInstructions::load_zero: cached in the Context (assumption A2): one cell
 * outside the stream, fixed to 0. */
static av_t g_load_zero(oracle_ctx *c) {
    c->st.load_zero++;
    if (!c->whole) return ext_cell(0, ORACLE_CELL_ZERO);
    /* whole-digest mode, ASSUMPTION A4-iii (halo2-lib v0.2.x load_zero): the first call in a
     * Context assigns [Constant(0)] as a one-cell region and caches it in ctx.zero_cell */
    if (!c->zero_loaded) {
        av_t zero = K(0);
        c->cur_kind = 0;
        region(c, 1, -1, -1);
        c->zero_cell = cell(c, &zero, 0);
        c->zero_loaded = 1;
    }
    return ext_cell(0, c->zero_cell);
}
/* GateInstructions::add -> [a, b, 1, a+b] */
static av_t g_add(oracle_ctx *c, av_t a, av_t b) {
    c->st.add++;
    av_t one = K(1);
    c->cur_kind = 1;
    region(c, 4, 0, -1);
    cell(c, &a, 0); cell(c, &b, 0); cell(c, &one, 0);
    return witness_cell(c, fe_add(&a.v, &b.v));
}
/* GateInstructions::neg -> [a, -a, 1, 0] */
static av_t g_neg(oracle_ctx *c, av_t a) {
    c->st.neg++;
    av_t one = K(1), zero = K(0);
    c->cur_kind = 1;
    region(c, 4, 0, -1);
    cell(c, &a, 0);
    av_t out = witness_cell(c, fe_neg(&a.v));
    cell(c, &one, 0); cell(c, &zero, 0);
    return out;
}
/* GateInstructions::mul_add(a, b, c) = a*b + c -> [c, a, b, out] */
static av_t g_mul_add(oracle_ctx *c, av_t a, av_t b, av_t cc) {
    c->st.mul_add++;
    ofe_t ab = fe_mul(&a.v, &b.v);
    c->cur_kind = 1;
    region(c, 4, 0, -1);
    cell(c, &cc, 0); cell(c, &a, 0); cell(c, &b, 0);
    return witness_cell(c, fe_add(&ab, &cc.v));
}
/* GateInstructions::assert_equal: copy constraint only. */
static void g_assert_equal(oracle_ctx *c, av_t a, av_t b) {
    c->st.assert_equal++;
    if (c->check && !fe_eq(&a.v, &b.v)) ofail(c, "assert_equal violated", a.v.l[0], b.v.l[0]);
    rec_eq(c, a.cell, b.cell);
}
/* RangeConfig::enable_lookup: the cell is queued in ctx.cells_to_lookup and
 * copied into the lookup-advice column by finalize() in queue order (A3). */
static void r_enable_lookup(oracle_ctx *c, const av_t *v) {
    if (c->check && (!fe_is_u64(&v->v) || (v->v.l[0] >> c->lookup_bits) != 0))
        ofail(c, "lookup value outside the range table", v->v.l[0], (uint64_t)c->lookup_bits);
    if (c->lookup) {
        if (c->lookup_len < c->lookup_cap) c->lookup[c->lookup_len] = v->v;
        else ofail(c, "lookup buffer overflow", c->lookup_len, c->lookup_cap);
    }
    if (c->rec) {   /* lookup-column entry j is a copy of this cell */
        oracle_constraints_t *r = c->rec;
        if (r->lookup_src && r->n_lookup < r->lookup_cap) r->lookup_src[r->n_lookup] = v->cell;
        r->n_lookup++;
    }
    c->lookup_len++;
}
/* RangeInstructions::range_check(a, bits): constraint a < 2^bits.
 * halo2-lib v0.2.x range_check_simple (ASSUMPTION A3, source absent):
 *   k = ceil(bits / lookup_bits) limbs of lookup_bits bits;
 *   k == 1: no new cell, `a` itself is looked up;
 *   k  > 1: inner_product_left(limbs, [1, B, B^2..]) lays out
 *           [limb0, limb1, B, s1, limb2, B^2, s2, ...] (1 + 3(k-1) cells, overlapping
 *           gate rows), s_{k-1} copy-constrained to `a`; every limb is looked up;
 *   bits % lookup_bits = r > 1: one more row [0, last, 2^(lookup_bits-r), last*2^(..)]
 *           whose output is looked up (r == 1: the last limb is constrained boolean;
 *           not reachable from this gadget).
 * The cells are emitted only with oracle_set_internals(1); the lookup queue, the
 * recorded range bound and the range self-check always run. */
static void r_range_check_limbs(oracle_ctx *c, av_t a, int bits, av_t *limbs_out) {
    if (bits == 16) c->st.range_check16++;
    else if (bits == 32) c->st.range_check32++;
    else c->st.range_check_other++;
    if (c->check) {
        if (!fe_is_u64(&a.v) || (bits < 64 && (a.v.l[0] >> bits) != 0))
            ofail(c, "range_check violated", a.v.l[0], (uint64_t)bits);
    }
    rec_range(c, a.cell, bits);
    const int lb = c->lookup_bits;
    const int k = (bits + lb - 1) / lb, rem = bits % lb;
    av_t last = a;
    if (k == 1) {
        r_enable_lookup(c, &a);
        if (limbs_out) limbs_out[0] = a;
    } else {
        av_t limbs[8];
        for (int i = 0; i < k && i < 8; i++) {
            limbs[i].v = fe_u64((a.v.l[0] >> (lb * i)) & ((1ULL << lb) - 1));
            limbs[i].cell = ORACLE_CELL_HIDDEN;        /* halo2-base's own witness; a stream cell only with internals */
        }
        if (c->internals) {
            c->cur_kind = 1;                                   /* rows overlap: [s, a, b, s'] */
            region(c, 1 + 3 * (k - 1), 0, -1);
            const size_t at = c->gate_len;
            for (int i = 2; i < k; i++) region_row(c, at + 3 * (size_t)(i - 1));
            limbs[0] = witness_cell(c, limbs[0].v);
            av_t sum = limbs[0];
            for (int i = 1; i < k; i++) {
                av_t base = K(1ULL << (lb * i));
                ofe_t prod = fe_mul(&limbs[i].v, &base.v);
                c->cur_kind = 2;
                limbs[i] = witness_cell(c, limbs[i].v);
                cell(c, &base, 0);
                sum = witness_cell(c, fe_add(&sum.v, &prod));
            }
            if (c->check && !fe_eq(&sum.v, &a.v)) ofail(c, "range_check decomposition", sum.v.l[0], a.v.l[0]);
            rec_eq(c, sum.cell, a.cell);                       /* constrain_equal(a, acc) */
        }
        for (int i = 0; i < k; i++) r_enable_lookup(c, &limbs[i]);
        if (limbs_out) for (int i = 0; i < k && i < 8; i++) limbs_out[i] = limbs[i];
        last = limbs[k - 1];
    }
    if (rem > 1) {
        av_t zero = K(0), mult = K(1ULL << (lb - rem));
        av_t out; out.v = fe_mul(&last.v, &mult.v); out.cell = ORACLE_CELL_HIDDEN;
        if (c->internals) {
            c->cur_kind = 1;
            region(c, 4, 0, -1);
            cell(c, &zero, 0); cell(c, &last, 0); cell(c, &mult, 0);
            out = witness_cell(c, out.v);
        }
        r_enable_lookup(c, &out);
    }
}
static void r_range_check(oracle_ctx *c, av_t a, int bits) { r_range_check_limbs(c, a, bits, NULL); }

/* ------------------------------------------- halo2-base calls of lib.rs::digest
 * ASSUMPTION A4 (halo2-lib v0.2.x flex_gate.rs / range.rs, Vertical strategy; source
 * absent from the reference tree, so UNPINNED like A1-A3).  Only `digest`'s own
 * prologue / epilogue uses these (lib.rs:122-178, 294-341):
 *   mul(a, b)          -> [0, a, b, a*b]
 *   sub(a, b)          -> [a-b, b, 1, a]                       (returns cell 0)
 *   is_zero(a)         -> [z, a, inv, 1, 0, a, z, 0]           gates at 0 and 4, cell 6 = cell 0;
 *                         z = (a == 0), inv = a^-1 (1 if a == 0):  z + a*inv = 1,  0 + a*z = 0
 *   is_equal(a, b)     -> [a-b, 1, b, a] then is_zero(cell 0)
 *   select(a, b, sel)  -> [a-b, 1, b, a, b, sel, a-b, out]     gates at 0 and 4, cell 6 = cell 0;
 *                         out = (a-b)*sel + b
 *   is_less_than(a, b, n): pb = ceil(n / lookup_bits) * lookup_bits
 *                      -> [a+2^pb-b, b, 1, a+2^pb, -2^pb, 1, a] gates at 0 and 3,
 *                         range_check(cell 0, pb + lookup_bits), is_zero(top limb)
 *   is_less_than_safe(a, b: u64): n = bit_length(b) rounded up to a multiple of
 *                         lookup_bits; range_check(a, n); is_less_than(a, Constant(b), n)
 *   assert_is_const(a, k): a's cell fixed to k, no new cell. */
static av_t g_mul(oracle_ctx *c, av_t a, av_t b) {
    av_t zero = K(0);
    c->cur_kind = 1;
    region(c, 4, 0, -1);
    cell(c, &zero, 0); cell(c, &a, 0); cell(c, &b, 0);
    return witness_cell(c, fe_mul(&a.v, &b.v));
}
static av_t g_sub(oracle_ctx *c, av_t a, av_t b) {
    av_t one = K(1);
    c->cur_kind = 1;
    region(c, 4, 0, -1);
    av_t out = witness_cell(c, fe_sub(&a.v, &b.v));
    cell(c, &b, 0); cell(c, &one, 0); cell(c, &a, 0);
    return out;
}
static av_t g_is_zero(oracle_ctx *c, av_t a) {
    const int z = fe_is_zero(&a.v);
    av_t one = K(1), zero = K(0);
    c->cur_kind = 1;
    region(c, 8, 0, 4);
    av_t is_zero = witness_cell(c, fe_u64(z ? 1 : 0));
    cell(c, &a, 0);
    witness_cell(c, z ? fe_u64(1) : fe_inv(&a.v));
    cell(c, &one, 0);
    c->cur_kind = 1;
    cell(c, &zero, 0); cell(c, &a, 0); cell(c, &is_zero, 0); cell(c, &zero, 0);
    return is_zero;
}
static av_t g_is_equal(oracle_ctx *c, av_t a, av_t b) {
    av_t one = K(1);
    c->cur_kind = 1;
    region(c, 4, 0, -1);
    av_t diff = witness_cell(c, fe_sub(&a.v, &b.v));
    cell(c, &one, 0); cell(c, &b, 0); cell(c, &a, 0);
    return g_is_zero(c, diff);
}
static av_t g_select(oracle_ctx *c, av_t a, av_t b, av_t sel) {
    av_t one = K(1);
    ofe_t diff_v = fe_sub(&a.v, &b.v);
    ofe_t prod = fe_mul(&diff_v, &sel.v);
    c->cur_kind = 1;
    region(c, 8, 0, 4);
    av_t diff = witness_cell(c, diff_v);
    cell(c, &one, 0); cell(c, &b, 0); cell(c, &a, 0);
    c->cur_kind = 1;
    cell(c, &b, 0); cell(c, &sel, 0); cell(c, &diff, 0);
    return witness_cell(c, fe_add(&prod, &b.v));
}
static void g_assert_is_const(oracle_ctx *c, av_t a, uint64_t k) {
    if (c->check && !(fe_is_u64(&a.v) && a.v.l[0] == k)) ofail(c, "assert_is_const violated", a.v.l[0], k);
    rec_const(c, a.cell, k);
}
static av_t r_is_less_than(oracle_ctx *c, av_t a, av_t b, int num_bits) {
    const int lb = c->lookup_bits;
    const int k = (num_bits + lb - 1) / lb, padded_bits = k * lb;
    ofe_t pow_padded = fe_u64(1ULL << padded_bits);
    av_t one = K(1), neg_pow;
    neg_pow.v = fe_neg(&pow_padded); neg_pow.cell = CELL_CONST;
    ofe_t shift_a_val = fe_add(&a.v, &pow_padded);
    ofe_t shifted_val = fe_sub(&shift_a_val, &b.v);
    c->cur_kind = 1;
    region(c, 7, 0, 3);
    av_t shifted = witness_cell(c, shifted_val);
    cell(c, &b, 0); cell(c, &one, 0);
    witness_cell(c, shift_a_val);
    c->cur_kind = 2;
    cell(c, &neg_pow, 0); cell(c, &one, 0); cell(c, &a, 0);
    av_t limbs[8];
    r_range_check_limbs(c, shifted, padded_bits + lb, limbs);
    return g_is_zero(c, limbs[k]);
}
static int bit_length_u64(uint64_t x) { int n = 0; while (x) { n++; x >>= 1; } return n; }
static av_t r_is_less_than_safe(oracle_ctx *c, av_t a, uint64_t b) {
    const int lb = c->lookup_bits;
    const int range_bits = (bit_length_u64(b) + lb - 1) / lb * lb;
    r_range_check(c, a, range_bits);
    return r_is_less_than(c, a, K(b), range_bits);
}

/* -------------------------------------------------------------- utils.rs */
/* utils.rs:6-14 fe_to_bits_le(val, size): little-endian bits of the canonical
 * value, whole bytes, zero-extended to `size`; the reference underflows
 * (panics) if the byte-rounded bit length exceeds `size`. */
static int fe_to_bits_le(oracle_ctx *c, const ofe_t *v, int size, uint8_t *bits) {
    int nbytes = 32;
    while (nbytes > 1 && ((v->l[(nbytes - 1) / 8] >> (8 * ((nbytes - 1) % 8))) & 0xff) == 0) nbytes--;
    /* BigUint::to_bytes_le() of zero is [0] -> 8 bits */
    int nbits = nbytes * 8;
    if (nbits > size) { ofail(c, "fe_to_bits_le: value wider than size (reference panics)", v->l[0], (uint64_t)size); nbits = size; }
    for (int i = 0; i < nbits; i++) bits[i] = (uint8_t)((v->l[i / 64] >> (i % 64)) & 1);
    for (int i = nbits; i < size; i++) bits[i] = 0;
    return size;
}
/* utils.rs:16-29 bits_le_to_fe (callers pass <= 64 bits). */
static ofe_t bits_le_to_fe(const uint8_t *bits, int n) {
    ofe_t r = {{0, 0, 0, 0}};
    for (int i = 0; i < n; i++) if (bits[i]) r.l[i / 64] |= 1ULL << (i % 64);
    return r;
}

/* ------------------------------------------------------------- spread.rs */
/* spread.rs:211-218: interleave zeros -- bit i of val goes to bit 2i. */
static ofe_t spread_value_of(oracle_ctx *c, const ofe_t *val) {
    uint8_t vb[32], sb[64];
    fe_to_bits_le(c, val, 32, vb);
    memset(sb, 0, sizeof sb);
    for (int i = 0; i < 32; i++) sb[2 * i] = vb[i];
    return bits_le_to_fe(sb, 64);
}
uint64_t oracle_spread_table_entry(uint32_t i) {
    uint64_t s = 0;
    for (int b = 0; b < 32; b++) s |= (uint64_t)((i >> b) & 1) << (2 * b);
    return s;
}

/* spread.rs:196-233 spread_limb: two raw region.assign_advice cells at
 * (denses[col], row_offset) / (spreads[col], row_offset), one load_witness. */
static av_t sc_spread_limb(oracle_ctx *c, av_t limb) {
    c->st.spread_limb_calls++;
    uint64_t column_idx = c->num_limb_sum % (uint64_t)c->num_advice_columns;    /* :202 */
    ofe_t spread_value = spread_value_of(c, &limb.v);                            /* :211-218 */
    if (c->check) {
        /* the "spread lookup" (spread.rs:56-62): (dense, spread) must be a table row */
        if (!fe_is_u64(&limb.v) || limb.v.l[0] >= (1ULL << c->num_bits_lookup))
            ofail(c, "spread lookup: dense limb outside table", limb.v.l[0], 1ULL << c->num_bits_lookup);
        else if (spread_value.l[0] != oracle_spread_table_entry((uint32_t)limb.v.l[0]))
            ofail(c, "spread lookup: spread mismatch", spread_value.l[0], limb.v.l[0]);
    }
    if (c->dense && c->spread) {
        if (c->row_offset < c->row_base || c->row_offset - c->row_base >= c->col_stride)
            ofail(c, "chip row outside buffer", c->row_offset, c->row_base);
        else {
            size_t at = (size_t)column_idx * c->col_stride + (size_t)(c->row_offset - c->row_base);
            c->dense[at] = limb.v;                                               /* :203-208 */
            c->spread[at] = spread_value;                                        /* :219-224 */
        }
    }
    c->st.chip_cells += 2;
    av_t assigned_spread_value = g_load_witness(c, spread_value);                /* :225 */
    if (c->rec) {   /* :209-210 dense chip cell == limb's cell; :226-227 spread chip cell == the new witness */
        oracle_constraints_t *r = c->rec;
        if (r->chip && r->n_chip < r->chip_cap) {
            r->chip[2 * r->n_chip] = limb.cell;
            r->chip[2 * r->n_chip + 1] = assigned_spread_value.cell;
        }
        r->n_chip++;
    }
    c->num_limb_sum += 1;                                                        /* :228 */
    if (column_idx == (uint64_t)c->num_advice_columns - 1) c->row_offset += 1;   /* :229-231 */
    return assigned_spread_value;
}

/* spread.rs:76-123 SpreadConfig::spread */
static av_t sc_spread(oracle_ctx *c, av_t dense) {
    c->st.spread_calls++;
    int limb_bits = c->num_bits_lookup;                 /* :83 */
    int num_limbs = 16 / limb_bits;                     /* :84 */
    av_t assigned_limbs[16];
    /* :85 decompose(v, num_limbs, limb_bits): little-endian limb_bits-wide digits */
    for (int idx = 0; idx < num_limbs; idx++) {
        uint64_t limb = (dense.v.l[0] >> (limb_bits * idx)) & ((1ULL << limb_bits) - 1);
        assigned_limbs[idx] = g_load_witness(c, fe_u64(limb));          /* :86-88 */
    }
    {
        av_t limbs_sum = g_load_zero(c);                                /* :90 */
        for (int idx = 0; idx < num_limbs; idx++)                       /* :91-98 */
            limbs_sum = g_mul_add(c, assigned_limbs[idx], K(1ULL << (limb_bits * idx)), limbs_sum);
        g_assert_equal(c, limbs_sum, dense);                            /* :104-108 */
    }
    av_t assigned_spread = g_load_zero(c);                              /* :110 */
    for (int idx = 0; idx < num_limbs; idx++) {                         /* :112-121 */
        av_t spread_limb = sc_spread_limb(c, assigned_limbs[idx]);
        assigned_spread = g_mul_add(c, spread_limb, K(1ULL << (2 * limb_bits * idx)), assigned_spread);
    }
    return assigned_spread;
}

/* spread.rs:139-163 decompose_even_and_odd_unchecked */
static void sc_decompose_even_and_odd_unchecked(oracle_ctx *c, av_t spread, av_t *even, av_t *odd) {
    c->st.even_odd_calls++;
    uint8_t bits[32], eb[16], ob[16];
    fe_to_bits_le(c, &spread.v, 32, bits);              /* :145 */
    for (int i = 0; i < 16; i++) { eb[i] = bits[2 * i]; ob[i] = bits[2 * i + 1]; }   /* :146-153 */
    ofe_t even_val = bits_le_to_fe(eb, 16), odd_val = bits_le_to_fe(ob, 16);         /* :154-157 */
    *even = g_load_witness(c, even_val);                /* :158 */
    *odd = g_load_witness(c, odd_val);                  /* :159 */
    r_range_check(c, *even, 16);                        /* :160 */
    r_range_check(c, *odd, 16);                         /* :161 */
}

/* -------------------------------------------------------- compression.rs */
static const uint32_t ROUND_CONSTANTS[64] = {           /* FIPS 180-4 K; compression.rs:992-1001 */
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
    0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
    0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
    0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
    0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
static const uint32_t INIT_STATE[8] = {                 /* compression.rs:1003-1012 */
    0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};

typedef struct { av_t lo, hi; } spread_u32;             /* compression.rs:17 SpreadU32 */

/* compression.rs:215-246 */
static spread_u32 state_to_spread_u32(oracle_ctx *c, av_t x) {
    ofe_t lo = fe_u64(fe_lower_32(&x.v) & ((1u << 16) - 1));   /* :222-225 */
    ofe_t hi = fe_u64(fe_lower_32(&x.v) >> 16);                /* :226-229 */
    av_t assigned_lo = g_load_witness(c, lo);                   /* :230 */
    av_t assigned_hi = g_load_witness(c, hi);                   /* :231 */
    av_t composed = g_mul_add(c, assigned_hi, K(1ULL << 16), assigned_lo);   /* :232-237 */
    g_assert_equal(c, x, composed);                             /* :238-242 */
    spread_u32 r;
    r.lo = sc_spread(c, assigned_lo);                           /* :243 */
    r.hi = sc_spread(c, assigned_hi);                           /* :244 */
    return r;
}

/* compression.rs:266-295 */
static av_t mod_u32(oracle_ctx *c, av_t x) {
    ofe_t lo = fe_u64(fe_lower_32(&x.v));                                   /* :272-275 */
    ofe_t hi = fe_u64((fe_lower_64(&x.v) >> 32) & ((1ULL << 32) - 1));      /* :276-279 */
    av_t assigned_lo = g_load_witness(c, lo);                               /* :280 */
    av_t assigned_hi = g_load_witness(c, hi);                               /* :281 */
    r_range_check(c, assigned_lo, 32);                                      /* :282 */
    av_t composed = g_mul_add(c, assigned_hi, K(1ULL << 32), assigned_lo);  /* :283-288 */
    g_assert_equal(c, x, composed);                                         /* :289-293 */
    return assigned_lo;
}

/* compression.rs:521-530 */
static av_t three_add(oracle_ctx *c, av_t x, av_t y, av_t z) {
    av_t add1 = g_add(c, x, y);
    return g_add(c, add1, z);
}

/* the { spread(even); spread(odd); 2*odd+even == whole } block that ch, maj
 * and sigma_generic each repeat (compression.rs:344-354 and siblings) */
static void recheck_even_odd(oracle_ctx *c, av_t even, av_t odd, av_t whole) {
    av_t even_spread = sc_spread(c, even);
    av_t odd_spread = sc_spread(c, odd);
    av_t sum = g_mul_add(c, K(2), odd_spread, even_spread);
    g_assert_equal(c, sum, whole);
}

/* compression.rs:297-405 */
static av_t ch(oracle_ctx *c, spread_u32 x, spread_u32 y, spread_u32 z) {
    av_t p_lo = g_add(c, x.lo, y.lo);                           /* :309-313 */
    av_t p_hi = g_add(c, x.hi, y.hi);                           /* :314-318 */
    const uint64_t MASK_EVEN_32 = 0x55555555;                   /* :319 */
    av_t x_neg_lo = g_neg(c, x.lo);                             /* :320 */
    av_t x_neg_hi = g_neg(c, x.hi);                             /* :321 */
    av_t q_lo = three_add(c, K(MASK_EVEN_32), x_neg_lo, z.lo);  /* :322-328 */
    av_t q_hi = three_add(c, K(MASK_EVEN_32), x_neg_hi, z.hi);  /* :329-335 */
    av_t p_lo_even, p_lo_odd, p_hi_even, p_hi_odd, q_lo_even, q_lo_odd, q_hi_even, q_hi_odd;
    sc_decompose_even_and_odd_unchecked(c, p_lo, &p_lo_even, &p_lo_odd);    /* :336-337 */
    sc_decompose_even_and_odd_unchecked(c, p_hi, &p_hi_even, &p_hi_odd);    /* :338-339 */
    sc_decompose_even_and_odd_unchecked(c, q_lo, &q_lo_even, &q_lo_odd);    /* :340-341 */
    sc_decompose_even_and_odd_unchecked(c, q_hi, &q_hi_even, &q_hi_odd);    /* :342-343 */
    recheck_even_odd(c, p_lo_even, p_lo_odd, p_lo);             /* :344-354 */
    recheck_even_odd(c, p_hi_even, p_hi_odd, p_hi);             /* :355-365 */
    recheck_even_odd(c, q_lo_even, q_lo_odd, q_lo);             /* :366-376 */
    recheck_even_odd(c, q_hi_even, q_hi_odd, q_hi);             /* :377-387 */
    av_t out_lo = g_add(c, p_lo_odd, q_lo_odd);                 /* :388-392 */
    av_t out_hi = g_add(c, p_hi_odd, q_hi_odd);                 /* :393-397 */
    return g_mul_add(c, out_hi, K(1ULL << 16), out_lo);         /* :398-403 */
}

/* compression.rs:460-519 */
static av_t maj(oracle_ctx *c, spread_u32 x, spread_u32 y, spread_u32 z) {
    av_t m_lo = three_add(c, x.lo, y.lo, z.lo);                 /* :472-478 */
    av_t m_hi = three_add(c, x.hi, y.hi, z.hi);                 /* :479-485 */
    av_t m_lo_even, m_lo_odd, m_hi_even, m_hi_odd;
    sc_decompose_even_and_odd_unchecked(c, m_lo, &m_lo_even, &m_lo_odd);    /* :486-487 */
    sc_decompose_even_and_odd_unchecked(c, m_hi, &m_hi_even, &m_hi_odd);    /* :488-489 */
    recheck_even_odd(c, m_lo_even, m_lo_odd, m_lo);             /* :490-500 */
    recheck_even_odd(c, m_hi_even, m_hi_odd, m_hi);             /* :501-511 */
    return g_mul_add(c, m_hi_odd, K(1ULL << 16), m_lo_odd);     /* :512-517 */
}

/* compression.rs:702-882 */
typedef struct { int starts[4], ends[4]; uint64_t coeffs[4]; } sigma_params;
static av_t sigma_generic(oracle_ctx *c, const spread_u32 *x_spread, const sigma_params *sp) {
    const int *starts = sp->starts, *ends = sp->ends;
    const uint64_t *coeffs = sp->coeffs;
    uint8_t bits[64];
    fe_to_bits_le(c, &x_spread->lo.v, 32, bits);                /* :715 */
    fe_to_bits_le(c, &x_spread->hi.v, 32, bits + 32);           /* :716 */
    av_t assigned[4];
    for (int i = 0; i < 4; i++) {                               /* :719-734 assign_bits x4 */
        uint8_t piece[64];
        int n = 2 * ends[i] - 2 * starts[i];
        memcpy(piece, bits + 2 * starts[i], (size_t)n);         /* :722 */
        memset(piece + n, 0, (size_t)(64 - n));                 /* :723 */
        assigned[i] = g_load_witness(c, bits_le_to_fe(piece, 64));      /* :724-726 */
    }
    {
        av_t sum = assigned[0];                                 /* :736 */
        sum = g_mul_add(c, assigned[1], K(1ULL << (2 * starts[1])), sum);     /* :737-742 */
        sum = g_mul_add(c, assigned[2], K(1ULL << (2 * starts[2])), sum);     /* :743-748 */
        sum = g_mul_add(c, assigned[3], K(1ULL << (2 * starts[3])), sum);     /* :749-754 */
        av_t x_composed = g_mul_add(c, x_spread->hi, K(1ULL << 32), x_spread->lo);   /* :755-760 */
        g_assert_equal(c, x_composed, sum);                     /* :761-765 */
    }
    av_t r_spread;
    {
        av_t sum = g_load_zero(c);                              /* :780 */
        for (int i = 0; i < 4; i++)                             /* :785-808 */
            sum = g_mul_add(c, K(coeffs[i]), assigned[i], sum);
        r_spread = sum;
    }
    av_t r_lo, r_hi;
    {
        ofe_t lo = fe_u64(fe_lower_32(&r_spread.v));                              /* :812-815 */
        ofe_t hi = fe_u64((fe_lower_64(&r_spread.v) >> 32) & ((1ULL << 32) - 1)); /* :816-819 */
        av_t assigned_lo = g_load_witness(c, lo);               /* :820 */
        av_t assigned_hi = g_load_witness(c, hi);               /* :821 */
        r_range_check(c, assigned_lo, 32);                      /* :822 */
        r_range_check(c, assigned_hi, 32);                      /* :823 */
        av_t composed = g_mul_add(c, assigned_hi, K(1ULL << 32), assigned_lo);   /* :824-829 */
        g_assert_equal(c, r_spread, composed);                  /* :830-834 */
        r_lo = assigned_lo; r_hi = assigned_hi;
    }
    av_t r_lo_even, r_lo_odd, r_hi_even, r_hi_odd;
    sc_decompose_even_and_odd_unchecked(c, r_lo, &r_lo_even, &r_lo_odd);    /* :843-844 */
    sc_decompose_even_and_odd_unchecked(c, r_hi, &r_hi_even, &r_hi_odd);    /* :845-846 */
    recheck_even_odd(c, r_lo_even, r_lo_odd, r_lo);             /* :852-862 */
    recheck_even_odd(c, r_hi_even, r_hi_odd, r_hi);             /* :863-873 */
    return g_mul_add(c, r_hi_even, K(1ULL << 16), r_lo_even);   /* :874-879 */
}

#define P2(n) (1ULL << (n))
/* compression.rs:594-619 */
static av_t sigma_upper0(oracle_ctx *c, const spread_u32 *x) {
    static const sigma_params SP = {{0, 2, 13, 22}, {2, 13, 22, 32},
                                    {P2(60) + P2(38) + P2(20), P2(0) + P2(42) + P2(24),
                                     P2(22) + P2(0) + P2(46), P2(40) + P2(18) + P2(0)}};
    return sigma_generic(c, x, &SP);
}
/* compression.rs:621-646 */
static av_t sigma_upper1(oracle_ctx *c, const spread_u32 *x) {
    static const sigma_params SP = {{0, 6, 11, 25}, {6, 11, 25, 32},
                                    {P2(52) + P2(42) + P2(14), P2(0) + P2(54) + P2(26),
                                     P2(10) + P2(0) + P2(36), P2(38) + P2(28) + P2(0)}};
    return sigma_generic(c, x, &SP);
}
/* compression.rs:648-673 */
static av_t sigma_lower0(oracle_ctx *c, const spread_u32 *x) {
    static const sigma_params SP = {{0, 3, 7, 18}, {3, 7, 18, 32},
                                    {P2(50) + P2(28), P2(0) + P2(56) + P2(34),
                                     P2(8) + P2(0) + P2(42), P2(30) + P2(22) + P2(0)}};
    return sigma_generic(c, x, &SP);
}
/* compression.rs:675-700 */
static av_t sigma_lower1(oracle_ctx *c, const spread_u32 *x) {
    static const sigma_params SP = {{0, 10, 17, 19}, {10, 17, 19, 32},
                                    {P2(30) + P2(26), P2(0) + P2(50) + P2(46),
                                     P2(14) + P2(0) + P2(60), P2(18) + P2(4) + P2(0)}};
    return sigma_generic(c, x, &SP);
}

/* compression.rs:19-213 */
static void compression_core(oracle_ctx *c, const av_t assigned_input_bytes[64], const av_t pre_state_words[8],
                             av_t out_words[8]) {

    /* message schedule: :31-47 */
    av_t message_u32s[64];
    for (int w = 0; w < 16; w++) {
        const av_t *bytes = &assigned_input_bytes[4 * w];
        av_t sum = g_load_zero(c);                                          /* :34 */
        for (int idx = 0; idx < 4; idx++)                                   /* :35-42 */
            sum = g_mul_add(c, bytes[3 - idx], K(1ULL << (8 * idx)), sum);
        message_u32s[w] = sum;
    }
    spread_u32 message_spreads[64];
    for (int w = 0; w < 16; w++)                                            /* :53-56 */
        message_spreads[w] = state_to_spread_u32(c, message_u32s[w]);
    for (int idx = 16; idx < 64; idx++) {                                   /* :57-96 */
        av_t term1 = sigma_lower1(c, &message_spreads[idx - 2]);            /* :60 */
        av_t term3 = sigma_lower0(c, &message_spreads[idx - 15]);           /* :61 */
        av_t sum = g_add(c, term1, message_u32s[idx - 7]);                  /* :65-69 */
        sum = g_add(c, sum, term3);                                         /* :70-74 */
        sum = g_add(c, sum, message_u32s[idx - 16]);                        /* :75-79 */
        av_t new_w = mod_u32(c, sum);                                       /* :80 */
        message_u32s[idx] = new_w;                                          /* :89 */
        message_spreads[idx] = state_to_spread_u32(c, new_w);               /* :90-91 */
    }

    /* compression: :99-124 */
    av_t a = pre_state_words[0], b = pre_state_words[1], cc = pre_state_words[2],
         d = pre_state_words[3], e = pre_state_words[4], f = pre_state_words[5],
         g = pre_state_words[6], h = pre_state_words[7];
    spread_u32 a_spread = state_to_spread_u32(c, a);        /* :109 */
    spread_u32 b_spread = state_to_spread_u32(c, b);        /* :110 */
    spread_u32 c_spread = state_to_spread_u32(c, cc);       /* :111 */
    spread_u32 e_spread = state_to_spread_u32(c, e);        /* :113 */
    spread_u32 f_spread = state_to_spread_u32(c, f);        /* :114 */
    spread_u32 g_spread = state_to_spread_u32(c, g);        /* :115 */
    g_load_zero(c);                                         /* :123 */
    g_load_zero(c);                                         /* :124 */
    for (int idx = 0; idx < 64; idx++) {                    /* :125-196 */
        av_t t1, t2;
        {
            av_t sigma_term = sigma_upper1(c, &e_spread);                   /* :130 */
            av_t ch_term = ch(c, e_spread, f_spread, g_spread);             /* :131 */
            av_t add1 = g_add(c, h, sigma_term);                            /* :138-142 */
            av_t add2 = g_add(c, add1, ch_term);                            /* :143-147 */
            av_t add3 = g_add(c, add2, K(ROUND_CONSTANTS[idx]));            /* :148-152 */
            av_t add4 = g_add(c, add3, message_u32s[idx]);                  /* :153-157 */
            t1 = mod_u32(c, add4);                                          /* :158 */
        }
        {
            av_t sigma_term = sigma_upper0(c, &a_spread);                   /* :164 */
            av_t maj_term = maj(c, a_spread, b_spread, c_spread);           /* :165 */
            av_t add = g_add(c, sigma_term, maj_term);                      /* :166-170 */
            t2 = mod_u32(c, add);                                           /* :171 */
        }
        h = g;                                              /* :174 */
        g = f; g_spread = f_spread;                         /* :176-177 */
        f = e; f_spread = e_spread;                         /* :178-179 */
        {
            av_t add = g_add(c, d, t1);                     /* :181 */
            e = mod_u32(c, add);                            /* :182 */
        }
        e_spread = state_to_spread_u32(c, e);               /* :184 */
        d = cc;                                             /* :185 */
        cc = b; c_spread = b_spread;                        /* :187-188 */
        b = a; b_spread = a_spread;                         /* :189-190 */
        {
            av_t add = g_add(c, t1, t2);                    /* :192 */
            a = mod_u32(c, add);                            /* :193 */
        }
        a_spread = state_to_spread_u32(c, a);               /* :195 */
    }
    av_t new_states[8] = {a, b, cc, d, e, f, g, h};         /* :197 */
    for (int i = 0; i < 8; i++) {                           /* :198-211 */
        av_t add = g_add(c, new_states[i], pre_state_words[i]);
        out_words[i] = mod_u32(c, add);
    }
}

int oracle_sha256_compression(oracle_ctx *c, const uint8_t block[64],
                              const uint32_t pre_state[8], uint32_t next_state[8]) {
    c->block_start = c->gate_len;
    /* the caller's assigned_input_bytes / pre_state_words (lib.rs:162-173): cells outside the stream */
    av_t assigned_input_bytes[64], pre_state_words[8], out[8];
    for (int i = 0; i < 64; i++) assigned_input_bytes[i] = ext_cell(block[i], ORACLE_CELL_INPUT_BYTE0 - i);
    for (int i = 0; i < 8; i++) pre_state_words[i] = ext_cell(pre_state[i], ORACLE_CELL_PRE_STATE0 - i);
    compression_core(c, assigned_input_bytes, pre_state_words, out);
    for (int i = 0; i < 8; i++) {
        if (next_state) next_state[i] = fe_lower_32(&out[i].v);
        if (c->rec && c->rec->next_state_cells) c->rec->next_state_cells[i] = out[i].cell;
    }
    c->rec = NULL;      /* the structure is input independent: recorded for one block only */
    return c->failed;
}

int oracle_witness_blocks(oracle_ctx *c, const uint8_t *blocks, const uint32_t *pre_states,
                          size_t n, uint32_t *next_states) {
    for (size_t i = 0; i < n; i++) {
        uint32_t ns[8];
        oracle_sha256_compression(c, blocks + 64 * i, pre_states + 8 * i, ns);
        if (next_states) memcpy(next_states + 8 * i, ns, sizeof ns);
    }
    return c->failed;
}

/* ------------------------------------------------ plain SHA-256 (sha2 crate) */
static inline uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
void oracle_plain_compress(uint32_t state[8], const uint8_t block[64]) {
    uint32_t w[64];
    for (int i = 0; i < 16; i++)
        w[i] = ((uint32_t)block[4 * i] << 24) | ((uint32_t)block[4 * i + 1] << 16) |
               ((uint32_t)block[4 * i + 2] << 8) | block[4 * i + 3];
    for (int i = 16; i < 64; i++) {
        uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3);
        uint32_t s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
        w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = state[0], b = state[1], c = state[2], d = state[3], e = state[4], f = state[5],
             g = state[6], h = state[7];
    for (int i = 0; i < 64; i++) {
        uint32_t S1 = rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25);
        uint32_t chv = (e & f) ^ (~e & g);
        uint32_t t1 = h + S1 + chv + ROUND_CONSTANTS[i] + w[i];
        uint32_t S0 = rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22);
        uint32_t mj = (a & b) ^ (a & c) ^ (b & c);
        uint32_t t2 = S0 + mj;
        h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    state[0] += a; state[1] += b; state[2] += c; state[3] += d;
    state[4] += e; state[5] += f; state[6] += g; state[7] += h;
}

/* ------------------------------------------------------------------ lib.rs */
/* lib.rs:71-349, value-level.  The digest epilogue's own cells (length
 * constraints :122-151, is_equal/select :294-310, output bytes :311-341) are
 * not emitted -- they are the out-of-scope front-end; the
 * *selection rule* is followed so the digest is the reference's. */
int oracle_digest(oracle_ctx *c, const uint8_t *input, size_t input_byte_size,
                  size_t precomputed_input_len, size_t max_variable_byte_size,
                  uint8_t digest[32], uint8_t *blocks_out, uint32_t *pre_states_out,
                  uint32_t *next_states_out) {
    const size_t one_round_size = 64;                                       /* :48 */
    if (max_variable_byte_size % one_round_size != 0) return 10;           /* :57-59 */
    size_t input_byte_size_with_9 = input_byte_size + 9;                    /* :78 */
    size_t num_round = (input_byte_size_with_9 % one_round_size == 0)       /* :80-84 */
                           ? input_byte_size_with_9 / one_round_size
                           : input_byte_size_with_9 / one_round_size + 1;
    size_t padded_size = one_round_size * num_round;                        /* :85 */
    size_t max_variable_round = max_variable_byte_size / one_round_size;    /* :87 */
    if (precomputed_input_len % one_round_size != 0) return 11;             /* :89 */
    if (precomputed_input_len > padded_size ||
        padded_size - precomputed_input_len > max_variable_byte_size) return 12;   /* :90 */
    size_t zero_padding_byte_size = padded_size - input_byte_size_with_9;   /* :91 */
    size_t remaining_byte_size = max_variable_byte_size + precomputed_input_len - padded_size;  /* :92 */
    size_t precomputed_round = precomputed_input_len / one_round_size;      /* :93 */
    if (remaining_byte_size != one_round_size * (max_variable_round + precomputed_round - num_round))
        return 13;                                                          /* :94-97 */
    size_t total = max_variable_byte_size + precomputed_input_len;
    uint8_t *padded_inputs = (uint8_t *)calloc(total ? total : 1, 1);
    if (!padded_inputs) return 14;
    size_t n = 0;
    memcpy(padded_inputs, input, input_byte_size); n += input_byte_size;   /* :98 */
    padded_inputs[n++] = 0x80;                                              /* :99 */
    n += zero_padding_byte_size;                                            /* :100-102 */
    uint64_t bitlen = 8ULL * (uint64_t)input_byte_size;                     /* :103-108 */
    for (int i = 7; i >= 0; i--) padded_inputs[n++] = (uint8_t)(bitlen >> (8 * i));
    if (n != num_round * one_round_size) { free(padded_inputs); return 15; }   /* :110 */
    n += remaining_byte_size;                                               /* :111-113 (zeros) */
    if (n != total) { free(padded_inputs); return 16; }                     /* :114-117 */

    uint32_t last_state[8];                                                 /* :155-160 */
    memcpy(last_state, INIT_STATE, sizeof last_state);
    for (size_t r = 0; r < precomputed_round; r++)
        oracle_plain_compress(last_state, padded_inputs + r * one_round_size);

    /* :180-238: always max_variable_round compressions */
    size_t target_round = num_round - precomputed_round;                    /* :147-151 */
    uint32_t output_h_out[8] = {0, 0, 0, 0, 0, 0, 0, 0};                    /* :294-295 */
    if (target_round == 0) memcpy(output_h_out, last_state, sizeof last_state);     /* n_round 0 candidate */
    for (size_t r = 0; r < max_variable_round; r++) {
        const uint8_t *blk = padded_inputs + precomputed_input_len + r * one_round_size;
        uint32_t next[8];
        if (blocks_out) memcpy(blocks_out + r * 64, blk, 64);
        if (pre_states_out) memcpy(pre_states_out + r * 8, last_state, sizeof last_state);
        oracle_sha256_compression(c, blk, last_state, next);                /* :183-189 */
        if (next_states_out) memcpy(next_states_out + r * 8, next, sizeof next);
        memcpy(last_state, next, sizeof next);                              /* :236 */
        if (r + 1 == target_round) memcpy(output_h_out, next, sizeof next); /* :296-310 */
    }
    for (int i = 0; i < 8; i++) {                                           /* :311-341 be bytes */
        digest[4 * i] = (uint8_t)(output_h_out[i] >> 24);
        digest[4 * i + 1] = (uint8_t)(output_h_out[i] >> 16);
        digest[4 * i + 2] = (uint8_t)(output_h_out[i] >> 8);
        digest[4 * i + 3] = (uint8_t)output_h_out[i];
    }
    free(padded_inputs);
    return c->failed ? 1 : 0;
}

/* lib.rs:71-349 with EVERY cell digest() itself allocates, under
 * assumptions A1-A4: prologue (lib.rs:122-178), the Context's zero cell at its
 * first use (compression.rs:34 of the first block of a context), the block loop
 * (lib.rs:180-238), the epilogue (lib.rs:294-341).  Cell ids are absolute stream
 * indices, so an attached constraint recorder sees the copy constraints between
 * the sections (input bytes -> word packing, states -> next block / select). */
int oracle_digest_cells(oracle_ctx *c, const uint8_t *input, size_t input_byte_size,
                        size_t precomputed_input_len, size_t max_variable_byte_size,
                        int is_input_range_check, uint8_t digest[32], oracle_digest_layout_t *lay) {
    if (!c->internals) return 20;        /* the frame consists of halo2-base internals */
    const size_t one_round_size = 64;                                       /* :48 */
    if (max_variable_byte_size % one_round_size != 0) return 10;           /* :57-59 */
    size_t input_byte_size_with_9 = input_byte_size + 9;                    /* :78 */
    size_t num_round = (input_byte_size_with_9 + one_round_size - 1) / one_round_size;   /* :80-84 */
    size_t padded_size = one_round_size * num_round;                        /* :85 */
    size_t max_variable_round = max_variable_byte_size / one_round_size;    /* :87 */
    if (precomputed_input_len % one_round_size != 0) return 11;             /* :89 */
    if (precomputed_input_len > padded_size ||
        padded_size - precomputed_input_len > max_variable_byte_size) return 12;   /* :90 */
    size_t zero_padding_byte_size = padded_size - input_byte_size_with_9;   /* :91 */
    size_t precomputed_round = precomputed_input_len / one_round_size;      /* :93 */
    size_t total = max_variable_byte_size + precomputed_input_len;
    uint8_t *padded_inputs = (uint8_t *)calloc(total ? total : 1, 1);       /* :98-117 */
    av_t *assigned_input_bytes = (av_t *)calloc(max_variable_byte_size ? max_variable_byte_size : 1, sizeof(av_t));
    av_t (*states)[8] = (av_t (*)[8])calloc(max_variable_round + 1, sizeof(av_t[8]));
    if (!padded_inputs || !assigned_input_bytes || !states) { free(padded_inputs); free(assigned_input_bytes); free(states); return 14; }
    size_t n = 0;
    if (input_byte_size) memcpy(padded_inputs, input, input_byte_size);
    n += input_byte_size;
    padded_inputs[n++] = 0x80;
    n += zero_padding_byte_size;
    uint64_t bitlen = 8ULL * (uint64_t)input_byte_size;
    for (int i = 7; i >= 0; i--) padded_inputs[n++] = (uint8_t)(bitlen >> (8 * i));

    const int was_whole = c->whole;
    const size_t saved_block_start = c->block_start;
    c->whole = 1;
    c->block_start = 0;                   /* cell ids = absolute stream indices */
    const size_t g0 = c->gate_len, l0 = c->lookup_len;

    /* ---- prologue: lib.rs:122-178 ---- */
    av_t assigned_input_byte_size = g_load_witness(c, fe_u64(input_byte_size));                 /* :124-125 */
    av_t assigned_num_round = g_load_witness(c, fe_u64(num_round));                             /* :126 */
    av_t assigned_padded_size = g_mul(c, assigned_num_round, K(one_round_size));                /* :127-131 */
    av_t assigned_input_with_9_size = g_add(c, assigned_input_byte_size, K(9));                 /* :132-136 */
    av_t padding_size = g_sub(c, assigned_padded_size, assigned_input_with_9_size);             /* :137-141 */
    av_t padding_is_less_than_round = r_is_less_than_safe(c, padding_size, one_round_size);     /* :142-143 */
    g_assert_is_const(c, padding_is_less_than_round, 1);                                        /* :144 */
    av_t assigned_precomputed_round = g_load_witness(c, fe_u64(precomputed_round));             /* :145-146 */
    av_t assigned_target_round = g_sub(c, assigned_num_round, assigned_precomputed_round);      /* :147-151 */
    uint32_t last_state[8];                                                                     /* :155-160 */
    memcpy(last_state, INIT_STATE, sizeof last_state);
    for (size_t r = 0; r < precomputed_round; r++)
        oracle_plain_compress(last_state, padded_inputs + r * one_round_size);
    for (int i = 0; i < 8; i++) states[0][i] = g_load_witness(c, fe_u64(last_state[i]));       /* :162-165 */
    for (size_t i = 0; i < max_variable_byte_size; i++)                                         /* :170-173 */
        assigned_input_bytes[i] = g_load_witness(c, fe_u64(padded_inputs[precomputed_input_len + i]));
    if (is_input_range_check)                                                                   /* :174-178 */
        for (size_t i = 0; i < max_variable_byte_size; i++) r_range_check(c, assigned_input_bytes[i], 8);
    const size_t g1 = c->gate_len, l1 = c->lookup_len;

    /* ---- block loop: lib.rs:180-238 (the first load_zero of a context lands here) ---- */
    const int zero_before = c->zero_loaded;
    size_t zero_cells = 0, g2 = g1;
    for (size_t r = 0; r < max_variable_round; r++) {
        if (r == 0 && !c->zero_loaded) { g_load_zero(c); zero_cells = 1; g2 = c->gate_len; c->st.load_zero--; }
        compression_core(c, assigned_input_bytes + r * one_round_size, states[r], states[r + 1]);
    }
    (void)zero_before;
    const size_t g3 = c->gate_len, l3 = c->lookup_len;

    /* ---- epilogue: lib.rs:294-341 ---- */
    av_t zero = g_load_zero(c);                                                                 /* :294 */
    av_t output_h_out[8];
    for (int i = 0; i < 8; i++) output_h_out[i] = zero;                                         /* :295 */
    for (size_t n_round = 0; n_round <= max_variable_round; n_round++) {                        /* :296 */
        av_t selector = g_is_equal(c, K(n_round), assigned_target_round);                       /* :297-301 */
        for (int i = 0; i < 8; i++)                                                             /* :302-309 */
            output_h_out[i] = g_select(c, states[n_round][i], output_h_out[i], selector);
    }
    for (int w = 0; w < 8; w++) {                                                               /* :311-341 */
        const uint32_t word = fe_lower_32(&output_h_out[w].v);                                  /* :314-316 */
        av_t assigned_bytes[4];
        for (int idx = 0; idx < 4; idx++) {                                                     /* :317-324 */
            const uint8_t be = (uint8_t)(word >> (24 - 8 * idx));
            assigned_bytes[idx] = g_load_witness(c, fe_u64(be));
            r_range_check(c, assigned_bytes[idx], 8);
            digest[4 * w + idx] = be;
        }
        av_t sum = g_load_zero(c);                                                              /* :325 */
        for (int idx = 0; idx < 4; idx++)                                                       /* :326-333 */
            sum = g_mul_add(c, assigned_bytes[idx], K(1ULL << (24 - 8 * idx)), sum);
        g_assert_equal(c, output_h_out[w], sum);                                                /* :334-338 */
    }
    if (lay) {
        lay->prologue_cells = g1 - g0;
        lay->zero_cells = zero_cells;
        lay->block_cells = g3 - g2;
        lay->epilogue_cells = c->gate_len - g3;
        lay->prologue_lookups = l1 - l0;
        lay->block_lookups = l3 - l1;
        lay->epilogue_lookups = c->lookup_len - l3;
        lay->num_round = num_round;
        lay->target_round = num_round - precomputed_round;
        lay->n_blocks = max_variable_round;
        lay->input_len_cell = assigned_input_byte_size.cell;
    }
    c->whole = was_whole;
    c->block_start = saved_block_start;
    free(padded_inputs); free(assigned_input_bytes); free(states);
    return c->failed ? 1 : 0;
}
void oracle_set_context(oracle_ctx *c, int zero_cell_loaded) {
    c->zero_loaded = zero_cell_loaded != 0;
    c->zero_cell = ORACLE_CELL_ZERO;
}
void oracle_set_tape(oracle_ctx *c, uint8_t *call_lens, size_t call_cap, uint64_t *gate_rows, size_t rows_cap) {
    c->call_lens = call_lens; c->call_cap = call_cap; c->n_calls = 0;
    c->gate_rows = gate_rows; c->rows_cap = rows_cap; c->n_rows = 0;
}
size_t oracle_tape_calls(const oracle_ctx *c) { return c->n_calls; }
size_t oracle_tape_rows(const oracle_ctx *c) { return c->n_rows; }

/* Canonical -> Montgomery form (halo2curves bn256::Fr in-memory representation:
 * x * 2^256 mod p), by generic 512-bit multiply + bit-serial reduction --
 * deliberately NOT the Barrett shortcut the kernel uses. */
void oracle_to_montgomery(ofe_t *cells, size_t n) {
    /* R = 2^256 mod p, derived here rather than hard-coded: double 1 mod p 256 times */
    ofe_t r = fe_u64(1);
    for (int i = 0; i < 256; i++) r = fe_add(&r, &r);
    for (size_t i = 0; i < n; i++) cells[i] = fe_mul_slow(&cells[i], &r);
}

int oracle_measure_shape(int num_bits_lookup, int num_advice_columns,
                         uint64_t *gate_cells_per_block, uint64_t *limb_calls_per_block) {
    oracle_ctx *c = oracle_create(num_bits_lookup, num_advice_columns, 1);
    if (!c) return 1;
    uint8_t blk[64];
    for (int i = 0; i < 64; i++) blk[i] = (uint8_t)(i * 37 + 11);
    uint32_t next[8];
    oracle_sha256_compression(c, blk, INIT_STATE, next);
    if (gate_cells_per_block) *gate_cells_per_block = c->gate_len;
    if (limb_calls_per_block) *limb_calls_per_block = c->num_limb_sum;
    int failed = c->failed;
    oracle_destroy(c);
    return failed;
}
