//! GPU witness path for `Sha256DynamicConfig` (MI355X, `libhsw.so`).  NOT COMPILED in the build image (no
//! Rust toolchain, the crate's git dependencies are not vendored): source for a maintainer, written against
//! the halo2-lib v0.2.x API the reference itself uses.  Everything on the C side of it is built and tested
//! (`include/hsw.h`, `pytest -m gpu`); `examples/assigned_hash_result.c` is this file's flow in plain C and
//! runs in the GPU test suite.
//!
//! The gadget's surface does not change.  `Sha256DynamicConfig::digest` keeps its signature
//! (reference `src/lib.rs:71-76`) and still returns an `AssignedHashResult` (`lib.rs:31-36`); user circuits
//! (`lib.rs:394-485`, `benches/digest.rs:37-100`) are untouched.  Inside `digest` (see `lib_patch.rs`):
//!
//! ```ignore
//! pub fn digest<'a, 'b: 'a>(&'a mut self, ctx: &mut Context<'b, F>, input: &'a [u8],
//!                           precomputed_input_len: Option<usize>) -> Result<AssignedHashResult<F>, Error> {
//!     if hsw::witness_only_pass(ctx, self) {
//!         if let Some(r) = hsw::digest_gpu(self, ctx, input, precomputed_input_len)? { return Ok(r); }   // this file
//!     }
//!     self.digest_cpu(ctx, input, precomputed_input_len)                          // the reference's body, unchanged
//! }
//! ```
//!
//! `digest_gpu` NEVER turns a circuit that works on the CPU into an error: whatever the C side refuses (no device,
//! a layout with more than 17 columns, a `Context` that moved between two digests, ...) makes it return `Ok(None)`
//! before it has touched `ctx`, and `digest` runs the reference's own body.  Only errors of `Region::assign_advice`
//! itself propagate.
//!
//! WHERE the region starts.  The reference's `digest` takes whatever `Context` it is handed (lib.rs:71-76,
//! 351-360): a circuit that has used the gate / range chips before its first digest stands at
//! `ctx.advice_alloc[0] = (column, row) != (0, 0)`, usually caches a zero cell (`ctx.zero_cell`) and has cells
//! queued for the lookup column (`ctx.cells_to_lookup`).  All three go to `hsw_gadget_set_origin`; the GPU lays the
//! region out from there (column breaks included) exactly as `FlexGateConfig::assign_region` would.
//!
//! WHEN the GPU path may run.  Selectors, fixed cells and copy constraints do not depend on the witness;
//! halo2 records them during key generation (`keygen_vk` / `keygen_pk`) and `MockProver::run` checks them.
//! Those passes keep the CPU path.  `create_proof` synthesises against a `WitnessCollection`, which listens
//! to `assign_advice` ONLY (`assign_fixed`, `copy`, `enable_selector` are no-ops there): that pass -- the one
//! `benches/digest.rs:143-156` times -- takes every advice cell of the region from the GPU.
//! `witness_only_pass` tells the passes apart from inside the gadget (ASSUMPTION A5, halo2_proofs PSE
//! v2023_02_02 `plonk/keygen.rs`, `plonk/prover.rs`, `dev.rs`; unpinned like A1-A4):
//!
//! | backend            | `assign_fixed(.., || Value::unknown())`                       | =>  |
//! |--------------------|----------------------------------------------------------------|-----|
//! | keygen `Assembly`  | evaluates the closure: `Err(Synthesis)`, nothing stored        | CPU |
//! | `MockProver`       | evaluates the closure: `Err(Synthesis)`, nothing stored        | CPU |
//! | `WitnessCollection`| ignores fixed assignments: `Ok(())`                            | GPU |
//!
//! The probe writes NOTHING in any backend (round 2 probed with real assignments of 0: a fixed cell another region
//! had already finalised a constant into would have been zeroed in the proving key).
//!
//! `hsw::force(Some(bool))` overrides the detection (a process-wide switch for a prover driver that prefers to
//! say so itself); `HSW_DISABLE=1` in the environment pins the CPU path.
use halo2_base::halo2_proofs::{
    circuit::{AssignedCell, Cell, Region, Value},
    plonk::{Advice, Column, Error, Fixed},
};
use halo2_base::{AssignedValue, Context};
use halo2_ecc::fields::PrimeField;
use hsw_sys as sys;
use std::cell::RefCell;
use std::os::raw::c_void;
use std::ptr;
use std::sync::atomic::{AtomicI8, Ordering};

use crate::{AssignedHashResult, Sha256DynamicConfig};

fn check(rc: i32) -> Result<(), Error> {
    if rc == sys::HSW_OK { Ok(()) } else { Err(Error::Synthesis) }
}

/// Everything the C side hands over for one digest, fetched BEFORE `ctx` is touched: if any call fails the CPU
/// path can still run on an untouched Context.
struct Fetched {
    r: sys::hsw_hash_result,
    rc: sys::hsw_result_cells,
    view: sys::hsw_gadget_view,
    segs: Vec<(u64, u64, usize)>,          // (FlexGate column, first row, cells) of this digest's gate cells, in stream order
    end: (u64, u64),                       // (column, row) of the digest's last cell
    tape: sys::hsw_region_tape,            // which distinct value / constant every cell of the region holds (input-independent)
}

static FORCE: AtomicI8 = AtomicI8::new(-1);          // -1 = detect, 0 = CPU, 1 = GPU

/// Override the pass detection for the whole process (`None` = detect again).
pub fn force(gpu: Option<bool>) {
    FORCE.store(match gpu { None => -1, Some(false) => 0, Some(true) => 1 }, Ordering::SeqCst);
}

/// Is this synthesis pass `create_proof`'s witness collection (A5 above)?  One `assign_fixed` of an UNKNOWN value:
/// the backends that store fixed cells evaluate the closure and fail on it without storing anything; the witness
/// pass ignores fixed assignments and returns `Ok`.  No cell of any column is written by the probe.
pub fn witness_only_pass<F: PrimeField>(ctx: &mut Context<'_, F>, sha256: &Sha256DynamicConfig<F>) -> bool {
    if std::env::var_os("HSW_DISABLE").is_some() { return false; }
    match FORCE.load(Ordering::SeqCst) { 0 => return false, 1 => return true, _ => {} }
    let fix: Column<Fixed> = sha256.range().gate.constants[0];
    ctx.region.assign_fixed(|| "hsw probe", fix, 0, || Value::<F>::unknown()).is_ok()
}

/// One engine + one whole-digest gadget per (prover thread, circuit shape), reused by every synthesis.
struct Backend {
    engine: *mut sys::hsw_engine,
    gadget: *mut sys::hsw_gadget,
    key: (Vec<usize>, usize, usize, bool, u64),       // max sizes, table bits, chip columns, range checks, max_rows
    // where the Context must stand for the gadget's NEXT digest of this pass: (advice_alloc[0], cells_to_lookup.len())
    // as the previous digest left them; None = the pass is not (or no longer) on the GPU
    expect: Option<((usize, usize), usize)>,
    chip_columns: usize,
    // pinned host buffer (hsw_host_alloc) for the region's DISTINCT values: ~40 % of its cells -- the rest repeat
    // one of them or hold a gate constant, at input-independent positions the tape names (hsw.h, distinct-value
    // delivery: 0.33 ms instead of 0.81 ms for the bench circuit's region)
    distinct: *mut [u64; 4],
    distinct_cells: usize,
}

thread_local! { static BACKEND: RefCell<Option<Backend>> = RefCell::new(None); }

impl Backend {
    fn new<F: PrimeField>(sha256: &Sha256DynamicConfig<F>, key: (Vec<usize>, usize, usize, bool, u64)) -> Result<Self, Error> {
        let device = std::env::var("HSW_DEVICE").ok().and_then(|s| s.parse().ok()).unwrap_or(0);
        // `be` owns whatever exists so far: an early `?` drops it and Drop releases engine / gadget / staging
        let mut be = Self { engine: ptr::null_mut(), gadget: ptr::null_mut(), key, expect: None,
                            chip_columns: sha256.spread_config.num_advice_columns,
                            distinct: ptr::null_mut(), distinct_cells: 0 };
        check(unsafe { sys::hsw_engine_create_ex(device, ptr::null_mut(), be.key.1 as u32, be.key.2 as u32,
                                                 sys::HSW_MODE_HALO2_INTERNALS, &mut be.engine) })?;
        check(unsafe { sys::hsw_gadget_create_ex(be.engine, be.key.0.as_ptr(), be.key.0.len(), be.key.3 as i32,
                                                 sys::HSW_GADGET_WHOLE_DIGEST, &mut be.gadget) })?;
        // halo2curves' in-memory Fr IS the cell format: no from_repr (a Montgomery multiplication) per cell
        check(unsafe { sys::hsw_gadget_set_repr(be.gadget, sys::HSW_REPR_MONTGOMERY) })?;
        // the column image from (0, 0); every synthesis pass re-bases it on where its Context stands (set_origin)
        let mut columns = 0u64;
        check(unsafe { sys::hsw_gadget_set_columns(be.gadget, be.key.4, &mut columns) })?;
        // where the chip columns sit relative to the gate stream is worth up to 8 % of an HBM-bound batch (DESIGN.md
        // 5.1): for circuits of a few hundred blocks or more let the gadget try three allocations, once
        if be.key.0.iter().sum::<usize>() / 64 >= 256 {
            check(unsafe { sys::hsw_gadget_place(be.gadget, 3, ptr::null_mut(), ptr::null_mut()) })?;
        }
        let mut tape = unsafe { std::mem::zeroed::<sys::hsw_region_tape>() };
        check(unsafe { sys::hsw_gadget_region_tape(be.gadget, &mut tape) })?;
        be.distinct_cells = tape.distinct_capacity as usize + 1;             // every digest of the circuit
        let mut p: *mut c_void = ptr::null_mut();
        check(unsafe { sys::hsw_host_alloc(be.distinct_cells * 32, &mut p) })?;
        be.distinct = p as *mut [u64; 4];
        Ok(be)
    }
}

impl Drop for Backend {
    fn drop(&mut self) {
        // (all three accept NULL)
        unsafe { sys::hsw_host_free(self.distinct as *mut c_void); sys::hsw_gadget_destroy(self.gadget); sys::hsw_engine_destroy(self.engine); }
    }
}

/// `[u64; 4]` in Montgomery form -> `F`.  For `F = bn256::Fr` this is the type's own memory layout
/// (halo2curves `Fr(pub(crate) [u64; 4])`), hence a plain 32-byte copy.
#[inline(always)]
fn fe<F: PrimeField>(cell: &[u64; 4]) -> F {
    debug_assert_eq!(std::mem::size_of::<F>(), 32);
    unsafe { std::mem::transmute_copy::<[u64; 4], F>(cell) }
}

/// `Sha256DynamicConfig::digest` with every advice cell taken from the GPU.  Same cells, same positions as
/// the CPU path under A1-A4; returns the same `AssignedHashResult`.  `Ok(None)` = the GPU path does not apply
/// (or the C side refused): nothing has been touched, the caller runs `digest_cpu`.
pub fn digest_gpu<'a, 'b: 'a, F: PrimeField>(sha256: &'a mut Sha256DynamicConfig<F>, ctx: &mut Context<'b, F>, input: &'a [u8],
                                             precomputed_input_len: Option<usize>) -> Result<Option<AssignedHashResult<F>>, Error> {
    let key = (sha256.max_variable_byte_sizes.clone(), sha256.spread_config.num_bits_lookup, sha256.spread_config.num_advice_columns,
               sha256.is_input_range_check, sha256.range().gate.max_rows as u64);
    BACKEND.with(|slot| {
        let mut slot = slot.borrow_mut();
        if slot.as_ref().map(|b| b.key != key).unwrap_or(true) {
            match Backend::new(sha256, key.clone()) { Ok(b) => *slot = Some(b), Err(_) => return Ok(None) }   // no device, no library: CPU
        }
        let be = slot.as_mut().unwrap();
        let here = (ctx.advice_alloc[0], ctx.cells_to_lookup.len());
        // ---- phase 1: the C side only.  Any refusal => Ok(None), ctx untouched.
        let fetched = match fetch(be, sha256, here, ctx.zero_cell.is_some(), input, precomputed_input_len) {
            Some(f) => f,
            None => { be.expect = None; return Ok(None); }
        };
        // ---- phase 2: hand the cells to halo2.  Errors here are halo2's own and propagate.
        let Fetched { r, rc, view, segs, end, tape } = fetched;
        let c = be.chip_columns as u64;
        let gate_cols: Vec<Column<Advice>> = sha256.range().gate.basic_gates[0].iter().map(|g| g.value).collect();
        // cell i of a stream holds value(code[i]): a gate constant or one of the distinct values fetch() brought over
        let (distinct, consts) = (be.distinct as *const [u64; 4], tape.consts as *const [u64; 4]);
        let value = |code: u32| -> F {
            fe(unsafe { &*(if code & sys::HSW_TAPE_CONST != 0 { consts.add((code & !sys::HSW_TAPE_CONST) as usize) } else { distinct.add(code as usize) }) })
        };
        let mut want: Vec<(u64, usize)> = vec![(rc.input_len_cell, usize::MAX)];            // (stream cell, slot in `got`)
        want.extend((0..rc.n_input_bytes).map(|i| (rc.input_bytes_cell0 + i, usize::MAX)));
        want.extend(rc.output_byte_cells.iter().map(|&c| (c, usize::MAX)));
        let mut got: Vec<Option<AssignedCell<F, F>>> = vec![None; want.len()];
        for (k, w) in want.iter_mut().enumerate() { w.1 = k; }
        want.sort_unstable();
        let mut next_want = 0usize;
        let mut cell = r.prologue_cell;
        for &(col, row, n) in &segs {
            for i in 0..n {
                let v: F = value(unsafe { *tape.gate_code.add(cell as usize + i) });
                let a = ctx.region.assign_advice(|| "hsw", gate_cols[col as usize], row as usize + i, || Value::known(v))?;
                while next_want < want.len() && want[next_want].0 == cell + i as u64 { got[want[next_want].1] = Some(a.clone()); next_want += 1; }
            }
            cell += n as u64;
        }
        // ---- the spread-chip columns of this digest's blocks (spread.rs:196-233): limb call n sits in column n % c,
        // row n / c
        let limbs = r.n_blocks as u64 * 2060 * (16 / sha256.spread_config.num_bits_lookup as u64);   // 2,060 spread() calls per block x limbs
        for n in r.spread_cursor0..r.spread_cursor0 + limbs {
            let (k, row) = ((n % c) as usize, (n / c) as usize);
            let d: F = value(unsafe { *tape.chip_dense_code.add(n as usize) });
            let sp: F = value(unsafe { *tape.chip_spread_code.add(n as usize) });
            ctx.region.assign_advice(|| "hsw chip", sha256.spread_config.denses[k], row, || Value::known(d))?;
            ctx.region.assign_advice(|| "hsw chip", sha256.spread_config.spreads[k], row, || Value::known(sp))?;
        }
        // ---- the lookup-advice column: RangeConfig::finalize (lib.rs:469) copies ctx.cells_to_lookup into it, in
        // queue order, after the circuit's last gadget.  Queue this digest's entries with their VALUES behind the
        // ones the circuit queued before (the gadget's own lookup stream starts at that index: set_origin): finalize
        // then fills the column exactly as in the CPU path, also when the circuit queues more lookups afterwards.
        let n = (r.epilogue_lookup + 64 - r.prologue_lookup) as usize;        // prologue | blocks | 64 epilogue entries (2 per digest byte)
        let j0 = (r.prologue_lookup - view.origin_lookups) as usize;          // the tape counts the gadget's own entries
        // every entry is a copy of a gate cell of this digest; in this pass `copy` is a no-op and only the value
        // matters, so the entries carry the handle of the digest's first cell
        let any_cell: Cell = got[0].as_ref().unwrap().cell();
        for i in 0..n {
            let v: F = value(unsafe { *tape.lookup_code.add(j0 + i) });
            ctx.cells_to_lookup.push(assigned(any_cell, v, 0));
        }
        // ---- the Context's own bookkeeping, as the CPU path leaves it: next free (column, row), the cached zero cell
        ctx.advice_alloc[0] = (end.0 as usize, end.1 as usize + 1);
        ctx.total_advice += (r.end_cell - r.prologue_cell) as usize;
        if ctx.zero_cell.is_none() {                       // load_zero caches one cell per Context (A4-iii)
            ctx.zero_cell = Some(assigned(any_cell, F::zero(), 0));
        }
        sha256.cur_hash_idx += 1;                          // lib.rs:347
        sha256.spread_config.num_limb_sum += limbs as usize;                 // spread.rs:228
        sha256.spread_config.row_offset = ((r.spread_cursor0 + limbs) / c) as usize;   // spread.rs:229-231
        be.expect = Some((ctx.advice_alloc[0], ctx.cells_to_lookup.len()));

        let pick = |slot: usize| -> AssignedValue<F> {
            let a = got[slot].as_ref().expect("result cell inside the digest's stream");
            let mut v = F::zero();
            a.value().map(|x| v = *x);
            assigned(a.cell(), v, 0)
        };
        Ok(Some(AssignedHashResult {
            input_len: pick(0),                                                           // lib.rs:124-125
            input_bytes: (0..rc.n_input_bytes as usize).map(|i| pick(1 + i)).collect(),    // lib.rs:170-173
            output_bytes: (0..32).map(|i| pick(1 + rc.n_input_bytes as usize + i)).collect(),   // lib.rs:317-324
        }))
    })
}

/// Phase 1 of `digest_gpu`: position the gadget where the Context stands, run the digest on the GPU, fetch the
/// positions.  `None` on ANY refusal of the C side -- the gadget is then out of step with the circuit and stays
/// unused until the next synthesis pass starts (cur_hash_idx == 0).
fn fetch<F: PrimeField>(be: &mut Backend, sha256: &Sha256DynamicConfig<F>, here: ((usize, usize), usize), zero_loaded: bool,
                        input: &[u8], precomputed_input_len: Option<usize>) -> Option<Fetched> {
    let ok = |rc: i32| if rc == sys::HSW_OK { Some(()) } else { None };
    if sha256.cur_hash_idx == 0 {
        // a new synthesis pass (= config.sha256.clone(), lib.rs:440): all cursors back, the region starts where the
        // Context stands now.  HSW_ERR_TOO_LARGE (more than 17 columns from this row) and friends => CPU path.
        ok(unsafe { sys::hsw_gadget_reset(be.gadget) })?;
        ok(unsafe { sys::hsw_gadget_set_origin(be.gadget, (here.0).0 as u64, (here.0).1 as u64, zero_loaded as i32, here.1 as u64) })?;
    } else if be.expect != Some(here) {
        // the circuit used the gate / range chips between two digests: the gadget's layout (fixed when the pass
        // started) no longer describes the region.  INTEGRATION.md section 3 -- the CPU path takes over.
        return None;
    }
    let mut view = unsafe { std::mem::zeroed::<sys::hsw_gadget_view>() };
    ok(unsafe { sys::hsw_gadget_streams(be.gadget, &mut view) })?;
    if (view.origin_column + view.columns) as usize > sha256.range().gate.basic_gates[0].len() { return None; }   // NUM_ADVICE too small: let the CPU path say so
    let h = sha256.cur_hash_idx;
    let mut r = unsafe { std::mem::zeroed::<sys::hsw_hash_result>() };
    ok(unsafe { sys::hsw_gadget_digest(be.gadget, input.as_ptr(), input.len(), precomputed_input_len.unwrap_or(0), &mut r) })?;
    // Debug builds re-check the region on the device before any cell is handed to halo2: every gate row, copy,
    // range bound, lookup entry and chip tie at the place the constraint structure expects it (0.3 ms for the
    // bench circuit -- not free: release builds skip it).
    #[cfg(debug_assertions)]
    {
        let mut rep = unsafe { std::mem::zeroed::<sys::hsw_verify_report>() };
        ok(unsafe { sys::hsw_gadget_verify(be.gadget, &mut rep) })?;
        if rep.violations != 0 { return None; }
    }
    let mut rc = unsafe { std::mem::zeroed::<sys::hsw_result_cells>() };
    ok(unsafe { sys::hsw_gadget_result_cells(be.gadget, h, &mut rc) })?;
    ok(unsafe { sys::hsw_gadget_streams(be.gadget, &mut view) })?;
    // this digest's gate cells: stream cells [prologue_cell, end_cell), column segment by column segment
    let pos = |cell: u64| -> Option<(u64, u64)> {
        let (mut c, mut r_) = (0u64, 0u64);
        ok(unsafe { sys::hsw_gadget_cell_position(be.gadget, cell, &mut c, &mut r_) })?;
        Some((c, r_))
    };
    // the new witnesses of everything assigned so far in this pass (this digest's are the tail of the array; a
    // circuit of many digests would fetch that tail alone -- the reference's circuits have one or two)
    let mut tape = unsafe { std::mem::zeroed::<sys::hsw_region_tape>() };
    ok(unsafe { sys::hsw_gadget_region_tape(be.gadget, &mut tape) })?;
    let mut n_distinct = 0usize;
    ok(unsafe { sys::hsw_gadget_download_region_distinct(be.gadget, be.distinct as *mut c_void, be.distinct_cells, &mut n_distinct) })?;
    let end = pos(r.end_cell - 1)?;
    let mut segs = Vec::new();
    let mut cell = r.prologue_cell;
    while cell < r.end_cell {
        let (col, row) = pos(cell)?;
        // cells of this column that belong to the digest: up to the column's last used row or the digest's end
        let n = if end.0 == col { (end.1 - row + 1) as usize } else { column_used_rows(be, col)? - row as usize };
        if n == 0 { return None; }
        segs.push((col, row, n));
        cell += n as u64;
    }
    Some(Fetched { r, rc, view, segs, end, tape })
}

/// halo2-base v0.2.x `AssignedValue` (halo2-pse feature): { cell, value, row_offset, context_id }.
fn assigned<F: PrimeField>(cell: Cell, v: F, row_offset: usize) -> AssignedValue<F> {
    AssignedValue { cell, value: Value::known(v), row_offset, context_id: 0 }
}

/// Rows of FlexGate column `col` that hold cells (the last rows of a column stay unassigned when the next call did
/// not fit: A3-iii): the row of the last stream cell placed in it, plus one.
fn column_used_rows(be: &Backend, col: u64) -> Option<usize> {
    let ok = |rc: i32| if rc == sys::HSW_OK { Some(()) } else { None };
    // binary search for the last stream cell whose position is in `col` (positions grow with the cell index)
    let mut view = unsafe { std::mem::zeroed::<sys::hsw_gadget_view>() };
    ok(unsafe { sys::hsw_gadget_streams(be.gadget, &mut view) })?;
    let (mut lo, mut hi) = (0u64, view.gate_capacity);       // invariant: position(lo).col <= col
    while hi - lo > 1 {
        let mid = (lo + hi) / 2;
        let (mut c, mut r) = (0u64, 0u64);
        ok(unsafe { sys::hsw_gadget_cell_position(be.gadget, mid, &mut c, &mut r) })?;
        if c <= col { lo = mid } else { hi = mid }
    }
    let (mut c, mut r) = (0u64, 0u64);
    ok(unsafe { sys::hsw_gadget_cell_position(be.gadget, lo, &mut c, &mut r) })?;
    if c != col { return None; }
    Some(r as usize + 1)
}
