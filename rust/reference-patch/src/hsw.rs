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
//!     if hsw::witness_only_pass(ctx, self)? {
//!         return hsw::digest_gpu(self, ctx, input, precomputed_input_len);      // this file
//!     }
//!     self.digest_cpu(ctx, input, precomputed_input_len)                          // the reference's body, unchanged
//! }
//! ```
//!
//! WHEN the GPU path may run.  Selectors, fixed cells and copy constraints do not depend on the witness;
//! halo2 records them during key generation (`keygen_vk` / `keygen_pk`) and `MockProver::run` checks them.
//! Those passes keep the CPU path.  `create_proof` synthesises against a `WitnessCollection`, which listens
//! to `assign_advice` ONLY (`assign_fixed`, `copy`, `enable_selector` are no-ops there): that pass -- the one
//! `benches/digest.rs:143-156` times -- takes every advice cell of the region from the GPU.
//! `witness_only_pass` tells the passes apart from inside the gadget (ASSUMPTION A5, halo2_proofs PSE
//! v2023_02_02 `plonk/keygen.rs`, `plonk/prover.rs`, `dev.rs`; unpinned like A1-A4):
//!
//! | backend            | `assign_advice` closure | `assign_fixed` closure | => value known? (advice, fixed) |
//! |--------------------|-------------------------|------------------------|---------------------------------|
//! | keygen `Assembly`  | not called              | called                 | (no, yes)  -> CPU                |
//! | `MockProver`       | called                  | called                 | (yes, yes) -> CPU                |
//! | `WitnessCollection`| called                  | not called             | (yes, no)  -> GPU                |
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

static FORCE: AtomicI8 = AtomicI8::new(-1);          // -1 = detect, 0 = CPU, 1 = GPU

/// Override the pass detection for the whole process (`None` = detect again).
pub fn force(gpu: Option<bool>) {
    FORCE.store(match gpu { None => -1, Some(false) => 0, Some(true) => 1 }, Ordering::SeqCst);
}

/// Is this synthesis pass `create_proof`'s witness collection (A5 above)?  Probes with two assignments that
/// change nothing: advice cell (gate column 0, row 0) is assigned by `digest` itself right afterwards
/// (`load_witness(input_byte_size)`, lib.rs:124-125, is the first cell of the region when the gadget opens it;
/// otherwise the probe goes to the next free row, `ctx.advice_alloc[0][0].1`, which is also assigned next), and
/// the fixed cell gets 0, what an unassigned fixed cell holds (and what `finalize` overwrites if it needs the row).
pub fn witness_only_pass<F: PrimeField>(ctx: &mut Context<'_, F>, sha256: &Sha256DynamicConfig<F>) -> Result<bool, Error> {
    if std::env::var_os("HSW_DISABLE").is_some() { return Ok(false); }
    match FORCE.load(Ordering::SeqCst) { 0 => return Ok(false), 1 => return Ok(true), _ => {} }
    let gate = &sha256.range().gate;
    let (col, row) = ctx.advice_alloc[0];                                 // (column index, next free row) of context 0
    let adv: Column<Advice> = gate.basic_gates[0][col].value;
    let fix: Column<Fixed> = gate.constants[0];
    let a = ctx.region.assign_advice(|| "hsw probe", adv, row, || Value::known(F::zero()))?;
    let f = ctx.region.assign_fixed(|| "hsw probe", fix, 0, || Value::known(F::zero()))?;
    let known = |v: Value<&F>| { let mut k = false; v.map(|_| k = true); k };
    Ok(known(a.value()) && !known(f.value()))
}

/// One engine + one whole-digest gadget per (prover thread, circuit shape), reused by every synthesis.
struct Backend {
    engine: *mut sys::hsw_engine,
    gadget: *mut sys::hsw_gadget,
    key: (Vec<usize>, usize, usize, bool, u64),       // max sizes, table bits, chip columns, range checks, max_rows
    max_rows: u64,
    chip_columns: usize,
    chip_col_stride: usize,
    // pinned host staging (hsw_host_alloc), sized for the largest digest of the circuit
    stage: *mut [u64; 4],
    stage_cells: usize,
}

thread_local! { static BACKEND: RefCell<Option<Backend>> = RefCell::new(None); }

impl Backend {
    fn new<F: PrimeField>(sha256: &Sha256DynamicConfig<F>, key: (Vec<usize>, usize, usize, bool, u64)) -> Result<Self, Error> {
        let device = std::env::var("HSW_DEVICE").ok().and_then(|s| s.parse().ok()).unwrap_or(0);
        let mut engine = ptr::null_mut();
        check(unsafe { sys::hsw_engine_create_ex(device, ptr::null_mut(), key.1 as u32, key.2 as u32,
                                                 sys::HSW_MODE_HALO2_INTERNALS, &mut engine) })?;
        let mut gadget = ptr::null_mut();
        check(unsafe { sys::hsw_gadget_create_ex(engine, key.0.as_ptr(), key.0.len(), key.3 as i32,
                                                 sys::HSW_GADGET_WHOLE_DIGEST, &mut gadget) })?;
        // halo2curves' in-memory Fr IS the cell format: no from_repr (a Montgomery multiplication) per cell
        check(unsafe { sys::hsw_gadget_set_repr(gadget, sys::HSW_REPR_MONTGOMERY) })?;
        let mut columns = 0u64;
        check(unsafe { sys::hsw_gadget_set_columns(gadget, key.4, &mut columns) })?;
        if (columns as usize) > sha256.range().gate.basic_gates[0].len() { return Err(Error::Synthesis); }   // NUM_ADVICE too small
        let mut view = unsafe { std::mem::zeroed::<sys::hsw_gadget_view>() };
        check(unsafe { sys::hsw_gadget_streams(gadget, &mut view) })?;
        let biggest = key.0.iter().copied().max().unwrap_or(64) / 64;
        let stage_cells = biggest * 70_000 + 8 * biggest * 64 + 4096;          // a digest's gate cells (69,348 per block + frame)
        let mut p: *mut c_void = ptr::null_mut();
        check(unsafe { sys::hsw_host_alloc(stage_cells * 32, &mut p) })?;
        Ok(Self { engine, gadget, key, max_rows: view.max_rows, chip_columns: sha256.spread_config.num_advice_columns,
                  chip_col_stride: view.chip_col_stride, stage: p as *mut [u64; 4], stage_cells })
    }
}

impl Drop for Backend {
    fn drop(&mut self) {
        unsafe { sys::hsw_host_free(self.stage as *mut c_void); sys::hsw_gadget_destroy(self.gadget); sys::hsw_engine_destroy(self.engine); }
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
/// the CPU path under A1-A4; returns the same `AssignedHashResult`.
pub fn digest_gpu<'a, 'b: 'a, F: PrimeField>(sha256: &'a mut Sha256DynamicConfig<F>, ctx: &mut Context<'b, F>, input: &'a [u8],
                                             precomputed_input_len: Option<usize>) -> Result<AssignedHashResult<F>, Error> {
    let key = (sha256.max_variable_byte_sizes.clone(), sha256.spread_config.num_bits_lookup, sha256.spread_config.num_advice_columns,
               sha256.is_input_range_check, sha256.range().gate.max_rows as u64);
    BACKEND.with(|slot| {
        let mut slot = slot.borrow_mut();
        if slot.as_ref().map(|b| b.key != key).unwrap_or(true) { *slot = Some(Backend::new(sha256, key.clone())?); }
        let be = slot.as_mut().unwrap();
        // The gadget lays the region out from (column 0, row 0): the reference's circuits open the region with
        // their first digest (lib.rs:454-459, benches/digest.rs:92-93).  A circuit that assigns other cells
        // first must pass its start row to hsw_pack_plan_query / hsw_witness_blocks_ex instead.
        if sha256.cur_hash_idx == 0 {
            if ctx.advice_alloc[0] != (0, 0) { return Err(Error::Synthesis); }
            check(unsafe { sys::hsw_gadget_reset(be.gadget) })?;                 // = config.sha256.clone(), lib.rs:440
        }
        let h = sha256.cur_hash_idx;
        let mut r = unsafe { std::mem::zeroed::<sys::hsw_hash_result>() };
        check(unsafe { sys::hsw_gadget_digest(be.gadget, input.as_ptr(), input.len(), precomputed_input_len.unwrap_or(0), &mut r) })?;
        // Debug builds re-check the region on the device before any cell is handed to halo2: every gate row, copy,
        // range bound, lookup entry and chip tie at the place the constraint structure expects it (0.3 ms for the
        // bench circuit).
        #[cfg(debug_assertions)]
        {
            let mut rep = unsafe { std::mem::zeroed::<sys::hsw_verify_report>() };
            check(unsafe { sys::hsw_gadget_verify(be.gadget, &mut rep) })?;
            if rep.violations != 0 { return Err(Error::Synthesis); }
        }
        let mut rc = unsafe { std::mem::zeroed::<sys::hsw_result_cells>() };
        check(unsafe { sys::hsw_gadget_result_cells(be.gadget, h, &mut rc) })?;
        let mut view = unsafe { std::mem::zeroed::<sys::hsw_gadget_view>() };
        check(unsafe { sys::hsw_gadget_streams(be.gadget, &mut view) })?;

        // ---- this digest's gate cells: stream cells [prologue_cell, end_cell), column segment by column segment
        let gate_cols: Vec<Column<Advice>> = sha256.range().gate.basic_gates[0].iter().map(|g| g.value).collect();
        let mut want: Vec<(u64, usize)> = vec![(rc.input_len_cell, usize::MAX)];            // (stream cell, slot in `got`)
        want.extend((0..rc.n_input_bytes).map(|i| (rc.input_bytes_cell0 + i, usize::MAX)));
        want.extend(rc.output_byte_cells.iter().map(|&c| (c, usize::MAX)));
        let mut got: Vec<Option<AssignedCell<F, F>>> = vec![None; want.len()];
        for (k, w) in want.iter_mut().enumerate() { w.1 = k; }
        want.sort_unstable();
        let mut next_want = 0usize;
        let (mut cell, end) = (r.prologue_cell, r.end_cell);
        while cell < end {
            let (mut col, mut row) = (0u64, 0u64);
            check(unsafe { sys::hsw_gadget_cell_position(be.gadget, cell, &mut col, &mut row) })?;
            // cells of this column that belong to the digest: up to the column's last used row or the digest's end
            let (mut ecol, mut erow) = (0u64, 0u64);
            check(unsafe { sys::hsw_gadget_cell_position(be.gadget, end - 1, &mut ecol, &mut erow) })?;
            let n = if ecol == col { (erow - row + 1) as usize } else { column_used_rows(be, col)? - row as usize };
            assert!(n <= be.stage_cells);
            check(unsafe { sys::hsw_download(be.engine, be.stage as *mut c_void,
                                             (view.d_gate as *const u8).add(((col * be.max_rows + row) * 32) as usize) as *const c_void, n * 32) })?;
            for i in 0..n {
                let v: F = fe(unsafe { &*be.stage.add(i) });
                let a = ctx.region.assign_advice(|| "hsw", gate_cols[col as usize], row as usize + i, || Value::known(v))?;
                while next_want < want.len() && want[next_want].0 == cell + i as u64 { got[want[next_want].1] = Some(a.clone()); next_want += 1; }
            }
            cell += n as u64;
        }
        // ---- the spread-chip columns of this digest's blocks (spread.rs:196-233): rows [cursor0 / c, (cursor0 + limbs) / c)
        let c = be.chip_columns as u64;
        let limbs = r.n_blocks as u64 * 2060 * (16 / sha256.spread_config.num_bits_lookup as u64);   // 2,060 spread() calls per block x limbs
        let (row0, row1) = (r.spread_cursor0 / c, (r.spread_cursor0 + limbs + c - 1) / c);
        for k in 0..be.chip_columns {
            for (cols, base) in [(&sha256.spread_config.denses, view.d_chip_dense), (&sha256.spread_config.spreads, view.d_chip_spread)] {
                let n = (row1 - row0) as usize;
                check(unsafe { sys::hsw_download(be.engine, be.stage as *mut c_void,
                                                 (base as *const u8).add((k * be.chip_col_stride + row0 as usize) * 32) as *const c_void, n * 32) })?;
                for i in 0..n {
                    let v: F = fe(unsafe { &*be.stage.add(i) });
                    ctx.region.assign_advice(|| "hsw chip", cols[k], row0 as usize + i, || Value::known(v))?;
                }
            }
        }
        // ---- the lookup-advice column: RangeConfig::finalize (lib.rs:469) copies ctx.cells_to_lookup into it, in
        // queue order, after the circuit's last gadget.  Queue this digest's entries with their VALUES (the cell
        // handle only matters to `copy`, a no-op in this pass): finalize then fills the column exactly as in the
        // CPU path, also when the circuit queues lookups of its own after the gadget.
        let n = (r.epilogue_lookup + 64 - r.prologue_lookup) as usize;        // prologue | blocks | 64 epilogue entries (2 per digest byte)
        check(unsafe { sys::hsw_download(be.engine, be.stage as *mut c_void,
                                         (view.d_lookup as *const u8).add((r.prologue_lookup * 32) as usize) as *const c_void, n * 32) })?;
        let any_cell: Cell = got[0].as_ref().unwrap().cell();
        for i in 0..n {
            let v: F = fe(unsafe { &*be.stage.add(i) });
            ctx.cells_to_lookup.push(assigned(any_cell, v, 0));
        }
        // ---- the Context's own bookkeeping, as the CPU path leaves it: next free (column, row), the cached zero cell
        let (mut ncol, mut nrow) = (0u64, 0u64);
        check(unsafe { sys::hsw_gadget_cell_position(be.gadget, r.end_cell - 1, &mut ncol, &mut nrow) })?;
        ctx.advice_alloc[0] = (ncol as usize, nrow as usize + 1);
        ctx.total_advice += (r.end_cell - r.prologue_cell) as usize;
        if ctx.zero_cell.is_none() {                       // load_zero caches one cell per Context (A4-iii)
            ctx.zero_cell = Some(assigned(any_cell, F::zero(), 0));
        }
        sha256.cur_hash_idx += 1;                          // lib.rs:347
        sha256.spread_config.num_limb_sum += limbs as usize;                 // spread.rs:228
        sha256.spread_config.row_offset = ((r.spread_cursor0 + limbs) / c) as usize;   // spread.rs:229-231

        let pick = |slot: usize| -> AssignedValue<F> {
            let a = got[slot].as_ref().expect("result cell inside the digest's stream");
            let mut v = F::zero();
            a.value().map(|x| v = *x);
            assigned(a.cell(), v, 0)
        };
        Ok(AssignedHashResult {
            input_len: pick(0),                                                           // lib.rs:124-125
            input_bytes: (0..rc.n_input_bytes as usize).map(|i| pick(1 + i)).collect(),    // lib.rs:170-173
            output_bytes: (0..32).map(|i| pick(1 + rc.n_input_bytes as usize + i)).collect(),   // lib.rs:317-324
        })
    })
}

/// halo2-base v0.2.x `AssignedValue` (halo2-pse feature): { cell, value, row_offset, context_id }.
fn assigned<F: PrimeField>(cell: Cell, v: F, row_offset: usize) -> AssignedValue<F> {
    AssignedValue { cell, value: Value::known(v), row_offset, context_id: 0 }
}

/// Rows of column `col` that hold cells (the last rows of a column stay unassigned when the next call did not
/// fit: A3-iii): the row of the last stream cell placed in it, plus one.
fn column_used_rows(be: &Backend, col: u64) -> Result<usize, Error> {
    // binary search for the last stream cell whose position is in `col`
    let mut view = unsafe { std::mem::zeroed::<sys::hsw_gadget_view>() };
    check(unsafe { sys::hsw_gadget_streams(be.gadget, &mut view) })?;
    let (mut lo, mut hi) = (0u64, view.gate_capacity);       // invariant: position(lo).col <= col
    while hi - lo > 1 {
        let mid = (lo + hi) / 2;
        let (mut c, mut r) = (0u64, 0u64);
        check(unsafe { sys::hsw_gadget_cell_position(be.gadget, mid, &mut c, &mut r) })?;
        if c <= col { lo = mid } else { hi = mid }
    }
    let (mut c, mut r) = (0u64, 0u64);
    check(unsafe { sys::hsw_gadget_cell_position(be.gadget, lo, &mut c, &mut r) })?;
    debug_assert_eq!(c, col);
    Ok(r as usize + 1)
}
