//! GPU witness path for `Sha256DynamicConfig` (MI355X, `libhsw.so`).  NOT COMPILED in the build image.
//!
//! Usage inside `Circuit::synthesize`, replacing the body of the region closure for PROVING only
//! (key generation keeps `sha256.digest(..)`, lib.rs:455-466):
//!
//! ```ignore
//! let mut hsw = HswRegion::new(0, &config.sha256, true)?;          // once per prover thread
//! layouter.assign_region(|| "dynamic sha2 (gpu)", |mut region| {
//!     let digests = hsw.assign_witness(&mut region, &config.sha256, &self.test_inputs, &self.precomputed_input_lens)?;
//!     Ok(digests)
//! })?;
//! ```
use halo2_base::halo2_proofs::{
    circuit::{Region, Value},
    plonk::{Advice, Column, Error},
};
use halo2_ecc::fields::PrimeField;
use hsw_sys as sys;
use std::os::raw::c_void;
use std::ptr;

use crate::Sha256DynamicConfig;

fn check(rc: i32) -> Result<(), Error> {
    if rc == sys::HSW_OK { Ok(()) } else { Err(Error::Synthesis) }
}

/// One engine + one whole-digest gadget, laid out for the circuit's FlexGate columns.
pub struct HswRegion {
    engine: *mut sys::hsw_engine,
    gadget: *mut sys::hsw_gadget,
    max_rows: u64,
    columns: u64,
    chip_columns: usize,
    // pinned host images, reused by every synthesis
    gate: *mut [u64; 4],
    lookup: *mut [u64; 4],
    dense: *mut [u64; 4],
    spread: *mut [u64; 4],
    chip_col_stride: usize,
}

impl HswRegion {
    /// `sha256` must be configured like the CPU path: same `max_variable_byte_sizes`, table width
    /// (`num_bits_lookup`), number of spread columns and `is_input_range_check`.
    pub fn new<F: PrimeField>(device: i32, sha256: &Sha256DynamicConfig<F>, is_input_range_check: bool) -> Result<Self, Error> {
        let num_bits_lookup = sha256.spread_config.num_bits_lookup as u32;       // spread.rs:24
        let chip_columns = sha256.spread_config.num_advice_columns;              // spread.rs:25
        let max_rows = sha256.range().gate.max_rows as u64;                      // lib.rs:355
        let mut engine = ptr::null_mut();
        check(unsafe {
            sys::hsw_engine_create_ex(device, ptr::null_mut(), num_bits_lookup, chip_columns as u32,
                                      sys::HSW_MODE_HALO2_INTERNALS, &mut engine)
        })?;
        let sizes: Vec<usize> = sha256.max_variable_byte_sizes.clone();
        let mut gadget = ptr::null_mut();
        check(unsafe {
            sys::hsw_gadget_create_ex(engine, sizes.as_ptr(), sizes.len(), is_input_range_check as i32,
                                      sys::HSW_GADGET_WHOLE_DIGEST, &mut gadget)
        })?;
        let mut columns = 0u64;
        check(unsafe { sys::hsw_gadget_set_columns(gadget, max_rows, &mut columns) })?;
        // the circuit must have configured at least this many gate advice columns (RangeConfig::configure's NUM_ADVICE)
        if (columns as usize) > sha256.range().gate.basic_gates[0].len() {
            return Err(Error::Synthesis);
        }
        let mut view = unsafe { std::mem::zeroed::<sys::hsw_gadget_view>() };
        check(unsafe { sys::hsw_gadget_streams(gadget, &mut view) })?;
        let pinned = |cells: u64| -> Result<*mut [u64; 4], Error> {
            let mut p: *mut c_void = ptr::null_mut();
            check(unsafe { sys::hsw_host_alloc((cells.max(1) * 32) as usize, &mut p) })?;
            unsafe { ptr::write_bytes(p as *mut u8, 0, (cells.max(1) * 32) as usize) };   // unassigned rows are zero
            Ok(p as *mut [u64; 4])
        };
        let stride = view.chip_col_stride as u64;
        Ok(Self {
            engine, gadget, max_rows, columns, chip_columns,
            gate: pinned(columns * max_rows)?,
            lookup: pinned(view.lookup_capacity)?,
            dense: pinned(chip_columns as u64 * stride)?,
            spread: pinned(chip_columns as u64 * stride)?,
            chip_col_stride: stride as usize,
        })
    }

    /// Hashes `inputs` on the GPU (lib.rs:71-349 for each) and assigns EVERY advice cell of the region.
    /// Returns the digests; cell handles of input / output bytes are at the positions
    /// `hsw_gadget_cell_position` gives for `hsw_hash_result::{prologue_cell + 46.., epilogue_cell + ..}`.
    pub fn assign_witness<F: PrimeField>(&mut self, region: &mut Region<'_, F>, sha256: &Sha256DynamicConfig<F>,
                                         inputs: &[Vec<u8>], precomputed_input_lens: &[usize]) -> Result<Vec<[u8; 32]>, Error> {
        check(unsafe { sys::hsw_gadget_reset(self.gadget) })?;            // = config.sha256.clone(), lib.rs:440
        let mut digests = Vec::with_capacity(inputs.len());
        for (input, pre) in inputs.iter().zip(precomputed_input_lens) {
            let mut r = unsafe { std::mem::zeroed::<sys::hsw_hash_result>() };
            check(unsafe { sys::hsw_gadget_digest(self.gadget, input.as_ptr(), input.len(), *pre, &mut r) })?;
            digests.push(r.output_bytes);
        }
        let dst = sys::hsw_region_host {
            gate: self.gate as *mut c_void, lookup: self.lookup as *mut c_void,
            chip_dense: self.dense as *mut c_void, chip_spread: self.spread as *mut c_void,
        };
        check(unsafe { sys::hsw_gadget_download_region(self.gadget, &dst) })?;
        let mut view = unsafe { std::mem::zeroed::<sys::hsw_gadget_view>() };
        check(unsafe { sys::hsw_gadget_streams(self.gadget, &mut view) })?;

        // canonical little-endian limbs -> F (with HSW_REPR_MONTGOMERY the cells could be transmuted instead)
        let fe = |cell: &[u64; 4]| -> F {
            let mut repr = F::Repr::default();
            for (i, limb) in cell.iter().enumerate() {
                repr.as_mut()[8 * i..8 * i + 8].copy_from_slice(&limb.to_le_bytes());
            }
            F::from_repr(repr).unwrap()
        };
        let mut assign = |column: Column<Advice>, cells: *const [u64; 4], rows: usize| -> Result<(), Error> {
            for row in 0..rows {
                let v = fe(unsafe { &*cells.add(row) });
                region.assign_advice(|| "hsw", column, row, || Value::known(v))?;
            }
            Ok(())
        };
        // FlexGate advice columns (basic_gates[0][c].value): column c of the image, used rows only
        let (mut last_col, mut last_row) = (0u64, 0u64);
        if view.gate_cells > 0 {
            check(unsafe { sys::hsw_gadget_cell_position(self.gadget, view.gate_cells - 1, &mut last_col, &mut last_row) })?;
        }
        for c in 0..=last_col as usize {
            let rows = if (c as u64) < last_col { self.max_rows as usize } else { last_row as usize + 1 };
            assign(sha256.range().gate.basic_gates[0][c].value, unsafe { self.gate.add(c * self.max_rows as usize) }, rows)?;
        }
        // the lookup-advice column RangeConfig::finalize would fill (lib.rs:469)
        assign(sha256.range().lookup_advice[0][0], self.lookup, view.lookup_cells as usize)?;
        // SpreadConfig's chip columns (spread.rs:20-21): row r of column c is limb call r * columns + c
        let chip_rows = ((view.num_limb_sum + self.chip_columns as u64 - 1) / self.chip_columns as u64) as usize;
        for c in 0..self.chip_columns {
            assign(sha256.spread_config.denses[c], unsafe { self.dense.add(c * self.chip_col_stride) }, chip_rows)?;
            assign(sha256.spread_config.spreads[c], unsafe { self.spread.add(c * self.chip_col_stride) }, chip_rows)?;
        }
        let _ = self.columns;
        Ok(digests)
    }
}

impl Drop for HswRegion {
    fn drop(&mut self) {
        unsafe {
            sys::hsw_host_free(self.gate as *mut c_void);
            sys::hsw_host_free(self.lookup as *mut c_void);
            sys::hsw_host_free(self.dense as *mut c_void);
            sys::hsw_host_free(self.spread as *mut c_void);
            sys::hsw_gadget_destroy(self.gadget);
            sys::hsw_engine_destroy(self.engine);
        }
    }
}
