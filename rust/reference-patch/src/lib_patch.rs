//! The edits to the reference's `src/lib.rs` (NOT COMPILED here; shown as the code that changes, the rest of
//! the file stays as it is):
//!
//! 1. `pub mod hsw;` next to the other modules (lib.rs:1-4) and `hsw-sys = { path = "../hsw-sys" }` in
//!    `Cargo.toml`.
//! 2. The body of `digest` (lib.rs:77-348) moves, unchanged, into a private `digest_cpu` with the same
//!    signature; `digest` itself becomes the dispatcher below.  Signature, return type and error type are the
//!    reference's own (lib.rs:71-76, 31-36): nothing a user circuit sees changes.
//! 3. `spread.rs:20-27`: the fields `denses`, `spreads`, `num_bits_lookup`, `num_advice_columns`,
//!    `num_limb_sum`, `row_offset` of `SpreadConfig` become `pub(crate)` (the GPU path assigns the chip
//!    columns itself and advances the cursor the way `spread_limb` does, spread.rs:228-231).
impl<F: PrimeField> Sha256DynamicConfig<F> {
    pub fn digest<'a, 'b: 'a>(
        &'a mut self,
        ctx: &mut Context<'b, F>,
        input: &'a [u8],
        precomputed_input_len: Option<usize>,
    ) -> Result<AssignedHashResult<F>, Error> {
        // create_proof's witness pass: every advice cell of this digest comes from the GPU (hsw.rs).
        // Key generation and MockProver -- the passes that record / check selectors, fixed cells and copy
        // constraints -- run the reference's own code.
        // The region starts wherever `ctx` stands (hsw_gadget_set_origin); whatever the GPU side cannot do
        // (Ok(None): no device, more than 17 columns, a Context that moved between two digests) falls through
        // to the reference's own code on an untouched Context -- the GPU path never adds a failure mode.
        if crate::hsw::witness_only_pass(ctx, self) {
            if let Some(r) = crate::hsw::digest_gpu(self, ctx, input, precomputed_input_len)? {
                return Ok(r);
            }
        }
        self.digest_cpu(ctx, input, precomputed_input_len)
    }
}
