// hsw_tape.hpp -- the sequence of halo2-base assign_region call LENGTHS of one
// block's gate stream (1 for load_witness, 4 for add / neg / mul_add, 4 for the
// inner product of a range_check(a, 32) in internals mode).
//
// FlexGate never lets one call straddle a column (assumption A3: halo2-lib
// v0.2.x assign_region moves to the next column when row + len >= max_rows),
// so packing the linear stream into advice columns needs the call boundaries.
// The sequence is input-independent; it is built compositionally from the same
// call structure as the kernel (reference lines cited per function) and is
// checked against the oracle's per-cell tape in tests/test_tape_and_packing.py.
#ifndef HSW_TAPE_HPP
#define HSW_TAPE_HPP

#include <cstdint>
#include <vector>

namespace hsw {

class TapeBuilder {
  public:
    TapeBuilder(int limbs, bool internals) : L(limbs), rc(internals) {}
    std::vector<uint8_t> block() {                       // compression.rs:19-213
        lens.clear();
        for (int w = 0; w < 16; w++) rep(4, 4);          // :31-47  4 mul_add per word
        for (int w = 0; w < 16; w++) state_to_spread();  // :53-56
        for (int i = 16; i < 64; i++) {                  // :57-96
            sigma(); sigma();
            rep(3, 4);                                   // three add
            mod_u32();
            state_to_spread();
        }
        for (int i = 0; i < 6; i++) state_to_spread();   // :109-115
        for (int r = 0; r < 64; r++) {                   // :125-196
            sigma(); ch(); rep(4, 4); mod_u32();
            sigma(); maj(); rep(1, 4); mod_u32();
            rep(1, 4); mod_u32(); state_to_spread();
            rep(1, 4); mod_u32(); state_to_spread();
        }
        for (int i = 0; i < 8; i++) { rep(1, 4); mod_u32(); }   // :197-212
        return lens;
    }

  private:
    int L;
    bool rc;
    std::vector<uint8_t> lens;
    void rep(int n, int len) { for (int i = 0; i < n; i++) lens.push_back((uint8_t)len); }
    void range_check32() { if (rc) rep(1, 4); }          // [limb0, limb1, 2^16, a]
    void spread() {                                      // spread.rs:76-123
        rep(L, 1);                                       // limbs
        rep(L, 4);                                       // limb sum
        for (int j = 0; j < L; j++) { rep(1, 1); rep(1, 4); }   // spread_limb + mul_add
    }
    void state_to_spread() { rep(2, 1); rep(1, 4); spread(); spread(); }   // compression.rs:215-246
    void mod_u32() { rep(2, 1); range_check32(); rep(1, 4); }              // compression.rs:266-295
    void recheck() { spread(); spread(); rep(1, 4); }                      // :344-354 and siblings
    void sigma() {                                       // compression.rs:702-882
        rep(4, 1); rep(3, 4); rep(1, 4); rep(4, 4);
        rep(2, 1); range_check32(); range_check32(); rep(1, 4);
        rep(4, 1); recheck(); recheck(); rep(1, 4);
    }
    void ch() {                                          // compression.rs:297-405
        rep(2, 4); rep(2, 4); rep(4, 4);                 // 2 add, 2 neg, 2 three_add
        rep(8, 1);
        recheck(); recheck(); recheck(); recheck();
        rep(2, 4); rep(1, 4);
    }
    void maj() {                                         // compression.rs:460-519
        rep(4, 4); rep(4, 1); recheck(); recheck(); rep(1, 4);
    }
};

}  // namespace hsw
#endif
