// hsw_flush_bounds.hpp -- where a flush of the tile sits relative to the FlexGate column breaks of its block
// (hsw_expand.hpp flush_tile).  Plain integer functions, host- and device-callable, so that the property they
// must have -- "a flush / a row that is not placed piece by piece is shifted by exactly the gaps its cells have
// passed" -- is checked on the CPU over random geometries (tests/test_flush_bounds.py).  Round 2's placement
// bug lived here: the lower bound wrapped below zero for a block's first tile.
#ifndef HSW_FLUSH_BOUNDS_HPP
#define HSW_FLUSH_BOUNDS_HPP
#include <cstdint>
#if defined(__HIPCC__)
#define HSW_HD __host__ __device__ __forceinline__
#else
#define HSW_HD inline
#endif

namespace hsw {

constexpr uint32_t HSW_NO_BREAK = 0xffffffffu;

// Cells at block-local index >= brk1 / brk2 are shifted by gap1 / gap2 more cells (brk1 < brk2; none = HSW_NO_BREAK).
struct BlockBreaks { uint32_t brk1, gap1, brk2, gap2; };

HSW_HD uint32_t packed_cell_of(const BlockBreaks &b, uint32_t cl) {
    return cl + (cl >= b.brk1 ? b.gap1 : 0u) + (cl >= b.brk2 ? b.gap2 : 0u);
}

// Block-local cell held by LDS column 0 of row 0 of flush number fl (T cells per tile) of a phase-part whose
// first unit starts at cell_base and whose units start `skew` cells past a 128-byte line.  The block's very
// first tile (cell_base = 0, fl = 0) has nothing in its columns below `skew`: clamped, not wrapped.
HSW_HD uint32_t flush_lo_cell(uint32_t cell_base, uint32_t fl, uint32_t T, uint32_t skew) {
    const uint32_t lo_raw = cell_base + fl * T;
    return lo_raw >= skew ? lo_raw - skew : 0u;
}

// A run of cells [lo, hi): the shift all of them share, or straddles = true if a break lies inside it.
HSW_HD uint32_t run_shift(const BlockBreaks &b, uint32_t lo, uint32_t hi, bool &straddles) {
    straddles = false;
    if (b.brk1 == HSW_NO_BREAK || b.brk1 >= hi) return 0u;
    if (b.brk1 <= lo && b.brk2 >= hi) return b.gap1;
    if (b.brk2 <= lo) return b.gap1 + b.gap2;
    straddles = true;
    return 0u;
}

// The whole flush (rows 0 .. nrows-1, unit_cells apart; T columns + up to 8 carried / appended ones per row).
HSW_HD uint32_t flush_shift(const BlockBreaks &b, uint32_t lo_c, uint32_t nrows, uint32_t unit_cells, uint32_t T, bool &packed) {
    return run_shift(b, lo_c, lo_c + (nrows ? nrows - 1u : 0u) * unit_cells + T + 8u, packed);
}
// Row r of a flush that straddles a break.
HSW_HD uint32_t flush_row_shift(const BlockBreaks &b, uint32_t lo_c, uint32_t r, uint32_t unit_cells, uint32_t T, bool &strad) {
    const uint32_t row_lo = lo_c + r * unit_cells;
    return run_shift(b, row_lo, row_lo + T + 8u, strad);
}

}  // namespace hsw
#endif
