// Property check of csrc/hsw_flush_bounds.hpp on the CPU (tests/test_flush_bounds.py builds and runs it):
// for random tile geometries and column breaks, every cell a flush may write must land where the cell-by-cell
// placement (packed_cell_of) puts it whenever the flush -- or the row -- is declared "shifted as a whole".
//   argv[1] = number of random geometries, argv[2] = seed, argv[3] = "wrap" to use the pre-fix lower bound
//   (cell_base + fl*T - skew in unsigned arithmetic) and so demonstrate that the check catches it.
#include <cstdio>
#include <cstdlib>
#include <random>

#include "hsw_flush_bounds.hpp"

using namespace hsw;

int main(int argc, char **argv) {
    const long n = argc > 1 ? atol(argv[1]) : 200000;
    std::mt19937_64 rng(argc > 2 ? atoll(argv[2]) : 1);
    const bool wrap = argc > 3 && argv[3][0] == 'w';
    auto rnd = [&](uint64_t lo, uint64_t hi) { return (uint32_t)(lo + rng() % (hi - lo + 1)); };
    long bad = 0, whole = 0, rows_whole = 0, strads = 0;
    const uint32_t G = 69348;
    for (long it = 0; it < n; it++) {
        const uint32_t T = (uint32_t[]){32, 64, 128}[rnd(0, 2)];
        const uint32_t unit_cells = rnd(8, 800), nrows = rnd(1, 2048 / T), skew = rnd(0, 3);
        // a phase-part somewhere in the block; one time in four the block's very first one
        const uint32_t cell_base = rnd(0, 3) == 0 ? 0u : rnd(0, G - 1);
        const uint32_t tiles = (unit_cells + T - 1) / T, fl = rnd(0, tiles - 1);
        BlockBreaks b{HSW_NO_BREAK, 0, HSW_NO_BREAK, 0};
        const uint32_t kind = rnd(0, 9);
        if (kind >= 1) {                                     // aim the first break at this flush's neighbourhood
            const uint32_t around = cell_base + fl * T;
            b.brk1 = rnd(0, 3) ? rnd(around > 300 ? around - 300 : 1, around + nrows * unit_cells + 300) : rnd(1, G - 1);
            b.gap1 = rnd(1, 9);
            if (kind >= 7) { b.brk2 = b.brk1 + rnd(1, 3000); b.gap2 = rnd(1, 9); }
        }
        const uint32_t lo_c = wrap ? cell_base + fl * T - skew : flush_lo_cell(cell_base, fl, T, skew);
        bool packed = false;
        const uint32_t shift = flush_shift(b, lo_c, nrows, unit_cells, T, packed);
        // cells row r may write in this flush: LDS columns [c0, T + 4) -- columns below skew of the first tile are
        // empty, the 4 past T are the appended heads of the next unit -- column c holding cell base_r + c - skew
        for (uint32_t r = 0; r < nrows; r++) {
            const uint32_t c0 = fl == 0 ? skew : 0u;
            bool strad = false;
            const uint32_t rs = packed ? flush_row_shift(b, lo_c, r, unit_cells, T, strad) : shift;
            if (packed && strad) { strads++; continue; }     // placed piece by piece: nothing to check here
            (packed ? rows_whole : whole)++;
            for (uint32_t c = c0; c < T + 4; c++) {
                const uint32_t cell = cell_base + fl * T + r * unit_cells + c - skew;   // (>= cell_base: c >= skew when fl = 0 ... or a carried cell)
                if (packed_cell_of(b, cell) != cell + rs) {
                    if (bad++ < 5)
                        printf("MISMATCH T=%u unit=%u nrows=%u skew=%u cell_base=%u fl=%u brk=(%u,+%u,%u,+%u) row %u column %u: cell %u -> %u, flush says +%u\n",
                               T, unit_cells, nrows, skew, cell_base, fl, b.brk1, b.gap1, b.brk2, b.gap2, r, c, cell,
                               packed_cell_of(b, cell), rs);
                }
            }
        }
    }
    printf("%ld geometries: %ld rows of whole flushes, %ld rows shifted on their own, %ld straddling rows, %ld mismatches\n",
           n, whole, rows_whole, strads, bad);
    return bad ? 1 : 0;
}
