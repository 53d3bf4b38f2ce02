// Row/column reordering into compact tiles and the tile-local storage of the fine-level operator -- declarations.
// (SURVEY.md 7 "x-gather locality": permute on upload, un-permute outputs inside the ABI.)
#pragma once
#include "wae_internal.h"

struct TilePlan {
    std::vector<int> perm;       // new -> old row/column index (empty: identity, no tiles)
    std::vector<int> iperm;      // old -> new
    std::vector<int> row_ptr;    // tile t owns the new rows row_ptr[t] .. row_ptr[t+1]-1  (<= tile_rows each)
    int wmax = 0;                // largest window (distinct columns touched by a tile's rows)
};

// Pattern of sum_k |A_k| (N orientation), rows -> sorted columns.
struct Pattern {
    int64_t n = 0;
    std::vector<int> ptr, col;
};
Pattern union_pattern(const std::vector<CsrZ> &planes);

// Nested breadth-first ordering (three stages: shells of a pseudo-peripheral BFS, strips inside a shell, bricks inside a
// strip) followed by the cut into tiles of at most tile_rows rows whose window fits wcap columns; rows inside a tile are
// sorted by decreasing length.  Returns an empty plan (identity) when the matrix is too small or a single row exceeds wcap.
TilePlan plan_tiles(const Pattern &U, int tile_rows, int wcap, int thick);

// B = P A P^T with P the permutation new -> old (columns re-sorted)
CsrZ permute_symmetric(const CsrZ &A, const std::vector<int> &perm, const std::vector<int> &iperm);

// Tile-local storage of one pattern group: per (tile, wavefront) a slice of 64 / lpr rows, lpr lanes per row (lane lpr i + h
// holds the entries h, h + lpr, ... of row i), padded to 1 / lpr of the longest row of the slice and stored entry-major
// ([k][lane]) so that the lanes read coalesced; column indices are 16-bit positions in the tile's window.
constexpr int TILE_SLICES = 8;                 // wavefronts per tile (default; build_tile_group takes the number)
struct TileGroupHost {
    std::vector<int> sptr;                 // slices*ntiles + 1 entry offsets (multiples of 64)
    std::vector<unsigned short> sidx;      // local column index per entry
    std::vector<double> svals;             // [entry][nplanes] doubles (real group) or [entry][nplanes][2]
    std::vector<unsigned short> dslot;     // per row: window slot of the row's own column, 0xFFFF if the row has no such entry
};
struct TileWindows {
    std::vector<int> win_ptr, win_cols;    // per tile: sorted distinct (new) columns of all groups
};
TileWindows build_windows(const Pattern &U, const std::vector<int> &row_ptr);
// mats: the planes of one pattern group (same pattern); is_real: store real parts only
TileGroupHost build_tile_group(const std::vector<const CsrZ *> &mats, bool is_real, const std::vector<int> &row_ptr, const TileWindows &W, int lpr,
                               int slices = TILE_SLICES);
