// Tile order of the persistent 256x256 GEMM launches (gemm256p.hip), round 5: WHICH output tile a workgroup computes in WHICH round.
//
// Round r of a persistent launch = gridDim tiles computed at the same time, gridDim / 8 of them on each XCD (workgroup b sits on XCD b % 8).
// Every K step a workgroup fetches one K slice of its A row panel and of its B column panel; slices requested by several workgroups of an
// XCD are served by its 4 MB L2 after the first request, everything else comes over the fabric from the Infinity Cache (256 MB, memory
// side) or from HBM.  The K loop tolerates ~1 K tile (1.3 us) of load latency (two 64 KiB K-tile buffers are all the LDS there is), which
// covers an Infinity-Cache hit and does NOT cover an HBM access under load: tools/gemm_ktile_probe.py - the same instruction stream runs a K
// tile in 1.29 us at [12288, 4096, 4096] (operands resident in the Infinity Cache) and in 1.6 - 1.9 us once A + B + C outgrow it.
//
// map 0 (rounds 1-4): XCD x owns a contiguous range of the tile list (8 tile rows at a time, column-major inside): its 32 tiles of a round
//   are 8 rows x 4 columns (12 panels, L2 hit rate 72-81 %), but the eight XCDs sit at eight unrelated places of the output: up to 96
//   panels = 192 MB (K = 4096) per round, re-read at distances beyond the Infinity Cache -> most L2 misses go to HBM.
// map 1 (this file): the 8 XCD blocks (4 rows x 8 columns each, 12 panels as before) of a round are STACKED into one super-block of
//   gridDim tiles - 32 rows x 8 columns, 16 x 16 or 8 x 32 by the height of the row band - so that they share their B panels (and the
//   launch walks a band column by column: the A panels of a band are re-read one round later, 32 + 8 panels = 80 MB in between).  Tiles of
//   ragged edges are simply fewer per block: the list stays dense (bijective), consecutive runs of gridDim / 8 tiles go to the XCDs, so
//   every round is full except the last.
#pragma once
#ifdef __HIPCC__
#define VLR_HD __host__ __device__
#else
#define VLR_HD
#endif

// linear tile index pl of a [tm x tn] tile grid -> (row, col): bands of 32, then 16, then 8 tile rows, then the remaining (< 8) rows;
// inside a band super-columns of 256 / h columns; inside a super-column XCD blocks of 4 rows x 8 columns, the blocks of one block
// column first (they share the B panels), a block's tiles row-fastest
VLR_HD inline void vlr_tile_of_shared(int pl, int tm, int tn, int* row, int* col, int rotate = 0) {
    int r0 = 0, h;
    for (;;) {
        const int left = tm - r0;
        h = left >= 32 ? 32 : left >= 16 ? 16 : left >= 8 ? 8 : left;
        const int cnt = h * tn;
        if (pl < cnt || left == h) break;
        pl -= cnt;
        r0 += h;
    }
    if (h < 8) { *row = r0 + pl % h; *col = pl / h; return; }
    const int nr = h >> 2;                 // XCD blocks stacked in a super-column
    const int W = 64 / nr;                 // its columns: 8, 16, 32 (h * W = 256 tiles)
    const int s0 = pl >> 8, l2 = pl & 255; // full super-columns hold exactly 256 tiles; only the last one of a band may be narrower
    const int s = rotate ? s0 : 0;         // (rotation of the rows, below)
    const int c0 = s0 * W, ws = (tn - c0) < W ? (tn - c0) : W;
    const int fb = ws >> 3, fullt = fb * h * 8;
    // rotate = 1: rows rotate with the super-column (block rows by s, the 4 rows inside a block by s too) and the 4 rows of an XCD block are
    // nr apart instead of adjacent (short tiles come in runs - the text rows of a sequence - and a block of adjacent rows is all short or all long).  The k-th run of a round always
    // goes to the same XCD and its j-th tile to the same workgroup, so WITHOUT the rotation a workgroup computes the same tile row in every
    // round of a band - fine when all tiles are equally long (and measured 1-3 ms better on the default step: the XCD's A panels survive
    // partly in its L2), wrong when they are not: adapter-segment launches whose all-text row tiles skip K tiles (vlr_gemm_seg_rowskip)
    // would last as long as the workgroups that own the long rows (the skip gained 0.7 ms without the rotation, 3-4 ms with it).
    if (l2 < fullt) {
        const int blk = l2 >> 5, t = l2 & 31;
        *row = rotate ? r0 + (blk + s) % nr + nr * ((t + s) & 3) : r0 + (blk % nr) * 4 + (t & 3);      // rotating form: an XCD block's 4 rows are nr apart
        *col = c0 + (blk / nr) * 8 + (t >> 2);
    } else {                               // the narrow last block column of the band (1 .. 7 columns)
        const int l3 = l2 - fullt, bs = 4 * (ws - fb * 8);
        const int t = l3 % bs;
        *row = rotate ? r0 + (l3 / bs + s) % nr + nr * ((t + s) & 3) : r0 + (l3 / bs) * 4 + (t & 3);
        *col = c0 + fb * 8 + (t >> 2);
    }
}
