// C-ABI of libfastmatch_hip.so (include/fastmatch_hip.h), part 4: Grid_Cache over a target given as PRE-EXTRACTED
// features -- which keypoints fall into which cell's crop.  Host code (one pass over the keypoints, no device work):
// the expander's target bank is "every cell's descriptors, cell after cell", and building that list in NumPy cost
// 6 ms per 12.5k-keypoint image (200 ms at 300k), fifteen times the device loop's share of a 64-pair launch.
#include "ctx_internal.h"

#include <algorithm>
#include <cmath>
#include <functional>
#include <new>
#include <vector>

using namespace fm;

namespace {

// Cells along one axis whose crop holds coordinate v.  The crop of cell i is [lo, hi) with
//   lo = i * cell - margin (0 for the first cell),  hi = lo + cell + 2 * margin (the image's limit for the last cell)
// -- the bounds Grid_Cache hands its caching function (cache.pyx:128-131).  At most kMaxSpan cells.
constexpr int kMaxSpan = 64;

inline int axis_cells(double v, int cell, int margin, int n_cells, int limit, int* out)
{
    // candidates: a cell's crop reaches `margin` to either side of the cell, so the cells of v - margin .. v + margin and
    // one more on each side (floating-point division at the edges, the first cell's clipped start); the test itself is
    // exact -- integer bounds against v
    const double f0 = std::floor((v - margin) / (double)cell), f1 = std::floor((v + margin) / (double)cell);
    if (!(f0 > -4.0e9 && f1 < 4.0e9)) return 0;                  // NaN / far outside: in no cell
    int64_t i0 = (int64_t)f0 - 1, i1 = (int64_t)f1 + 1;
    if (i0 < 0) i0 = 0;
    if (i1 > n_cells - 1) i1 = n_cells - 1;
    int k = 0;
    auto test = [&](int64_t i) {
        const int64_t lo = i * cell - (i > 0 ? margin : 0);
        const int64_t hi = (i + 1 < n_cells) ? lo + cell + 2 * (int64_t)margin : (int64_t)limit;
        if (v >= (double)lo && v < (double)hi && k < kMaxSpan) out[k++] = (int)i;
    };
    // the first cell's crop starts at 0 and keeps its width (it reaches 2 * margin past the cell), the last one ends at
    // the image's limit: both are tested whatever the range says
    if (i0 > i1) i0 = i1 = -1;                                    // (nothing in range)
    if (i0 != 0) test(0);
    for (int64_t i = i0 < 0 ? 1 : i0; i <= i1; ++i) test(i);
    if (n_cells > 1 && i1 != n_cells - 1) test(n_cells - 1);
    return k;
}

}  // namespace

extern "C" int fm_grid_pack_cells(const double* positions, int64_t n, int32_t width, int32_t height, int32_t cell_w, int32_t cell_h,
                                  int32_t rows, int32_t cols, int32_t margin, int64_t capacity, int64_t* cell_off, int64_t* n_rows,
                                  int32_t* src_row, double* target_pos)
{
    if (n < 0 || (n > 0 && !positions) || !cell_off || !n_rows)
        return fail(nullptr, FM_EINVAL, "fm_grid_pack_cells: NULL argument");
    if (cell_w < 1 || cell_h < 1 || rows < 1 || cols < 1 || margin < 0 || width < 0 || height < 0 || n > INT32_MAX)
        return fail(nullptr, FM_EINVAL, "fm_grid_pack_cells: bad geometry");
    if (2.0 * margin / cell_w + 4 > kMaxSpan || 2.0 * margin / cell_h + 4 > kMaxSpan)
        return fail(nullptr, FM_EUNSUPPORTED, "fm_grid_pack_cells: margin more than ~30 cells wide");
    const int64_t ncells = (int64_t)rows * cols;
  try {
    std::vector<int64_t> cnt((size_t)ncells + 1, 0);
    std::vector<int32_t> member;                  // the cells of keypoint 0, of keypoint 1, ... (pass 2 walks it again)
    std::vector<uint8_t> n_member((size_t)n);
    member.reserve((size_t)n * 4 + 16);
    int rx[kMaxSpan], cy[kMaxSpan];
    // pass 1: rows per cell (cell id = col * rows + row; "rows" counts cells along x, cache.pyx:41-42)
    for (int64_t p = 0; p < n; ++p) {
        const int nx = axis_cells(positions[2 * p], cell_w, margin, rows, width, rx);
        const int ny = nx ? axis_cells(positions[2 * p + 1], cell_h, margin, cols, height, cy) : 0;
        if (nx * ny > 255) return fail(nullptr, FM_EUNSUPPORTED, "fm_grid_pack_cells: a keypoint in more than 255 cells");
        n_member[(size_t)p] = (uint8_t)(nx * ny);
        for (int a = 0; a < nx; ++a)
            for (int b = 0; b < ny; ++b) {
                const int64_t c = (int64_t)cy[b] * rows + rx[a];
                ++cnt[(size_t)c];
                member.push_back((int32_t)c);
            }
    }
    int64_t total = 0;
    for (int64_t c = 0; c < ncells; ++c) { cell_off[c] = total; total += cnt[(size_t)c]; }
    cell_off[ncells] = total;
    *n_rows = total;
    if (!src_row || !target_pos || capacity < total) return FM_OK;      // (the caller looks at *n_rows and comes again)
    // pass 2: keypoints in ascending order, so every cell's rows are in ascending keypoint index
    for (int64_t c = 0; c < ncells; ++c) cnt[(size_t)c] = cell_off[c];
    size_t at_m = 0;
    for (int64_t p = 0; p < n; ++p) {
        const double x = positions[2 * p], y = positions[2 * p + 1];
        for (int k = 0; k < (int)n_member[(size_t)p]; ++k) {
            const int32_t c = member[at_m++];
            const int col = c / rows, row = c - col * rows;          // (32-bit: one short division per row of the packed bank)
            const int64_t at = cnt[(size_t)c]++;
            src_row[at] = (int32_t)p;
            // crop-local position (the crop starts at x_min = max(row * cell_w - margin, 0) along x), then the offset
            // match_position adds, which always subtracts the margin (fastmatch.pyx:157-158, cache.pyx:67-68)
            const double x_min = (double)((int64_t)row * cell_w - (row > 0 ? margin : 0));
            const double y_min = (double)((int64_t)col * cell_h - (col > 0 ? margin : 0));
            const double lx = x - x_min, ly = y - y_min;
            target_pos[2 * at]     = lx + (double)((int64_t)row * cell_w - margin);
            target_pos[2 * at + 1] = ly + (double)((int64_t)col * cell_h - margin);
        }
    }
  } catch (const std::bad_alloc&) {
    return fail(nullptr, FM_ENOMEM, "fm_grid_pack_cells: out of host memory");
  }
    return FM_OK;
}

// ---- fm_self_dist: the workgroup table of the triangular self sweep (kernel: rowreduce.hip, TRI) --------------------------
namespace fm {

// Workgroups of the triangular sweep of a bank of n_pad rows, in two launches.
//   A  every output chunk k (512 rows = 4 stages) against its own rows: the square block on the diagonal, masked.
//      After it every row's word of bound[] holds the best of 511 candidates.
//   B  chunk k against the stages from 4 k + 4 on, cut into pieces of S stages counted from there, piece-number
//      major: piece i of every chunk, then piece i + 1.  The column direction of a tile only pays when the
//      streamed rows' bounds are already good -- a row that is visited by many workgroups at once (a slice-major
//      order: 190 visits of 512 candidates each against the same stale word) fires ~1000 atomics per row instead
//      of a handful -- so a row's visits must be spread over the whole sweep: in this order the pieces that run
//      together stream rows at the same DISTANCE from their chunks, i.e. different rows, and a row meets its
//      partners in order of that distance from both directions at once.
// S = target, or (target 0) the S in 20 .. 72 whose pieces leave the last wave of 512 resident workgroups fullest (plan_tri).
static int tri_pieces(int nstages, int S, std::vector<int>* table, int* n_diag)
{
    const int nchunks = (nstages + 3) / 4;
    int n = 0;
    for (int k = 0; k < nchunks; ++k) {
        if (table) { table->push_back(k); table->push_back(4 * k); table->push_back(std::min(nstages, 4 * k + 4)); table->push_back(0); }
        ++n;
    }
    *n_diag = n;
    for (int i = 0; 4 + i * S < nstages; ++i)
        for (int k = 0; k < nchunks; ++k) {
            const int lo = 4 * k + 4 + i * S, hi = std::min(nstages, lo + S);
            if (lo >= hi) break;                       // (later chunks are shorter still)
            if (table) { table->push_back(k); table->push_back(lo); table->push_back(hi); table->push_back(0); }
            ++n;
        }
    return n;
}

TriPlan plan_tri(int64_t n_pad, int target, std::vector<int>* table)
{
    TriPlan pl;
    const int nstages = (int)(n_pad / kStageRows);
    pl.nchunks = (nstages + 3) / 4;
    pl.ncols_alloc = pl.nchunks * 512;
    int S = target;
    if (S <= 0) {
        // Piece length by arithmetic, O(chunks) per candidate (r05, last: the first version list-scheduled every candidate's
        // pieces on a heap -- 2.3 ms of host time for a 100k-row bank, four times the kernel it plans, 340 ms for 1M rows;
        // a dataset's images all differ in size, so the plan of a new size must cost microseconds).  Launch B's pieces are
        // dispatched round by round onto 512 resident workgroups and all but each chunk's last one take S stages (+ ~2 for
        // prologue and hand-over): what decides is how full the LAST wave of 512 is, so the candidate with the smallest
        // ceil(pieces / 512) * (S + 2), i.e. the least idle tail, is taken; ties go to the longer piece (fewer prologues).
        double best = 1e300;
        for (int c = 20; c <= 72; ++c) {
            long long np = 0;
            for (int k = 0; k < pl.nchunks; ++k) {
                const int rem = nstages - (4 * k + 4);
                if (rem <= 0) break;
                np += (rem + c - 1) / c;
            }
            const double waves = (double)((np + 511) / 512);
            const double cost = waves * (c + 2.0);
            if (cost <= best) { best = cost; S = c; }
        }
    }
    if (S < 4) S = 4;
    pl.stages = S;
    if (table) {
        table->clear();
        pl.npieces = tri_pieces(nstages, S, table, &pl.ndiag);
        return pl;
    }
    // the counts alone by arithmetic, in 64 bits (counted in an int by walking every piece, the count of a
    // bank beyond ~140M rows went negative; such a bank is kept off this sweep altogether: kTriMaxRows)
    long long np = pl.nchunks;
    for (int k = 0; k < pl.nchunks; ++k) {
        const int rem = nstages - (4 * k + 4);
        if (rem <= 0) break;
        np += (rem + S - 1) / S;
    }
    pl.ndiag = pl.nchunks;
    pl.npieces = np <= INT32_MAX ? (int)np : 0;         // (0: launch_rowreduce_tri refuses the plan)
    return pl;
}


}  // namespace fm
