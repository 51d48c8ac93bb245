// ASAN / UBSAN harness for the host-only cell planner (fast-match_amd/csrc/api_grid.hip): random geometries and keypoints,
// including NaN / infinite / far-away coordinates, capacities that are too small, degenerate grids.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>
#include "ctx_internal.h"
int fm::fail(fm_ctx*, int code, const std::string&) { return code; }
extern "C" int fm_grid_pack_cells(const double*, int64_t, int32_t, int32_t, int32_t, int32_t, int32_t, int32_t, int32_t, int64_t, int64_t*, int64_t*, int32_t*, double*);
int main()
{
    std::mt19937_64 rng(12345);
    long checked = 0;
    for (int it = 0; it < 4000; ++it) {
        const int w = 1 + (int)(rng() % 1500), h = 1 + (int)(rng() % 1500);
        const int cw = 1 + (int)(rng() % 130), ch = 1 + (int)(rng() % 130), margin = (int)(rng() % 70);
        const int rows = w / cw + 1, cols = h / ch + 1;
        const int64_t n = (int64_t)(rng() % 3000);
        std::vector<double> pos((size_t)n * 2);
        for (auto& v : pos) {
            const unsigned k = (unsigned)(rng() % 64);
            const double u = (double)(rng() % 2000000) / 1000.0 - 200.0;
            v = k == 0 ? NAN : k == 1 ? INFINITY : k == 2 ? -INFINITY : k == 3 ? 1e300 : k == 4 ? std::floor(u) : u;
        }
        std::vector<int64_t> off((size_t)rows * cols + 1);
        int64_t nt = -1;
        int rc = fm_grid_pack_cells(pos.data(), n, w, h, cw, ch, rows, cols, margin, 0, off.data(), &nt, nullptr, nullptr);
        if (rc == FM_EUNSUPPORTED || rc == FM_EINVAL) continue;
        if (rc != 0 || nt < 0 || off.back() != nt) { printf("count pass failed rc %d\n", rc); return 1; }
        const int64_t cap = (rng() % 4 == 0 && nt > 0) ? nt - 1 : nt;      // sometimes one row short: nothing may be written
        std::vector<int32_t> src((size_t)(cap > 0 ? cap : 1), -7);
        std::vector<double> tp((size_t)(cap > 0 ? cap : 1) * 2, -7.0);
        int64_t nt2 = -1;
        rc = fm_grid_pack_cells(pos.data(), n, w, h, cw, ch, rows, cols, margin, cap, off.data(), &nt2, src.data(), tp.data());
        if (rc != 0 || nt2 != nt) { printf("fill pass failed\n"); return 1; }
        if (cap < nt) { if (src[0] != -7) { printf("wrote into a short buffer\n"); return 1; } continue; }
        for (int64_t c = 0; c < (int64_t)rows * cols; ++c)
            for (int64_t i = off[(size_t)c]; i < off[(size_t)c + 1]; ++i) {
                const int32_t p = src[(size_t)i];
                if (p < 0 || p >= n || (i > off[(size_t)c] && src[(size_t)i - 1] >= p)) { printf("bad row list\n"); return 1; }
                const int row = (int)(c % rows), col = (int)(c / rows);
                const double x = pos[(size_t)p * 2], y = pos[(size_t)p * 2 + 1];
                const double x0 = row * cw - (row > 0 ? margin : 0), y0 = col * ch - (col > 0 ? margin : 0);
                const double x1 = row + 1 < rows ? x0 + cw + 2 * margin : w, y1 = col + 1 < cols ? y0 + ch + 2 * margin : h;
                if (!(x >= x0 && x < x1 && y >= y0 && y < y1)) { printf("keypoint outside its cell's crop\n"); return 1; }
                ++checked;
            }
    }
    printf("ok: %ld packed rows checked\n", checked);
    return 0;
}
