// ASAN / UBSAN harness for the host-side plan of the triangular self-distance sweep (plan_tri, rowreduce.hip, built for
// the HOST): every padded size from 128 to 40 000 rows in steps of 128 plus a few large ones, piece lengths 0 (the search),
// 4 .. 80 and one absurd one; checks the coverage property on the fly (chunk k meets every stage >= 4 k exactly once).
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>
#include "ctx_internal.h"
int fm::fail(fm_ctx*, int code, const std::string&) { return code; }      // (api_grid.hip's error path: not exercised here)
int main()
{
    long checked = 0;
    std::vector<long long> sizes;
    for (long long n = 128; n <= 40000; n += 128) sizes.push_back(n);
    for (long long n : {100096LL, 300032LL, 1000064LL}) sizes.push_back(n);
    const int targets[] = {0, 4, 5, 31, 32, 80, 100000};
    for (long long n_pad : sizes)
        for (int t : targets) {
            if (n_pad > 40000 && t != 0 && t != 32) continue;
            std::vector<int> table;
            const fm::TriPlan pl = fm::plan_tri(n_pad, t, &table);
            const int nstages = (int)(n_pad / 128), nchunks = (nstages + 3) / 4;
            if (pl.nchunks != nchunks || pl.ndiag != nchunks || (int)table.size() != 4 * pl.npieces || pl.stages < 4) { printf("bad plan %lld %d\n", n_pad, t); return 1; }
            std::vector<int> cover((size_t)nchunks, 0);       // stages covered per chunk
            std::vector<long long> sum((size_t)nchunks, 0);
            for (int i = 0; i < pl.npieces; ++i) {
                const int k = table[4 * i], s0 = table[4 * i + 1], s1 = table[4 * i + 2];
                if (k < 0 || k >= nchunks || s0 < 4 * k || s0 >= s1 || s1 > nstages) { printf("bad piece %lld %d %d\n", n_pad, t, i); return 1; }
                cover[(size_t)k] += s1 - s0;
                sum[(size_t)k] += (long long)(s0 + s1 - 1) * (s1 - s0) / 2;          // sum of the stage numbers
            }
            for (int k = 0; k < nchunks; ++k) {
                const long long a = 4LL * k, b = nstages;                               // every stage in [4 k, nstages) once
                if (cover[(size_t)k] != b - a || sum[(size_t)k] != (a + b - 1) * (b - a) / 2) { printf("coverage %lld %d chunk %d\n", n_pad, t, k); return 1; }
            }
            ++checked;
        }
    printf("ok: %ld plans\n", checked);
    return 0;
}
