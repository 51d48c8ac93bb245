// K7 -- device-resident expansion loop: fastmatch.pyx:56-103 (do_iter + get_neighbors) and
// :145-169 (match_position) for a pre-extracted image pair, one persistent workgroup per
// pair, no host round trip between rounds.
//
// The loop is order dependent (neighbours are PREPENDED: depth first; the query subset of a
// (cell, query-cell) key is centred on the first seed that reaches it -- SURVEY.md fact 9),
// so one workgroup replays it sequentially in exactly the reference's order; parallelism is
// inside a round (radius query, sort, MFMA cross-check, hash probes) and across independent
// pairs (one workgroup each).  Per round:
//   1. pop the next (query_pos, target_pos) -- pending stack first, then the seed list;
//      skip it if its key (col,row,qcol,qrow) was already matched        fastmatch.pyx:68-72
//   2. radius query on the query keypoints around the truncated position, sorted by
//      (dx^2+dy^2 in float64, index), boundary inclusive                 cache.pyx:173-188
//   3. cross-checked 1-NN of that subset against the cell's descriptors  fastmatch.pyx:161-162
//      (x1_round: int8 MFMA), ratio = float64(dist)/selfdist             fastmatch.pyx:165
//   4. accepted = ratio < tau; for every accepted match the 4-neighbour cell on the side its
//      target point lies in (Grid_Cache.get_neighbor, cache.pyx:72-92); pushed so that the
//      first accepted match's neighbour is visited next                  fastmatch.pyx:75-77
//   5. accepted matches not seen before under (ratio, int-truncated positions) are appended
//      to the output in order                                            fastmatch.pyx:82-86
// Two reductions that cannot change the outcome keep the stack short: a neighbour whose key
// is already matched when it is pushed, or equals the key of an earlier neighbour of the same
// round, would be skipped when popped anyway, so it is not pushed.
#include "round_body.h"
#include "expand_pair.h"

namespace fm {

constexpr int kExpCand = 2048;            // radius-subset capacity per round

enum { kExpOk = 0, kExpStackFull = 1, kExpCandFull = 2, kExpOutOfBounds = 3, kExpMatchFull = 4, kExpTableFull = 5 };

__device__ __forceinline__ unsigned long long mix64(unsigned long long x)
{
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull;
    x ^= x >> 27; x *= 0x94d049bb133111ebull;
    x ^= x >> 31;
    return x;
}

__device__ __forceinline__ unsigned long long pack4x16(int a, int b, int c, int d)
{
    return ((unsigned long long)(unsigned short)a << 48) | ((unsigned long long)(unsigned short)b << 32) |
           ((unsigned long long)(unsigned short)c << 16) | (unsigned long long)(unsigned short)d;
}

__device__ __forceinline__ bool set_contains(const unsigned long long* tab, long long cap, unsigned long long key)
{
    long long p = (long long)(mix64(key) & (unsigned long long)(cap - 1));
    for (long long n = 0; n < cap; ++n) {
        const unsigned long long v = tab[p];
        if (v == key) return true;
        if (v == ~0ull) return false;
        p = (p + 1) & (cap - 1);
    }
    return false;
}

// Single-writer insert (thread 0 only).  Returns false when the table is full.
__device__ __forceinline__ bool set_insert(unsigned long long* tab, long long cap, unsigned long long key)
{
    long long p = (long long)(mix64(key) & (unsigned long long)(cap - 1));
    for (long long n = 0; n < cap; ++n) {
        const unsigned long long v = tab[p];
        if (v == key) return true;
        if (v == ~0ull) { tab[p] = key; return true; }
        p = (p + 1) & (cap - 1);
    }
    return false;
}

__device__ __forceinline__ bool found_contains(const unsigned long long* tab, long long cap,
                                               unsigned long long k0, unsigned long long k1)
{
    long long p = (long long)(mix64(k0 ^ mix64(k1)) & (unsigned long long)(cap - 1));
    for (long long n = 0; n < cap; ++n) {
        const unsigned long long v0 = tab[2 * p];
        if (v0 == ~0ull) return false;
        if (v0 == k0 && tab[2 * p + 1] == k1) return true;
        p = (p + 1) & (cap - 1);
    }
    return false;
}

// Concurrent insert of keys known to be absent and mutually distinct (claim an empty slot).
__device__ __forceinline__ bool found_insert(unsigned long long* tab, long long cap,
                                             unsigned long long k0, unsigned long long k1)
{
    long long p = (long long)(mix64(k0 ^ mix64(k1)) & (unsigned long long)(cap - 1));
    for (long long n = 0; n < cap; ++n) {
        if (atomicCAS(&tab[2 * p], ~0ull, k0) == ~0ull) { tab[2 * p + 1] = k1; return true; }
        p = (p + 1) & (cap - 1);
    }
    return false;
}

// Grid_Cache geometry (cache.pyx:95-99, 116-121, 72-92) in the reference's arithmetic:
// float64 division / multiplication, int() truncation toward zero.
__device__ __forceinline__ int blk(double v, int cell) { return (int)(v / (double)cell); }
__device__ __forceinline__ int center_coord(int i, int cell, int limit)
{
    const int c = (int)(((double)i + 0.5) * (double)cell);
    return c < limit - 1 ? c : limit - 1;
}

// Exclusive scan of per-thread counts over the 256-thread block; returns the total.
__device__ __forceinline__ int block_exclusive_scan(int v, int* my_offset, int* wave_tot /*LDS[4]*/)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(inc, d);
        if (lane >= d) inc += o;
    }
    __syncthreads();
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) { if (w < wave) base += wave_tot[w]; tot += wave_tot[w]; }
    *my_offset = base + inc - v;
    return tot;
}

__global__ __launch_bounds__(256)
void expand_kernel(const ExpandPair* __restrict__ pairs)
{
    __shared__ __attribute__((aligned(16))) char smem[kStageBytes];
    __shared__ unsigned long long keys[kExpCand];     // sort keys (d2 bits), then the qbest table
    __shared__ int cand[kExpCand];                    // candidate / sorted query rows
    __shared__ int tix[kExpCand];                     // matched local train index per slot (-1 none)
    __shared__ unsigned long long nkey[kExpCand];     // per slot: neighbour round key / dedup key part
    __shared__ double cur[4];                         // query_pos, target_pos of the round
    __shared__ int sh_i[8];
    __shared__ long long sh_top;
    __shared__ int wave_tot[4];

    const ExpandPair& P = pairs[blockIdx.x];
    const int tid = threadIdx.x;

    long long top = 0;            // stack height: owned by thread 0, published in sh_top each round
    long long seed_i = 0;
    long long n_matches = 0, n_rounds = 0, n_pairs = 0;
    long long seen_n = 0;
    int status = kExpOk;

    for (;;) {
        // ---- 1. next unseen (query_pos, target_pos) -----------------------------------------
        if (tid == 0) {
            int have = 0;
            while (status == kExpOk) {
                double e[4];
                if (top > 0) {
                    --top;
                    for (int k = 0; k < 4; ++k) e[k] = P.stack[top * 4 + k];
                } else if (seed_i < P.n_seeds) {
                    for (int k = 0; k < 4; ++k) e[k] = P.seeds[seed_i * 4 + k];
                    ++seed_i;
                } else {
                    break;
                }
                const int col = blk(e[3], P.cell_h), row = blk(e[2], P.cell_w);
                const int qcol = blk(e[1], P.cell_h), qrow = blk(e[0], P.cell_w);
                const unsigned long long key = pack4x16(col, row, qcol, qrow);
                if (set_contains(P.seen, P.seen_cap, key)) continue;
                if (2 * (seen_n + 1) > P.seen_cap || !set_insert(P.seen, P.seen_cap, key)) { status = kExpTableFull; break; }
                ++seen_n;
                for (int k = 0; k < 4; ++k) cur[k] = e[k];
                sh_i[1] = col; sh_i[2] = row;
                have = 1;
                break;
            }
            sh_i[0] = have;
            sh_i[3] = status;
            sh_i[7] = 0;
            sh_top = top;
        }
        __syncthreads();
        status = sh_i[3];
        if (!sh_i[0] || status != kExpOk) break;
        const int col = sh_i[1], row = sh_i[2];
        // C-int truncation of the positions (fastmatch.pyx:147-150)
        const int qx = (int)cur[0], qy = (int)cur[1], tx = (int)cur[2], ty = (int)cur[3];
        if (tx > P.width || ty > P.height) { status = kExpOutOfBounds; break; }     // cache.pyx:56-57
        // the cell actually fetched is the one of the TRUNCATED target position (target.get)
        const int gcol = blk((double)ty, P.cell_h), grow = blk((double)tx, P.cell_w);
        ++n_rounds;

        // ---- 2. radius query (Position_Index.radius) ------------------------------------------
        if (tid == 0) sh_i[4] = 0;
        __syncthreads();
        {
            const double r = (double)P.radius, b = P.idx_bucket;
            int bx0 = (int)floor(((double)qx - r - P.idx_x0) / b), bx1 = (int)floor(((double)qx + r - P.idx_x0) / b);
            int by0 = (int)floor(((double)qy - r - P.idx_y0) / b), by1 = (int)floor(((double)qy + r - P.idx_y0) / b);
            bx0 = max(bx0, 0); by0 = max(by0, 0);
            bx1 = min(bx1, P.idx_nbx - 1); by1 = min(by1, P.idx_nby - 1);
            const double r2 = r * r;
            if (P.idx_nbx > 0 && bx1 >= bx0) {
                for (int by = by0; by <= by1; ++by) {
                    const int s = P.idx_start[by * P.idx_nbx + bx0], e = P.idx_start[by * P.idx_nbx + bx1 + 1];
                    for (int i = s + tid; i < e; i += 256) {
                        const int qi = P.idx_order[i];
                        const double dx = P.q_pos[2 * qi] - (double)qx, dy = P.q_pos[2 * qi + 1] - (double)qy;
                        const double d2 = __dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy));   // no fma: NumPy order
                        if (d2 <= r2) {
                            const int slot = atomicAdd(&sh_i[4], 1);
                            if (slot < kExpCand) { keys[slot] = (unsigned long long)__double_as_longlong(d2); cand[slot] = qi; }
                        }
                    }
                }
            }
        }
        __syncthreads();
        const int nq = sh_i[4];
        if (nq > kExpCand) { status = kExpCandFull; break; }
        // bitonic sort of (d2 bits, index): non-negative doubles order like their bit patterns
        int npow = 1;
        while (npow < nq) npow <<= 1;
        for (int i = nq + tid; i < npow; i += 256) { keys[i] = ~0ull; cand[i] = 0x7fffffff; }
        __syncthreads();
        for (int k = 2; k <= npow; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = tid; i < npow; i += 256) {
                    const int l = i ^ j;
                    if (l > i) {
                        const unsigned long long ki = keys[i], kl = keys[l];
                        const int ci = cand[i], cl = cand[l];
                        const bool gt = ki > kl || (ki == kl && ci > cl);
                        if (gt == ((i & k) == 0)) { keys[i] = kl; keys[l] = ki; cand[i] = cl; cand[l] = ci; }
                    }
                }
                __syncthreads();
            }

        // ---- 3. cross-checked 1-NN against the cell ---------------------------------------------
        const int cell = gcol * P.rows + grow;
        const int64_t t0 = P.cell_off[cell];
        const int nt = (int)(P.cell_off[cell + 1] - t0);
        if (nt == 0 || nq == 0) continue;                   // match_position returns empty arrays
        n_pairs += (long long)nq * nt;
        for (int i = tid; i < nq; i += 256) keys[i] = ~0ull;     // keys[] becomes the qbest table
        x1_round<1>(P.q_rows8, P.q_norm, cand, nq, P.t_rows8, P.t_norm, t0, nt, smem, keys);
        __syncthreads();

        // ---- 4./5. accepted matches: neighbours and new results, in slot order ------------------
        const int ccx = center_coord(row, P.cell_w, P.width), ccy = center_coord(col, P.cell_h, P.height);
        int n_push = 0, n_emit = 0;          // this thread's counts (slots tid, tid+256, ...)
        // pass A: per slot flags; nkey = neighbour key (or ~0), keys[] keeps qbest for pass B
        for (int i = tid; i < nq; i += 256) {
            const unsigned long long qb = keys[i];
            int t_local = -1;
            unsigned long long nk = ~0ull;
            if (qb != ~0ull) {
                const float d = sqrtf((float)(unsigned)(qb >> 32));
                const double ratio = (double)d / P.q_selfdist[cand[i]];
                if (ratio < P.tau) {
                    t_local = (int)(unsigned)qb;
                    const double px = P.t_pos[2 * (t0 + t_local)], py = P.t_pos[2 * (t0 + t_local) + 1];
                    const int xd = (int)px - ccx, yd = (int)py - ccy;
                    int ncol = col, nrow = row;
                    if (yd < xd && yd < -xd) ncol = col - 1;
                    else if (xd > yd) nrow = row + 1;
                    else if (yd > -xd) ncol = col + 1;
                    else nrow = row - 1;
                    if (ncol >= 0 && ncol < P.cols && nrow >= 0 && nrow < P.rows) {
                        // pushed entry = (query position of the match, centre of the neighbour cell)
                        const double nx = (double)center_coord(nrow, P.cell_w, P.width);
                        const double ny = (double)center_coord(ncol, P.cell_h, P.height);
                        const double mqx = P.q_pos[2 * cand[i]], mqy = P.q_pos[2 * cand[i] + 1];
                        nk = pack4x16(blk(ny, P.cell_h), blk(nx, P.cell_w), blk(mqy, P.cell_h), blk(mqx, P.cell_w));
                        if (set_contains(P.seen, P.seen_cap, nk)) nk = ~0ull;      // would be skipped when popped
                    }
                }
            }
            tix[i] = t_local;
            nkey[i] = nk;
        }
        __syncthreads();
        // pass B: drop neighbours whose key an earlier slot of this round already pushes; count
        for (int i = tid; i < nq; i += 256) {
            unsigned long long nk = nkey[i];
            if (nk != ~0ull) {
                bool dup = false;
                for (int j = 0; j < i && !dup; ++j) dup = nkey[j] == nk;
                if (!dup) ++n_push;
                else tix[i] |= 0x40000000;          // mark "no push" (bit 30; train indices are small)
            }
        }
        __syncthreads();
        // stack push, first accepted slot on top: entry for rank k goes to top + (total-1-k)
        {
            int off;
            const int total = block_exclusive_scan(n_push, &off, wave_tot);
            if (tid == 0) sh_i[5] = (sh_top + total > P.stack_cap) ? 1 : 0;
            __syncthreads();
            if (sh_i[5]) { status = kExpStackFull; break; }
            // ranks must follow SLOT order, and a thread owns slots tid, tid+256, ...: do it per
            // 256-slot pass with a scan per pass
            const long long base = sh_top;
            int pushed_before = 0;
            for (int s0 = 0; s0 < nq; s0 += 256) {
                const int i = s0 + tid;
                const bool doit = i < nq && nkey[i] != ~0ull && !(tix[i] & 0x40000000);
                int o;
                const int cnt = block_exclusive_scan(doit ? 1 : 0, &o, wave_tot);
                if (doit) {
                    const int rank = pushed_before + o;
                    const long long dst = base + (total - 1 - rank);
                    const int ti = tix[i] & 0x3fffffff;
                    const double px = P.t_pos[2 * (t0 + ti)], py = P.t_pos[2 * (t0 + ti) + 1];
                    const int xd = (int)px - ccx, yd = (int)py - ccy;
                    int ncol = col, nrow = row;
                    if (yd < xd && yd < -xd) ncol = col - 1;
                    else if (xd > yd) nrow = row + 1;
                    else if (yd > -xd) ncol = col + 1;
                    else nrow = row - 1;
                    P.stack[dst * 4 + 0] = P.q_pos[2 * cand[i]];
                    P.stack[dst * 4 + 1] = P.q_pos[2 * cand[i] + 1];
                    P.stack[dst * 4 + 2] = (double)center_coord(nrow, P.cell_w, P.width);
                    P.stack[dst * 4 + 3] = (double)center_coord(ncol, P.cell_h, P.height);
                }
                pushed_before += cnt;
            }
            if (tid == 0) top += total;
        }
        __syncthreads();
        // results: dedup against earlier rounds (table lookups) and earlier slots (pairwise)
        for (int i = tid; i < nq; i += 256) {
            unsigned long long k1 = ~0ull;
            const int tl = tix[i];
            if (tl >= 0) {
                const int ti = tl & 0x3fffffff;
                const double mqx = P.q_pos[2 * cand[i]], mqy = P.q_pos[2 * cand[i] + 1];
                const double px = P.t_pos[2 * (t0 + ti)], py = P.t_pos[2 * (t0 + ti) + 1];
                k1 = pack4x16((int)mqx, (int)mqy, (int)px, (int)py);
            }
            nkey[i] = k1;                              // ratio part is recomputed below (keys[] = qbest)
        }
        __syncthreads();
        for (int s0 = 0; s0 < nq; s0 += 256) {
            const int i = s0 + tid;
            bool emit = false;
            unsigned long long k0 = 0, k1 = 0;
            double ratio = 0.0;
            if (i < nq && tix[i] >= 0) {
                const float d = sqrtf((float)(unsigned)(keys[i] >> 32));
                ratio = (double)d / P.q_selfdist[cand[i]];
                k0 = (unsigned long long)__double_as_longlong(ratio);
                k1 = nkey[i];
                emit = !found_contains(P.found, P.found_cap, k0, k1);
                for (int j = 0; j < i && emit; ++j) {
                    if (tix[j] >= 0 && nkey[j] == k1) {
                        const float dj = sqrtf((float)(unsigned)(keys[j] >> 32));
                        const double rj = (double)dj / P.q_selfdist[cand[j]];
                        if ((unsigned long long)__double_as_longlong(rj) == k0) emit = false;
                    }
                }
            }
            int o;
            const int cnt = block_exclusive_scan(emit ? 1 : 0, &o, wave_tot);
            if (tid == 0) sh_i[6] = (n_matches + n_emit + cnt > P.match_cap || 2 * (n_matches + n_emit + cnt) > P.found_cap) ? 1 : 0;
            __syncthreads();
            if (sh_i[6]) { status = kExpMatchFull; break; }
            if (emit) {
                const long long dst = n_matches + n_emit + o;
                const int ti = tix[i] & 0x3fffffff;
                P.m_index[dst] = cand[i];
                P.m_pos[dst * 4 + 0] = P.q_pos[2 * cand[i]];
                P.m_pos[dst * 4 + 1] = P.q_pos[2 * cand[i] + 1];
                P.m_pos[dst * 4 + 2] = P.t_pos[2 * (t0 + ti)];
                P.m_pos[dst * 4 + 3] = P.t_pos[2 * (t0 + ti) + 1];
                P.m_ratio[dst] = ratio;
                if (!found_insert(P.found, P.found_cap, k0, k1)) sh_i[7] = 1;
            }
            n_emit += cnt;
        }
        if (status != kExpOk) break;
        n_matches += n_emit;
        __threadfence_block();
        __syncthreads();           // table / stack writes visible before the next round reads them
        if (sh_i[7]) { status = kExpTableFull; break; }
    }
    if (tid == 0) {
        P.result[0] = n_matches;
        P.result[1] = n_rounds;
        P.result[2] = n_pairs;
        P.result[3] = status;
    }
}

hipError_t launch_expand(const void* d_pairs, int n_pairs, hipStream_t stream)
{
    hipLaunchKernelGGL(expand_kernel, dim3(n_pairs), dim3(256), 0, stream, (const ExpandPair*)d_pairs);
    return hipGetLastError();
}

int expand_cand_cap() { return kExpCand; }

}  // namespace fm
