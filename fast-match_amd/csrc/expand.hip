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
constexpr int kExpSR = 512;               // query rows gathered per staging step
constexpr int kExpStageBytes = kExpSR * kDim + kExpSR / 32 * 256;
constexpr int kExpLdsBytes = kExpStageBytes + kExpCand * (8 + 8 + 4 + 4) + (2 * 1024 + 16) * 4 + 128 * 8;

enum { kExpOk = 0, kExpStackFull = 1, kExpCandFull = 2, kExpOutOfBounds = 3, kExpMatchFull = 4, kExpTableFull = 5,
       kExpListFull = 6 };
static_assert(kRF_StageBytes <= kExpStageBytes, "the float32 round's gather image must fit the stage buffer");
// float32 route: candidate list of x1_round_f32 aliases the sort scratch (nkey + tix + hist), free during step 3
constexpr int kExpClistCap = (kExpCand * (8 + 4) + 2 * 1024 * 4) / 4;

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

// Single-writer insert-if-absent in one probe sequence: 1 = inserted, 0 = was present, -1 = full.
__device__ __forceinline__ int set_insert_new(unsigned long long* tab, long long cap, unsigned long long key)
{
    long long p = (long long)(mix64(key) & (unsigned long long)(cap - 1));
    for (long long n = 0; n < cap; ++n) {
        const unsigned long long v = tab[p];
        if (v == key) return 0;
        if (v == ~0ull) { tab[p] = key; return 1; }
        p = (p + 1) & (cap - 1);
    }
    return -1;
}

// Grid_Cache geometry (cache.pyx:95-99, 116-121, 72-92) in the reference's arithmetic:
// float64 division / multiplication, int() truncation toward zero.
__device__ __forceinline__ int blk(double v, int cell) { return (int)(v / (double)cell); }
__device__ __forceinline__ int center_coord(int i, int cell, int limit)
{
    const int c = (int)(((double)i + 0.5) * (double)cell);
    return c < limit - 1 ? c : limit - 1;
}

// Sort n <= kExpCand (key, idx) pairs held in LDS ascending by (key, idx); pairs are unique.
// keys are the bit patterns of squared distances in [0, r2]: inside a disc they are spread
// uniformly, so a counting sort over kSortBuckets linear buckets leaves ~n / kSortBuckets
// elements per bucket and the exact order inside a bucket is fixed by a handful of
// comparisons.  bucket(d2) is monotone in d2, so bucket order + in-bucket order = total order.
// Scratch: k2 (u64[kExpCand]), i2 (int[kExpCand]), hist (int[2 * kSortBuckets + 8]).
constexpr int kSortBuckets = 1024;
__device__ __forceinline__ int block_exclusive_scan_nosync(int v, int* my_offset, int* wave_tot);

__device__ __forceinline__ void block_sort_pairs(unsigned long long* keys, int* idx, unsigned long long* k2,
                                                 int* i2, int* hist, int n, double r2)
{
    const int tid = threadIdx.x;
    int* start = hist;                       // [kSortBuckets + 1] after the scan
    int* cursor = hist + kSortBuckets + 4;   // [kSortBuckets]
    for (int b = tid; b < kSortBuckets; b += 256) { start[b] = 0; cursor[b] = 0; }
    __syncthreads();
    const double scale = r2 > 0.0 ? (double)kSortBuckets / r2 : 0.0;
    int myb[kExpCand / 256];
#pragma unroll
    for (int s = 0; s < kExpCand / 256; ++s) {
        const int i = s * 256 + tid;
        myb[s] = 0;
        if (i < n) {
            const double d2 = __longlong_as_double((long long)keys[i]);
            int b = (int)(d2 * scale);
            b = b < kSortBuckets - 1 ? b : kSortBuckets - 1;
            myb[s] = b;
            atomicAdd(&start[b], 1);
        }
    }
    __syncthreads();
    // exclusive scan of the bucket counts: 4 buckets per thread
    {
        int c[4], s = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) { c[q] = start[tid * 4 + q]; s += c[q]; }
        int off;
        block_exclusive_scan_nosync(s, &off, hist + 2 * kSortBuckets + 8 /* tail: wave totals */);
#pragma unroll
        for (int q = 0; q < 4; ++q) { start[tid * 4 + q] = off; off += c[q]; }
        if (tid == 255) start[kSortBuckets] = off;
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < kExpCand / 256; ++s) {
        const int i = s * 256 + tid;
        if (i < n) {
            const int p = start[myb[s]] + atomicAdd(&cursor[myb[s]], 1);
            k2[p] = keys[i];
            i2[p] = idx[i] | 0;          // (bucket order; order inside a bucket is arbitrary here)
        }
    }
    __syncthreads();
    // final position = bucket start + number of smaller pairs inside the bucket
#pragma unroll
    for (int s = 0; s < kExpCand / 256; ++s) {
        const int p = s * 256 + tid;
        if (p < n) {
            const unsigned long long k = k2[p];
            const int v = i2[p];
            double d2 = __longlong_as_double((long long)k);
            int b = (int)(d2 * scale);
            b = b < kSortBuckets - 1 ? b : kSortBuckets - 1;
            const int s0 = start[b], s1 = start[b + 1];
            int rank = s0;
            for (int j = s0; j < s1; ++j) {
                const unsigned long long kj = k2[j];
                rank += (kj < k || (kj == k && i2[j] < v)) ? 1 : 0;
            }
            keys[rank] = k;
            idx[rank] = v;
        }
    }
    __syncthreads();
}

// Exclusive scan of per-thread counts over the 256-thread block; returns the total.
__device__ __forceinline__ int block_exclusive_scan(int v, int* my_offset, int* wave_tot /*LDS[4]*/);
__device__ __forceinline__ int block_exclusive_scan_nosync(int v, int* my_offset, int* wave_tot)
{
    return block_exclusive_scan(v, my_offset, wave_tot);
}
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
    // dynamic LDS (kExpLdsBytes): a 512-row gather stage, the sort keys / qbest table, the
    // candidate rows, and two scratch arrays
    extern __shared__ __attribute__((aligned(16))) char dyn_lds[];
    char* smem = dyn_lds;                                                         // kExpStageBytes
    unsigned long long* keys = (unsigned long long*)(dyn_lds + kExpStageBytes);   // sort keys, then qbest
    int* cand = (int*)(keys + kExpCand);                                          // candidate / sorted query rows
    unsigned long long* nkey = (unsigned long long*)(cand + kExpCand);            // ratio bits of accepted matches
    int* tix  = (int*)(nkey + kExpCand);                                          // sort scratch, then accepted list
    int* hist = tix + kExpCand;                                                   // counting-sort buckets
    // (nkey, tix, hist are contiguous: the float32 round's candidate list aliases them during step 3)
    __shared__ double cur[4];                         // query_pos, target_pos of the round
    __shared__ int sh_i[8];
    __shared__ long long sh_top;
    __shared__ int wave_tot[4];
    __shared__ int sh_rf[4];                          // per-wave candidate counts of the float32 round

    const ExpandPair& P = pairs[blockIdx.x];
    const int tid = threadIdx.x;

    long long top = 0;            // stack height: owned by thread 0, published in sh_top each round
    long long seed_i = 0;
    long long n_matches = 0, n_rounds = 0, n_pairs = 0;
    long long seen_n = 0;
    int status = kExpOk;
    long long pt[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long tstamp = P.prof ? wall_clock64() : 0;
#define EXP_STAMP(k) do { if (P.prof && tid == 0) { const long long _n = wall_clock64(); pt[k] += _n - tstamp; tstamp = _n; } } while (0)

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
                if (2 * (seen_n + 1) > P.seen_cap) {                 // (only a NEW key needs room)
                    if (set_contains(P.seen, P.seen_cap, key)) continue;
                    status = kExpTableFull;
                    break;
                }
                const int ins = set_insert_new(P.seen, P.seen_cap, key);
                if (ins == 0) continue;
                if (ins < 0) { status = kExpTableFull; break; }
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
        const long long sh_i_top_before = sh_top;
        EXP_STAMP(0);
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
        EXP_STAMP(1);
        // sort by (d2 bits, index): non-negative doubles order like their bit patterns
        block_sort_pairs(keys, cand, nkey, tix, hist, nq, (double)P.radius * (double)P.radius);

        EXP_STAMP(2);
        // ---- 3. cross-checked 1-NN against the cell ---------------------------------------------
        const int cell = gcol * P.rows + grow;
        const int64_t t0 = P.cell_off[cell];
        const int nt = (int)(P.cell_off[cell + 1] - t0);
        if (nt == 0 || nq == 0) continue;                   // match_position returns empty arrays
        n_pairs += (long long)nq * nt;
        for (int i = tid; i < nq; i += 256) keys[i] = ~0ull;     // keys[] becomes the qbest table
        if (P.f32) {
            // descriptors that are not integer valued: fp16 MFMA filter + exact float32 chain (round_body_f32.h)
            __syncthreads();
            const bool ok = x1_round_f32(P.rf, cand, nq, t0, nt, smem, keys, (unsigned*)nkey, kExpClistCap,
                                         (unsigned long long*)(hist + 2 * kSortBuckets + 16), sh_rf,
                                         P.prof ? pt : nullptr, &tstamp);
            if (!ok) { status = kExpListFull; break; }
        } else {
            x1_round_wsplit<kExpSR>(P.q_rows8, P.q_norm, cand, nq, P.t_rows8, P.t_norm, t0, nt, smem, keys,
                                    (unsigned long long*)(hist + 2 * kSortBuckets + 16), P.prof ? pt : nullptr, &tstamp);
        }
        __syncthreads();

        EXP_STAMP(3);
        // ---- 4./5. accepted matches: neighbours and new results, in slot order ------------------
        // (a) compact the accepted slots in order:  tix[k] = slot | t_local << 11,
        //     nkey[k] = ratio bits  (k < na)
        const int ccx = center_coord(row, P.cell_w, P.width), ccy = center_coord(col, P.cell_h, P.height);
        int na = 0;
        for (int s0 = 0; s0 < nq; s0 += 256) {
            const int i = s0 + tid;
            bool acc = false;
            double ratio = 0.0;
            int t_local = 0;
            if (i < nq) {
                const unsigned long long qb = keys[i];
                if (qb != ~0ull) {
                    // high word: exact integer d^2 (int8 route) or the float32 distance bits (float32 route)
                    const float d = P.f32 ? __uint_as_float((unsigned)(qb >> 32)) : sqrtf((float)(unsigned)(qb >> 32));
                    ratio = (double)d / P.q_selfdist[cand[i]];
                    acc = ratio < P.tau;
                    t_local = (int)(unsigned)qb;
                }
            }
            int o;
            const int cnt = block_exclusive_scan(acc ? 1 : 0, &o, wave_tot);
            // keys[] (qbest) of slots < s0 + 256 are consumed: entries na+o <= i never clobber unread ones
            __syncthreads();
            if (acc) { tix[na + o] = i | (t_local << 11); nkey[na + o] = (unsigned long long)__double_as_longlong(ratio); }
            na += cnt;
        }
        __syncthreads();
        EXP_STAMP(4);
        // (b) per accepted match: neighbour key + seen probe, result key + found probe.
        //     keys[k] = neighbour key (or ~0), rk[k] = result key (int-truncated positions)
        unsigned long long* rk = (unsigned long long*)smem;        // stage buffer is free now
        int n_emit = 0;
        for (int k0 = 0; k0 < na; k0 += 256) {
            const int k = k0 + tid;
            const bool live = k < na;
            double mqx = 0, mqy = 0, px = 0, py = 0, nx = 0, ny = 0;
            unsigned long long nk = ~0ull, k1 = 0, rbits = 0;
            int qrow_idx = 0;
            bool known = false;
            if (live) {
                const int slot = tix[k] & 2047, t_local = tix[k] >> 11;
                qrow_idx = cand[slot];
                rbits = nkey[k];
                mqx = P.q_pos[2 * qrow_idx]; mqy = P.q_pos[2 * qrow_idx + 1];
                px = P.t_pos[2 * (t0 + t_local)]; py = P.t_pos[2 * (t0 + t_local) + 1];
                const int xd = (int)px - ccx, yd = (int)py - ccy;          // Grid_Cache.get_neighbor
                int ncol = col, nrow = row;
                if (yd < xd && yd < -xd) ncol = col - 1;
                else if (xd > yd) nrow = row + 1;
                else if (yd > -xd) ncol = col + 1;
                else nrow = row - 1;
                if (ncol >= 0 && ncol < P.cols && nrow >= 0 && nrow < P.rows) {
                    nx = (double)center_coord(nrow, P.cell_w, P.width);
                    ny = (double)center_coord(ncol, P.cell_h, P.height);
                    nk = pack4x16(blk(ny, P.cell_h), blk(nx, P.cell_w), blk(mqy, P.cell_h), blk(mqx, P.cell_w));
                }
                k1 = pack4x16((int)mqx, (int)mqy, (int)px, (int)py);
                // two independent probes: would the neighbour be skipped when popped? is the
                // result already in the list?
                if (nk != ~0ull && set_contains(P.seen, P.seen_cap, nk)) nk = ~0ull;
                known = found_contains(P.found, P.found_cap, rbits, k1);
                keys[k] = nk;
                rk[k] = k1;
            }
            __syncthreads();
            // earlier entries of this round with the same key win (lists are in slot order)
            bool push = live && nk != ~0ull, emit = live && !known;
            if (live) {
                for (int j = 0; j < k && (push || emit); ++j) {
                    if (push && keys[j] == nk) push = false;
                    if (emit && rk[j] == k1 && nkey[j] == rbits) emit = false;
                }
            }
            // stack push, first accepted match on top: entry of rank r goes to top + (total-1-r);
            // chunks of 256 accepted matches are pushed in reverse chunk order below
            int po;
            const int ptot = block_exclusive_scan(push ? 1 : 0, &po, wave_tot);
            int eo;
            const int etot = block_exclusive_scan(emit ? 1 : 0, &eo, wave_tot);
            if (tid == 0) {
                sh_i[5] = (sh_top + ptot > P.stack_cap) ? 1 : 0;
                sh_i[6] = (n_matches + n_emit + etot > P.match_cap || 2 * (n_matches + n_emit + etot) > P.found_cap) ? 1 : 0;
            }
            __syncthreads();
            if (sh_i[5]) { status = kExpStackFull; break; }
            if (sh_i[6]) { status = kExpMatchFull; break; }
            if (push) {
                // The first accepted match must be popped first, i.e. sit on top.  One chunk
                // (na <= 256, the usual case): write in reverse rank order.  More: chunks are
                // written in ascending order and the whole region is reversed afterwards.
                const long long dst = (na <= 256) ? sh_top + (ptot - 1 - po) : sh_top + po;
                P.stack[dst * 4 + 0] = mqx; P.stack[dst * 4 + 1] = mqy;
                P.stack[dst * 4 + 2] = nx;  P.stack[dst * 4 + 3] = ny;
            }
            if (emit) {
                const long long dst = n_matches + n_emit + eo;
                P.m_index[dst] = qrow_idx;
                P.m_pos[dst * 4 + 0] = mqx; P.m_pos[dst * 4 + 1] = mqy;
                P.m_pos[dst * 4 + 2] = px;  P.m_pos[dst * 4 + 3] = py;
                P.m_ratio[dst] = __longlong_as_double((long long)rbits);
                if (!found_insert(P.found, P.found_cap, rbits, k1)) sh_i[7] = 1;
            }
            n_emit += etot;
            __syncthreads();
            if (tid == 0) { sh_top += ptot; top += ptot; }
            __syncthreads();
        }
        if (status != kExpOk) break;
        // More than one chunk: the pushed region [top_before, top) is in ascending slot order;
        // reverse it in place.
        if (na > 256) {
            const long long lo = sh_i_top_before, hi = sh_top;
            const long long cntp = hi - lo;
            for (long long x = tid; x < cntp / 2; x += 256) {
                const long long a = lo + x, b = hi - 1 - x;
                for (int c = 0; c < 4; ++c) { const double t = P.stack[a * 4 + c]; P.stack[a * 4 + c] = P.stack[b * 4 + c]; P.stack[b * 4 + c] = t; }
            }
        }
        EXP_STAMP(5);
        n_matches += n_emit;
        __threadfence_block();
        __syncthreads();           // table / stack writes visible before the next round reads them
        if (sh_i[7]) { status = kExpTableFull; break; }
        EXP_STAMP(6);
    }
    if (tid == 0) {
        P.result[0] = n_matches;
        P.result[1] = n_rounds;
        P.result[2] = n_pairs;
        P.result[3] = status;
        if (P.prof) for (int k = 0; k < 12; ++k) P.result[4 + k] = pt[k];
    }
}

hipError_t launch_expand(const void* d_pairs, int n_pairs, hipStream_t stream)
{
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)expand_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kExpLdsBytes);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(expand_kernel, dim3(n_pairs), dim3(256), kExpLdsBytes, stream, (const ExpandPair*)d_pairs);
    return hipGetLastError();
}

int expand_cand_cap() { return kExpCand; }

}  // namespace fm
