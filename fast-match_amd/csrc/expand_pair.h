// Device-side descriptor of one image pair for expand_kernel (expand.hip); filled by api_expand.hip.
#pragma once
#include <stdint.h>
#include "round_body_f32.h"

namespace fm {

struct ExpandPair {
    // query side
    const int8_t*  q_rows8;
    const int32_t* q_norm;
    const double*  q_selfdist;
    const double*  q_pos;          // [nq][2]
    const double*  q_pos_ord;      // [nq][2] the same positions in the position index's order (q_pos[idx_order[j]])
    const int32_t* idx_order;      // keypoints sorted by bucket
    const int32_t* idx_start;      // [nbx*nby + 1]
    double idx_bucket, idx_x0, idx_y0;
    int    idx_nbx, idx_nby;
    int    metric;                 // radius query: 0 = Euclidean (d2 <= r2), 1 = Manhattan, 2 = Chebyshev (d <= r)
    // target side: every grid cell's descriptors packed back to back
    const int8_t*  t_rows8;
    const int32_t* t_norm;
    const int64_t* cell_off;       // [cols*rows + 1], cell id = col * rows + row
    const double*  t_pos;          // [nt_total][2] full-image coordinates (offset() applied)
    int width, height, cell_w, cell_h, rows, cols, margin, radius;
    // float32 route (banks that are not integer valued): planes and scale terms of both banks
    int      f32;                  // non-zero: x1_round_f32 instead of the int8 round
    int      tie_guard;            // int8 round: the banks' norms allow d2 >= kSqrtTieMin (tile_ops.h)
    RoundF32 rf;
    // run inputs
    const double* seeds;           // [n_seeds][2][2]
    int64_t n_seeds;
    double  tau;
    // work memory
    double* stack;                 // [stack_cap][4]
    int64_t stack_cap;
    unsigned long long* seen;      // open-addressing set of round keys, capacity seen_cap (pow2)
    int64_t seen_cap;
    unsigned long long* found;     // [found_cap][2] (ratio bits, packed int positions)
    int64_t found_cap;
    // outputs
    int32_t* m_index;              // [match_cap]
    double*  m_pos;                // [match_cap][4]
    double*  m_ratio;              // [match_cap]
    int64_t  match_cap;
    // huge tier (a radius subset beyond the LDS tables, expand.hip): per-run scratch in global memory, or null
    int32_t* h_cand;               // [nq] the round's sorted query rows, all slots
    unsigned long long* h_pkey;    // [nq] the subset's sort keys partitioned by chunk (slot ranges as h_cand)
    int32_t* h_ucand;              // [nq] the round's subset as the radius query found it (unsorted; its keys sit in h_qbest until the election)
    unsigned long long* h_qbest;   // [nq] cross-check table of the round, all slots
    unsigned long long* h_tbest;   // [largest cell] per train row: running (d2 << 32 | slot) minimum over the chunks
    // lazy targets (expand.hip, LAZY): cells arrive one by one; per cell its first row, row count and a ready flag
    const int64_t* cell_start;     // [cols*rows]
    const int32_t* cell_cnt;       // [cols*rows]
    const int32_t* cell_ready;     // [cols*rows] 0 = not computed yet
    long long* resume_state;       // [14] loop state of a parked run (top, seed cursor, counters, the popped entry; a delegated
                                   // cross-check: + subset size, first train row, train rows; [13] log entries so far)
    int        resume;             // 1: restore resume_state and take its entry first; 2: ... and go straight to steps 4 / 5
                                   // (the round's cross-checked keys are in h_qbest)
    long long  delegate_min;       // > 0: rounds of at least this many descriptor pairs whose subset does not fit LDS are parked
                                   // for a dense cross-check on the whole GPU (expand.hip, DELEGATED)
    // Result words of a run (kExpResultWords of them): [0] n_matches, [1] n_rounds, [2] n_pairs, [3] status,
    // [4] the cell a parked run waits for, [5 .. 7] a delegated cross-check's subset size / first train row / train rows,
    // [8] log entries written, [16 .. 27] the 12 phase timers (prof), [32 .. 39] FM_PARK_PROF builds' phase sums.
    long long* result;
    int        prof;               // non-zero: thread 0 accumulates per-phase 100 MHz ticks
    // Per-round log (options["log"], fastmatch.pyx:79-80, 172-180), or null: one record per processed round --
    // the popped entry (query_pos, target_pos as float64 bits), the cell the round fetched (col * rows + row: the
    // host derives Grid_Cache.last from the order in which cells first appear, cache.pyx:102-106) and the number of
    // accepted matches -- and, per accepted match in the round's order BEFORE the result dedup (log_round keeps
    // result_pos[ratios < tau], fastmatch.pyx:177, 179), its query row, its row of the packed target bank and its ratio.
    long long* lg_round;           // [lg_round_cap][6]
    int32_t*   lg_q;               // [lg_entry_cap]
    int32_t*   lg_t;               // [lg_entry_cap]
    double*    lg_ratio;           // [lg_entry_cap]
    long long  lg_round_cap, lg_entry_cap;
};
constexpr int kExpResultWords = 64;


}  // namespace fm
