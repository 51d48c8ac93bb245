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
    long long* resume_state;       // [13] loop state of a parked run (top, seed cursor, counters, the popped entry; a delegated
                                   // cross-check: + subset size, first train row, train rows)
    int        resume;             // 1: restore resume_state and take its entry first; 2: ... and go straight to steps 4 / 5
                                   // (the round's cross-checked keys are in h_qbest)
    long long  delegate_min;       // > 0: rounds of at least this many descriptor pairs whose subset does not fit LDS are parked
                                   // for a dense cross-check on the whole GPU (expand.hip, DELEGATED)
    long long* result;             // [12]: n_matches, n_rounds, n_pairs, status, 8 phase timers (lazy: [4] = the cell wanted)
    int        prof;               // non-zero: thread 0 accumulates per-phase 100 MHz ticks
};


}  // namespace fm
