// Internal declarations shared by the HIP translation units of libfastmatch_hip.so.
// gfx950 (MI355X / CDNA4) only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include "../../include/fastmatch_hip.h"

namespace fm {

// ---- bank layout in HBM -------------------------------------------------------------
// rows8 : int8 [n_pad][128]   descriptor bytes XOR 0x80 (u8 -> i8 shift by 128; L2 is
//                             shift invariant), zero rows for padding, n_pad % 128 == 0
// norm  : int32 [n_pad]       nm = sum of squares of the int8 row (<= 2^21)
// aux   : int32 [n_pad/32][64] per 32-row unit = two 16-row MFMA tiles (sub = 0, 1), in the
//         accumulator order of v_mfma_i32_16x16x64_i8 (tile row rr: lane group rr >> 2,
//         register rr & 3):
//           aux[unit][32*sub +      rr] = -(nm >> 1)                       ("cinit")
//           aux[unit][32*sub + 16 + rr] = (1 - (nm & 1)) << 4 | (15 - id)  ("low": parity
//                                         npar and inverted in-lane order id = 4*sub + (rr&3))
//         Padding rows carry cinit = -2^25 (far below any real accumulator value, and
//         (cinit << 5) still fits int32) so they never beat a real row.
constexpr int kDim        = 128;
constexpr int kStageRows  = 128;                 // rows staged into LDS per pipeline step
constexpr int kTileRows   = 32;                  // one MFMA M-tile
constexpr int kAuxPerTile = 64;
constexpr int kPadCinit   = -(1 << 25);

struct Bank {
    int      kind   = 0;       // FM_BANK_I8 / FM_BANK_F32
    int64_t  n      = 0;
    int      dim    = 0;
    int64_t  n_pad  = 0;       // multiple of kStageRows (>= kStageRows so empty banks stage)
    int64_t  cap_pad = 0;      // n_pad the arrays were allocated for (fm_bank_refill_u8_async keeps within it)
    void*    stage  = nullptr; // refill staging: cap_pad * 128 source bytes + the preparation kernel's flag words
    int8_t*  rows8  = nullptr;
    int32_t* norm   = nullptr;
    int32_t* aux    = nullptr;
    int32_t  usq_max = 0;      // FM_BANK_I8: largest |row|^2 of the uint8 rows (sqrt_tie_possible below)
    float*   rowsf  = nullptr; // FM_BANK_F32: [n_pad][128] float32, zero padded
    // FM_BANK_F32, for the fp16 filter (filter_f16.hip); "scaled" = times 2^kscale:
    uint16_t* rowsh = nullptr; // [n_pad][128] fp16 of the scaled rows
    float*   normf  = nullptr; // [n_pad] |scaled row|^2
    float*   auxf   = nullptr; // [n_pad] accumulator init -|scaled row|^2 / 2 (padding rows: -3.4e38)
    float    nm_max = 0.f;     // max |scaled row|^2
    int      kscale = 0;       // largest scaled magnitude lies in [2^13, 2^14)
    bool     filt_ok = false;  // every value is finite
    double*  selfdist = nullptr;
};

// OpenCV orders candidates by the float32 root of d2; the integer route orders by d2, which is the same
// order unless a d2 reaches 4 197 200, where two integers start to share a float32 root (kSqrtTieMin,
// tile_ops.h).  Rows are non-negative, so d2(a, b) <= |a|^2 + |b|^2: a bank pair whose largest row
// norms stay below that can skip every repair step (all SIFT banks: |d|^2 ~ 2.6e5).
inline bool sqrt_tie_possible(const Bank& a, const Bank& b)
{
    return (int64_t)a.usq_max + (int64_t)b.usq_max >= 4197200;
}

// ---- per-context tuning (fm_ctx_set_option; the FM_* environment variables only seed the defaults
// when a context is created) -------------------------------------------------------------------
struct Tuning {
    int nb = 0, nsplit = 0, nw = 0;   // K1 launch shape overrides, 0 = the built-in rule
    int nbuf = 0;                     // K1 / K2 LDS stage buffers: 0 = 3 (top-2 shapes other than 4 blocks per wave: 2)
    int prio = 1;                     // s_setprio around a unit's MFMA burst (three-buffer kernel)
    int glds = 1;                     // LDS-DMA staging (0: through registers)
    int coop = 1;                     // cross-workgroup K-th-best bounds
    int f32_filter = 1;               // float32 route: 0 = K5 only, 1 = fp16 filter for large calls, 2 = always
    int f32_nw = 0, f32_nsplit = 0;   // K8 launch shape overrides
    int f32_fused = -1;               // K8: rescoring inside the filter (-1 = when the plan has one split)
    int f32_lpc = 0;                  // K8 rescoring: lanes per output row (0 = rule)
    int f32_bound_every = 4;          // K8: re-read the shared bounds every n-th stage once a sweep is 8 stages old (a power of two)
    int batch_group = 8;              // fm_match_accepted_batch: most pairs per distance-kernel launch (<= kRRBatchMax)
    int batch_tail = 2;               // ... pairs of the short launch a run ends with (0 = none)
    int async_time_every = 4;         // async calls: every n-th call carries kernel timing events (0 = none)
    int expand_big = 1;               // K7: re-run pairs that overflow the 2048-row round in the 4096-row variant
    int expand_huge = 1;              // K7: re-run pairs that still overflow in the variant that chunks a radius subset of any size
    int delegated_rounds = 0;         // (counter, saturating) rounds whose cross-check the dense kernels ran since it was last set to 0
    int expand_delegate = 1500000;    // K7: a chunked round of at least this many descriptor pairs is parked and its cross-check run
                                      // by the dense kernels on the whole GPU (0 = never: the round's own workgroup does it)
    int expand_grow = 2;              // K7: how often a run that fills its stack / result list / table is repeated in a
                                      // state four times as large (0 = never: the status goes to the caller)
    int expand_prof = 0;              // K7: per-phase timers of pair 0 on stderr
    int bound_every = 16;             // K1 / K2: re-read the shared bounds every n-th stage once a sweep is 8 stages old (a power of
                                      // two; r04 A/B on two boxes, ms per pair in the 12-pair launch: 1 -> 0.8356 / 0.8501, 8 -> 0.8290 /
                                      // 0.8399, 32 -> 0.8277, 1024 -> 0.8340: profiles/r04c_k1_bound_every_*)
    int refill_grid = 128;            // fm_bank_refill_u8_async: workgroups of the preparation kernel (each walks its share of the tiles)
    int self_tri = 1;                 // fm_self_dist: 1 = the triangular sweep (every distance once) from 32768 rows on (integer banks) /
                                      // 65536 (float32-route banks, r06), 0 = the masked full sweep always, 2 = the triangular sweep always (tests)
    int tri_stages = 0;               // ... stages (128 rows) per slice of the triangular sweep (0 = plan_tri's search)
    int k1_order = 0;                 // K1: workgroup -> (chunk, split) mapping (rowreduce.hip, map_block): 0 split major,
                                      // 1 an XCD owns output chunks, 2 an XCD owns a contiguous share of the split-major order
};

// ---- K1: row-reduce kernel launcher ---------------------------------------------------
// For every row c of bank `cols`, reduce over all rows m of bank `red` the key
// (d2(c,m), m) lexicographically and keep the KTOP smallest.  Writes, per split of the
// reduction range, packed candidates  (uint64)d2 << 32 | m  (~0 = none) to
// partial[(split*ncols_alloc + c)*KTOP + k].
struct RowReducePlan {
    int nb;            // blocks of 16 output rows per wave (4 or 8)
    int nw = 4;        // waves per workgroup (4, 8 or 16)
    int ncols_alloc;   // columns covered by the grid (multiple of 16*nb*nw)
    int nchunks;
    int nsplit;
    int stages_per_split;
    int nbuf = 0;      // stage buffers (0 = rule: rowreduce.hip nbuf_choice)
    int prio = 1;      // s_setprio around the MFMA burst
    int order = 0;     // Tuning::k1_order
    int bound_every = 1;   // Tuning::bound_every
    size_t partial_bytes(int ktop) const { return (size_t)nsplit * ncols_alloc * ktop * 8; }
    size_t bound_bytes() const { return (size_t)ncols_alloc * 4 * 2; }   // (top-2 launches keep two arrays)
};
RowReducePlan plan_rowreduce(int64_t ncols_pad, int64_t nred_pad, const Tuning& tn);
hipError_t launch_rowreduce(const Bank& cols, const Bank& red, int ktop, const RowReducePlan& plan,
                            unsigned long long* partial, int* bound, bool use_glds, hipStream_t stream);
int rowreduce_grid(const RowReducePlan& plan);      // workgroups per bank pair (padded under orders 1 and 2)
// fm_self_dist: top-1 of every row over the OTHER rows of its own bank (masked diagonal)
RowReducePlan plan_rowreduce_self(int64_t n_pad, const Tuning& tn);
hipError_t launch_rowreduce_self(const Bank& bank, const RowReducePlan& plan, unsigned long long* partial, int* bound,
                                 bool use_glds, hipStream_t stream);

// fm_self_dist on large integer banks: the triangular sweep (rowreduce.hip, "TRI") -- every distance of a bank
// against itself once, bound[] carries the result (d2(i) = norm[i] + 1 - bound[i]; <= kTriNoBoundHost: none).
struct TriPlan {
    int nchunks = 0, ncols_alloc = 0, npieces = 0, stages = 0;
    int ndiag = 0;                     // the first ndiag workgroups of the table are launch A (the diagonal blocks)
    int bound_every = 16;              // Tuning::bound_every (a power of two)
};
TriPlan plan_tri(int64_t n_pad, int target_stages, std::vector<int>* table);
constexpr int64_t kTriMaxRows = (int64_t)1 << 26;      // banks beyond this take the masked full sweep (the workgroup count of
                                                        // the triangular one grows with the square of the rows: int32 from ~75M on)
hipError_t launch_rowreduce_tri(int n, const Bank* const* banks, const TriPlan* plans, int* const* bound, bool prio, hipStream_t stream);
constexpr int kTriNoBoundHost = -(1 << 25);     // (a real pair's word is >= -3 * 2^21, a padding row's or the masked diagonal's ~ -2^26)

// ---- K5: float32 route (dist_f32.hip); partial keys carry float32 distance bits ----------
// run_flag: device word; the kernel returns at once when *run_flag == 0 (null = always run) and
// counts its runs in run_flag[2].
RowReducePlan plan_rowreduce_f32(int64_t ncols_pad, int64_t nred_pad, int force_nsplit);
hipError_t launch_rowreduce_f32(const Bank& cols, const Bank& red, int ktop, const RowReducePlan& plan,
                                unsigned long long* partial, const int* run_flag, hipStream_t stream, bool self = false);

// ---- K8: fp16 MFMA filter + exact rescoring for the float32 route (filter_f16.hip) --------
// Writes the same keys as K5 into split 0 of `partial` (the caller presets the other splits to
// ~0).  flag: device words [0] K5 must redo the call, [1] output rows rescanned in full,
// [2] K5 runs, [3] rescans in total, [4 .. 4 + 256) the rescanned rows, then tickets + rescan scratch
// (filter_flag_bytes() in all); [0] and [1] are reset per call.
struct FilterPlan {
    int nw;            // waves per workgroup (4 or 8)
    int nc;            // blocks of 16 output rows per wave (2 or 4)
    int ncols_alloc;
    int nchunks;
    int nsplit;
    int stages_per_split;   // 128-row stages
    int fused = -1, lpc = 0; // Tuning::f32_fused / f32_lpc
    int bound_every = 4;     // Tuning::f32_bound_every
    int64_t aux_elems = 0;   // floats of the rescaled accumulator inits (banks of different scales)
    size_t slots_bytes() const { return (size_t)nsplit * ncols_alloc * 16 * 8; }
    size_t aux_bytes() const { return ((size_t)aux_elems * 4 + 255) & ~(size_t)255; }
    size_t bound_bytes() const { return (size_t)ncols_alloc * 8; }   // best and 2nd best
};
FilterPlan plan_filter(int64_t ncols_pad, int64_t nred_pad, const Tuning& tn);
int filter_empty_bound();
size_t filter_flag_bytes();       // size of the device words launch_filter's `flag` points at
bool filter_usable(const Bank& cols, const Bank& red);   // both banks carry filter planes of compatible scale
hipError_t launch_filter(const Bank& cols, const Bank& red, int ktop, const FilterPlan& plan,
                         unsigned long long* slots, int* bound, int* flag,
                         unsigned long long* partial, hipStream_t stream, bool self = false, float* aux_scratch = nullptr);

// fm_self_dist of a float32-route bank by the triangular sweep (filter_f16.hip, TRI): every distance once
size_t filter_tri_bytes(int ncols_alloc);
hipError_t launch_filter_tri(const Bank& bank, const TriPlan& plan, int bound_every, void* ws, int* bound, int* flag,
                             unsigned long long* partial, hipStream_t stream);

// ---- K9: exact k-NN lists for k up to 8 on the vector ALUs (knn_k.hip; the reference's own calls -- k = 1, 2 -- stay on K1 / K8)
size_t knnk_partial_bytes(int64_t nq, int64_t nt, int k);
hipError_t launch_knnk(const Bank& q, const Bank& t, int k, unsigned long long* partial, int32_t* d_idx, float* d_dist, hipStream_t stream);

// ---- K4: one workgroup per expansion round (rounds.hip) --------------------------------
hipError_t launch_rounds(const Bank& q, const Bank& t, const int32_t* d_q_rows, const int64_t* d_q_off,
                         const int64_t* d_t_off, int64_t n_rounds, int32_t* d_tidx, float* d_dist,
                         double* d_ratio, hipStream_t stream);
int round_qcap();
struct RoundF32;   // round_body_f32.h
hipError_t launch_rounds_f32(const RoundF32& rf, const double* q_selfdist, const int32_t* d_q_rows, const int64_t* d_q_off,
                             const int64_t* d_t_off, int64_t n_rounds, int32_t* d_tidx, float* d_dist,
                             double* d_ratio, hipStream_t stream);

// ---- K7: device-resident expansion loop (expand.hip) ------------------------------------
constexpr int kRRBatchMax = 16;           // bank pairs per batched row-reduce launch
hipError_t launch_rowreduce_batch(int n, const Bank* const* cols, const Bank* const* red, const RowReducePlan* plans,
                                  unsigned long long* const* partial, int* const* bound, hipStream_t stream, bool self = false);
hipError_t launch_expand(const void* d_pairs, int n_pairs, bool f32, int tier, hipStream_t stream);   // all pairs of one kind / capacity tier (expand.hip)
int expand_cand_cap();
int expand_cand_cap_big();

// ---- result gather over RCCL (comm.hip; RCCL is dlopen'ed on first use) -----------------------
int comm_unique_id(void* id128, std::string* err);
int comm_init(int device, int nranks, int rank, const void* id128, void** comm_out, std::string* err);
int comm_destroy(void* comm);
int comm_gather(void* comm, const int32_t* d_rows, const int64_t* d_count, int64_t cap,
                int32_t* d_all_rows, int64_t* d_all_counts, hipStream_t stream, std::string* err);

}  // namespace fm

struct fm_bank : fm::Bank {};
