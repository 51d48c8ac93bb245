// C-ABI of libfastmatch_hip.so (include/fastmatch_hip.h), part 3: host glue of K7, the device-resident
// expansion loop (expand.hip) -- pairs, run states, launches, result fetch.
#include "ctx_internal.h"

#include <chrono>

using namespace fm;

// ---------------------------------------------------------------------------------------
// K7 entry points
// ---------------------------------------------------------------------------------------
// An fm_expand holds what an image pair's runs SHARE and never change (banks, position index, cell
// offsets, target positions) plus a pool of run states (pending stack, seen / found tables, result
// arrays, seed buffer).  One launch may hold several runs of one pair -- the reference is driven as
// pairs x thresholds (turntable.py:59-60) -- each run in a state of its own, one workgroup each.
struct ExpandRun {
    void* blob = nullptr;          // stack | seen | found | m_index | m_pos | m_ratio | result
    double* stack = nullptr;
    unsigned long long* seen = nullptr;
    unsigned long long* found = nullptr;
    int32_t* m_index = nullptr;
    double* m_pos = nullptr;
    double* m_ratio = nullptr;
    long long* result = nullptr;
    long long* resume_state = nullptr;   // loop state of a parked run (lazy targets, delegated cross-checks)
    double* d_seeds = nullptr;     // grown on demand
    int64_t seeds_cap = 0;
    // capacities of THIS run state: they start at the pair's defaults and are multiplied by four when a run
    // ends with the matching FM_EXPAND_*_FULL status (fm_expand_run then repeats the run)
    int64_t match_cap = 0, stack_cap = 0, seen_cap = 0, found_cap = 0;
    // huge tier: the tables of a round whose radius subset does not fit LDS (created when a run first needs them)
    void* huge = nullptr;          // h_cand i32[nq] | h_qbest u64[nq] | h_tbest u64[largest cell] | h_ucand i32[nq] | h_pkey u64[nq]
    // per-round log (fm_expand_set_log): round records | query rows | target rows | ratios; grown fourfold on FM_EXPAND_LOG_FULL
    void* logb = nullptr;
    int64_t lg_round_cap = 0, lg_entry_cap = 0;
};

struct fm_expand {
    ExpandPair dev{};              // device pointers + parameters (run state, seeds and tau filled per run)
    void* blob = nullptr;          // the shared arrays
    int64_t nq = 0;
    int64_t tmax = 0;              // rows of the largest cell
    int tier_hint = 0;             // capacity tier (launch_expand) the pair's last complete run needed: the next run starts there
    // lazy targets (fm_expand_desc.lazy): the target bank grows as the host adds cells
    bool lazy = false;
    fm_bank* lazy_target = nullptr;    // (borrowed) the capacity bank whose arrays dev.t_rows8 / t_norm point into
    const fm_bank* query = nullptr;
    const fm_bank* target = nullptr;
    void* lazy_blob = nullptr;     // cell_start i64[ncells] | cell_cnt i32[ncells] | cell_ready i32[ncells] | resume state
    int64_t ncells = 0, t_cap = 0;
    double* d_tpos = nullptr;      // (inside blob) [t_cap][2]
    int64_t match_cap = 0, stack_cap = 0, seen_cap = 0;       // defaults of a new run state
    bool want_log = false;         // fm_expand_set_log: the runs of this pair write the per-round log
    int64_t log_cap0 = 0;          // ... first capacity of both log arrays (0 = the defaults below)
    // fm_expand_run_lazy: slot 0 is parked at a missing cell (the last launch ended with FM_EXPAND_NEED_CELL) and may be resumed;
    // the seeds of the run that parked (a resume re-uses them: they are on the device)
    bool parked = false;
    int64_t parked_seeds = 0;
    std::vector<ExpandRun> runs;   // run slot k = the k-th run of this pair inside one launch
};

static inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }
static inline int64_t pow2_at_least(int64_t x) { int64_t p = 1; while (p < x) p <<= 1; return p; }

static void expand_run_free(ExpandRun& r)
{
    if (r.blob) (void)hipFree(r.blob);
    if (r.d_seeds) (void)hipFree(r.d_seeds);
    if (r.huge) (void)hipFree(r.huge);
    if (r.logb) (void)hipFree(r.logb);
    r = ExpandRun{};
}

// (Re)allocate the arrays of a run state for its current capacities.
static int expand_run_alloc(fm_ctx* ctx, ExpandRun& r)
{
    if (r.blob) { (void)hipFree(r.blob); r.blob = nullptr; }
    r.found_cap = pow2_at_least(4 * r.match_cap);
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off += al256(bytes > 0 ? bytes : 1); return o; };
    // (seen and found are neighbours: one fill resets both)
    const size_t o_stack = carve((size_t)r.stack_cap * 32), o_seen = carve((size_t)r.seen_cap * 8), o_found = carve((size_t)r.found_cap * 16);
    const size_t o_mi = carve((size_t)r.match_cap * 4), o_mp = carve((size_t)r.match_cap * 32), o_mr = carve((size_t)r.match_cap * 8);
    const size_t o_res = carve((size_t)kExpResultWords * 8), o_resume = carve(256);
    hipError_t e = hipMalloc(&r.blob, off);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        r.blob = nullptr;
        return fail(ctx, FM_ENOMEM, std::string("fm_expand: run state (") + std::to_string(off >> 20) + " MiB): " + hipGetErrorString(e));
    }
    char* b = (char*)r.blob;
    r.stack = (double*)(b + o_stack); r.seen = (unsigned long long*)(b + o_seen); r.found = (unsigned long long*)(b + o_found);
    r.m_index = (int32_t*)(b + o_mi); r.m_pos = (double*)(b + o_mp); r.m_ratio = (double*)(b + o_mr);
    r.result = (long long*)(b + o_res);
    r.resume_state = (long long*)(b + o_resume);
    // (a resume flag without a parked run is refused on the host; the words are defined all the same)
    if (hipMemsetAsync(b + o_res, 0, (size_t)kExpResultWords * 8 + 256, ctx->stream) != hipSuccess) (void)hipGetLastError();
    return FM_OK;
}

// The log arrays of a run state (created with the state's current capacities; 0 = the pair's defaults).
static int expand_run_log(fm_ctx* ctx, const fm_expand* ex, ExpandRun& r)
{
    if (!ex->want_log || r.logb) return FM_OK;
    if (r.lg_round_cap <= 0) r.lg_round_cap = ex->log_cap0 > 0 ? ex->log_cap0 : std::max<int64_t>(4096, 4 * ex->ncells);
    if (r.lg_entry_cap <= 0) r.lg_entry_cap = ex->log_cap0 > 0 ? ex->log_cap0 : std::max<int64_t>(65536, 4 * ex->nq);
    const size_t bytes = al256((size_t)r.lg_round_cap * 48) + al256((size_t)r.lg_entry_cap * 4) * 2 + al256((size_t)r.lg_entry_cap * 8);
    hipError_t e = hipMalloc(&r.logb, bytes);
    if (e != hipSuccess) { (void)hipGetLastError(); r.logb = nullptr; return fail(ctx, FM_ENOMEM, std::string("fm_expand: log arrays: ") + hipGetErrorString(e)); }
    return FM_OK;
}

// Run slot `slot` of the pair exists after this call (slots are created in order).
static int expand_ensure_run(fm_ctx* ctx, fm_expand* ex, size_t slot)
{
    while (ex->runs.size() <= slot) {
        ExpandRun r;
        r.match_cap = ex->match_cap; r.stack_cap = ex->stack_cap; r.seen_cap = ex->seen_cap;
        int rc = expand_run_alloc(ctx, r);
        if (rc != FM_OK) return rc;
        ex->runs.push_back(r);
    }
    return FM_OK;
}

// The global tables of the chunked rounds (expand.hip, HUGE) for one run state.
// (a lazy target's cells are not known yet: room for the whole target bank's capacity)
static size_t tm_of(const fm_expand* ex) { return (size_t)(ex->lazy ? ex->t_cap : (ex->tmax > 0 ? ex->tmax : 1)); }

static int expand_run_huge(fm_ctx* ctx, const fm_expand* ex, ExpandRun& r)
{
    if (r.huge) return FM_OK;
    const size_t nq = (size_t)(ex->nq > 0 ? ex->nq : 1), tm = tm_of(ex);
    hipError_t e = hipMalloc(&r.huge, al256(nq * 4) + al256(nq * 8) + al256(tm * 8) + al256(nq * 4) + al256(nq * 8));
    if (e != hipSuccess) { (void)hipGetLastError(); r.huge = nullptr; return fail(ctx, FM_ENOMEM, std::string("fm_expand: chunked-round tables: ") + hipGetErrorString(e)); }
    return FM_OK;
}

static void expand_bind_run(ExpandPair& P, const ExpandRun& r, const fm_expand* ex = nullptr)
{
    P.h_cand = nullptr; P.h_qbest = nullptr; P.h_tbest = nullptr; P.h_ucand = nullptr; P.h_pkey = nullptr;
    if (r.huge && ex) {
        const size_t nq = (size_t)(ex->nq > 0 ? ex->nq : 1);
        P.h_cand = (int32_t*)r.huge;
        P.h_qbest = (unsigned long long*)((char*)r.huge + al256(nq * 4));
        P.h_tbest = (unsigned long long*)((char*)r.huge + al256(nq * 4) + al256(nq * 8));
        P.h_ucand = (int32_t*)((char*)r.huge + al256(nq * 4) + al256(nq * 8) + al256(tm_of(ex) * 8));
        P.h_pkey = (unsigned long long*)((char*)r.huge + al256(nq * 4) + al256(nq * 8) + al256(tm_of(ex) * 8) + al256(nq * 4));
    }
    P.stack = r.stack; P.stack_cap = r.stack_cap;
    P.seen = r.seen; P.seen_cap = r.seen_cap;
    P.found = r.found; P.found_cap = r.found_cap;
    P.m_index = r.m_index; P.m_pos = r.m_pos; P.m_ratio = r.m_ratio; P.match_cap = r.match_cap;
    P.result = r.result;
    P.resume_state = r.resume_state;
    P.lg_round = nullptr; P.lg_q = nullptr; P.lg_t = nullptr; P.lg_ratio = nullptr; P.lg_round_cap = 0; P.lg_entry_cap = 0;
    if (r.logb) {
        char* lb = (char*)r.logb;
        P.lg_round = (long long*)lb;
        P.lg_q = (int32_t*)(lb + al256((size_t)r.lg_round_cap * 48));
        P.lg_t = (int32_t*)(lb + al256((size_t)r.lg_round_cap * 48) + al256((size_t)r.lg_entry_cap * 4));
        P.lg_ratio = (double*)(lb + al256((size_t)r.lg_round_cap * 48) + 2 * al256((size_t)r.lg_entry_cap * 4));
        P.lg_round_cap = r.lg_round_cap; P.lg_entry_cap = r.lg_entry_cap;
    }
}

// A run that ended with FM_EXPAND_LOG_FULL gets log arrays four times as large (both: which one filled is not told).
// The larger arrays are allocated FIRST; the old ones and their capacities stand when that fails (the run then keeps its
// FM_EXPAND_LOG_FULL status and the caller is not asked to try again at 16 times the size).
static int expand_log_grow(fm_ctx* ctx, const fm_expand* ex, ExpandRun& r)
{
    const int64_t limit = (int64_t)1 << 28;
    if (r.lg_entry_cap >= limit || r.lg_round_cap >= limit) return FM_ENOMEM;
    ExpandRun bigger = r;
    bigger.logb = nullptr;
    bigger.lg_round_cap = r.lg_round_cap * 4; bigger.lg_entry_cap = r.lg_entry_cap * 4;
    const int rc = expand_run_log(ctx, ex, bigger);
    if (rc != FM_OK) return rc;
    if (r.logb) (void)hipFree(r.logb);
    r.logb = bigger.logb; r.lg_round_cap = bigger.lg_round_cap; r.lg_entry_cap = bigger.lg_entry_cap;
    return FM_OK;
}

extern "C" int fm_expand_create(fm_ctx* ctx, const fm_expand_desc* d, fm_expand** out)
{
    if (!ctx) return fail(nullptr, FM_EINVAL, "fm_expand_create: ctx is NULL");
    if (!d || !out) return fail(ctx, FM_EINVAL, "fm_expand_create: NULL argument");
    *out = nullptr;
    if (!d->query || !d->target) return fail(ctx, FM_EINVAL, "fm_expand_create: NULL bank");
    if (d->query->kind != d->target->kind)
        return fail(ctx, FM_EINVAL, "fm_expand_create: query/target kind mismatch");
    const bool f32 = d->query->kind == FM_BANK_F32;
    if (f32 && !filter_usable(*d->target, *d->query))
        return fail(ctx, FM_EUNSUPPORTED, "fm_expand_create: float32 banks without usable fp16 filter planes (non-finite values or scales more than 2^40 apart)");
    if (d->query->dim != d->target->dim) return fail(ctx, FM_EINVAL, "fm_expand_create: dim mismatch");
    if (!d->query->selfdist) return fail(ctx, FM_EINVAL, "fm_expand_create: query bank has no self distances");
    const int64_t nq = d->query->n, nt = d->target->n;
    const int64_t ncells = (int64_t)d->rows * d->cols;
    if (d->width < 1 || d->height < 1 || d->cell_w < 1 || d->cell_h < 1 || d->rows < 1 || d->cols < 1 || d->radius < 0)
        return fail(ctx, FM_EINVAL, "fm_expand_create: bad grid parameters");
    if (d->metric < FM_METRIC_EUCLIDEAN || d->metric > FM_METRIC_CHEBYSHEV) return fail(ctx, FM_EINVAL, "fm_expand_create: unknown metric");
    if (d->rows > 65535 || d->cols > 65535 || d->width > 65535 || d->height > 65535)
        return fail(ctx, FM_EUNSUPPORTED, "fm_expand_create: image or grid too large for 16-bit cell keys");
    const bool lazy = d->lazy != 0;
    if (lazy && d->target->cap_pad <= 0) return fail(ctx, FM_EUNSUPPORTED, "fm_expand_create: lazy targets need a target bank with capacity (fm_bank_create_u8_cap / fm_bank_create_f32_cap)");
    if ((nq > 0 && (!d->query_pos || !d->index_order)) || !d->index_start || (!lazy && (!d->cell_off || (nt > 0 && !d->target_pos))))
        return fail(ctx, FM_EINVAL, "fm_expand_create: NULL array");
    const int64_t nb = (int64_t)d->index_nbx * d->index_nby;
    if (d->index_nbx < 0 || d->index_nby < 0 || d->index_start[nb] != nq || !(d->index_bucket > 0.0))
        return fail(ctx, FM_EINVAL, "fm_expand_create: inconsistent position index");
    if (!lazy) {
        if (d->cell_off[0] != 0 || d->cell_off[ncells] != nt) return fail(ctx, FM_EINVAL, "fm_expand_create: cell_off must cover the target bank");
        for (int64_t c = 0; c < ncells; ++c)
            if (d->cell_off[c + 1] < d->cell_off[c]) return fail(ctx, FM_EINVAL, "fm_expand_create: cell_off not monotonic");
    }
    for (int64_t i = 0; i < nq; ++i) {
        if (d->index_order[i] < 0 || d->index_order[i] >= nq) return fail(ctx, FM_EINVAL, "fm_expand_create: index_order out of range");
        const double x = d->query_pos[2 * i], y = d->query_pos[2 * i + 1];
        if (!(x >= 0.0) || !(y >= 0.0) || x / d->cell_w >= 65535.0 || y / d->cell_h >= 65535.0)
            return fail(ctx, FM_EUNSUPPORTED, "fm_expand_create: query position outside the 16-bit cell-key range");
    }
    // the positions once more in the index's order: the radius query reads them beside the keypoint indices
    std::vector<double> pos_ord((size_t)nq * 2);
    for (int64_t j = 0; j < nq; ++j) {
        pos_ord[2 * j] = d->query_pos[2 * (size_t)d->index_order[j]];
        pos_ord[2 * j + 1] = d->query_pos[2 * (size_t)d->index_order[j] + 1];
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    fm_expand* ex = new (std::nothrow) fm_expand();
    if (!ex) return fail(ctx, FM_ENOMEM, "fm_expand_create: out of host memory");
    ex->nq = nq;
    if (!lazy) for (int64_t c = 0; c < ncells; ++c) ex->tmax = std::max<int64_t>(ex->tmax, d->cell_off[c + 1] - d->cell_off[c]);
    ex->lazy = lazy; ex->lazy_target = lazy ? const_cast<fm_bank*>(d->target) : nullptr; ex->query = d->query; ex->target = d->target;
    ex->ncells = ncells; ex->t_cap = lazy ? d->target->cap_pad : nt;
    ex->match_cap = d->match_cap > 0 ? d->match_cap : (4 * nq > 1024 ? 4 * nq : 1024);
    ex->stack_cap = d->stack_cap > 0 ? d->stack_cap : (64 * ncells > 65536 ? 64 * ncells : 65536);
    ex->seen_cap = pow2_at_least(16 * ncells > 65536 ? 16 * ncells : 65536);
    // the shared arrays in one allocation
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off += al256(bytes > 0 ? bytes : 1); return o; };
    const size_t o_qpos = carve((size_t)nq * 16), o_order = carve((size_t)nq * 4), o_start = carve((size_t)(nb + 1) * 4);
    const size_t o_coff = carve((size_t)(ncells + 1) * 8), o_tpos = carve((size_t)(lazy ? d->target->cap_pad : nt) * 16), o_qord = carve((size_t)nq * 16);
    const size_t o_cstart = carve(lazy ? (size_t)ncells * 8 : 0), o_ccnt = carve(lazy ? (size_t)ncells * 4 : 0), o_cready = carve(lazy ? (size_t)ncells * 4 : 0);
    const size_t o_resume = carve(lazy ? 128 : 0);
    hipError_t e = hipMalloc(&ex->blob, off);
    if (e != hipSuccess) { (void)hipGetLastError(); delete ex; return fail(ctx, FM_ENOMEM, std::string("fm_expand_create: hipMalloc: ") + hipGetErrorString(e)); }
    char* b = (char*)ex->blob;
    auto bail = [&](int code) { for (auto& r : ex->runs) expand_run_free(r); (void)hipFree(ex->blob); delete ex; return code; };
#define ETRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { (void)hipGetLastError(); \
        return bail(fail(ctx, FM_EDEVICE, std::string(#expr " failed: ") + hipGetErrorString(_e))); } } while (0)
    if (nq) ETRY(hipMemcpyAsync(b + o_qpos, d->query_pos, (size_t)nq * 16, hipMemcpyHostToDevice, ctx->stream));
    if (nq) ETRY(hipMemcpyAsync(b + o_order, d->index_order, (size_t)nq * 4, hipMemcpyHostToDevice, ctx->stream));
    if (nq) ETRY(hipMemcpyAsync(b + o_qord, pos_ord.data(), (size_t)nq * 16, hipMemcpyHostToDevice, ctx->stream));
    ETRY(hipMemcpyAsync(b + o_start, d->index_start, (size_t)(nb + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
    if (!lazy) {
        ETRY(hipMemcpyAsync(b + o_coff, d->cell_off, (size_t)(ncells + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
        if (nt) ETRY(hipMemcpyAsync(b + o_tpos, d->target_pos, (size_t)nt * 16, hipMemcpyHostToDevice, ctx->stream));
    } else {
        // no cell is there yet: start 0, count 0, not ready
        ETRY(hipMemsetAsync(b + o_cstart, 0, (o_resume + 128) - o_cstart, ctx->stream));
    }
    ETRY(hipStreamSynchronize(ctx->stream));
#undef ETRY
    ExpandPair& P = ex->dev;
    P.q_rows8 = d->query->rows8; P.q_norm = d->query->norm; P.q_selfdist = d->query->selfdist;
    P.q_pos = (const double*)(b + o_qpos); P.q_pos_ord = (const double*)(b + o_qord);
    P.idx_order = (const int32_t*)(b + o_order); P.idx_start = (const int32_t*)(b + o_start);
    P.idx_bucket = d->index_bucket; P.idx_x0 = d->index_x0; P.idx_y0 = d->index_y0;
    P.idx_nbx = d->index_nbx; P.idx_nby = d->index_nby;
    P.metric = d->metric;
    P.t_rows8 = d->target->rows8; P.t_norm = d->target->norm;
    P.f32 = f32 ? 1 : 0;
    P.tie_guard = (!f32 && sqrt_tie_possible(*d->query, *d->target)) ? 1 : 0;
    P.rf = RoundF32{};
    if (f32) fill_round_f32(&P.rf, *d->query, *d->target);
    P.cell_off = (const int64_t*)(b + o_coff); P.t_pos = (const double*)(b + o_tpos);
    P.cell_start = lazy ? (const int64_t*)(b + o_cstart) : nullptr; P.cell_cnt = lazy ? (const int32_t*)(b + o_ccnt) : nullptr;
    P.cell_ready = lazy ? (const int32_t*)(b + o_cready) : nullptr; P.resume_state = lazy ? (long long*)(b + o_resume) : nullptr;
    P.resume = 0;
    ex->d_tpos = (double*)(b + o_tpos);
    P.width = d->width; P.height = d->height; P.cell_w = d->cell_w; P.cell_h = d->cell_h;
    P.rows = d->rows; P.cols = d->cols; P.margin = d->margin; P.radius = d->radius;
    P.seeds = nullptr; P.n_seeds = 0; P.tau = 0.0;
    P.prof = 0;                        // (set per run from the context's expand_prof option)
    // the first run state exists from the start (a pair that cannot get one fails here, not at its first run)
    int rc = expand_ensure_run(ctx, ex, 0);
    if (rc != FM_OK) return bail(rc);
    expand_bind_run(P, ex->runs[0]);
    *out = ex;
    return FM_OK;
}

// A computed cell of a lazy target: its rows [first_row, first_row + n_rows) of the target bank (fm_bank_append_u8) and their
// full-image positions; n_rows may be 0 (a cell without features, fastmatch.pyx:155-156).  The cell is ready afterwards.
extern "C" int fm_expand_set_cell(fm_ctx* ctx, fm_expand* ex, int32_t cell, int64_t first_row, int64_t n_rows, const double* pos)
{
    if (!ctx || !ex) return fail(ctx, FM_EINVAL, "fm_expand_set_cell: NULL argument");
    if (!ex->lazy) return fail(ctx, FM_EINVAL, "fm_expand_set_cell: the pair was not created with lazy targets");
    if (cell < 0 || cell >= ex->ncells || first_row < 0 || n_rows < 0 || n_rows > INT32_MAX || first_row + n_rows > ex->lazy_target->n ||
        first_row + n_rows > ex->t_cap || (n_rows > 0 && !pos))
        return fail(ctx, FM_EINVAL, "fm_expand_set_cell: cell or row range out of bounds");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int32_t cnt = (int32_t)n_rows, one = 1;
    if (n_rows) HIP_TRY(ctx, hipMemcpyAsync(ex->d_tpos + 2 * first_row, pos, (size_t)n_rows * 16, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync((void*)(ex->dev.cell_start + cell), &first_row, 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync((void*)(ex->dev.cell_cnt + cell), &cnt, 4, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync((void*)(ex->dev.cell_ready + cell), &one, 4, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ex->tmax = std::max<int64_t>(ex->tmax, n_rows);
    return FM_OK;
}

// One run of a lazy pair (run slot 0): from the start (resume == 0: tables cleared, seeds uploaded) or from where the last
// launch parked (resume != 0, after fm_expand_set_cell of the cell it asked for).  status FM_EXPAND_NEED_CELL: *need_cell is
// the cell (col * rows + row) to compute; 0: done, fetch with fm_expand_fetch; anything else: the device gave up (a
// capacity of the first kernel: lazy runs are not repeated in larger variants -- host loop).
extern "C" int fm_expand_run_lazy(fm_ctx* ctx, fm_expand* ex, const double* seeds, int64_t n_seeds, double tau, int32_t resume,
                                  int64_t* n_matches, int64_t* n_rounds, int64_t* n_pairs, int32_t* status, int32_t* need_cell)
{
    if (!ctx || !ex) return fail(ctx, FM_EINVAL, "fm_expand_run_lazy: NULL argument");
    if (!ex->lazy) return fail(ctx, FM_EINVAL, "fm_expand_run_lazy: the pair was not created with lazy targets");
    if (n_seeds < 0 || (n_seeds > 0 && !seeds && !resume)) return fail(ctx, FM_EINVAL, "fm_expand_run_lazy: bad seeds");
    //  a resume restores the loop state the LAST launch saved: only a run that parked at a missing cell has one
    if (resume && !ex->parked) return fail(ctx, FM_EINVAL, "fm_expand_run_lazy: resume without a parked run (the last launch did not end with FM_EXPAND_NEED_CELL)");
    if (resume) n_seeds = ex->parked_seeds;          // (they are on the device since the run began)
    ex->parked = false;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = expand_ensure_run(ctx, ex, 0);
    if (rc != FM_OK) return rc;
    ExpandRun* r = &ex->runs[0];
    if ((rc = expand_run_log(ctx, ex, *r)) != FM_OK) return rc;
    if ((rc = ws_ensure(ctx, &ctx->ws_in, &ctx->ws_in_bytes, sizeof(ExpandPair) + 64)) != FM_OK) return rc;
    CallScope cs(ctx);
    if (!resume) {
        if (n_seeds > r->seeds_cap) {
            if (r->d_seeds) { HIP_TRY(ctx, hipFree(r->d_seeds)); r->d_seeds = nullptr; r->seeds_cap = 0; }
            const int64_t cap = n_seeds + n_seeds / 2 + 64;
            HIP_TRY(ctx, hipMalloc((void**)&r->d_seeds, (size_t)cap * 32));
            r->seeds_cap = cap;
        }
        if (n_seeds) HIP_TRY(ctx, hipMemcpyAsync(r->d_seeds, seeds, (size_t)n_seeds * 32, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemsetAsync(r->seen, 0xff, (size_t)((char*)r->found - (char*)r->seen) + (size_t)r->found_cap * 16, ctx->stream));
    }
    if ((rc = expand_run_huge(ctx, ex, *r)) != FM_OK) return rc;       // the chunked rounds' tables (the lazy kernel is the chunked one)
    ExpandPair host = ex->dev;
    expand_bind_run(host, *r, ex);
    host.seeds = r->d_seeds; host.n_seeds = n_seeds; host.tau = tau; host.prof = 0;
    host.resume = resume ? 1 : 0;
    const bool f32 = ex->query->kind == FM_BANK_F32;
    // (the target bank has grown since the pair was made: the float32-root guard / the float32 round's scale terms are re-derived)
    host.tie_guard = (!f32 && sqrt_tie_possible(*ex->query, *ex->lazy_target)) ? 1 : 0;
    if (f32) {
        if (!filter_usable(*ex->lazy_target, *ex->query)) return fail(ctx, FM_EUNSUPPORTED, "fm_expand_run_lazy: the growing float32 bank lost its fp16 planes");
        fill_round_f32(&host.rf, *ex->query, *ex->lazy_target);
    }
    // big rounds' cross-checks go to the dense kernels, as in fm_expand_run: the run parks with status 8 and is resumed here
    host.delegate_min = host.tie_guard ? 0 : ctx->tune.expand_delegate;
    HIP_TRY(ctx, hipEventRecord(ctx->ev_k0, ctx->stream));
    long long res[16] = {0};
    bool log_stuck = false;              // no memory for a larger log: FM_EXPAND_LOG_FULL stands
    for (int grown = 0;;) {
        HIP_TRY(ctx, hipMemcpyAsync(ctx->ws_in, &host, sizeof(ExpandPair), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, launch_expand(ctx->ws_in, 1, f32, 3, ctx->stream));
        HIP_TRY(ctx, hipEventRecord(ctx->ev_k1, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(res, r->result, sizeof(res), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (res[3] == 9 && grown < 12 && !log_stuck && (log_stuck = expand_log_grow(ctx, ex, *r) != FM_OK) == false) {
            // the log filled: larger arrays and the run from its start (the cells computed so far stay)
            ++grown;
            HIP_TRY(ctx, hipMemsetAsync(r->seen, 0xff, (size_t)((char*)r->found - (char*)r->seen) + (size_t)r->found_cap * 16, ctx->stream));
            expand_bind_run(host, *r, ex);
            host.resume = 0;
            continue;
        }
        if (res[3] != 8) break;
        if ((rc = round_xcheck_dense(ctx, *ex->query, host.h_cand, res[5], *ex->lazy_target, res[6], res[7], host.h_qbest)) != FM_OK) return rc;
        if (ctx->tune.delegated_rounds < INT32_MAX) ++ctx->tune.delegated_rounds;
        host.resume = 2;
    }
    ctx->kernel_timed = true;
    if ((rc = cs.finish()) != FM_OK) return rc;
    if (n_matches) *n_matches = res[0];
    if (n_rounds) *n_rounds = res[1];
    if (n_pairs) *n_pairs = res[2];
    if (status) *status = (int32_t)res[3];
    if (need_cell) *need_cell = (int32_t)res[4];
    ex->parked = res[3] == 7;
    ex->parked_seeds = n_seeds;
    return FM_OK;
}

// ---- per-round log (options["log"], fastmatch.pyx:79-80, 172-180) on the device ---------------------------------------
extern "C" int fm_expand_set_log(fm_ctx* ctx, fm_expand* ex, int32_t enable)
{
    if (!ctx || !ex) return fail(ctx, FM_EINVAL, "fm_expand_set_log: NULL argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ex->want_log = enable != 0;
    ex->log_cap0 = enable > 1 ? enable : 0;
    if (!ex->want_log || ex->log_cap0 > 0)
        for (auto& r : ex->runs) if (r.logb) { (void)hipFree(r.logb); r.logb = nullptr; r.lg_round_cap = 0; r.lg_entry_cap = 0; }
    return FM_OK;
}

static int expand_log_slot(fm_ctx* ctx, const fm_expand* ex, int32_t slot, const ExpandRun** r, const char* who)
{
    if (!ctx || !ex) return fail(ctx, FM_EINVAL, std::string(who) + ": NULL argument");
    if (slot < 0 || (size_t)slot >= ex->runs.size()) return fail(ctx, FM_EINVAL, std::string(who) + ": the pair has no such run slot");
    if (!ex->runs[(size_t)slot].logb) return fail(ctx, FM_EINVAL, std::string(who) + ": the run wrote no log (fm_expand_set_log)");
    *r = &ex->runs[(size_t)slot];
    return FM_OK;
}

extern "C" int fm_expand_log_counts(fm_ctx* ctx, const fm_expand* ex, int32_t slot, int64_t* n_rounds, int64_t* n_entries)
{
    const ExpandRun* r = nullptr;
    int rc = expand_log_slot(ctx, ex, slot, &r, "fm_expand_log_counts");
    if (rc != FM_OK) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    long long res[16];
    HIP_TRY(ctx, hipMemcpyAsync(res, r->result, sizeof(res), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (n_rounds) *n_rounds = res[1];
    if (n_entries) *n_entries = res[8];
    return FM_OK;
}

extern "C" int fm_expand_fetch_log(fm_ctx* ctx, const fm_expand* ex, int32_t slot, int64_t n_rounds, int64_t n_entries,
                                   int64_t* rounds, int32_t* query_row, int32_t* target_row, double* ratio)
{
    const ExpandRun* r = nullptr;
    int rc = expand_log_slot(ctx, ex, slot, &r, "fm_expand_fetch_log");
    if (rc != FM_OK) return rc;
    if (n_rounds < 0 || n_rounds > r->lg_round_cap || n_entries < 0 || n_entries > r->lg_entry_cap)
        return fail(ctx, FM_EINVAL, "fm_expand_fetch_log: counts out of range");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ExpandPair P{};
    expand_bind_run(P, *r);
    if (n_rounds && rounds) HIP_TRY(ctx, hipMemcpyAsync(rounds, P.lg_round, (size_t)n_rounds * 48, hipMemcpyDeviceToHost, ctx->stream));
    if (n_entries && query_row) HIP_TRY(ctx, hipMemcpyAsync(query_row, P.lg_q, (size_t)n_entries * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (n_entries && target_row) HIP_TRY(ctx, hipMemcpyAsync(target_row, P.lg_t, (size_t)n_entries * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (n_entries && ratio) HIP_TRY(ctx, hipMemcpyAsync(ratio, P.lg_ratio, (size_t)n_entries * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return FM_OK;
}

// Bytes one run state of the pair takes at its first capacities (stack + tables + result arrays), and how many
// states exist: a caller that puts pairs x thresholds into one launch sizes the launch against free memory with it.
extern "C" int fm_expand_info(const fm_expand* ex, int64_t* state_bytes, int32_t* n_slots)
{
    if (!ex) return fail(nullptr, FM_EINVAL, "fm_expand_info: NULL pair");
    if (state_bytes) {
        const int64_t found_cap = pow2_at_least(4 * ex->match_cap);
        *state_bytes = (int64_t)(al256((size_t)ex->stack_cap * 32) + al256((size_t)ex->seen_cap * 8) + al256((size_t)found_cap * 16) +
                                 al256((size_t)ex->match_cap * 4) + al256((size_t)ex->match_cap * 32) + al256((size_t)ex->match_cap * 8) +
                                 al256((size_t)kExpResultWords * 8) + 256);       // (result words + resume state: expand_run_alloc)
        if (ex->want_log) {              // the log arrays at their first capacities (expand_run_log)
            const int64_t rc0 = ex->log_cap0 > 0 ? ex->log_cap0 : std::max<int64_t>(4096, 4 * ex->ncells);
            const int64_t ec0 = ex->log_cap0 > 0 ? ex->log_cap0 : std::max<int64_t>(65536, 4 * ex->nq);
            *state_bytes += (int64_t)(al256((size_t)rc0 * 48) + 2 * al256((size_t)ec0 * 4) + al256((size_t)ec0 * 8));
        }
    }
    if (n_slots) *n_slots = (int32_t)ex->runs.size();
    return FM_OK;
}

// Free the run states from slot `keep` on (keep >= 1: slot 0 exists as long as the pair does).
extern "C" int fm_expand_trim(fm_ctx* ctx, fm_expand* ex, int32_t keep)
{
    if (!ctx || !ex) return fail(ctx, FM_EINVAL, "fm_expand_trim: NULL argument");
    if (keep < 1) return fail(ctx, FM_EINVAL, "fm_expand_trim: keep < 1");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    while ((int32_t)ex->runs.size() > keep) { expand_run_free(ex->runs.back()); ex->runs.pop_back(); }
    return FM_OK;
}

extern "C" int fm_mem_info(fm_ctx* ctx, int64_t* free_bytes, int64_t* total_bytes)
{
    if (!ctx) return fail(nullptr, FM_EINVAL, "fm_mem_info: ctx is NULL");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    size_t f = 0, t = 0;
    HIP_TRY(ctx, hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = (int64_t)f;
    if (total_bytes) *total_bytes = (int64_t)t;
    return FM_OK;
}

extern "C" int fm_expand_destroy(fm_ctx* ctx, fm_expand* ex)
{
    if (!ex) return FM_OK;
    if (ctx) { (void)hipSetDevice(ctx->device); (void)hipStreamSynchronize(ctx->stream); }
    for (auto& r : ex->runs) expand_run_free(r);
    if (ex->blob) (void)hipFree(ex->blob);
    delete ex;
    return FM_OK;
}

// Run slot of entry i of a launch: how many earlier entries name the same pair.
static int expand_slot_of(fm_expand* const* pairs, int i)
{
    int k = 0;
    for (int j = 0; j < i; ++j) k += pairs[j] == pairs[i] ? 1 : 0;
    return k;
}

extern "C" int fm_expand_run(fm_ctx* ctx, int32_t n, fm_expand* const* pairs, const double* const* seeds,
                             const int64_t* n_seeds, const double* tau, int64_t* n_matches,
                             int64_t* n_rounds, int64_t* n_pairs, int32_t* status)
{
    if (!ctx) return fail(nullptr, FM_EINVAL, "fm_expand_run: ctx is NULL");
    if (n < 0) return fail(ctx, FM_EINVAL, "fm_expand_run: n < 0");
    if (n == 0) return FM_OK;
    if (!pairs || !seeds || !n_seeds || !tau) return fail(ctx, FM_EINVAL, "fm_expand_run: NULL argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    std::vector<ExpandPair> host((size_t)n);
    std::vector<ExpandRun*> run((size_t)n);
    int rc;
    {
        // slots by occurrence (a map instead of expand_slot_of's quadratic scan: a launch may hold thousands of runs)
        std::map<fm_expand*, int> seen_pairs;
        for (int i = 0; i < n; ++i) {
            fm_expand* ex = pairs[i];
            if (!ex) return fail(ctx, FM_EINVAL, "fm_expand_run: NULL pair");
            if (ex->lazy) return fail(ctx, FM_EINVAL, "fm_expand_run: a lazy pair is driven with fm_expand_run_lazy");
            if (n_seeds[i] < 0 || (n_seeds[i] > 0 && !seeds[i])) return fail(ctx, FM_EINVAL, "fm_expand_run: bad seeds");
            const int slot = seen_pairs[ex]++;
            if ((rc = expand_ensure_run(ctx, ex, (size_t)slot)) != FM_OK) return rc;
            if ((rc = expand_run_log(ctx, ex, ex->runs[(size_t)slot])) != FM_OK) return rc;
        }
        seen_pairs.clear();
        for (int i = 0; i < n; ++i) {                      // (pointers into runs[] are taken once the vectors stopped growing)
            fm_expand* ex = pairs[i];
            ExpandRun* r = &ex->runs[(size_t)seen_pairs[ex]++];
            run[(size_t)i] = r;
            if (n_seeds[i] > r->seeds_cap) {
                if (r->d_seeds) { HIP_TRY(ctx, hipFree(r->d_seeds)); r->d_seeds = nullptr; r->seeds_cap = 0; }
                const int64_t cap = n_seeds[i] + n_seeds[i] / 2 + 64;
                HIP_TRY(ctx, hipMalloc((void**)&r->d_seeds, (size_t)cap * 32));
                r->seeds_cap = cap;
            }
        }
    }
    rc = ws_ensure(ctx, &ctx->ws_in, &ctx->ws_in_bytes, (size_t)n * sizeof(ExpandPair) + 64);
    if (rc != FM_OK) return rc;
    CallScope cs(ctx);
    for (int i = 0; i < n; ++i) {
        fm_expand* ex = pairs[i];
        ExpandRun* r = run[(size_t)i];
        if (n_seeds[i]) HIP_TRY(ctx, hipMemcpyAsync(r->d_seeds, seeds[i], (size_t)n_seeds[i] * 32, hipMemcpyHostToDevice, ctx->stream));
        // (the two tables are neighbours in the run's allocation: one fill)
        HIP_TRY(ctx, hipMemsetAsync(r->seen, 0xff, (size_t)((char*)r->found - (char*)r->seen) + (size_t)r->found_cap * 16, ctx->stream));
        host[i] = ex->dev;
        expand_bind_run(host[i], *r);
        host[i].seeds = r->d_seeds;
        host[i].n_seeds = n_seeds[i];
        host[i].tau = tau[i];
        host[i].prof = ctx->tune.expand_prof;
        host[i].resume = 0;
        host[i].delegate_min = 0;
    }
    // the int8 and the float32 pairs are different kernels, and so are the capacity tiers of the int8 one (a pair starts
    // in the tier its last complete run needed): descriptors grouped by kernel, one launch each
    std::vector<char> big((size_t)n, 0);          // capacity tier of the run's last launch (launch_expand)
    auto huge_ok = [&](int i) { return ctx->tune.expand_huge != 0; };
    for (int i = 0; i < n; ++i) {
        int t = pairs[i]->tier_hint;
        if (t == 2 && (!huge_ok(i) || expand_run_huge(ctx, pairs[i], *run[(size_t)i]) != FM_OK)) t = 1;
        if (t == 1 && (host[i].f32 || !ctx->tune.expand_big)) t = 0;
        big[(size_t)i] = (char)t;
        if (t == 2) expand_bind_run(host[i], *run[(size_t)i], pairs[i]);
    }
    // Chunked runs (both descriptor kinds) hand a big round's cross-check to the dense kernels (expand.hip, DELEGATED); phase timers off then
    // (a parked run's timers would restart)
    auto set_delegate = [&](int i) {
        host[i].delegate_min = (big[(size_t)i] == 2 && !host[i].tie_guard && !host[i].prof) ? ctx->tune.expand_delegate : 0;
    };
    for (int i = 0; i < n; ++i) set_delegate(i);
    // kernel of a run: int8 tiers 0 / 1 / 2, float32 tiers 0 / 2
    auto group_of = [&](int i) { return host[i].f32 ? (big[(size_t)i] == 2 ? 4 : 3) : (int)big[(size_t)i]; };
    // (function scope: the asynchronous copy below reads it until the next stream synchronisation -- ADVICE r04)
    std::vector<ExpandPair> grouped;
    auto launch_groups = [&](const std::vector<int>& idx) -> int {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));       // the previous launch's copy out of `grouped` / into ws_in is done
        grouped.clear();
        grouped.reserve(idx.size());
        int cnt[5] = {0, 0, 0, 0, 0};
        for (int g = 0; g < 5; ++g)
            for (int i : idx) if (group_of(i) == g) { grouped.push_back(host[i]); ++cnt[g]; }
        HIP_TRY(ctx, hipMemcpyAsync(ctx->ws_in, grouped.data(), grouped.size() * sizeof(ExpandPair), hipMemcpyHostToDevice, ctx->stream));
        size_t at = 0;
        for (int g = 0; g < 5; ++g) {
            if (cnt[g] > 0)
                HIP_TRY(ctx, launch_expand((const char*)ctx->ws_in + at * sizeof(ExpandPair), cnt[g], g >= 3, g == 4 ? 2 : (g == 3 ? 0 : g), ctx->stream));
            at += (size_t)cnt[g];
        }
        return FM_OK;
    };
    std::vector<int> all_runs((size_t)n);
    for (int i = 0; i < n; ++i) all_runs[(size_t)i] = i;
    HIP_TRY(ctx, hipEventRecord(ctx->ev_k0, ctx->stream));
    if ((rc = launch_groups(all_runs)) != FM_OK) return rc;
    HIP_TRY(ctx, hipEventRecord(ctx->ev_k1, ctx->stream));
    ctx->kernel_timed = true;
    // per run: n_matches, n_rounds, n_pairs, status | (a parked run) cell, subset size, first train row, train rows (expand_pair.h)
    std::vector<long long> res((size_t)n * 8);
    for (int i = 0; i < n; ++i)
        HIP_TRY(ctx, hipMemcpyAsync(&res[(size_t)i * 8], run[(size_t)i]->result, 64, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    // Parked runs (status 8: a round's cross-check is wanted from the dense kernels): per run the subset's rows are in its
    // h_cand[], the round's (subset size, first train row, train rows) in its resume state; K1 + the election fill its
    // h_qbest[] on the whole GPU, then the parked runs are launched again with resume = 2 -- until none parks.
    int64_t delegated = 0;
    const bool dbg = ctx->dbg_expand;
    double t_dense = 0.0, t_launch = 0.0, t_wait = 0.0;      // host seconds: enqueueing the dense kernels / the resumed runs / waiting
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    auto settle_parked = [&](const std::vector<int>& among) -> int {
        std::vector<int> parked;
        for (int i : among) if (res[(size_t)i * 8 + 3] == 8) parked.push_back(i);
        while (!parked.empty()) {
            const double t0 = dbg ? now() : 0.0;
            for (size_t k = 0; k < parked.size(); ++k) {
                const int i = parked[k];
                const long long* pk = &res[(size_t)i * 8 + 5];          // subset size, first train row, train rows (beside the results)
                int rc2 = round_xcheck_dense(ctx, *pairs[i]->query, host[i].h_cand, pk[0], *pairs[i]->target, pk[1], pk[2], host[i].h_qbest);
                if (rc2 != FM_OK) return rc2;
                host[i].resume = 2;
                ++delegated;
            }
            const double t1 = dbg ? now() : 0.0;
            int rc2 = launch_groups(parked);
            if (rc2 != FM_OK) return rc2;
            HIP_TRY(ctx, hipEventRecord(ctx->ev_k1, ctx->stream));
            for (int i : parked) {
                host[i].resume = 0;
                HIP_TRY(ctx, hipMemcpyAsync(&res[(size_t)i * 8], run[(size_t)i]->result, 64, hipMemcpyDeviceToHost, ctx->stream));
            }
            const double t2 = dbg ? now() : 0.0;
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            if (dbg) { t_dense += t1 - t0; t_launch += t2 - t1; t_wait += now() - t2; }
            std::vector<int> again;
            for (int i : parked) if (res[(size_t)i * 8 + 3] == 8) again.push_back(i);
            parked.swap(again);
        }
        return FM_OK;
    };
    if ((rc = settle_parked(all_runs)) != FM_OK) return rc;
    // Runs that ended on a capacity run again, from the start: a radius subset beyond the kernel's 2048 rows
    // (status 2, int8 banks) in the 4096-row variant of the kernel and, beyond that, in the variant that takes a
    // subset of any size in chunks (option expand_huge; not for pairs under the float32-root guard); a full pending stack, result list or
    // hash table (status 1, 4, 5: thresholds above 1 accept nearly every cross-checked pair and the
    // expansion heads for every (cell, query cell) combination) in a run state four times as large, at most
    // `expand_grow` times over (default 2; the status stands after that).  The other runs keep their results.
    // (the budgets are per run and per kind: a tier step or a larger log must not use up the run state's growth steps)
    std::vector<int> cap_grows((size_t)n * 3, 0), log_grows((size_t)n, 0);       // [run][stack | result list | table]
    for (int pass = 0; pass < 40; ++pass) {
        std::vector<int> redo;
        for (int i = 0; i < n; ++i) {
            const long long st = res[(size_t)i * 8 + 3];
            ExpandRun* r = run[(size_t)i];
            if (st == 2 && !host[i].f32 && ctx->tune.expand_big && big[(size_t)i] == 0) { big[(size_t)i] = 1; redo.push_back(i); continue; }
            if (st == 2 && huge_ok(i) && big[(size_t)i] <= 1) {
                if (expand_run_huge(ctx, pairs[i], *r) == FM_OK) {
                    big[(size_t)i] = 2;
                    expand_bind_run(host[i], *r, pairs[i]);
                    set_delegate(i);
                    redo.push_back(i);
                }
                continue;                                 // (no memory for the tables: the status stands)
            }
            if (st == 9 && log_grows[(size_t)i] < 12) {                 // the log filled: larger arrays, the run again
                if (expand_log_grow(ctx, pairs[i], *r) != FM_OK) { log_grows[(size_t)i] = 12; continue; }   // (no memory: the status stands)
                ++log_grows[(size_t)i];
                expand_bind_run(host[i], *r, big[(size_t)i] == 2 ? pairs[i] : nullptr);
                redo.push_back(i);
                continue;
            }
            if ((st == 1 || st == 4 || st == 5) && cap_grows[(size_t)i * 3 + (st == 1 ? 0 : st == 4 ? 1 : 2)] < ctx->tune.expand_grow) {
                ++cap_grows[(size_t)i * 3 + (st == 1 ? 0 : st == 4 ? 1 : 2)];
                const int64_t limit = (int64_t)1 << 28;
                if (st == 1) { if (r->stack_cap >= limit) continue; r->stack_cap *= 4; }
                if (st == 4) { if (r->match_cap >= limit) continue; r->match_cap *= 4; }
                if (st == 5) { if (r->seen_cap >= limit) continue; r->seen_cap *= 4; }
                if (expand_run_alloc(ctx, *r) != FM_OK) {                 // no memory for the larger state: the status stands
                    r->match_cap = pairs[i]->match_cap; r->stack_cap = pairs[i]->stack_cap; r->seen_cap = pairs[i]->seen_cap;
                    if ((rc = expand_run_alloc(ctx, *r)) != FM_OK) return rc;
                    continue;
                }
                expand_bind_run(host[i], *r, pairs[i]);
                redo.push_back(i);
            }
        }
        if (redo.empty()) break;
        for (int i : redo)
            HIP_TRY(ctx, hipMemsetAsync(run[(size_t)i]->seen, 0xff, (size_t)((char*)run[(size_t)i]->found - (char*)run[(size_t)i]->seen) +
                                        (size_t)run[(size_t)i]->found_cap * 16, ctx->stream));
        if ((rc = launch_groups(redo)) != FM_OK) return rc;
        HIP_TRY(ctx, hipEventRecord(ctx->ev_k1, ctx->stream));
        for (int i : redo)
            HIP_TRY(ctx, hipMemcpyAsync(&res[(size_t)i * 8], run[(size_t)i]->result, 64, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));         // (ws_in is reused by the next pass)
        if ((rc = settle_parked(redo)) != FM_OK) return rc;
    }
    for (int i = 0; i < n; ++i) {
        ctx->pending_pairs += res[(size_t)i * 8 + 2];
        if (res[(size_t)i * 8 + 3] == 0 && (int)big[(size_t)i] > pairs[i]->tier_hint) pairs[i]->tier_hint = (int)big[(size_t)i];
    }
    ctx->tune.delegated_rounds = (int)std::min<int64_t>((int64_t)INT32_MAX, (int64_t)ctx->tune.delegated_rounds + delegated);
    if (ctx->dbg_park) {               // (a library built with -DFM_PARK_PROF: expand.hip)
        long long pr[kExpResultWords];
        (void)hipMemcpy(pr, run[0]->result, sizeof(pr), hipMemcpyDeviceToHost);
        static const char* hn[7] = {"pop + radius walk", "list + histogram + bounds", "partition", "chunk sorts (+ x-check)", "step 4", "step 5", "rounds that fit"};
        fprintf(stderr, "[fm_expand_run, run 0] %lld rounds beyond the LDS tables; thread 0's clock, ms: ", pr[39]);
        for (int k = 0; k < 7; ++k) fprintf(stderr, "%s %.2f  ", hn[k], pr[32 + k] * 1e-5);
        fprintf(stderr, "\n");
    }
    if (delegated > 0 && dbg)
        fprintf(stderr, "[fm_expand_run] %lld cross-checks delegated to the dense kernels; host seconds: enqueue dense %.4f, enqueue resume %.4f, "
                        "wait %.4f\n", (long long)delegated, t_dense, t_launch, t_wait);
    if (ctx->tune.expand_prof) {
        long long pr[kExpResultWords];
        (void)hipMemcpy(pr, run[0]->result, sizeof(pr), hipMemcpyDeviceToHost);
        static const char* names[12] = {"pop:barrier", "radius", "sort", "x1_tail", "compact", "neigh+push+emit", "end", "pop:thread0",
                                        "x1:bfrag+barrier", "x1:gather", "x1:mfma", "x1:merge"};
        fprintf(stderr, "[fm_expand prof, run 0, %lld rounds] ", pr[1]);
        for (int k = 0; k < 12; ++k) fprintf(stderr, "%s %.2f us  ", names[k], pr[1] ? pr[16 + k] * 0.01 / (double)pr[1] : 0.0);
        fprintf(stderr, "\n");
    }
    rc = cs.finish();
    if (rc != FM_OK) return rc;
    for (int i = 0; i < n; ++i) {
        if (n_matches) n_matches[i] = res[(size_t)i * 8 + 0];
        if (n_rounds) n_rounds[i] = res[(size_t)i * 8 + 1];
        if (n_pairs) n_pairs[i] = res[(size_t)i * 8 + 2];
        if (status) status[i] = (int32_t)res[(size_t)i * 8 + 3];
    }
    return FM_OK;
}

extern "C" int fm_expand_fetch(fm_ctx* ctx, const fm_expand* ex, int64_t n, int32_t* index, double* positions, double* ratio)
{
    if (!ctx || !ex) return fail(ctx, FM_EINVAL, "fm_expand_fetch: NULL argument");
    const int32_t slot = 0;
    const fm_expand* one = ex;
    int32_t* ip = index; double* pp = positions; double* rp = ratio;
    return fm_expand_fetch_many(ctx, 1, &one, &slot, &n, index ? &ip : nullptr, positions ? &pp : nullptr, ratio ? &rp : nullptr);
}

// The results of several runs of one fm_expand_run in ONE pass: every copy is enqueued, then a single
// synchronisation (a synchronisation per run: 64 runs = 2 ms of a 32 ms call).
extern "C" int fm_expand_fetch_many(fm_ctx* ctx, int32_t n_ex, const fm_expand* const* ex, const int32_t* slot, const int64_t* n,
                                    int32_t* const* index, double* const* positions, double* const* ratio)
{
    if (!ctx) return fail(nullptr, FM_EINVAL, "fm_expand_fetch_many: ctx is NULL");
    if (n_ex < 0) return fail(ctx, FM_EINVAL, "fm_expand_fetch_many: n_ex < 0");
    if (n_ex == 0) return FM_OK;
    if (!ex || !n) return fail(ctx, FM_EINVAL, "fm_expand_fetch_many: NULL argument");
    size_t total = 0;
    std::vector<const ExpandRun*> run((size_t)n_ex);
    for (int i = 0; i < n_ex; ++i) {
        if (!ex[i]) return fail(ctx, FM_EINVAL, "fm_expand_fetch_many: NULL pair");
        const int s = slot ? slot[i] : expand_slot_of((fm_expand* const*)ex, i);
        if (s < 0 || (size_t)s >= ex[i]->runs.size()) return fail(ctx, FM_EINVAL, "fm_expand_fetch_many: the pair has no such run slot");
        run[(size_t)i] = &ex[i]->runs[(size_t)s];
        if (n[i] < 0 || n[i] > run[(size_t)i]->match_cap) return fail(ctx, FM_EINVAL, "fm_expand_fetch_many: n out of range");
        total += (size_t)n[i] * 44 + 192;
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    struct StageGuard {      // drop the staged copies on every exit path (their targets die with the call)
        fm_ctx* c;
        ~StageGuard() { c->staged.clear(); c->h_stage_used = 0; }
    } guard{ctx};
    ctx->staged.clear();
    ctx->h_stage_used = 0;
    if (total > ctx->h_stage_bytes) {          // one staging area for all of it (d2h grows it only while nothing is staged)
        if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
        ctx->h_stage = nullptr;
        ctx->h_stage_bytes = 0;
        const size_t want = total + total / 4 + (1 << 20);
        if (hipHostMalloc((void**)&ctx->h_stage, want, hipHostMallocDefault) == hipSuccess) ctx->h_stage_bytes = want;
        else { (void)hipGetLastError(); ctx->h_stage = nullptr; }
    }
    for (int i = 0; i < n_ex; ++i) {
        if (n[i] == 0) continue;
        if (index && index[i]) HIP_TRY(ctx, d2h(ctx, index[i], run[(size_t)i]->m_index, (size_t)n[i] * 4));
        if (positions && positions[i]) HIP_TRY(ctx, d2h(ctx, positions[i], run[(size_t)i]->m_pos, (size_t)n[i] * 32));
        if (ratio && ratio[i]) HIP_TRY(ctx, d2h(ctx, ratio[i], run[(size_t)i]->m_ratio, (size_t)n[i] * 8));
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (const auto& c : ctx->staged) memcpy(c.dst, ctx->h_stage + c.off, c.bytes);
    return FM_OK;
}
