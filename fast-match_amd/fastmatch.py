"""Fast-Match on the MI355X: ``match(query_cache, target_img, options)(tau)``.

Mirrors ``fastmatch.pyx`` of the reference (same function names, option keys, return
shapes and log schema):

* ``match``          reference ``fastmatch.pyx:32-53``  -- options, Grid_Cache, seeding,
                     returns the closure ``get_matches(tau)``
* ``match_thumbs``   reference ``fastmatch.pyx:107-141`` -- thumbnail seeding
* ``do_iter``        reference ``fastmatch.pyx:56-89``  -- depth-first cell expansion
* ``get_neighbors``  reference ``fastmatch.pyx:92-103``
* ``match_position`` reference ``fastmatch.pyx:145-169`` -- one expansion round
* ``log_round``      reference ``fastmatch.pyx:172-180``

What runs where: the whole expansion loop runs on the device (K7, ``fm_expand_*``: one persistent
workgroup per run replays the reference's exact depth-first order -- the match set depends on it,
SURVEY.md fact 9): for a ``cache.Feature_Image`` target (every feature known up front) in one
launch per call, any number of (pair, threshold) runs side by side; for the reference's PIXEL
target, whose cells are computed when the loop first reaches them (cache.pyx:102-106, 124-138), the
kernel parks at a missing cell and the host computes it and resumes (``lazy_device_loop``).  A per-round
``log`` (the README's flow) is written by the kernel too, and float32 descriptors on a pixel target stay on the
device as well (r05).  Only when a capacity of the device loop is exceeded does the host replay the loop (``do_iter``): each round's arithmetic --
OpenCV's ``BFMatcher(NORM_L2, crossCheck=True).knnMatch`` plus the float64 ratio -- is then one launch
of the HIP round kernel on banks that stay resident (the query bank with its self distances, one bank
per computed grid cell); the radius subset's row indices go up, (train index, distance, ratio) come back.

``target_img`` is either the reference's ``uint8[H, W, 3]`` array (SIFT through OpenCV on
the host, needs ``cv2``) or a ``cache.Feature_Image`` carrying pre-extracted features.
Extra option keys (additions): ``"context"``/``"device"`` select the GPU,
``"stats"`` (a dict) receives round/pair counters; ``"log"`` (the reference's own key) is filled from the device loop's records, ``"feature_function"`` replaces
``matchutil.get_features`` (cv2 SIFT) for the thumbnail and the lazily computed grid cells
(e.g. ``standin.standin_features`` where cv2 is absent).
"""
from collections import deque

import os
import numpy as np

from . import matchutil
from .cache import Grid_Cache, Metric_Cache, Feature_Image, keypoint_positions   # noqa: F401
from .imaging import get_thumbnail, get_size


def match(query_cache, target_img, options={}):
    thumb_x, thumb_y = options.get("thumb_size", (400, 400))
    grid_x, grid_y = options.get("grid_size", (50, 50))
    thumb_strategy = options.get("thumb_strategy", lambda n: n)
    log = options.get("log", None)
    grid_margin = options.get("grid_margin", 25)
    radius = options.get("radius", 100)
    stats = options.get("stats", None)
    context = matchutil._context(options)

    if isinstance(target_img, Feature_Image):
        cell_features = target_img
    else:
        cell_features = options.get("feature_function", matchutil.get_features)
    target_cache = Grid_Cache(target_img, (grid_x, grid_y), cell_features, margin=grid_margin)
    thumb_positions, thumb_ratios = match_thumbs(target_img, query_cache, thumb_x=thumb_x, thumb_y=thumb_y,
                                                 context=context, feature_function=cell_features)

    # The whole expansion loop runs on the device when every target feature is known up
    # front (a Feature_Image) and no per-round log is wanted; otherwise (lazy SIFT per cell,
    # logging) the host replays the loop and only each round's arithmetic runs on the device.
    # r05: a per-round ``log`` no longer sends the run to the host loop -- the kernel records the rounds
    # (fm_expand_set_log) and ``_append_device_log`` rebuilds the reference's dicts from them.
    use_device_loop = isinstance(target_img, Feature_Image) and options.get("device_loop", True)
    # r04: a pixel target (cells computed on demand, the reference's own mode: cache.pyx:102-106, 124-138) runs the loop on
    # the device too -- the kernel parks when it reaches a cell that has not been computed, the host computes it (SIFT on the
    # crop), adds it to the growing target bank and resumes (lazy_device_loop below)
    use_lazy_loop = not isinstance(target_img, Feature_Image) and options.get("device_loop", True)
    state = {"expander": None, "lazy": None}

    def seeds_for(tau):
        return thumb_positions[thumb_ratios < thumb_strategy(tau)]

    def expander():
        """The pair's device-resident expansion state (built on first use), or None."""
        if not use_device_loop:
            return None
        if state["expander"] is None:
            state["expander"] = make_expander(query_cache, target_cache, radius, context)
            if state["expander"] and log is not None:
                state["expander"].set_log(True, first_capacity=options.get("log_first_capacity", 0))
        return state["expander"] or None

    def host_loop(tau):
        return do_iter(iter(seeds_for(tau)), query_cache, target_cache, tau=tau, thumb_tau=thumb_strategy(tau),
                       radius=radius, log=log, context=context, stats=stats)

    def lazy_device_loop(tau):
        """The device-resident loop on a target whose cells are computed on demand; None = not possible (float
        descriptors, a capacity of the device loop, the target bank's room used up): the host loop takes the run."""
        if not use_lazy_loop or state["lazy"] is False:
            return None
        if state["lazy"] is None:
            state["lazy"] = make_lazy_expander(query_cache, target_cache, radius, context, options.get("lazy_capacity"),
                                               float_route=state.get("lazy_float", False))
            if state["lazy"] is False:
                return None
            if log is not None:
                state["lazy"][0].set_log(True, first_capacity=options.get("log_first_capacity", 0))
        ex, t_bank = state["lazy"]
        seeds, resume, added = seeds_for(tau), False, 0
        from . import _ffi
        grid_before = _grid_state(target_cache) if log is not None else None
        while True:
            n_matches, n_rounds, n_pairs, status, need = ex.run_lazy(seeds, tau, resume)
            if status == _ffi.FM_EXPAND_NEED_CELL:
                col, row = divmod(need, target_cache.rows)
                value = target_cache.get_cell(col, row)        # the caching function runs here (SIFT on the crop)
                kp, ds = value if isinstance(value, tuple) else (None, None)
                try:
                    if ds is None or len(ds) == 0:
                        ex.set_cell(need, 0, np.zeros((0, 2)))
                    else:
                        ds = np.asarray(ds)
                        if t_bank.kind == _ffi.FM_BANK_F32:
                            ds = ds.astype(np.float32, copy=False)              # the float32 route takes any finite values
                        elif ds.dtype != np.uint8:
                            u8 = ds.astype(np.uint8)
                            if not np.array_equal(u8.astype(ds.dtype), ds):
                                # not integer valued, the query bank is: the pair moves to the float32 route (r06: it used
                                # to leave the device loop here).  A fresh lazy pair on the query bank's float32 twin, the
                                # run from its start; the cells computed so far are in the Grid_Cache, so the function is
                                # not called again for them -- and the log is rebuilt from the run that finishes.
                                if state.get("lazy_float", False):
                                    state["lazy"] = False
                                    return None
                                state["lazy_float"] = True
                                old_ex = state["lazy"][0]
                                state["lazy"] = make_lazy_expander(query_cache, target_cache, radius, context,
                                                                   options.get("lazy_capacity"), float_route=True)
                                old_ex.close()
                                if state["lazy"] is False:
                                    return None
                                if log is not None:
                                    state["lazy"][0].set_log(True, first_capacity=options.get("log_first_capacity", 0))
                                ex, t_bank = state["lazy"]
                                resume, added = False, 0
                                continue
                            ds = u8
                        off = np.array([row * target_cache.cell_width - target_cache.margin,
                                        col * target_cache.cell_height - target_cache.margin], dtype=np.float64)
                        ex.set_cell(need, t_bank.append(ds), keypoint_positions(kp) + off)
                except _ffi.FastMatchHipError:                 # the bank's capacity is used up
                    state["lazy"] = False
                    return None
                resume, added = True, added + 1
                continue
            if stats is not None:
                stats["lazy_cells"] = stats.get("lazy_cells", 0) + added
            if status != 0:
                if stats is not None:
                    stats["device_fallbacks"] = stats.get("device_fallbacks", 0) + 1
                return None
            if stats is not None:
                stats["device_loops"] = stats.get("device_loops", 0) + 1
                stats["rounds"] = stats.get("rounds", 0) + n_rounds
                stats["pairs"] = stats.get("pairs", 0) + n_pairs
            if log is not None:
                _append_device_log(log, ex, 0, target_cache, radius, grid_before, mark_computed=False)
            index, pos, ratio = ex.fetch(n_matches)
            if options.get("return_arrays", False):
                return index, pos, ratio
            return [(int(a), {"positions": p, "ratio": float(r)}) for a, p, r in zip(index, pos, ratio)]

    # A function where tau can be varied to get different results.  Addition: a LIST of thresholds
    # (the reference's driver asks one pair for many, turntable.py:59-60) returns the list of their
    # results from ONE launch of the device loop -- the runs are independent, one workgroup each.
    def get_matches(tau):
        if np.ndim(tau) > 0:
            taus = [float(t) for t in tau]
            res = [None] * len(taus)
            ex = expander()
            if ex is not None and taus:
                res = run_device_loops(context, [ex] * len(taus), [seeds_for(t) for t in taus], taus, stats=stats,
                                       as_arrays=options.get("return_arrays", False),
                                       logs=None if log is None else [(log, target_cache, radius)] * len(taus))
            elif use_lazy_loop:
                res = [lazy_device_loop(t) for t in taus]      # (one after the other: they share the cells computed so far)
            return [r if r is not None else host_loop(t) for r, t in zip(res, taus)]
        if use_lazy_loop:
            res = lazy_device_loop(tau)
            if res is not None:
                return res
        ex = expander()
        if ex is not None:
            res = run_device_loops(context, [ex], [seeds_for(tau)], [tau], stats=stats,
                                   as_arrays=options.get("return_arrays", False),
                                   logs=None if log is None else [(log, target_cache, radius)])[0]
            if res is not None:
                return res
        return host_loop(tau)

    # (for drivers that put the runs of SEVERAL pairs into one launch: evaluate.evaluate)
    get_matches.expander = expander
    get_matches.seeds_for = seeds_for
    get_matches.host_loop = host_loop
    get_matches.context = context
    return get_matches


def make_expander(query_cache, target_grid, radius, context, match_cap=0, stack_cap=0, plan=None):
    """Device-resident expansion state for (query_cache, target_grid), or False when the
    pair cannot use the device loop (oversize geometry, float32 banks the fp16 filter cannot
    take).  Integer-valued descriptors run the int8 round, others (RootSIFT-style float32)
    the float32 round -- both banks then take the float32 route."""
    from . import _ffi
    q_bank = query_cache.bank(context)
    if hasattr(target_grid.fun, "pack_plan") and target_grid.fun is target_grid.data:
        # pre-extracted target: each descriptor crosses PCIe once, the cells' copies are made by the upload kernel
        # (``plan``: the same triple computed ahead of time -- match_many plans the pairs of a dataset on several host threads)
        src_row, t_pos, cell_off = plan if plan is not None else target_grid.fun.pack_plan(target_grid)
        t_bank = context.bank_gather(target_grid.fun.descriptors, src_row, float_route=(q_bank.kind == _ffi.FM_BANK_F32))
    else:
        descs, t_pos, cell_off = target_grid.pack_cells()
        t_bank = context.bank(descs, float_route=(q_bank.kind == _ffi.FM_BANK_F32))
    if t_bank.kind != q_bank.kind:                 # integer-valued query bank, target not: pair on the float route
        q_bank = context.bank(query_cache.original["descriptors"], float_route=True)
        q_bank.set_selfdist(query_cache.original["distances"])
    if t_bank.kind != q_bank.kind or t_bank.dim != q_bank.dim:
        return False
    grid = {"width": target_grid.width, "height": target_grid.height, "cell_w": target_grid.cell_width,
            "cell_h": target_grid.cell_height, "rows": target_grid.rows, "cols": target_grid.cols,
            "margin": target_grid.margin}
    try:
        return _ffi.Expander(context, q_bank, query_cache.original["positions"],
                             query_cache.original["position_tree"], t_bank, cell_off, t_pos, grid, radius,
                             match_cap=match_cap, stack_cap=stack_cap)
    except _ffi.FastMatchHipError:
        return False


def _launch_plan(context, expanders, mem_fraction=0.5):
    """Cut a list of runs into launches that fit the device's memory.  fm_expand_run gives every run of a launch a run
    state of its own (the k-th run naming an expander uses its slot k; ~210 MB each for a 300k-keypoint pair) and keeps
    it: a dataset of pairs x 15 thresholds in ONE launch can ask for more than the device has.  A launch is closed when
    the NEW states it would create exceed ``mem_fraction`` of the free memory; the next launch re-uses slots 0..k.
    Returns a list of index lists (order preserved)."""
    try:
        free, _ = context.mem_info()
    except Exception:
        return [list(range(len(expanders)))]
    budget = mem_fraction * free
    info = {}
    launches, cur, used, new_bytes = [], [], {}, 0.0
    for i, ex in enumerate(expanders):
        if id(ex) not in info:
            info[id(ex)] = ex.info()
        state_bytes, n_slots = info[id(ex)]
        k = used.get(id(ex), 0)
        need = state_bytes if k >= n_slots else 0
        if cur and need and new_bytes + need > budget:
            launches.append(cur)
            for e in {id(expanders[j]): expanders[j] for j in cur}.values():    # what that launch created exists now
                info[id(e)] = (info[id(e)][0], max(info[id(e)][1], sum(1 for j in cur if expanders[j] is e)))
            cur, used, new_bytes = [], {}, 0.0
            k = 0
            need = state_bytes if k >= info[id(ex)][1] else 0
        cur.append(i)
        used[id(ex)] = k + 1
        new_bytes += need
    if cur:
        launches.append(cur)
    return launches


def _expand_launch(context, expanders, seeds, taus):
    """One fm_expand_run launch + fetch; a launch the device has no memory for is halved (down to single runs, whose
    failure is reported as status -1: the caller's host loop takes them)."""
    from . import _ffi
    try:
        results = context.expand_run(expanders, seeds, taus)
    except _ffi.FastMatchHipError as e:
        if getattr(e, "code", None) != -3:                 # FM_ENOMEM
            raise
        for ex in set(expanders):                          # give back what the earlier launches hold beyond slot 0
            ex.trim(1)
        if len(expanders) == 1:
            return [(0, 0, 0, -1)], {}
        h = len(expanders) // 2
        r1, f1 = _expand_launch(context, expanders[:h], seeds[:h], taus[:h])
        r2, f2 = _expand_launch(context, expanders[h:], seeds[h:], taus[h:])
        f1.update({k + h: v for k, v in f2.items()})
        return r1 + r2, f1
    slots = context.expand_slots(expanders)
    ok = [i for i, r in enumerate(results) if r[3] == 0]
    fetched = dict(zip(ok, context.expand_fetch_many([expanders[i] for i in ok], [results[i][0] for i in ok],
                                                     slots=[slots[i] for i in ok])))
    return results, fetched


def make_lazy_expander(query_cache, target_grid, radius, context, capacity=None, float_route=False):
    """(expander, growing target bank) for a target whose cells are computed on demand, or False (oversize geometry;
    r05: a float32 query bank gets a growing float32-route target bank).  ``capacity``: rows the target bank has room for; default 6 x the query's keypoints (a cell's
    crop includes its margins: a keypoint lands in up to four cells) + 32 per cell (cells start at multiples of 32 rows).
    ``float_route`` (r06): an integer-valued query bank is paired through its float32-route twin -- the feature function
    of the target returned descriptors that are not integer valued (make_expander does the same for pre-extracted targets)."""
    from . import _ffi
    q_bank = query_cache.bank(context)
    if float_route and q_bank.kind == _ffi.FM_BANK_I8:
        q_bank = context.bank(query_cache.original["descriptors"], float_route=True)
        q_bank.set_selfdist(query_cache.original["distances"])
    ncells = target_grid.rows * target_grid.cols
    if capacity is None:
        capacity = 6 * max(q_bank.n, 4096) + 32 * ncells + 4096
    grid = {"width": target_grid.width, "height": target_grid.height, "cell_w": target_grid.cell_width,
            "cell_h": target_grid.cell_height, "rows": target_grid.rows, "cols": target_grid.cols,
            "margin": target_grid.margin}
    try:
        if q_bank.kind == _ffi.FM_BANK_I8:
            t_bank = context.bank_with_capacity(np.zeros((0, q_bank.dim), dtype=np.uint8), int(capacity))
        else:            # r05: descriptors that are not integer valued (RootSIFT-style): a growing float32-route bank
            t_bank = context.bank_f32_with_capacity(q_bank.dim, int(capacity), q_bank)
        ex = _ffi.Expander(context, q_bank, query_cache.original["positions"], query_cache.original["position_tree"],
                           t_bank, None, None, grid, radius, lazy=True)
    except _ffi.FastMatchHipError:
        return False
    return ex, t_bank


def _grid_state(grid):
    """What Grid_Cache.last depends on (cache.pyx:102-106): the cells computed so far and the current value."""
    return {"known": set((col, row) for col, rows in grid.grid.items() for row in rows), "last": grid.last}


def _append_device_log(log, ex, slot, grid, radius, before, mark_computed):
    """The reference's per-round records (log_round, fastmatch.pyx:172-180) from what the device loop wrote
    (fm_expand_set_log): positions of the accepted matches from the banks' host-side positions, and
    ``target_grid`` = Grid_Cache.last as the host loop would have seen it -- the crop of the most recently COMPUTED
    cell, i.e. it changes at a round whose cell appears for the first time (``before``: the grid's state when the run
    began; the runs of one launch are replayed in order).  ``mark_computed``: a pre-extracted target's cells are
    computed here, in that order, so that the Grid_Cache ends in the state the host loop leaves it in."""
    q_pos, t_pos, cell, n_acc, q_row, t_row, ratio = ex.fetch_log(slot)
    tp = ex.target_positions()
    known, last = before["known"], before["last"]
    at = 0
    for i in range(len(cell)):
        col, row = divmod(int(cell[i]), grid.rows)
        if (col, row) not in known:
            known.add((col, row))
            last = grid.cell_bounds(col, row)
            if mark_computed:
                grid.get_cell(col, row)
        k = int(n_acc[i])
        if k < 0:                    # -1 a cell without features, -2 no cross-checked pair: match_position's arrays of
            matches, ratios = np.array([]), np.array([])     # shape (0,) (fastmatch.pyx:155-156, 162-167)
        else:
            matches = np.stack([ex.q_pos[q_row[at:at + k]], tp[t_row[at:at + k]]], axis=1) if k else np.zeros((0, 2, 2))
            ratios = ratio[at:at + k].copy()
            at += k
        log.append({"query_pos": q_pos[i].copy(), "target_pos": t_pos[i].copy(), "target_grid": last, "matches": matches,
                    "radius": radius, "ratios": ratios, "margin": grid.margin})
    before["last"] = last


def run_device_loops(context, expanders, seeds, taus, stats=None, as_arrays=False, logs=None):
    """The device-resident loop for several independent runs: (expander, seeds, tau) triples, an expander may appear
    several times (several thresholds of one pair).  All runs go into ONE launch when the run states fit the device's
    memory, else into as few launches as do (``_launch_plan``).  Returns, per run, the match list in do_iter's format
    (or (index, positions, ratio) arrays), or None where the device gave up (caller falls back to the host loop)."""
    out = [None] * len(expanders)
    plan = _launch_plan(context, expanders)
    if stats is not None and len(plan) > 1:
        stats["device_launches"] = stats.get("device_launches", 0) + len(plan)
    grid_states = {}                 # per Grid_Cache: its state as the runs of this call are replayed in order
    for idx in plan:
        results, fetched = _expand_launch(context, [expanders[i] for i in idx], [seeds[i] for i in idx], [taus[i] for i in idx])
        slots = context.expand_slots([expanders[i] for i in idx])
        for j, (i, (n_matches, n_rounds, n_pairs, status)) in enumerate(zip(idx, results)):
            if status != 0:
                if stats is not None:
                    stats["device_fallbacks"] = stats.get("device_fallbacks", 0) + 1
                continue
            if logs is not None and logs[i] is not None:
                lg, grid, radius = logs[i]
                st = grid_states.setdefault(id(grid), _grid_state(grid))
                _append_device_log(lg, expanders[i], slots[j], grid, radius, st, mark_computed=True)
            if stats is not None:
                stats["device_loops"] = stats.get("device_loops", 0) + 1
                stats["rounds"] = stats.get("rounds", 0) + n_rounds
                stats["pairs"] = stats.get("pairs", 0) + n_pairs
            index, pos, ratio = fetched[j]
            if as_arrays:
                out[i] = (index, pos, ratio)
            else:
                out[i] = [(int(a), {"positions": p, "ratio": float(r)}) for a, p, r in zip(index, pos, ratio)]
    return out


def match_many(pairs, tau, options={}):
    """Match several independent (query_cache, Feature_Image) pairs at one threshold with a
    single launch of the device-resident loop (one workgroup per pair).  Addition: the
    reference maps its matcher over pairs one after the other (turntable.py:59)."""
    context = matchutil._context(options)
    thumb_strategy = options.get("thumb_strategy", lambda n: n)
    grid_x, grid_y = options.get("grid_size", (50, 50))
    thumb_x, thumb_y = options.get("thumb_size", (400, 400))
    margin, radius = options.get("grid_margin", 25), options.get("radius", 100)
    prepared = options.get("prepared")
    if prepared is None:
        prepared = []
        grids = [Grid_Cache(target, (grid_x, grid_y), target, margin=margin) for _, target in pairs]
        # The cell plans of the pairs (which keypoint lands in which cells: host code of the library, 0.8 ms per 12.5k keypoints,
        # twice the device loop's share of such a pair) on a few host threads: the call leaves the interpreter lock.
        plans = [None] * len(pairs)
        packable = [i for i, g in enumerate(grids) if hasattr(g.fun, "pack_plan") and g.fun is g.data]
        if len(packable) >= 4:
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
                for i, pl in zip(packable, pool.map(lambda i: grids[i].fun.pack_plan(grids[i]), packable)):
                    plans[i] = pl
        for (query_cache, target), grid, plan in zip(pairs, grids, plans):
            pos, ratios = match_thumbs(target, query_cache, thumb_x=thumb_x, thumb_y=thumb_y, context=context)
            prepared.append({"query": query_cache, "grid": grid, "seeds": pos, "ratios": ratios,
                             "expander": make_expander(query_cache, grid, radius, context, plan=plan)})
        if "prepared_out" in options:
            options["prepared_out"].extend(prepared)
    thumb_tau = thumb_strategy(tau)
    seeds = [p["seeds"][p["ratios"] < thumb_tau] for p in prepared]
    on_dev = [i for i, p in enumerate(prepared) if p["expander"] not in (None, False)]
    results = [None] * len(prepared)
    stats = options.get("stats")
    if on_dev:
        got = run_device_loops(context, [prepared[i]["expander"] for i in on_dev], [seeds[i] for i in on_dev],
                               [tau] * len(on_dev), stats=stats, as_arrays=options.get("return_arrays", False))
        for i, g in zip(on_dev, got):
            results[i] = g
    for i, p in enumerate(prepared):
        if results[i] is None:
            results[i] = do_iter(iter(seeds[i]), p["query"], p["grid"], tau=tau, radius=radius, context=context,
                                 stats=stats)
    return results


def do_iter(positions, cache, target_grid, tau, thumb_tau=None, radius=100, log=None, context=None,
            stats=None):
    """Depth-first expansion.  ``positions`` yields [2,2] arrays (query_pos, target_pos);
    neighbours found in a round are visited before the remaining seeds, in the order the
    accepted matches produced them (reference fastmatch.pyx:75-77)."""
    context = context or matchutil._context({})
    pending = deque()
    seeds = iter(positions)
    matches = []
    has_matched = set()
    found_matches = {}
    n_rounds = n_pairs = 0
    while True:
        if pending:
            query_pos, target_pos = pending.popleft()
        else:
            try:
                query_pos, target_pos = next(seeds)
            except StopIteration:
                break
        col, row = target_grid.block(target_pos[0], target_pos[1])
        query_col, query_row = target_grid.block(query_pos[0], query_pos[1])
        key = (col, row, query_col, query_row)
        if key in has_matched:
            continue
        has_matched.add(key)
        result_pos, ratios, query_idx, pairs = _match_position((query_pos, target_pos), cache, target_grid,
                                                               radius, context)
        n_rounds += 1
        n_pairs += pairs
        accepted = ratios < tau
        acc_pos = result_pos[accepted]
        # For each match we keep, the neighbouring cell on its side is examined next
        neighbors = get_neighbors(target_pos, acc_pos, target_grid)
        if len(neighbors) > 0:
            # a neighbour whose key is already consumed would be skipped when popped (keys are
            # only ever added), so dropping it here cannot change the visiting order
            fresh = [nb for nb in neighbors
                     if target_grid.block(nb[1][0], nb[1][1]) + target_grid.block(nb[0][0], nb[0][1])
                     not in has_matched]
            pending.extendleft(reversed(fresh))
        if log is not None:
            log.append(log_round(query_pos, target_pos, result_pos, target_grid, ratios, tau, radius))
        for p, r, index in zip(acc_pos, ratios[accepted], query_idx[accepted]):
            p_tuple = [int(p[0, 0]), int(p[0, 1]), int(p[1, 0]), int(p[1, 1])]
            r = float(r)
            seen = found_matches.get(r)
            if seen is None:
                found_matches[r] = [p_tuple]
            elif p_tuple in seen:
                continue
            else:
                seen.append(p_tuple)
            matches.append((int(index), {"positions": p, "ratio": r}))
    if stats is not None:
        stats["rounds"] = stats.get("rounds", 0) + n_rounds
        stats["pairs"] = stats.get("pairs", 0) + n_pairs
    return matches


def get_neighbors(target_pos, result_pos, target_grid):
    """For every kept match, the centre of the 4-neighbour cell on the side of the current
    cell its target point lies closest to (reference fastmatch.pyx:92-103 +
    Grid_Cache.get_neighbor cache.pyx:72-92), vectorised over the matches; order kept."""
    if len(result_pos) == 0:
        return []
    g = target_grid
    col, row = g.block(target_pos[0], target_pos[1])
    cx, cy = g.center(col, row)
    pt = np.asarray(result_pos)[:, 1, :]
    x_diff = pt[:, 0].astype(np.int64) - int(cx)          # int() truncation (cache.pyx:82-83)
    y_diff = pt[:, 1].astype(np.int64) - int(cy)
    up = (y_diff < x_diff) & (y_diff < -x_diff)
    right = ~up & (x_diff > y_diff)
    down = ~up & ~right & (y_diff > -x_diff)
    left = ~up & ~right & ~down
    ncol = col + np.where(down, 1, 0) - np.where(up, 1, 0)
    nrow = row + np.where(right, 1, 0) - np.where(left, 1, 0)
    ok = (ncol >= 0) & (ncol < g.cols) & (nrow >= 0) & (nrow < g.rows)
    if not ok.any():
        return []
    # centre of the neighbour cell, clipped like Grid_Cache.center (cache.pyx:116-121)
    nx = np.minimum(((nrow + 0.5) * g.cell_width).astype(np.int64), g.width - 1)
    ny = np.minimum(((ncol + 0.5) * g.cell_height).astype(np.int64), g.height - 1)
    out = np.empty((int(ok.sum()), 2, 2), dtype=np.float64)
    out[:, 0, :] = np.asarray(result_pos)[ok, 0, :]
    out[:, 1, 0] = nx[ok]
    out[:, 1, 1] = ny[ok]
    return list(out)


def _thumb_features(img, thumb_x, thumb_y, feature_function=None):
    """(thumbnail positions [n,2], descriptors, (thumb_w, thumb_h)) of the target."""
    if isinstance(img, Feature_Image):
        if img.thumb is None:
            raise ValueError("Feature_Image has no thumbnail features (thumb_positions/thumb_descriptors)")
        return img.thumb["positions"], img.thumb["descriptors"], img.thumb["size"]
    target = get_thumbnail(img, (thumb_x, thumb_y))
    t_keypoints, t_descriptors = (feature_function or matchutil.get_features)(target)
    return keypoint_positions(t_keypoints), t_descriptors, (target.shape[1], target.shape[0])


def match_thumbs(img, query_cache, thumb_x=400, thumb_y=400, context=None, feature_function=None):
    """Seeding: cross-checked 1-NN between the thumbnail banks, ratio against the query
    thumbnail's self distances, positions scaled to full resolution, sorted by ratio
    (stable sort; the reference's quicksort leaves equal ratios in unspecified order)."""
    context = context or matchutil._context({})
    t_orig_x, t_orig_y = get_size(img)
    t_thumb_pos, t_descriptors, t_size = _thumb_features(img, thumb_x, thumb_y, feature_function)
    q_thumb_pos = query_cache.thumb["positions"]
    if t_descriptors is None or len(t_descriptors) == 0 or len(q_thumb_pos) == 0:
        return np.zeros((0, 2, 2), dtype=np.float64), np.zeros(0, dtype=np.float64)

    from . import _ffi
    q_thumb = query_cache.thumb_bank(context)
    t_bank = context.bank(np.asarray(t_descriptors), float_route=(q_thumb.kind == _ffi.FM_BANK_F32))
    if t_bank.kind != q_thumb.kind:                # query thumbnail integer valued, target not: pair on the float route
        q_thumb = context.bank(query_cache.thumb["descriptors"], float_route=True)
        q_thumb.set_selfdist(query_cache.thumb["distances"])
    try:
        tidx, dist, ratio, _, _ = context.match_ratio(q_thumb, t_bank, np.inf)
    finally:
        t_bank.close()
        if q_thumb is not query_cache.thumb_bank(context):
            q_thumb.close()
    m = tidx >= 0                                 # non-empty inner lists, query order
    ratios = ratio[m]
    t_pos = np.asarray(t_thumb_pos, dtype=np.float64)[tidx[m]]
    q_pos = np.asarray(q_thumb_pos, dtype=np.float64)[m]

    t_ratio = np.array([t_orig_x / float(t_size[0]), t_orig_y / float(t_size[1])])
    q_ratio = np.array([query_cache.original["size"][0] / float(query_cache.thumb["size"][0]),
                        query_cache.original["size"][1] / float(query_cache.thumb["size"][1])])
    pos_scaled = np.stack([q_pos * q_ratio, t_pos * t_ratio], axis=1) if len(ratios) else \
        np.zeros((0, 2, 2), dtype=np.float64)
    indices = np.argsort(ratios, kind="stable")
    return pos_scaled[indices], ratios[indices]


def _match_position(pos, query_cache, target, radius, context):
    """One round; returns (positions [m,2,2], ratios [m], indices [m], pairs evaluated)."""
    # the reference declares these as C ints: truncation toward zero (fastmatch.pyx:147-150)
    query_x, query_y = int(pos[0][0]), int(pos[0][1])
    target_x, target_y = int(pos[1][0]), int(pos[1][1])
    empty = (np.array([]), np.array([]), np.array([]), 0)

    query_idx = query_cache.radius_indices(query_x, query_y, radius)
    target_kp, target_ds = target.get(target_x, target_y)
    if target_ds is None or len(target_ds) == 0:
        return empty
    from . import _ffi
    col, row = target.block(target_x, target_y)
    q_bank = query_cache.bank(context)
    float_route = q_bank.kind == _ffi.FM_BANK_F32
    t_bank = target.cell_bank(col, row, context, float_route=float_route)
    nq, nt = len(query_idx), len(target_ds)
    if nq == 0:                      # knnMatch of an empty query set: matches = [], numpy.array([]) (fastmatch.pyx:162-167)
        return empty
    offset_x, offset_y = target.offset(target_x, target_y)

    tidx = None
    if nq <= 4096 and t_bank.kind == q_bank.kind:
        # one launch of the round kernel on the resident banks (int8 round, or the float32 round
        # for descriptors that are not integer valued)
        try:
            tidx, dist, ratio = context.xcheck1_batched(q_bank, query_idx, [0, nq], t_bank, [0, nt])
            if len(tidx) and tidx[0] == -2:          # float32 round gave up (candidate list overflow)
                tidx = None
        except _ffi.FastMatchHipError:
            if not float_route:
                raise
            tidx = None                              # banks the fp16 filter cannot take: dense path
    if tidx is None:
        # oversize radius subset, or an integer-valued query bank against a cell that is not:
        # gather the subset into a bank of its own and use the dense path
        if not float_route and t_bank.kind != _ffi.FM_BANK_I8:
            t_bank = target.cell_bank(col, row, context, float_route=True)
            float_route = True
        sub = context.bank(query_cache.original["descriptors"][query_idx], float_route=float_route)
        try:
            sub.set_selfdist(query_cache.original["distances"][query_idx])
            tidx, dist, ratio, _, _ = context.match_ratio(sub, t_bank, np.inf)
        finally:
            sub.close()
    m = tidx >= 0
    if not m.any():                  # no cross-checked pair: the reference builds its arrays from an empty list -> shape (0,)
        return (np.array([]), np.array([]), np.array([]), nq * nt)
    target_pos = keypoint_positions(target_kp) + np.array([offset_x, offset_y], dtype=np.float64)
    positions = np.stack([query_cache.original["positions"][query_idx[m]], target_pos[tidx[m]]], axis=1)
    return positions, ratio[m], query_idx[m], nq * nt


def match_position(pos, query_cache, target, radius=100, context=None):
    context = context or matchutil._context({})
    positions, ratios, indices, _ = _match_position(pos, query_cache, target, radius, context)
    return positions, ratios, indices


def log_round(query_pos, target_pos, result_pos, target_grid, ratios, tau, radius):
    keep = ratios < tau
    return {
        "query_pos": query_pos,
        "target_pos": target_pos,
        "target_grid": target_grid.last,
        "matches": result_pos[keep],
        "radius": radius,
        "ratios": ratios[keep],
        "margin": target_grid.margin}
