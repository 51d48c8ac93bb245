"""ctypes binding of libfastmatch_hip.so (C-ABI: include/fastmatch_hip.h).

This is the only bridge between the Python surface (fastmatch / cache / matchutil) and
the HIP kernels.  There is no CPU fallback: if the library is missing, or no gfx950
device is present, every compute entry point raises ``FastMatchHipError``.
"""
import ctypes
import weakref
import importlib.util
import os
import sys
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libfastmatch_hip.so")

FM_BANK_I8 = 1
FM_BANK_F32 = 2


class FastMatchHipError(RuntimeError):
    """Raised for every failure of the HIP path (the analogue of cv2.error).  ``code`` = the FM_E* status
    (-3 = FM_ENOMEM) when the library returned one."""
    code = None


class fm_stats(ctypes.Structure):
    _fields_ = [("kernel_ms", ctypes.c_double), ("total_ms", ctypes.c_double),
                ("kernel_launches", ctypes.c_int64), ("pairs", ctypes.c_int64),
                ("calls", ctypes.c_int64)]


class fm_stats_ex(ctypes.Structure):
    _fields_ = [("struct_bytes", ctypes.c_int64), ("kernel_ms", ctypes.c_double), ("total_ms", ctypes.c_double),
                ("kernel_launches", ctypes.c_int64), ("pairs", ctypes.c_int64), ("calls", ctypes.c_int64),
                ("bytes_moved", ctypes.c_int64)]


FM_ABI_VERSION = 8          # include/fastmatch_hip.h: the revision this binding was written against


class fm_expand_desc(ctypes.Structure):
    _fields_ = [("query", ctypes.c_void_p), ("query_pos", ctypes.c_void_p),
                ("index_bucket", ctypes.c_double), ("index_x0", ctypes.c_double), ("index_y0", ctypes.c_double),
                ("index_nbx", ctypes.c_int32), ("index_nby", ctypes.c_int32),
                ("index_order", ctypes.c_void_p), ("index_start", ctypes.c_void_p),
                ("target", ctypes.c_void_p), ("cell_off", ctypes.c_void_p), ("target_pos", ctypes.c_void_p),
                ("width", ctypes.c_int32), ("height", ctypes.c_int32),
                ("cell_w", ctypes.c_int32), ("cell_h", ctypes.c_int32),
                ("rows", ctypes.c_int32), ("cols", ctypes.c_int32),
                ("margin", ctypes.c_int32), ("radius", ctypes.c_int32),
                ("match_cap", ctypes.c_int64), ("stack_cap", ctypes.c_int64), ("metric", ctypes.c_int32),
                ("lazy", ctypes.c_int32)]


EXPAND_STATUS = {0: "ok", 1: "pending stack full", 2: "a radius subset the device could not take (see FM_EXPAND_SUBSET_FULL)",
                 3: "target position outside the image", 4: "result list full", 5: "hash table full",
                 6: "float32 round: candidate list full", 7: "lazy target: a cell is wanted", 9: "per-round log full"}
FM_EXPAND_NEED_CELL = 7     # include/fastmatch_hip.h

# name -> (restype, argtypes); every symbol include/fastmatch_hip.h declares
_P = ctypes.c_void_p
_I64 = ctypes.c_int64
_INT = ctypes.c_int
_I32 = ctypes.c_int32
SYMBOLS = {
    "fm_ctx_create": (_INT, [_INT, ctypes.POINTER(_P)]),
    "fm_ctx_destroy": (_INT, [_P]),
    "fm_ctx_set_option": (_INT, [_P, ctypes.c_char_p, _I64]),
    "fm_ctx_get_option": (_INT, [_P, ctypes.c_char_p, ctypes.POINTER(_I64)]),
    "fm_last_error": (ctypes.c_char_p, [_P]),
    "fm_sync": (_INT, [_P]),
    "fm_abi_version": (_INT, []),
    "fm_get_stats": (_INT, [_P, ctypes.POINTER(fm_stats)]),
    "fm_get_stats_ex": (_INT, [_P, ctypes.POINTER(fm_stats_ex), _I64]),
    "fm_reset_stats": (_INT, [_P]),
    "fm_device_name": (_INT, [_P, ctypes.c_char_p, _INT]),
    "fm_f32_filter_stats": (_INT, [_P, ctypes.POINTER(_I64), ctypes.POINTER(_I64)]),
    "fm_host_alloc": (_INT, [_P, _I64, ctypes.POINTER(_P)]),
    "fm_host_free": (_INT, [_P, _P]),
    "fm_bank_create_u8": (_INT, [_P, _P, _I64, _INT, ctypes.POINTER(_P)]),
    "fm_bank_create_f32": (_INT, [_P, _P, _I64, _INT, ctypes.POINTER(_P)]),
    "fm_bank_create_f32_route": (_INT, [_P, _P, _I64, _INT, ctypes.POINTER(_P)]),
    "fm_bank_destroy": (_INT, [_P, _P]),
    "fm_bank_info": (_INT, [_P, ctypes.POINTER(_I64), ctypes.POINTER(_INT), ctypes.POINTER(_INT)]),
    "fm_bank_set_selfdist": (_INT, [_P, _P, _P]),
    "fm_bank_refill_u8_async": (_INT, [_P, _P, _P, _I64]),
    "fm_bank_create_u8_cap": (_INT, [_P, _P, _I64, _INT, _I64, ctypes.POINTER(_P)]),
    "fm_grid_pack_cells": (_INT, [_P, _I64, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I64, _P, ctypes.POINTER(_I64), _P, _P]),
    "fm_bank_create_u8_gather": (_INT, [_P, _P, _I64, _INT, _P, _I64, ctypes.POINTER(_P)]),
    "fm_bank_create_f32_gather": (_INT, [_P, _P, _I64, _INT, _INT, _P, _I64, ctypes.POINTER(_P)]),
    "fm_bank_append_u8": (_INT, [_P, _P, _P, _I64, ctypes.POINTER(_I64)]),
    "fm_bank_create_f32_cap": (_INT, [_P, _INT, _I64, _P, ctypes.POINTER(_P)]),
    "fm_bank_append_f32": (_INT, [_P, _P, _P, _I64, ctypes.POINTER(_I64)]),
    "fm_expand_set_cell": (_INT, [_P, _P, ctypes.c_int32, _I64, _I64, _P]),
    "fm_expand_run_lazy": (_INT, [_P, _P, _P, _I64, ctypes.c_double, ctypes.c_int32, ctypes.POINTER(_I64), ctypes.POINTER(_I64),
                           ctypes.POINTER(_I64), ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32)]),
    "fm_upload_fence": (_INT, [_P]),
    "fm_self_dist_batch": (_INT, [_P, ctypes.c_int32, _P, _P]),
    "fm_knn2": (_INT, [_P, _P, _P, _P, _P]),
    "fm_knn": (_INT, [_P, _P, _P, ctypes.c_int32, _P, _P]),
    "fm_xcheck1_keys": (_INT, [_P, _P, _P, _I64, _P]),
    "fm_xcheck1_keys_dev": (_INT, [_P, _P, _P, _I64, _P]),
    "fm_knn2_ratio": (_INT, [_P, _P, _P, ctypes.c_double, _I64, _P, _P, _P, _P, ctypes.POINTER(_I64)]),
    "fm_self_dist": (_INT, [_P, _P, _P]),
    "fm_self_dist_plan": (_INT, [_I64, _I32, _P, _I64, ctypes.POINTER(_I32), ctypes.POINTER(_I32), ctypes.POINTER(_I32)]),
    "fm_xcheck1": (_INT, [_P, _P, _P, _P, _P]),
    "fm_ratio_filter": (_INT, [_P, _P, _P, _P, _I64, ctypes.c_double, _P, _P, ctypes.POINTER(_I64)]),
    "fm_match_ratio": (_INT, [_P, _P, _P, ctypes.c_double, _P, _P, _P, _P, ctypes.POINTER(_I64)]),
    "fm_match_accepted": (_INT, [_P, _P, _P, ctypes.c_double, _I64, _P, _P, _P, _P, ctypes.POINTER(_I64)]),
    "fm_match_accepted_async": (_INT, [_P, _P, _P, ctypes.c_double, _I64, _P, _P, _P, _P, _P]),
    "fm_mark": (_INT, [_P, ctypes.POINTER(_I64)]),
    "fm_wait": (_INT, [_P, _I64]),
    "fm_expand_fetch_many": (_INT, [_P, ctypes.c_int32, _P, _P, ctypes.POINTER(_I64), _P, _P, _P]),
    "fm_match_accepted_batch": (_INT, [_P, ctypes.c_int32, _P, _P, ctypes.c_double, _I64, _P, _P, _P, _P, _P]),
    "fm_match_accepted_dev_batch": (_INT, [_P, ctypes.c_int32, _P, _P, ctypes.c_double, _I64, _P, _P, _P, _P]),
    "fm_match_accepted_dev": (_INT, [_P, _P, _P, ctypes.c_double, _I64, _P, _P, ctypes.POINTER(_I64)]),
    "fm_match_accepted_dev_async": (_INT, [_P, _P, _P, ctypes.c_double, _I64, _P, _P, _P, _P]),
    "fm_xcheck1_batched": (_INT, [_P, _P, _P, _P, _P, _P, _I64, _P, _P, _P]),
    "fm_expand_create": (_INT, [_P, ctypes.POINTER(fm_expand_desc), ctypes.POINTER(_P)]),
    "fm_expand_destroy": (_INT, [_P, _P]),
    "fm_expand_run": (_INT, [_P, ctypes.c_int32, _P, _P, _P, _P, _P, _P, _P, _P]),
    "fm_expand_fetch": (_INT, [_P, _P, _I64, _P, _P, _P]),
    "fm_expand_info": (_INT, [_P, ctypes.POINTER(_I64), ctypes.POINTER(ctypes.c_int32)]),
    "fm_expand_trim": (_INT, [_P, _P, ctypes.c_int32]),
    "fm_expand_set_log": (_INT, [_P, _P, ctypes.c_int32]),
    "fm_expand_log_counts": (_INT, [_P, _P, ctypes.c_int32, ctypes.POINTER(_I64), ctypes.POINTER(_I64)]),
    "fm_expand_fetch_log": (_INT, [_P, _P, ctypes.c_int32, _I64, _I64, _P, _P, _P, _P]),
    "fm_mem_info": (_INT, [_P, ctypes.POINTER(_I64), ctypes.POINTER(_I64)]),
    "fm_comm_unique_id": (_INT, [_P]),
    "fm_comm_init": (_INT, [_P, _INT, _INT, _P]),
    "fm_comm_destroy": (_INT, [_P]),
    "fm_gather_matches": (_INT, [_P, _P, _P, _I64, _P, _P, _INT]),
    "fm_gather_matches_counted": (_INT, [_P, _P, _P, _I64, _P, _P, ctypes.POINTER(_I64)]),
}

_lib = None
_lib_lock = threading.Lock()


def _preload_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so.7 (and
    HSA runtime) under torch/lib; if this library pulled in /opt/rocm's copy first, a later
    ``import torch`` would mix the two sets and find no GPU.  So when torch is installed but
    not imported yet, its bundled runtime is loaded first and both share it (the order
    ``import torch`` -> this library already behaves that way).  FM_SYSTEM_HIP_RUNTIME=1 skips it."""
    if "torch" in sys.modules or os.environ.get("FM_SYSTEM_HIP_RUNTIME") == "1":
        return
    try:
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(path):
            ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
    except Exception:
        pass


def load_library():
    """dlopen libfastmatch_hip.so and declare every prototype.  Raises if missing."""
    global _lib
    with _lib_lock:
        if _lib is not None:
            return _lib
        _preload_torch_hip_runtime()
        if not os.path.exists(LIB_PATH):
            raise FastMatchHipError(
                "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % LIB_PATH)
        try:
            lib = ctypes.CDLL(LIB_PATH)
        except OSError as e:
            raise FastMatchHipError("cannot load %s: %s" % (LIB_PATH, e))
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)     # AttributeError here = ABI drift, fail loudly
            fn.restype = res
            fn.argtypes = args
        if lib.fm_abi_version() != FM_ABI_VERSION:
            raise FastMatchHipError("%s has ABI revision %d, this binding was written for %d: rebuild the library"
                                    % (LIB_PATH, lib.fm_abi_version(), FM_ABI_VERSION))
        _lib = lib
        return lib


def _ptr(a):
    return None if a is None else a.ctypes.data


def self_dist_plan(n_pad, stages=0):
    """fm_self_dist_plan (host code of the library): the workgroups of the triangular self-distance sweep of a bank
    padded to ``n_pad`` rows -> (table int32[n, 4] = (chunk, first stage, end stage, 0), n_diag, stages_used)."""
    lib = load_library()
    nwg, nd, su = _I32(), _I32(), _I32()
    rc = lib.fm_self_dist_plan(int(n_pad), int(stages), None, 0, ctypes.byref(nwg), ctypes.byref(nd), ctypes.byref(su))
    if rc == 0:
        table = np.empty((nwg.value, 4), dtype=np.int32)
        rc = lib.fm_self_dist_plan(int(n_pad), int(stages), _ptr(table), nwg.value, ctypes.byref(nwg), ctypes.byref(nd), ctypes.byref(su))
    if rc != 0:
        msg = lib.fm_last_error(None)
        e = FastMatchHipError("fm_self_dist_plan: %s" % (msg.decode() if msg else rc))
        e.code = rc
        raise e
    return table, nd.value, su.value


def grid_pack_cells(positions, width, height, cell_w, cell_h, rows, cols, margin):
    """fm_grid_pack_cells (host code of the library): (src_row int32[nt], target_pos f64[nt, 2], cell_off int64[rows * cols + 1])
    for a Grid_Cache over pre-extracted keypoints -- every cell's keypoints, cell after cell (cell id = col * rows + row)."""
    lib = load_library()
    pos = np.ascontiguousarray(positions, dtype=np.float64).reshape(-1, 2)
    n = pos.shape[0]
    cell_off = np.zeros(int(rows) * int(cols) + 1, dtype=np.int64)
    cap = 5 * n + 16
    while True:
        src_row, t_pos, nt = np.empty(cap, dtype=np.int32), np.empty((cap, 2), dtype=np.float64), _I64()
        rc = lib.fm_grid_pack_cells(_ptr(pos), n, int(width), int(height), int(cell_w), int(cell_h), int(rows), int(cols), int(margin),
                                    cap, _ptr(cell_off), ctypes.byref(nt), _ptr(src_row), _ptr(t_pos))
        if rc != 0:
            msg = lib.fm_last_error(None)
            e = FastMatchHipError("fm_grid_pack_cells: %s" % (msg.decode() if msg else rc))
            e.code = rc
            raise e
        if nt.value <= cap:
            return src_row[:nt.value], t_pos[:nt.value], cell_off
        cap = nt.value


def _stream_arg(stream):
    """consumer_stream of the C-ABI: None -> FM_NO_STREAM, an integer handle (0 = the null stream) as is."""
    return _P(-1 & 0xFFFFFFFFFFFFFFFF) if stream is None else _P(int(stream))


class _LockedLib(object):
    """The library with every call serialised on one lock.  An fm_ctx is not re-entrant (it
    owns shared workspaces, the staging list and the timing events) and ctypes releases the
    GIL during a call, so two Python threads sharing a Context must not overlap inside it."""

    def __init__(self, lib):
        self._lib = lib
        self._lock = threading.RLock()
        self._wrapped = {}

    def __getattr__(self, name):
        fn = self._wrapped.get(name)
        if fn is None:
            raw, lock = getattr(self._lib, name), self._lock

            def fn(*args):
                with lock:
                    return raw(*args)
            self._wrapped[name] = fn
        return fn


class Bank(object):
    """Device-resident descriptor bank (fm_bank)."""

    def __init__(self, ctx, handle, n, dim, kind):
        self.ctx, self.handle, self.n, self.dim, self.kind = ctx, handle, n, dim, kind
        self.has_selfdist = False
        ctx._children.add(self)           # (a context that is closed first destroys what was made through it)

    def set_selfdist(self, selfdist):
        sd = np.ascontiguousarray(selfdist, dtype=np.float64)
        if sd.shape != (self.n,):
            raise ValueError("selfdist must have shape (%d,)" % self.n)
        self.ctx._check(self.ctx.lib.fm_bank_set_selfdist(self.ctx.handle, self.handle, _ptr(sd)))
        self.has_selfdist = True

    def refill_async(self, rows):
        """A new image's uint8 descriptors into this bank without allocation or host synchronisation
        (``fm_bank_refill_u8_async``): ``rows`` = [n, dim] uint8 from ``Context.pinned_empty`` with n within the
        bank's first size.  Nothing enqueued earlier may still read the bank; call ``Context.upload_fence()``
        before the first use, and recompute the self distances (``Context.self_dist_batch``)."""
        a = np.asarray(rows)
        if a.dtype != np.uint8 or a.ndim != 2 or a.shape[1] != self.dim or not a.flags.c_contiguous:
            raise ValueError("rows must be a contiguous [n, %d] uint8 array" % self.dim)
        self.ctx._check(self.ctx.lib.fm_bank_refill_u8_async(self.ctx.handle, self.handle, _ptr(a), a.shape[0]))
        self.n = a.shape[0]
        self.has_selfdist = False
        self._refill_src = a          # (the copy is asynchronous: keep the source alive)

    def append(self, rows):
        """More uint8 rows into a bank made with room for them (``Context.bank_with_capacity``), placed at the next
        multiple of 32 rows; returns the index of the first one."""
        if self.kind == FM_BANK_F32:            # a growing float32-route bank (Context.bank_f32_with_capacity)
            a = np.ascontiguousarray(rows, dtype=np.float32)
            if a.ndim != 2 or a.shape[1] != self.dim:
                raise ValueError("rows must be [n, %d]" % self.dim)
            first = _I64(0)
            self.ctx._check(self.ctx.lib.fm_bank_append_f32(self.ctx.handle, self.handle, _ptr(a), a.shape[0], ctypes.byref(first)))
            if a.shape[0]:
                self.n = int(first.value) + a.shape[0]
            return int(first.value)
        a = np.ascontiguousarray(rows)
        if a.dtype != np.uint8 or a.ndim != 2 or a.shape[1] != self.dim:
            raise ValueError("rows must be [n, %d] uint8" % self.dim)
        first = _I64(0)
        self.ctx._check(self.ctx.lib.fm_bank_append_u8(self.ctx.handle, self.handle, _ptr(a), a.shape[0], ctypes.byref(first)))
        if a.shape[0]:
            self.n = int(first.value) + a.shape[0]
        return int(first.value)

    def close(self):
        if self.handle is not None and self.ctx.handle is not None:
            self.ctx.lib.fm_bank_destroy(self.ctx.handle, self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Expander(object):
    """Device-resident expansion state of one image pair (fm_expand)."""

    def __init__(self, ctx, q_bank, q_pos, index, t_bank, cell_off, t_pos, grid, radius,
                 match_cap=0, stack_cap=0, lazy=False):
        """``lazy``: the target's cells are added one by one (``set_cell``) as the loop asks for them (``run_lazy``);
        ``t_bank`` is then a bank from ``Context.bank_with_capacity`` and ``cell_off`` / ``t_pos`` are None."""
        self.ctx = ctx
        self.handle = None
        self.lazy = bool(lazy)
        # keep every host array alive until fm_expand_create has copied it
        q_pos = np.ascontiguousarray(q_pos, dtype=np.float64).reshape(-1, 2)
        order = np.ascontiguousarray(index.order, dtype=np.int32)
        start = np.ascontiguousarray(index.start, dtype=np.int32)
        cell_off = None if lazy else np.ascontiguousarray(cell_off, dtype=np.int64)
        t_pos = None if lazy else np.ascontiguousarray(t_pos, dtype=np.float64).reshape(-1, 2)
        d = fm_expand_desc()
        d.query, d.query_pos = q_bank.handle, _ptr(q_pos)
        d.index_bucket, d.index_x0, d.index_y0 = index.bucket, index.x0, index.y0
        d.index_nbx, d.index_nby = index.nbx, index.nby
        d.index_order, d.index_start = _ptr(order), _ptr(start)
        d.target, d.cell_off, d.target_pos = t_bank.handle, _ptr(cell_off), _ptr(t_pos)
        d.width, d.height = grid["width"], grid["height"]
        d.cell_w, d.cell_h = grid["cell_w"], grid["cell_h"]
        d.rows, d.cols, d.margin, d.radius = grid["rows"], grid["cols"], grid["margin"], int(radius)
        d.match_cap, d.stack_cap = int(match_cap), int(stack_cap)
        d.metric = int(getattr(index, "metric", 0))         # Position_Index.metric = FM_METRIC_*
        d.lazy = 1 if lazy else 0
        h = _P()
        ctx._check(ctx.lib.fm_expand_create(ctx.handle, ctypes.byref(d), ctypes.byref(h)))
        self.handle = h
        ctx._children.add(self)
        self._banks = (q_bank, t_bank)            # the banks must outlive the expander
        # host copies of the positions the log's records are rebuilt from (fetch_log)
        self.q_pos = q_pos
        self.t_pos = t_pos
        self._lazy_pos = {}                       # lazy targets: first row -> positions of the cell added there
        self.logging = False

    def set_log(self, enable=True, first_capacity=0):
        """Runs of this pair record the reference's per-round log on the device (fm_expand_set_log);
        ``first_capacity`` > 1: the log arrays start that small (they grow fourfold when a run fills them)."""
        if bool(enable) != self.logging or first_capacity > 1:
            self.ctx._check(self.ctx.lib.fm_expand_set_log(self.ctx.handle, self.handle,
                                                           int(first_capacity) if enable and first_capacity > 1 else (1 if enable else 0)))
            self.logging = bool(enable)

    def fetch_log(self, slot=0):
        """The log of the last run in ``slot``: (query_pos f64[n, 2], target_pos f64[n, 2], cell i64[n], n_accepted i64[n]
        (-1: the cell had no features, -2: no cross-checked pair), query_row i32[m], target_row i32[m], ratio f64[m]) -- rounds in order, the accepted
        matches of all rounds back to back."""
        nr, ne = _I64(0), _I64(0)
        self.ctx._check(self.ctx.lib.fm_expand_log_counts(self.ctx.handle, self.handle, int(slot), ctypes.byref(nr), ctypes.byref(ne)))
        rounds = np.empty((nr.value, 6), dtype=np.int64)
        q, t, ratio = np.empty(ne.value, dtype=np.int32), np.empty(ne.value, dtype=np.int32), np.empty(ne.value, dtype=np.float64)
        self.ctx._check(self.ctx.lib.fm_expand_fetch_log(self.ctx.handle, self.handle, int(slot), nr.value, ne.value,
                                                         _ptr(rounds), _ptr(q), _ptr(t), _ptr(ratio)))
        pos = np.ascontiguousarray(rounds[:, :4]).view(np.float64)
        return pos[:, 0:2], pos[:, 2:4], rounds[:, 4].copy(), rounds[:, 5].copy(), q, t, ratio

    def target_positions(self):
        """Full-image positions of the target bank's rows (lazy targets: of the cells added so far)."""
        if not self.lazy:
            return self.t_pos
        n = max([r + len(p) for r, p in self._lazy_pos.items()] + [0])
        out = np.zeros((n, 2), dtype=np.float64)
        for r, p in self._lazy_pos.items():
            out[r:r + len(p)] = p
        return out

    def set_cell(self, cell, first_row, positions):
        """Register a computed cell of a lazy target: rows [first_row, first_row + len(positions)) of the target bank."""
        pos = np.ascontiguousarray(positions, dtype=np.float64).reshape(-1, 2)
        self.ctx._check(self.ctx.lib.fm_expand_set_cell(self.ctx.handle, self.handle, int(cell), int(first_row), pos.shape[0],
                                                        _ptr(pos) if pos.shape[0] else None))
        if pos.shape[0]:
            self._lazy_pos[int(first_row)] = pos

    def run_lazy(self, seeds, tau, resume):
        """One launch of a lazy pair: (n_matches, n_rounds, n_pairs, status, need_cell); status 7 = compute ``need_cell``
        (col * rows + row), ``set_cell`` it and call again with ``resume=True``."""
        seeds = np.ascontiguousarray(seeds, dtype=np.float64).reshape(-1, 2, 2)
        nm, nr, npairs = _I64(0), _I64(0), _I64(0)
        st, need = ctypes.c_int32(0), ctypes.c_int32(-1)
        self.ctx._check(self.ctx.lib.fm_expand_run_lazy(self.ctx.handle, self.handle, _ptr(seeds) if seeds.shape[0] else None,
                                                        seeds.shape[0], float(tau), 1 if resume else 0, ctypes.byref(nm),
                                                        ctypes.byref(nr), ctypes.byref(npairs), ctypes.byref(st), ctypes.byref(need)))
        return int(nm.value), int(nr.value), int(npairs.value), int(st.value), int(need.value)

    def info(self):
        """(bytes of one run state, run states that exist): see fm_expand_info."""
        b, k = _I64(0), ctypes.c_int32(0)
        self.ctx._check(self.ctx.lib.fm_expand_info(self.handle, ctypes.byref(b), ctypes.byref(k)))
        return int(b.value), int(k.value)

    def trim(self, keep=1):
        """Free the run states from slot ``keep`` on (they are re-created on demand)."""
        self.ctx._check(self.ctx.lib.fm_expand_trim(self.ctx.handle, self.handle, int(keep)))

    def fetch(self, n):
        index = np.empty(n, dtype=np.int32)
        pos = np.empty((n, 2, 2), dtype=np.float64)
        ratio = np.empty(n, dtype=np.float64)
        self.ctx._check(self.ctx.lib.fm_expand_fetch(self.ctx.handle, self.handle, n, _ptr(index), _ptr(pos), _ptr(ratio)))
        return index, pos, ratio

    def close(self):
        if self.handle is not None and self.ctx.handle is not None:
            self.ctx.lib.fm_expand_destroy(self.ctx.handle, self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context(object):
    """One fm_ctx (= one device + one HIP stream)."""

    def __init__(self, device=0):
        self.lib = _LockedLib(load_library())
        self.handle = None
        h = _P()
        rc = self.lib.fm_ctx_create(int(device), ctypes.byref(h))
        if rc != 0:
            msg = self.lib.fm_last_error(None)
            raise FastMatchHipError("fm_ctx_create(%d) failed (%d): %s"
                                    % (device, rc, msg.decode() if msg else "?"))
        self.handle = h
        self.device = int(device)
        # banks and expanders made through this context (weak references): fm_bank_destroy / fm_expand_destroy need a live
        # context, so close() destroys the ones still open first -- r06's memory soak found 9.5 MB of device memory per
        # context left behind by `c.close()` with banks open (their own close() could no longer reach the library)
        self._children = weakref.WeakSet()

    def _check(self, rc):
        if rc != 0:
            msg = self.lib.fm_last_error(self.handle) if self.handle is not None else None
            e = FastMatchHipError("libfastmatch_hip error %d: %s" % (rc, msg.decode() if msg else "?"))
            e.code = rc
            raise e

    def close(self):
        if self.handle is not None:
            kids = list(getattr(self, "_children", ()))
            for k in kids:                        # expanders first: they borrow their banks
                if isinstance(k, Expander):
                    k.close()
            for k in kids:
                if not isinstance(k, Expander):
                    k.close()
            for p in getattr(self, "_pinned", []):
                self.lib.fm_host_free(self.handle, p)
            self._pinned = []
            self.lib.fm_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, name, value):
        """Per-context tuning / batch shape (fm_ctx_set_option): "batch_group", "batch_tail", "nsplit", "nb",
        "nw", "nbuf", "prio", "glds", "coop", "f32_filter", "f32_nw", "f32_nsplit", "f32_fused", "f32_lpc", "f32_bound_every",
        "async_time_every", "k1_order", "bound_every", "self_tri", "tri_stages", "refill_grid", "expand_big", "expand_huge", "expand_delegate", "expand_grow", "expand_prof".  Results never depend on them."""
        self._check(self.lib.fm_ctx_set_option(self.handle, name.encode(), int(value)))

    def get_option(self, name):
        v = _I64(0)
        self._check(self.lib.fm_ctx_get_option(self.handle, name.encode(), ctypes.byref(v)))
        return int(v.value)

    def pinned_empty(self, shape, dtype):
        """ndarray in page-locked host memory (lives as long as the context): pass it as an
        ``out=`` buffer so results arrive by direct DMA."""
        dt = np.dtype(dtype)
        shape = (shape,) if np.isscalar(shape) else tuple(shape)
        nbytes = int(np.prod(shape)) * dt.itemsize
        p = _P()
        self._check(self.lib.fm_host_alloc(self.handle, nbytes, ctypes.byref(p)))
        if not hasattr(self, "_pinned"):
            self._pinned = []
        self._pinned.append(p)
        buf = (ctypes.c_char * max(nbytes, 1)).from_address(p.value)
        return np.frombuffer(buf, dtype=dt, count=int(np.prod(shape))).reshape(shape)

    # -- banks -------------------------------------------------------------------------
    def bank(self, rows, float_route=False):
        """Upload an [n, dim] uint8 or float32 matrix.  Other dtypes are converted to
        float32 first (cv2 would reject them).  ``float_route``: keep the float32 route even
        if the values are integers (to pair with a bank that is not integer valued)."""
        a = np.asarray(rows)
        if a.ndim != 2:
            raise ValueError("descriptor bank must be 2-D [n, dim]")
        h = _P()
        if float_route:
            a = np.ascontiguousarray(a, dtype=np.float32)
            self._check(self.lib.fm_bank_create_f32_route(self.handle, _ptr(a), a.shape[0], a.shape[1], ctypes.byref(h)))
        elif a.dtype == np.uint8:
            a = np.ascontiguousarray(a)
            self._check(self.lib.fm_bank_create_u8(self.handle, _ptr(a), a.shape[0], a.shape[1], ctypes.byref(h)))
        else:
            a = np.ascontiguousarray(a, dtype=np.float32)
            self._check(self.lib.fm_bank_create_f32(self.handle, _ptr(a), a.shape[0], a.shape[1], ctypes.byref(h)))
        n, dim, kind = _I64(), _INT(), _INT()
        self._check(self.lib.fm_bank_info(h, ctypes.byref(n), ctypes.byref(dim), ctypes.byref(kind)))
        return Bank(self, h, n.value, dim.value, kind.value)

    def bank_gather(self, rows, src_row, float_route=False):
        """The bank ``rows[src_row]`` without building that matrix on the host: the n_src rows are uploaded once and
        gathered by the upload kernel (fm_bank_create_*_gather).  dtype handling as ``bank``."""
        a = np.asarray(rows)
        if a.ndim != 2:
            raise ValueError("descriptor bank must be 2-D [n, dim]")
        m = np.ascontiguousarray(src_row, dtype=np.int32).reshape(-1)
        h = _P()
        if float_route or a.dtype != np.uint8:
            a = np.ascontiguousarray(a, dtype=np.float32)
            self._check(self.lib.fm_bank_create_f32_gather(self.handle, _ptr(a), a.shape[0], a.shape[1], 1 if float_route else 0,
                                                           _ptr(m), m.shape[0], ctypes.byref(h)))
        else:
            a = np.ascontiguousarray(a)
            self._check(self.lib.fm_bank_create_u8_gather(self.handle, _ptr(a), a.shape[0], a.shape[1], _ptr(m), m.shape[0],
                                                          ctypes.byref(h)))
        n, dim, kind = _I64(), _INT(), _INT()
        self._check(self.lib.fm_bank_info(h, ctypes.byref(n), ctypes.byref(dim), ctypes.byref(kind)))
        return Bank(self, h, n.value, dim.value, kind.value)

    def bank_with_capacity(self, rows, capacity):
        """A uint8 bank with room for ``capacity`` rows (``Bank.append``): the growing target bank of a lazy pair."""
        a = np.ascontiguousarray(rows, dtype=np.uint8)
        if a.ndim != 2:
            raise ValueError("descriptor bank must be 2-D [n, dim]")
        h = _P()
        self._check(self.lib.fm_bank_create_u8_cap(self.handle, _ptr(a) if a.shape[0] else None, a.shape[0], a.shape[1],
                                                   int(max(capacity, a.shape[0])), ctypes.byref(h)))
        n, dim, kind = _I64(), _INT(), _INT()
        self._check(self.lib.fm_bank_info(h, ctypes.byref(n), ctypes.byref(dim), ctypes.byref(kind)))
        return Bank(self, h, n.value, dim.value, kind.value)

    def bank_f32_with_capacity(self, dim, capacity, scale_like):
        """An empty float32-route bank with room for ``capacity`` rows whose fp16 planes use ``scale_like``'s scale
        (``Bank.append`` takes float32 rows): the growing target bank of a lazy pair with non-integer descriptors."""
        h = _P()
        self._check(self.lib.fm_bank_create_f32_cap(self.handle, int(dim), int(capacity), scale_like.handle, ctypes.byref(h)))
        return Bank(self, h, 0, int(dim), FM_BANK_F32)

    # -- operators -----------------------------------------------------------------------
    def knn2(self, q, t):
        idx = np.empty((q.n, 2), dtype=np.int32)
        dist = np.empty((q.n, 2), dtype=np.float32)
        self._check(self.lib.fm_knn2(self.handle, q.handle, t.handle, _ptr(idx), _ptr(dist)))
        return idx, dist

    def knn(self, q, t, k):
        """``fm_knn``: k-NN lists for 1 <= k <= 8 (idx int32[nq, k], dist float32[nq, k]; -1 / inf where t has fewer rows)."""
        idx = np.empty((q.n, int(k)), dtype=np.int32)
        dist = np.empty((q.n, int(k)), dtype=np.float32)
        self._check(self.lib.fm_knn(self.handle, q.handle, t.handle, int(k), _ptr(idx), _ptr(dist)))
        return idx, dist

    def self_dist(self, bank):
        out = np.empty(bank.n, dtype=np.float64)
        self._check(self.lib.fm_self_dist(self.handle, bank.handle, _ptr(out)))
        return out

    def self_dist_batch(self, banks, want_host=True):
        """Metric_Cache builds of several images in one call (``fm_self_dist_batch``): the self distances of
        every bank, attached to it on the device; runs of integer banks (of any sizes, r05) share the
        triangular sweep's launches, up to ``batch_group`` banks each.
        ``want_host``: also return them (list of float64 arrays; synchronous); False = enqueue only."""
        n = len(banks)
        if n == 0:
            return []
        hs = (_P * n)(*[b.handle for b in banks])
        outs = [np.empty(b.n, dtype=np.float64) for b in banks] if want_host else None
        op = (_P * n)(*[_P(o.ctypes.data) if o.shape[0] else None for o in outs]) if want_host else None
        self._check(self.lib.fm_self_dist_batch(self.handle, n, hs, op))
        for b in banks:
            b.has_selfdist = True
        return outs

    def upload_fence(self):
        """Later calls on this context wait (on the device) for the refills enqueued so far."""
        self._check(self.lib.fm_upload_fence(self.handle))

    def xcheck1(self, q, t):
        tidx = np.empty(q.n, dtype=np.int32)
        dist = np.empty(q.n, dtype=np.float32)
        self._check(self.lib.fm_xcheck1(self.handle, q.handle, t.handle, _ptr(tidx), _ptr(dist)))
        return tidx, dist

    def xcheck1_keys(self, q, t, t_offset=0):
        """X1 election keys of a train-set shard (see sharding.xcheck1_sharded): uint64[nq]."""
        keys = np.empty(q.n, dtype=np.uint64)
        self._check(self.lib.fm_xcheck1_keys(self.handle, q.handle, t.handle, int(t_offset), _ptr(keys)))
        return keys

    def xcheck1_keys_dev(self, q, t, t_offset, keys_ptr):
        """``xcheck1_keys`` into device memory: ``keys_ptr`` = device address of a uint64/int64 [nq]
        buffer (``tensor.data_ptr()``)."""
        self._check(self.lib.fm_xcheck1_keys_dev(self.handle, q.handle, t.handle, int(t_offset), _P(int(keys_ptr))))

    def match_ratio(self, q, t, tau, out=None):
        """X1 + R1 fused.  ``out`` = optional (tidx i32[nq], dist f32[nq], ratio f64[nq],
        passed u8[nq]) buffers to fill (e.g. from ``pinned_empty``); then ``passed`` is
        returned as that uint8 array instead of a bool copy."""
        if out is not None:
            tidx, dist, ratio, passed = out
            for a, dt in ((tidx, np.int32), (dist, np.float32), (ratio, np.float64), (passed, np.uint8)):
                if a.dtype != dt or a.shape != (q.n,) or not a.flags.c_contiguous:
                    raise ValueError("out buffers must be contiguous [nq] int32/float32/float64/uint8")
            npass = _I64(0)
            self._check(self.lib.fm_match_ratio(self.handle, q.handle, t.handle, float(tau), _ptr(tidx),
                                                _ptr(dist), _ptr(ratio), _ptr(passed), ctypes.byref(npass)))
            return tidx, dist, ratio, passed, npass.value
        tidx = np.empty(q.n, dtype=np.int32)
        dist = np.empty(q.n, dtype=np.float32)
        ratio = np.empty(q.n, dtype=np.float64)
        passed = np.empty(q.n, dtype=np.uint8)
        npass = _I64(0)
        self._check(self.lib.fm_match_ratio(self.handle, q.handle, t.handle, float(tau), _ptr(tidx),
                                            _ptr(dist), _ptr(ratio), _ptr(passed), ctypes.byref(npass)))
        return tidx, dist, ratio, passed.astype(bool), npass.value

    def match_accepted(self, q, t, tau, out=None):
        """X1 + R1, returning only the accepted matches in ascending query index:
        (qidx i32[m], tidx i32[m], dist f32[m], ratio f64[m]).  ``out`` = optional buffers
        (qidx, tidx, dist, ratio) of equal capacity (e.g. pinned); views of them are returned."""
        if out is None:
            cap = q.n
            out = (np.empty(cap, np.int32), np.empty(cap, np.int32), np.empty(cap, np.float32), np.empty(cap, np.float64))
        qidx, tidx, dist, ratio = out
        cap = self._check_accepted_out(out)
        n = _I64(0)
        self._check(self.lib.fm_match_accepted(self.handle, q.handle, t.handle, float(tau), cap, _ptr(qidx),
                                               _ptr(tidx), _ptr(dist), _ptr(ratio), ctypes.byref(n)))
        m = min(n.value, cap)
        return qidx[:m], tidx[:m], dist[:m], ratio[:m]

    @staticmethod
    def _check_accepted_out(out):
        """Capacity of an (qidx, tidx, dist, ratio) output tuple; the library (or, for pinned
        buffers, the device) writes up to that many rows into each array."""
        qidx, tidx, dist, ratio = out
        cap = qidx.shape[0] if getattr(qidx, "ndim", 0) == 1 else -1
        for a, dt in ((qidx, np.int32), (tidx, np.int32), (dist, np.float32), (ratio, np.float64)):
            if not isinstance(a, np.ndarray) or a.dtype != dt or a.shape != (cap,) or not a.flags.c_contiguous \
                    or not a.flags.writeable:
                raise ValueError("out buffers must be writable contiguous 1-D int32/int32/float32/float64 of one length")
        return cap

    def match_accepted_async(self, q, t, tau, out, count):
        """Enqueue X1 + R1 + compaction and return at once.  ``out`` = (qidx, tidx, dist, ratio)
        and ``count`` (int64[1]) must come from ``pinned_empty``; they are valid after ``sync()``:
        ``m = int(count[0])`` accepted matches in ``out[i][:m]``."""
        cap = self._check_accepted_out(out)
        if not isinstance(count, np.ndarray) or count.dtype != np.int64 or count.size < 1:
            raise ValueError("count must be an int64 array (pinned_empty(1, np.int64))")
        qidx, tidx, dist, ratio = out
        self._check(self.lib.fm_match_accepted_async(self.handle, q.handle, t.handle, float(tau), cap, _ptr(qidx),
                                                     _ptr(tidx), _ptr(dist), _ptr(ratio), _ptr(count)))

    def prepare_batch(self, pairs, outs, counts):
        """Argument block of ``match_accepted_batch`` for a fixed list of (query bank, train bank) pairs
        and their output buffers, built once (the per-call cost of a batch sits in front of its first
        launch): ``outs[i]`` = (qidx, tidx, dist, ratio), ``counts[i]`` = int64[1], all ``pinned_empty``."""
        n = len(pairs)
        if len(outs) != n or len(counts) != n:
            raise ValueError("pairs, outs and counts must have the same length")
        cap = None
        for out, cnt in zip(outs, counts):
            c = self._check_accepted_out(out)
            cap = c if cap is None else min(cap, c)
            if not isinstance(cnt, np.ndarray) or cnt.dtype != np.int64 or cnt.size < 1:
                raise ValueError("count must be an int64 array (pinned_empty(1, np.int64))")
        arr = lambda vals: (_P * n)(*[_P(int(v)) if v is not None else None for v in vals])
        args = (n, arr([q.handle.value for q, _ in pairs]), arr([t.handle.value for _, t in pairs]), int(cap or 0),
                arr([_ptr(o[0]) for o in outs]), arr([_ptr(o[1]) for o in outs]), arr([_ptr(o[2]) for o in outs]),
                arr([_ptr(o[3]) for o in outs]), arr([_ptr(c) for c in counts]))
        return {"args": args, "keep": (list(pairs), list(outs), list(counts))}     # (keeps banks and buffers alive)

    def match_accepted_batch(self, pairs, tau, outs=None, counts=None):
        """``match_accepted_async`` for a list of (query bank, train bank) pairs in ONE call: consecutive
        pairs share distance-kernel launches (up to eight per launch; r05: of any sizes).  Either the lists
        (``outs[i]`` = (qidx, tidx, dist, ratio), ``counts[i]`` = int64[1], all from ``pinned_empty``) or
        the block ``prepare_batch`` built from them; results are valid after ``sync()``."""
        batch = pairs if isinstance(pairs, dict) else self.prepare_batch(pairs, outs, counts)
        n, qh, th, cap, a0, a1, a2, a3, ac = batch["args"]
        self._check(self.lib.fm_match_accepted_batch(self.handle, n, qh, th, float(tau), cap, a0, a1, a2, a3, ac))

    def match_accepted_dev_batch(self, pairs, tau, rows_ptr, counts_ptr, cap, h_counts=None, consumer_stream=None):
        """``match_accepted_batch`` with device outputs: ``rows_ptr`` = device address of an int32
        [n, cap, 3] block, ``counts_ptr`` of an int64 [n] array; ``h_counts`` = ``pinned_empty(n, np.int64)``
        or None; ``consumer_stream`` as in ``match_accepted_dev_async``.  ``pairs`` may be the list of
        (query bank, train bank) or a ``prepare_pairs`` block."""
        n, qh, th = pairs["args"] if isinstance(pairs, dict) else self.prepare_pairs(pairs)["args"]
        if h_counts is not None and (not isinstance(h_counts, np.ndarray) or h_counts.dtype != np.int64 or h_counts.size < n):
            raise ValueError("h_counts must be an int64 array of n words (pinned_empty(n, np.int64))")
        self._check(self.lib.fm_match_accepted_dev_batch(self.handle, n, qh, th, float(tau), int(cap), _P(int(rows_ptr)),
                                                         _P(int(counts_ptr)), _ptr(h_counts) if h_counts is not None else None,
                                                         _stream_arg(consumer_stream)))

    def prepare_pairs(self, pairs):
        """The bank-handle arrays of a fixed list of pairs, built once (``match_accepted_dev_batch``)."""
        n = len(pairs)
        arr = lambda vals: (_P * n)(*[_P(int(v)) for v in vals])
        return {"args": (n, arr([q.handle.value for q, _ in pairs]), arr([t.handle.value for _, t in pairs])), "keep": list(pairs)}

    def match_accepted_dev(self, q, t, tau, rows_ptr, count_ptr, cap):
        """X1 + R1 with the accepted matches left on the device: ``rows_ptr`` = device address of
        an int32 [cap, 3] buffer (query index, train index, float32 distance bits), ``count_ptr``
        = device address of an int64 word (e.g. ``tensor.data_ptr()`` of torch tensors on this
        context's device).  Returns the number accepted."""
        n = _I64(0)
        self._check(self.lib.fm_match_accepted_dev(self.handle, q.handle, t.handle, float(tau), int(cap),
                                                   _P(int(rows_ptr)), _P(int(count_ptr)), ctypes.byref(n)))
        return n.value

    def match_accepted_dev_async(self, q, t, tau, rows_ptr, count_ptr, cap, h_count=None, consumer_stream=None):
        """``match_accepted_dev`` enqueued without a synchronisation (``sync()`` later).  ``h_count``:
        a ``pinned_empty(1, np.int64)`` array that also receives the count, or None;
        ``consumer_stream``: the raw handle of the stream that will read the buffers
        (``torch.cuda.current_stream().cuda_stream`` -- 0 is a stream, PyTorch's default one), ordered
        against the fill in both directions; None = no such stream."""
        if h_count is not None and (not isinstance(h_count, np.ndarray) or h_count.dtype != np.int64 or h_count.size < 1):
            raise ValueError("h_count must be an int64 array (pinned_empty(1, np.int64))")
        self._check(self.lib.fm_match_accepted_dev_async(self.handle, q.handle, t.handle, float(tau), int(cap),
                                                         _P(int(rows_ptr)), _P(int(count_ptr)),
                                                         _ptr(h_count) if h_count is not None else None,
                                                         _stream_arg(consumer_stream)))

    def knn2_ratio(self, q, t, tau, out=None):
        """Classic Ratio-Match: 2-NN + d1/d2 < tau, accepted matches in ascending query index:
        (qidx i32[m], tidx i32[m], dist f32[m] (= d1), ratio f64[m])."""
        if out is None:
            cap = q.n
            out = (np.empty(cap, np.int32), np.empty(cap, np.int32), np.empty(cap, np.float32), np.empty(cap, np.float64))
        qidx, tidx, dist, ratio = out
        cap = self._check_accepted_out(out)
        n = _I64(0)
        self._check(self.lib.fm_knn2_ratio(self.handle, q.handle, t.handle, float(tau), cap, _ptr(qidx), _ptr(tidx),
                                           _ptr(dist), _ptr(ratio), ctypes.byref(n)))
        m = min(n.value, cap)
        return qidx[:m], tidx[:m], dist[:m], ratio[:m]

    def ratio_filter(self, dist, selfdist, tau, qrows=None):
        dist = np.ascontiguousarray(dist, dtype=np.float32)
        selfdist = np.ascontiguousarray(selfdist, dtype=np.float64)
        n = dist.shape[0]
        if qrows is not None:
            qrows = np.ascontiguousarray(qrows, dtype=np.int32)
            if qrows.shape[0] != n:
                raise ValueError("qrows and dist must have the same length")
            if n and int(qrows.max()) >= selfdist.shape[0]:
                raise ValueError("qrows index beyond selfdist")
        elif selfdist.shape[0] < n:
            raise ValueError("selfdist shorter than dist")
        ratio = np.empty(n, dtype=np.float64)
        passed = np.empty(n, dtype=np.uint8)
        npass = _I64(0)
        self._check(self.lib.fm_ratio_filter(self.handle, _ptr(dist), _ptr(selfdist), _ptr(qrows), n,
                                             float(tau), _ptr(ratio), _ptr(passed), ctypes.byref(npass)))
        return ratio, passed.astype(bool), npass.value

    def xcheck1_batched(self, q, q_rows, q_off, t, t_off):
        q_rows = np.ascontiguousarray(q_rows, dtype=np.int32)
        q_off = np.ascontiguousarray(q_off, dtype=np.int64)
        t_off = np.ascontiguousarray(t_off, dtype=np.int64)
        if q_off.shape != t_off.shape or q_off.ndim != 1 or q_off.shape[0] < 1:
            raise ValueError("q_off and t_off must both be [B+1]")
        nb = q_off.shape[0] - 1
        tot = int(q_off[-1])
        if q_rows.shape[0] != tot:
            raise ValueError("q_rows length must equal q_off[-1]")
        tidx = np.empty(tot, dtype=np.int32)
        dist = np.empty(tot, dtype=np.float32)
        ratio = np.empty(tot, dtype=np.float64)
        self._check(self.lib.fm_xcheck1_batched(self.handle, q.handle, _ptr(q_rows), _ptr(q_off), t.handle,
                                                _ptr(t_off), nb, _ptr(tidx), _ptr(dist), _ptr(ratio)))
        return tidx, dist, ratio

    @staticmethod
    def expand_slots(expanders):
        """Run slot of every entry of a launch: the k-th entry that names an Expander uses its slot k
        (fm_expand_run's rule; several thresholds of one pair run side by side, each in a state of its own)."""
        seen, slots = {}, []
        for e in expanders:
            k = seen.get(id(e), 0)
            slots.append(k)
            seen[id(e)] = k + 1
        return slots

    def expand_run(self, expanders, seeds, taus):
        """Run the device-resident expansion loop for several runs in one launch (one workgroup each); an
        Expander may appear several times (e.g. once per threshold).  Returns per run
        (n_matches, n_rounds, n_pairs, status)."""
        n = len(expanders)
        seeds = [np.ascontiguousarray(s, dtype=np.float64).reshape(-1, 2, 2) for s in seeds]
        hs = (_P * n)(*[e.handle for e in expanders])
        sp = (_P * n)(*[s.ctypes.data if s.shape[0] else None for s in seeds])
        ns = np.array([s.shape[0] for s in seeds], dtype=np.int64)
        tau = np.ascontiguousarray(taus, dtype=np.float64)
        nm = np.zeros(n, dtype=np.int64)
        nr = np.zeros(n, dtype=np.int64)
        npairs = np.zeros(n, dtype=np.int64)
        st = np.zeros(n, dtype=np.int32)
        self._check(self.lib.fm_expand_run(self.handle, n, hs, sp, _ptr(ns), _ptr(tau), _ptr(nm), _ptr(nr),
                                           _ptr(npairs), _ptr(st)))
        return [(int(nm[i]), int(nr[i]), int(npairs[i]), int(st[i])) for i in range(n)]

    # -- result gather over RCCL (fm_comm_*, fm_gather_matches) ----------------------------
    def comm_unique_id(self):
        """128 opaque bytes drawn by ONE rank; hand them to every rank's ``comm_init``."""
        buf = ctypes.create_string_buffer(128)
        rc = self.lib.fm_comm_unique_id(buf)
        if rc != 0:
            msg = self.lib.fm_last_error(None)
            raise FastMatchHipError("fm_comm_unique_id failed (%d): %s" % (rc, msg.decode() if msg else "?"))
        return buf.raw

    def comm_init(self, nranks, rank, unique_id):
        uid = ctypes.create_string_buffer(bytes(unique_id), 128)
        self._check(self.lib.fm_comm_init(self.handle, int(nranks), int(rank), uid))

    def comm_destroy(self):
        self._check(self.lib.fm_comm_destroy(self.handle))

    def gather_matches(self, rows_ptr, count_ptr, cap, all_rows_ptr, all_counts_ptr, wait=True):
        """All-gather the device rows / count left by ``match_accepted_dev`` into device buffers
        [nranks, cap, 3] int32 and [nranks] int64 (addresses, e.g. ``tensor.data_ptr()``)."""
        self._check(self.lib.fm_gather_matches(self.handle, _P(int(rows_ptr)), _P(int(count_ptr)), int(cap),
                                               _P(int(all_rows_ptr)), _P(int(all_counts_ptr)), 1 if wait else 0))

    def gather_matches_counted(self, rows_ptr, count_ptr, cap, all_rows_ptr, all_counts_ptr):
        """Two-phase form of ``gather_matches``: counts first, then only ``m`` = the fullest rank's rows per
        rank; the rows arrive as [nranks, m, 3] at ``all_rows_ptr``.  Returns m.  Synchronous."""
        m = _I64(0)
        self._check(self.lib.fm_gather_matches_counted(self.handle, _P(int(rows_ptr)), _P(int(count_ptr)), int(cap),
                                                       _P(int(all_rows_ptr)), _P(int(all_counts_ptr)), ctypes.byref(m)))
        return int(m.value)

    # -- bookkeeping ---------------------------------------------------------------------
    def stats(self):
        s = fm_stats_ex()
        self._check(self.lib.fm_get_stats_ex(self.handle, ctypes.byref(s), ctypes.sizeof(s)))
        return {"kernel_ms": s.kernel_ms, "total_ms": s.total_ms, "kernel_launches": s.kernel_launches,
                "pairs": s.pairs, "calls": s.calls, "bytes_moved": s.bytes_moved}

    def reset_stats(self):
        self._check(self.lib.fm_reset_stats(self.handle))

    def sync(self):
        self._check(self.lib.fm_sync(self.handle))

    def expand_fetch_many(self, expanders, counts, slots=None):
        """(index, positions, ratio) arrays of several runs after one ``expand_run``: every copy
        enqueued, one synchronisation (``Expander.fetch`` costs one per pair).  ``slots[i]`` = run slot of
        ``expanders[i]`` (``expand_slots`` of the launch's list); None = slots by appearance in THIS list."""
        n = len(expanders)
        out = [(np.empty(c, dtype=np.int32), np.empty((c, 2, 2), dtype=np.float64), np.empty(c, dtype=np.float64)) for c in counts]
        if n:
            arr = lambda vals: (_P * n)(*[_P(int(v)) if v is not None else None for v in vals])
            slots = self.expand_slots(expanders) if slots is None else slots
            sl = np.ascontiguousarray(slots, dtype=np.int32)
            self._check(self.lib.fm_expand_fetch_many(self.handle, n, arr([e.handle.value for e in expanders]), _ptr(sl),
                                                      (_I64 * n)(*[int(c) for c in counts]), arr([_ptr(o[0]) for o in out]),
                                                      arr([_ptr(o[1]) for o in out]), arr([_ptr(o[2]) for o in out])))
        return out

    def mark(self):
        """Ticket for "everything enqueued so far" (``wait(ticket)`` blocks until it is done, later work keeps running)."""
        t = _I64(0)
        self._check(self.lib.fm_mark(self.handle, ctypes.byref(t)))
        return t.value

    def wait(self, ticket):
        self._check(self.lib.fm_wait(self.handle, int(ticket)))

    def f32_filter_stats(self):
        """(row-reduces routed through the bf16x3 filter, of which redone by the all-pairs kernel)."""
        a, b = _I64(0), _I64(0)
        self._check(self.lib.fm_f32_filter_stats(self.handle, ctypes.byref(a), ctypes.byref(b)))
        return int(a.value), int(b.value)

    def mem_info(self):
        """(free, total) bytes of the context's device."""
        f, t = _I64(0), _I64(0)
        self._check(self.lib.fm_mem_info(self.handle, ctypes.byref(f), ctypes.byref(t)))
        return int(f.value), int(t.value)

    def device_name(self):
        buf = ctypes.create_string_buffer(256)
        self._check(self.lib.fm_device_name(self.handle, buf, 256))
        return buf.value.decode()


_default_ctx = {}
_default_lock = threading.Lock()


def default_context(device=None):
    """Process-wide context per device.  ``device=None`` uses LOCAL_RANK (one process per
    GPU under torch.distributed.run) or 0."""
    if device is None:
        device = int(os.environ.get("FM_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    with _default_lock:
        ctx = _default_ctx.get(device)
        if ctx is None or ctx.handle is None:
            ctx = Context(device)
            _default_ctx[device] = ctx
        return ctx
