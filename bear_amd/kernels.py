"""Thin torch-facing wrappers over the C ABI (one call = one asynchronous launch on the
current torch stream).  Tensors are passed as raw device pointers; nothing here computes."""
import ctypes
import warnings

import numpy as np
import torch

from . import _lib

EPSILON = 1e-7  # keras epsilon, bear_model/core.py:8


class Workspace:
    """Owns a bear_ws handle for one device (per-block partial sums)."""

    def __init__(self, device=None):
        if not torch.cuda.is_available():
            raise RuntimeError("bear_amd requires an MI355X (HIP) device; there is no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else torch.device(device).index or 0)
        h = ctypes.c_void_p()
        _lib.check(_lib.lib().bear_ws_create(self.device.index, ctypes.byref(h)), "bear_ws_create")
        self._h = h

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                _lib.lib().bear_ws_destroy(h)
            except Exception:
                pass

    @property
    def handle(self):
        return self._h


_default_ws = {}


def default_workspace(device, for_capture=False):
    """The workspace of (device, current stream).  A workspace serves ONE stream at a time (include/bear_hip.h: its block
    partials and the arrival counter of the "last block finalizes" kernels are per-launch state), so a side stream gets its own;
    a stream that is capturing a graph shares the device's first workspace (nothing may be allocated during capture, and a
    capture orders every launch of the step on that one stream anyway)."""
    idx = torch.device(device).index
    if idx is None:
        idx = torch.cuda.current_device()
    stream = torch.cuda.current_stream(idx)
    key = (idx, 0 if (for_capture or stream == torch.cuda.default_stream(idx) or torch.cuda.is_current_stream_capturing())
           else stream.cuda_stream)       # for_capture: the one a capturing stream will resolve to (created BEFORE the capture)
    ws = _default_ws.get(key)
    if ws is None:
        ws = _default_ws[key] = Workspace(torch.device("cuda", idx))
    return ws


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _check_rows(t, dtype, name):
    if not (t.is_cuda and t.dtype == dtype and t.dim() == 2 and t.shape[1] == 5 and t.is_contiguous()):
        raise ValueError(f"{name} must be a contiguous CUDA tensor of shape [N, 5] and dtype {dtype}")
    if t.data_ptr() % 16:
        # the C ABI wants 16-byte aligned rows; a slice of a [N,5] tensor may start on any row
        t = t.clone()
    return t


def counts_dtype():
    """Counts travel as 32-bit words (KMC's counter range, summarize.py:66-67). torch has no
    general uint32 arithmetic, so int32 storage is used and reinterpreted by the kernel."""
    return torch.int32


def dm_prior(counts, prior, h_signed, eps=EPSILON, train_ar=False, want_grad=False, out=None, ws=None):
    """sum LL and d/dh_signed over rows (bear_net._train_step arithmetic, bear_net.py:146-197).
    Returns (out[2] device tensor, grad_prior or None)."""
    counts = _check_rows(counts, torch.int32, "counts")
    prior = _check_rows(prior, torch.float64, "prior")
    if counts.shape[0] != prior.shape[0]:
        raise ValueError("counts and prior must have the same number of rows")
    ws = ws or default_workspace(counts.device)
    if out is None:
        out = torch.empty(2, dtype=torch.float64, device=counts.device)
    grad = torch.empty_like(prior) if want_grad else None
    with torch.cuda.device(counts.device):
        st = _lib.lib().bear_dm_prior_f64(ws.handle, _ptr(counts), _ptr(prior), counts.shape[0], float(h_signed), float(eps),
                                          int(bool(train_ar)), _ptr(out), _ptr(grad), _stream())
    _lib.check(st, "bear_dm_prior_f64")
    return out, grad


def dm_ref(train, ref, h_signed, tau_signed, nu_signed, eps=EPSILON, train_ar=False, out=None, ws=None):
    """[sum LL, d/dh_signed, d/dtau_signed, d/dnet_weight_signed] (bear_ref._train_step
    arithmetic with the stop net function, bear_ref.py:207-259)."""
    train = _check_rows(train, torch.int32, "train")
    ref = _check_rows(ref, torch.int32, "ref")
    if train.shape[0] != ref.shape[0]:
        raise ValueError("train and ref must have the same number of rows")
    ws = ws or default_workspace(train.device)
    if out is None:
        out = torch.empty(4, dtype=torch.float64, device=train.device)
    with torch.cuda.device(train.device):
        st = _lib.lib().bear_dm_ref_f64(ws.handle, _ptr(train), _ptr(ref), train.shape[0], float(h_signed), float(tau_signed),
                                        float(nu_signed), float(eps), int(bool(train_ar)), _ptr(out), _stream())
    _lib.check(st, "bear_dm_ref_f64")
    return out


class Plan:
    """Count-dependent part of the hot path for a table that stays resident across optimizer steps
    (work items sorted by count; include/bear_hip.h "Planned variants").  Keeps the count tensor alive."""

    def __init__(self, counts, ncol, ws=None, ref=None, rows_if_dense=False):
        """ref (ncol = 4 only): the reference column the planned mode-R entries will be called with -- the plan then folds the
        contexts without reference counts into a histogram (``bear_plan_create_ref``).
        rows_if_dense (ncol = 5, a plan for ``dm_prior_planned`` / ``dm_prior_planned_dev`` ONLY): ``bear_plan_create_auto`` -- a table
        whose cells are mostly beyond the product path gets the dense form (``self.rowwise``: nothing kept per item, the step streams
        the count and prior rows); the fused linear / convolutional steps refuse such a plan."""
        counts = _check_rows(counts, torch.int32, "counts")
        self.counts = counts
        self.ncol = int(ncol)
        self.ref = None
        self.rowwise = False
        self.ws = ws or default_workspace(counts.device)
        h = ctypes.c_void_p()
        with torch.cuda.device(counts.device):
            torch.cuda.current_stream().synchronize()  # plan construction runs on the default stream
            if ref is not None:
                ref = _check_rows(ref, torch.int32, "ref")
                if self.ncol != 4 or ref.shape[0] != counts.shape[0] or ref.data_ptr() % 16:
                    raise ValueError("a reference-aware plan needs ncol = 4 and a 16-byte aligned ref slab with one row per context")
                self.ref = ref
                st = _lib.lib().bear_plan_create_ref(self.ws.handle, _ptr(counts), _ptr(ref), counts.shape[0], ctypes.byref(h))
                _lib.check(st, "bear_plan_create_ref")
            elif rows_if_dense:
                if self.ncol != 5:
                    raise ValueError("rows_if_dense: five-column plans")
                rw = ctypes.c_int(0)
                st = _lib.lib().bear_plan_create_auto(self.ws.handle, _ptr(counts), counts.shape[0], ctypes.byref(rw), ctypes.byref(h))
                _lib.check(st, "bear_plan_create_auto")
                self.rowwise = bool(rw.value)
            else:
                st = _lib.lib().bear_plan_create(self.ws.handle, _ptr(counts), counts.shape[0], self.ncol, ctypes.byref(h))
                _lib.check(st, "bear_plan_create")
        self._h = h

    @property
    def nbytes(self):
        return int(_lib.lib().bear_plan_bytes(self._h))

    def pair_contexts(self, kmer_index, lag):
        """``bear_plan_pair_contexts``: ties this (five-column) plan to the index words ``kmer_index`` (``linear_index``) of its rows
        so that the fused linear step takes neighbouring contexts with equal leading letters two at a time.  Returns False when the
        table is too sparse for it (the plan is then unchanged).  The tensor is kept alive; it must not be modified afterwards."""
        _check_codes(kmer_index)
        if kmer_index.shape[0] != self.counts.shape[0] or kmer_index.data_ptr() % 16:
            raise ValueError("kmer_index: one 16-byte aligned index word per row of the plan's count slab")
        ok = ctypes.c_int(0)
        with torch.cuda.device(self.counts.device):
            st = _lib.lib().bear_plan_pair_contexts(self._h, _ptr(kmer_index), int(lag), ctypes.byref(ok), _stream())
            if st == _lib.ERR_NOMEM:
                # the lists come from hipMalloc: slabs the torch allocator has cached are out of its reach.  Given back only now --
                # emptying the cache on every call made a streamed epoch pay hipFree / hipMalloc (device-wide synchronisations that
                # serialise with the side stream's prefetch) on every batch load
                torch.cuda.empty_cache()
                st = _lib.lib().bear_plan_pair_contexts(self._h, _ptr(kmer_index), int(lag), ctypes.byref(ok), _stream())
        if st == _lib.ERR_NOMEM:         # an optional speed-up: a table close to the card's capacity runs the plain step
            warnings.warn("bear_plan_pair_contexts: out of device memory, the linear step keeps its plain lists")
            self.paired_codes = None
            return False
        _lib.check(st, "bear_plan_pair_contexts")
        self.paired_codes = kmer_index if ok.value else None
        return bool(ok.value)

    def count_total(self):
        """(total, bound) -- each [sum of all counts, cells that hold a count, largest count]: of the plan's table, and the values in
        force for the deterministic mode's fixed-point scale (``bear_plan_count_total``)."""
        a, b = (ctypes.c_double * 3)(), (ctypes.c_double * 3)()
        _lib.check(_lib.lib().bear_plan_count_total(self._h, a, b), "bear_plan_count_total")
        return list(a), list(b)

    def set_count_bound(self, bound):
        """``bear_plan_set_count_bound``: [sum of counts, non-zero cells, largest count] of EVERYTHING that is added into one gradient
        together with this plan's rows (BEAR_AMD_DETERMINISTIC: the scale of the fixed-point gradient tables; ranks that share a batch
        must agree on it: sums of the first two, maximum of the third)."""
        _lib.check(_lib.lib().bear_plan_set_count_bound(self._h, (ctypes.c_double * 3)(*[float(x) for x in bound])), "bear_plan_set_count_bound")

    def pair_info(self):
        """(tiles that take the paired form of the linear step, tiles that keep their plain list) -- ``bear_plan_pair_info``."""
        a, b = ctypes.c_uint64(0), ctypes.c_uint64(0)
        _lib.lib().bear_plan_pair_info(self._h, ctypes.byref(a), ctypes.byref(b))
        return int(a.value), int(b.value)

    def attach_cnn_levels(self, kmer_code, lag, filter_width):
        """``bear_plan_attach_cnn_levels``: prefix levels of the (k-mer-sorted) packed contexts ``kmer_code`` (``pack_kmers``) of this
        plan's rows for the convolutional training step -- a position is then evaluated once per distinct prefix.  Returns the
        number of levels attached (0: the step runs as before).  The tensor is kept alive; it must not be modified afterwards."""
        _check_codes(kmer_code)
        if kmer_code.shape[0] != self.counts.shape[0] or kmer_code.data_ptr() % 16:
            raise ValueError("kmer_code: one 16-byte aligned packed context per row of the plan's count slab")
        n = ctypes.c_int(0)
        with torch.cuda.device(self.counts.device):
            st = _lib.lib().bear_plan_attach_cnn_levels(self._h, _ptr(kmer_code), int(lag), int(filter_width), ctypes.byref(n), _stream())
            if st == _lib.ERR_NOMEM:
                torch.cuda.empty_cache()     # (as in pair_contexts: only when the first attempt ran out)
                st = _lib.lib().bear_plan_attach_cnn_levels(self._h, _ptr(kmer_code), int(lag), int(filter_width), ctypes.byref(n), _stream())
        if st == _lib.ERR_NOMEM:
            warnings.warn("bear_plan_attach_cnn_levels: out of device memory, the convolutional step runs without prefix levels")
            self.cnn_codes = None
            return 0
        _lib.check(st, "bear_plan_attach_cnn_levels")
        self.cnn_codes = kmer_code if (n.value or self.cnn_window_rows()) else None     # (kept alive: levels and window tables are tied to it)
        return int(n.value)

    def cnn_window_rows(self):
        """[(level, position, distinct windows)] of the window tables attached with the prefix levels (``bear_plan_cnn_window_rows``;
        level 0 = the contexts)."""
        n = int(_lib.lib().bear_plan_cnn_window_rows(self._h, None, None, None, 0))
        if n <= 0:
            return []
        rows, pos, lev = (ctypes.c_uint64 * n)(), (ctypes.c_int * n)(), (ctypes.c_int * n)()
        _lib.lib().bear_plan_cnn_window_rows(self._h, rows, pos, lev, n)
        return [(int(lev[q]), int(pos[q]), int(rows[q])) for q in range(n)]

    def cnn_level_rows(self, with_letters=False):
        """Rows of the attached prefix levels 1 .. n (``bear_plan_cnn_level_rows``); [] without levels.  ``with_letters``: (rows,
        prefix lengths)."""
        n = int(_lib.lib().bear_plan_cnn_level_rows(self._h, None, None, 0))
        if n <= 0:
            return ([], []) if with_letters else []
        rows, letters = (ctypes.c_uint64 * n)(), (ctypes.c_int * n)()
        _lib.lib().bear_plan_cnn_level_rows(self._h, rows, letters, n)
        return ([int(r) for r in rows], [int(x) for x in letters]) if with_letters else [int(r) for r in rows]

    def tiles(self):
        """Diagnostics (``bear_plan_tile_info``): (row0 [T] uint64, rows [T] uint32, items [T] uint32, stream_offset [T] uint64)."""
        import numpy as np
        n = int(_lib.lib().bear_plan_tile_count(self._h))
        row0, rows, items, off = np.empty(n, np.uint64), np.empty(n, np.uint32), np.empty(n, np.uint32), np.empty(n, np.uint64)
        _lib.check(_lib.lib().bear_plan_tile_info(self._h, 0, n, row0.ctypes.data, rows.ctypes.data, items.ctypes.data, off.ctypes.data),
                   "bear_plan_tile_info")
        return row0, rows, items, off

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                _lib.lib().bear_plan_destroy(h)
            except Exception:
                pass


def dm_prior_planned(plan, prior, h_signed, eps=EPSILON, out=None, normalized=False, want_grad=False, train_ar=False):
    """Planned twin of dm_prior (BEAR mode): [sum LL, d/dh_signed]; with want_grad also the gradient rows
    d sum LL / d prior, returned as (out, grad).  normalized=True asserts that every prior row sums to one
    (any softmax output)."""
    counts = plan.counts
    _check_rows(prior, torch.float64, "prior")
    if prior.data_ptr() % 16 or prior.shape[0] != counts.shape[0] or plan.ncol != 5:
        raise ValueError("prior must be 16-byte aligned with one row per planned context (plan ncol=5)")
    if out is None:
        out = torch.empty(2, dtype=torch.float64, device=counts.device)
    if want_grad:
        grad = torch.empty_like(prior)
        with torch.cuda.device(counts.device):
            st = _lib.lib().bear_dm_prior_plan_grad_f64(plan.ws.handle, plan._h, _ptr(counts), _ptr(prior), counts.shape[0],
                                                        float(h_signed), float(eps), int(bool(train_ar)), int(bool(normalized)),
                                                        _ptr(out), _ptr(grad), _stream())
        _lib.check(st, "bear_dm_prior_plan_grad_f64")
        return out, grad
    with torch.cuda.device(counts.device):
        st = _lib.lib().bear_dm_prior_plan_f64(plan.ws.handle, plan._h, _ptr(counts), _ptr(prior), counts.shape[0],
                                               float(h_signed), float(eps), int(bool(train_ar)), int(bool(normalized)), _ptr(out),
                                               _stream())
    _lib.check(st, "bear_dm_prior_plan_f64")
    return out


def dm_ref_planned(plan, ref, h_signed, tau_signed, nu_signed, eps=EPSILON, out=None, train_ar=False):
    """Planned twin of dm_ref (BEAR mode): [sum LL, d/dh_signed, d/dtau_signed, d/dnet_weight_signed]."""
    train = plan.counts
    _check_rows(ref, torch.int32, "ref")
    if ref.data_ptr() % 16 or ref.shape[0] != train.shape[0] or plan.ncol != 4:
        raise ValueError("ref must be 16-byte aligned with one row per planned context (plan ncol=4)")
    if out is None:
        out = torch.empty(4, dtype=torch.float64, device=train.device)
    with torch.cuda.device(train.device):
        st = _lib.lib().bear_dm_ref_plan_f64(plan.ws.handle, plan._h, _ptr(train), _ptr(ref), train.shape[0], float(h_signed),
                                             float(tau_signed), float(nu_signed), float(eps), int(bool(train_ar)), _ptr(out), _stream())
    _lib.check(st, "bear_dm_ref_plan_f64")
    return out


def dm_items(x, c, path=0, ws=None):
    """D = lgamma(x+c) - lgamma(x), P = digamma(x+c) - digamma(x) per item (diagnostic entry)."""
    if not (x.is_cuda and x.dtype == torch.float64 and x.is_contiguous() and c.is_cuda and c.dtype == torch.int32
            and c.is_contiguous() and x.shape == c.shape and x.dim() == 1):
        raise ValueError("x: contiguous CUDA float64 [n]; c: contiguous CUDA int32 [n]")
    ws = ws or default_workspace(x.device)
    D, P = torch.empty_like(x), torch.empty_like(x)
    with torch.cuda.device(x.device):
        st = _lib.lib().bear_dm_items_f64(ws.handle, _ptr(x), _ptr(c), x.shape[0], int(path), _ptr(D), _ptr(P), _stream())
    _lib.check(st, "bear_dm_items_f64")
    return D, P


LINEAR_MAX_LAG = 21   # LIN_MAX_LAG of kernels_linear.h (3 bits per letter in one 64-bit word)


def pack_kmers(codes):
    """int8 code matrix [n, lag] (core.encode_kmers) -> packed contexts, int64 storage [n] (3 bits per letter)."""
    if not (codes.is_cuda and codes.dtype == torch.int8 and codes.dim() == 2 and codes.is_contiguous()):
        raise ValueError("codes must be a contiguous CUDA int8 tensor [n, lag]")
    out = torch.empty(codes.shape[0], dtype=torch.int64, device=codes.device)
    with torch.cuda.device(codes.device):
        st = _lib.lib().bear_pack_kmers_u64(_ptr(codes), codes.shape[0], codes.shape[1], _ptr(out), _stream())
    _lib.check(st, "bear_pack_kmers_u64")
    return out


def linear_index(kmer_code, lag):
    """Packed contexts (``pack_kmers``) -> the table-row words the fused linear head reads (``bear_linear_index_u64``); once per
    batch, the contexts do not change between steps."""
    if not (kmer_code.is_cuda and kmer_code.dtype == torch.int64 and kmer_code.dim() == 1 and kmer_code.is_contiguous()):
        raise ValueError("kmer_code must be a contiguous CUDA int64 tensor [n_rows] (pack_kmers)")
    out = torch.empty_like(kmer_code)
    with torch.cuda.device(kmer_code.device):
        st = _lib.lib().bear_linear_index_u64(_ptr(kmer_code), kmer_code.shape[0], int(lag), _ptr(out), _stream())
    _lib.check(st, "bear_linear_index_u64")
    return out


def dm_linear(plan, kmer_index, mat, h_signed, eps=EPSILON, train_ar=False, out=None):
    """One launch of ``bear_dm_linear_f64``: the bear_net step with the linear AR function fused on the plan.
    ``kmer_index`` = ``linear_index(pack_kmers(codes), lag)``.
    Returns (out[2] = {sum LL, d/dh_signed}, grad_mat [lag,5,5] = d sum LL / d mat)."""
    n = plan.counts.shape[0]
    kmer_code = kmer_index
    if not (kmer_code.is_cuda and kmer_code.dtype == torch.int64 and kmer_code.dim() == 1 and kmer_code.is_contiguous()
            and kmer_code.shape[0] == n):
        raise ValueError("kmer_index must be a contiguous CUDA int64 tensor [n_rows] (linear_index)")
    if kmer_code.data_ptr() % 16:
        kmer_code = kmer_code.clone()
    if not (mat.is_cuda and mat.dtype == torch.float64 and mat.dim() == 3 and mat.shape[1:] == (5, 5) and mat.is_contiguous()):
        raise ValueError("mat must be a contiguous CUDA float64 tensor [lag, 5, 5]")
    if out is None:
        out = torch.empty(2, dtype=torch.float64, device=mat.device)
    grad = torch.empty_like(mat)
    with torch.cuda.device(mat.device):
        st = _lib.lib().bear_dm_linear_f64(plan.ws.handle, plan._h, _ptr(plan.counts), _ptr(kmer_code), _ptr(mat), mat.shape[0], n,
                                           float(h_signed), float(eps), int(bool(train_ar)), _ptr(out), _ptr(grad), _stream())
    _lib.check(st, "bear_dm_linear_f64")
    return out, grad


def _host_f64(values):
    a = np.ascontiguousarray(np.atleast_1d(np.asarray(values, dtype=np.float64)))
    return a, a.ctypes.data_as(ctypes.c_void_p)


def evaluate(test, prior, h, van_reg, train=None, eps=EPSILON, with_ar=True, noise_seed=0, row_base=0, ws=None):
    """One launch of ``bear_eval_f64``: the 7 partial sums of ``_evaluation_step`` (bear_net.py:323-371).
    Returns a device float64 vector {ll_ear[H], ll_arm, ll_van[V], cor_ear[H], cor_arm, cor_van[V], total_len}."""
    test = _check_rows(test, torch.int32, "test")
    n = test.shape[0]
    if train is not None:
        train = _check_rows(train, torch.int32, "train")
    if prior is not None:
        prior = _check_rows(prior, torch.float64, "prior")
    for t in (train, prior):
        if t is not None and t.shape[0] != n:
            raise ValueError("test, train and prior must have the same number of rows")
    hs, hp = _host_f64(h) if h is not None else (np.zeros(0), ctypes.c_void_p(0))
    vs, vp = _host_f64(van_reg) if van_reg is not None else (np.zeros(0), ctypes.c_void_p(0))
    ws = ws or default_workspace(test.device)
    out = torch.empty(2 * (hs.size + vs.size) + 3, dtype=torch.float64, device=test.device)
    with torch.cuda.device(test.device):
        st = _lib.lib().bear_eval_f64(ws.handle, _ptr(test), _ptr(train), _ptr(prior), n, hp, hs.size, int(bool(with_ar)), vp,
                                      vs.size, float(eps), int(noise_seed), int(row_base), _ptr(out), _stream())
    _lib.check(st, "bear_eval_f64")
    return out


class EvalPlan:
    """Sorted plan of a resident TEST column (``bear_eval_plan_create``): per tile the cells and row totals with a non-zero
    count, sorted by count, and the rows whose largest counts in the conditioning column ``train`` tie.  Keeps the count
    tensors alive; built asynchronously on the current stream."""

    def __init__(self, test, train=None, ws=None):
        test = _check_rows(test, torch.int32, "test")
        if train is not None:
            train = _check_rows(train, torch.int32, "train")
            if train.shape[0] != test.shape[0]:
                raise ValueError("test and train must have the same number of rows")
        self.test, self.train = test, train
        self.ws = ws or default_workspace(test.device)
        h = ctypes.c_void_p()
        with torch.cuda.device(test.device):
            st = _lib.lib().bear_eval_plan_create(self.ws.handle, _ptr(test), _ptr(train), test.shape[0], ctypes.byref(h), _stream())
        _lib.check(st, "bear_eval_plan_create")
        self._h = h

    @property
    def nbytes(self):
        return int(_lib.lib().bear_eval_plan_bytes(self._h))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                _lib.lib().bear_eval_plan_destroy(h)
            except Exception:
                pass


def evaluate_planned(plan, prior, h, van_reg, eps=EPSILON, with_ar=True, noise_seed=0, row_base=0, row_ids=None):
    """One ``bear_eval_plan_f64`` call: ``evaluate`` on a sorted plan of the test column (same output vector); the conditioning
    column is the one the plan was built with.  ``row_ids`` (int32 storage, uint32 values, [n]): the plan's buffers are a
    compacted batch whose row i is table row ``row_base + row_ids[i]`` (the key of the tie-breaking noise)."""
    test, train = plan.test, plan.train
    n = test.shape[0]
    if row_ids is not None and not (row_ids.is_cuda and row_ids.dtype == torch.int32 and row_ids.is_contiguous()
                                    and row_ids.shape == (n,) and row_ids.data_ptr() % 16 == 0):
        raise ValueError("row_ids must be a contiguous, 16-byte aligned CUDA int32 tensor [n_rows]")
    if prior is not None:
        _check_rows(prior, torch.float64, "prior")
        if prior.shape[0] != n or prior.data_ptr() % 16:
            raise ValueError("prior must be 16-byte aligned with one row per planned context")
    hs, hp = _host_f64(h) if h is not None else (np.zeros(0), ctypes.c_void_p(0))
    vs, vp = _host_f64(van_reg) if van_reg is not None else (np.zeros(0), ctypes.c_void_p(0))
    if vs.size and not (1750.0 * float(eps) < 0.5 and (vs >= 0.0).all() and (vs <= 2.0 ** 30).all()):
        raise ValueError("a planned evaluation decides the vanilla models' arg-max on the integer counts: it needs 1750 eps < 0.5 "
                         "and 0 <= van_reg <= 2^30 (kernels.evaluate takes any values)")
    out = torch.empty(2 * (hs.size + vs.size) + 3, dtype=torch.float64, device=test.device)
    with torch.cuda.device(test.device):
        st = _lib.lib().bear_eval_plan_f64(plan.ws.handle, plan._h, _ptr(test), _ptr(train), _ptr(prior), n, hp, hs.size, int(bool(with_ar)),
                                           vp, vs.size, float(eps), int(noise_seed), int(row_base), _ptr(row_ids), _ptr(out), _stream())
    _lib.check(st, "bear_eval_plan_f64")
    return out


def bmm(counts, alpha, ws=None):
    """sum_i lbeta(counts_i + alpha_k) - lbeta(alpha_k) for every alpha_k (dataloader.py:111-118): device [V]."""
    counts = _check_rows(counts, torch.int32, "counts")
    al, ap = _host_f64(alpha)
    ws = ws or default_workspace(counts.device)
    out = torch.empty(al.size, dtype=torch.float64, device=counts.device)
    with torch.cuda.device(counts.device):
        st = _lib.lib().bear_bmm_f64(ws.handle, _ptr(counts), counts.shape[0], ap, al.size, _ptr(out), _stream())
    _lib.check(st, "bear_bmm_f64")
    return out


def synth_counts(seed, row0, n_rows, device, dense=False, want=("train", "test", "ref")):
    """Rows [row0, row0+n_rows) of the synthetic k=13 table, generated on the device."""
    bufs = {k: torch.empty((n_rows, 5), dtype=torch.int32, device=device) for k in want}
    with torch.cuda.device(device):
        st = _lib.lib().bear_synth_counts_u32(int(seed), int(row0), int(n_rows), int(bool(dense)), _ptr(bufs.get("train")),
                                              _ptr(bufs.get("test")), _ptr(bufs.get("ref")), _stream())
    _lib.check(st, "bear_synth_counts_u32")
    return bufs


def synth_prior(seed, row0, n_rows, device):
    prior = torch.empty((n_rows, 5), dtype=torch.float64, device=device)
    with torch.cuda.device(device):
        st = _lib.lib().bear_synth_prior_f64(int(seed), int(row0), int(n_rows), _ptr(prior), _stream())
    _lib.check(st, "bear_synth_prior_f64")
    return prior


def synth_kmer_ids(seed, row0, n_rows, lag, device):
    """The k-mers of rows [row0, row0 + n_rows) of the synthetic table as integers in [0, 4^lag) (letter 0 most significant,
    two bits per letter): row index -> a FIXED BIJECTION of [0, 4^lag) -- the contexts of a table are DISTINCT,
    as the contexts of any count table are (summarize.py:429-449 writes one row per k-mer), in a scrambled order, and rows
    [a, b) are the same k-mers however the table is cut into shards.  The bijection: rounds of an odd multiplier + constant
    (mod 4^lag) and a right xor-shift, each invertible on 2 lag bits.  torch ops only: the same numbers on the CPU and the card
    (measurement tooling, like synth_counts; the parity tests draw their own small tables)."""
    lag = int(lag)
    if not 1 <= lag <= 31:
        raise ValueError("synth_kmer_ids: 1 <= lag <= 31")
    bits = 2 * lag
    if row0 < 0 or row0 + n_rows > (1 << bits):
        raise ValueError(f"rows [{row0}, {row0 + n_rows}) of a table of distinct {lag}-mers: only 4^{lag} = {1 << bits} exist")
    mask = (1 << bits) - 1
    x = torch.arange(int(row0), int(row0) + int(n_rows), dtype=torch.int64, device=device)
    s = (int(seed) * 0x9E3779B97F4A7C15 + 0x632BE59BD9B4E019) & ((1 << 62) - 1)
    for r in range(4):
        mul = ((s >> (7 * r)) | 1) & mask | 1                      # odd: invertible mod 2^bits
        add = (s >> (5 * r + 3)) & mask
        x = (x * mul + add) & mask                                 # (int64 products wrap: the low `bits` bits are exact)
        x = x ^ (x >> max(1, bits // 2 + (r & 1)))
    return x


def synth_kmer_codes(seed, row0, n_rows, lag, device, sort=False):
    """int8 letter codes [n_rows, lag] (0..3 = A, C, G, T: core.encode_kmers) of ``synth_kmer_ids``; ``sort``: in k-mer order
    (the order bear_net.train gives a batch) -- the ids ascending."""
    ids = synth_kmer_ids(seed, row0, n_rows, lag, device)
    if sort:
        ids = torch.sort(ids).values
    shifts = torch.arange(2 * (int(lag) - 1), -1, -2, dtype=torch.int64, device=ids.device)
    return ((ids[:, None] >> shifts[None, :]) & 3).to(torch.int8).contiguous()


def log_gamma(conc, n_samples, seed):
    """One launch of ``bear_log_gamma_f64``: device float64 [n_samples, n] of log Gamma(conc[i], 1) draws."""
    if not (conc.is_cuda and conc.dtype == torch.float64 and conc.dim() == 1 and conc.is_contiguous()):
        raise ValueError("conc must be a contiguous CUDA float64 vector")
    out = torch.empty((int(n_samples), conc.shape[0]), dtype=torch.float64, device=conc.device)
    with torch.cuda.device(conc.device):
        st = _lib.lib().bear_log_gamma_f64(_ptr(conc), conc.shape[0], int(n_samples), int(seed) & (2 ** 64 - 1), _ptr(out), _stream())
    _lib.check(st, "bear_log_gamma_f64")
    return out


def logdir_sample(counts, prior, h, vans, mc_samples, get_map=False, with_ar=False, seed=0, row_base=0, n_rows=None,
                  device=None):
    """One launch of ``bear_logdir_sample_f64``: normalised log transition probabilities
    [n_rows, 5, n_models, mc_samples] (get_var_probs.get_pdf, output='numpy').  counts None = all-zero rows."""
    n = None
    for t, dt, name in ((counts, torch.int32, "counts"), (prior, torch.float64, "prior")):
        if t is not None:
            _check_rows(t, dt, name)
            if n is not None and t.shape[0] != n:
                raise ValueError("counts and prior must have the same number of rows")
            n, device = t.shape[0], t.device
    if n is None:
        n, device = int(n_rows), torch.device(device or "cuda")
    hs, hp = _host_f64(h) if h is not None and np.size(h) else (np.zeros(0), ctypes.c_void_p(0))
    vs, vp = _host_f64(vans) if vans is not None and np.size(vans) else (np.zeros(0), ctypes.c_void_p(0))
    mc = 1 if get_map else int(mc_samples)
    M = int(bool(with_ar)) + hs.size + vs.size
    out = torch.empty((n, 5, M, mc), dtype=torch.float64, device=device)
    with torch.cuda.device(device):
        st = _lib.lib().bear_logdir_sample_f64(_ptr(counts), _ptr(prior), n, hp, hs.size, int(bool(with_ar)), vp, vs.size, mc,
                                               int(bool(get_map)), int(seed) & (2 ** 64 - 1), int(row_base), _ptr(out), _stream())
    _lib.check(st, "bear_logdir_sample_f64")
    return out


def shuffle_rows(src, seed):
    """One launch of ``bear_shuffle_rows``: a new tensor with ``dst[i] = src[perm_seed(i)]`` along dim 0."""
    if not (src.is_cuda and src.is_contiguous() and src.dim() >= 1):
        raise ValueError("src must be a contiguous CUDA tensor")
    dst = torch.empty_like(src)
    n = src.shape[0]
    row_bytes = src.element_size() * (src.numel() // n) if n else 0
    with torch.cuda.device(src.device):
        st = _lib.lib().bear_shuffle_rows(_ptr(src), _ptr(dst), n, row_bytes, int(seed) & (2 ** 64 - 1), _stream())
    _lib.check(st, "bear_shuffle_rows")
    return dst


def kmer_order(kmer_code, lag):
    """``bear_kmer_order_u64``: int32-storage permutation [n] that sorts packed contexts (``pack_kmers``) lexicographically, first
    letter most significant, stable.  Stream-ordered: the sort's scratch (``kmer_order_scratch_bytes``: ~20 B per row) is a torch
    tensor, so it comes out of -- and goes back to -- the caching allocator like every other slab."""
    _check_codes(kmer_code)
    n = kmer_code.shape[0]
    perm = torch.empty(n, dtype=torch.int32, device=kmer_code.device)
    nbytes = ctypes.c_uint64(kmer_order_scratch_bytes(n, lag))
    scratch = torch.empty(max(int(nbytes.value), 1), dtype=torch.uint8, device=kmer_code.device)
    with torch.cuda.device(kmer_code.device):
        st = _lib.lib().bear_kmer_order_u64(_ptr(kmer_code), n, int(lag), _ptr(perm), _ptr(scratch), ctypes.byref(nbytes), _stream())
    _lib.check(st, "bear_kmer_order_u64")
    return perm


def kmer_order_scratch_bytes(n_rows, lag):
    """Device scratch ``bear_kmer_order_u64`` asks of its caller for a batch of n_rows contexts."""
    nbytes = ctypes.c_uint64(0)
    _lib.check(_lib.lib().bear_kmer_order_u64(None, int(n_rows), int(lag), None, None, ctypes.byref(nbytes), None), "bear_kmer_order_u64")
    return int(nbytes.value)


def gather_rows(src, perm):
    """One launch of ``bear_gather_rows``: a new tensor with ``dst[i] = src[perm[i]]`` along dim 0 (perm: int32 storage)."""
    if not (src.is_cuda and src.is_contiguous() and src.dim() >= 1 and perm.is_cuda and perm.dtype == torch.int32
            and perm.is_contiguous() and perm.shape == (src.shape[0],)):
        raise ValueError("src: contiguous CUDA tensor; perm: contiguous CUDA int32 [n_rows]")
    dst = torch.empty_like(src)
    n = src.shape[0]
    row_bytes = src.element_size() * (src.numel() // n) if n else 1
    with torch.cuda.device(src.device):
        st = _lib.lib().bear_gather_rows(_ptr(src), _ptr(perm), _ptr(dst), n, row_bytes, _stream())
    _lib.check(st, "bear_gather_rows")
    return dst


def shuffle_source_row(i, n_rows, seed):
    """Host evaluation of the permutation: the source row of shuffled row i."""
    return int(_lib.lib().bear_shuffle_source_row(int(i), int(n_rows), int(seed) & (2 ** 64 - 1)))


CNN_NUM_FILTERS, CNN_LAYER1_WIDTH, CNN_MAX_LAG = 30, 16, 21   # CNN_NF / CNN_L1 / CNN_MAX_LAG of kernels_cnn.h


def cnn_supported(lag, alphabet_size, filter_width, num_filters, kmer_layer1_width):
    return (alphabet_size == 4 and num_filters == CNN_NUM_FILTERS and kmer_layer1_width == CNN_LAYER1_WIDTH
            and 1 <= filter_width <= lag <= CNN_MAX_LAG)


def cnn_param_count(lag, filter_width):
    n = _lib.lib().bear_cnn_param_count(int(lag), int(filter_width), CNN_NUM_FILTERS, CNN_LAYER1_WIDTH)
    _lib.check(min(n, 0), "bear_cnn_param_count")
    return n


def _check_codes(kmer_code):
    if not (kmer_code.is_cuda and kmer_code.dtype == torch.int64 and kmer_code.dim() == 1 and kmer_code.is_contiguous()):
        raise ValueError("kmer_code must be a contiguous CUDA int64 tensor [n_rows] (pack_kmers)")


def cnn_forward(kmer_code, flat_params, lag, filter_width, save=True, ws=None, plan=None):
    """One launch of ``bear_cnn_forward_f64``: (prior [n,5], t1 [n,16] or None).  ``plan``: a plan of the same rows -- the forward
    pass then runs over its prefix levels when they were attached for this tensor (``bear_cnn_forward_plan_f64``; t1 is kept)."""
    _check_codes(kmer_code)
    n = kmer_code.shape[0]
    if not (flat_params.is_cuda and flat_params.dtype == torch.float64 and flat_params.is_contiguous()
            and flat_params.numel() == cnn_param_count(lag, filter_width)):
        raise ValueError("flat_params must be the contiguous CUDA float64 parameter vector of bear_cnn_param_count elements")
    if plan is not None:
        if plan.counts.shape[0] != n:
            raise ValueError("plan: built for another number of rows than kmer_code holds")
        prior = torch.empty((n, 5), dtype=torch.float64, device=kmer_code.device)
        t1 = torch.empty((n, CNN_LAYER1_WIDTH), dtype=torch.float64, device=kmer_code.device)
        with torch.cuda.device(kmer_code.device):
            st = _lib.lib().bear_cnn_forward_plan_f64(plan.ws.handle, plan._h, _ptr(kmer_code), n, int(lag), int(filter_width), CNN_NUM_FILTERS,
                                                      CNN_LAYER1_WIDTH, _ptr(flat_params), _ptr(prior), _ptr(t1), _stream())
        _lib.check(st, "bear_cnn_forward_plan_f64")
        return prior, t1
    ws = ws or default_workspace(kmer_code.device)
    prior = torch.empty((n, 5), dtype=torch.float64, device=kmer_code.device)
    t1 = torch.empty((n, CNN_LAYER1_WIDTH), dtype=torch.float64, device=kmer_code.device) if save else None
    with torch.cuda.device(kmer_code.device):
        st = _lib.lib().bear_cnn_forward_f64(ws.handle, _ptr(kmer_code), n, int(lag), int(filter_width), CNN_NUM_FILTERS,
                                             CNN_LAYER1_WIDTH, _ptr(flat_params), _ptr(prior), _ptr(t1), _stream())
    _lib.check(st, "bear_cnn_forward_f64")
    return prior, t1


def cnn_backward(kmer_code, flat_params, lag, filter_width, t1, prior, grad_prior, ws=None):
    """One launch of ``bear_cnn_backward_f64``: d L / d flat_params."""
    _check_codes(kmer_code)
    n = kmer_code.shape[0]
    for t, w in ((t1, CNN_LAYER1_WIDTH), (prior, 5), (grad_prior, 5)):
        if not (t.is_cuda and t.dtype == torch.float64 and t.is_contiguous() and t.shape == (n, w)):
            raise ValueError("t1 [n,16], prior [n,5] and grad_prior [n,5] must be contiguous CUDA float64 tensors")
    ws = ws or default_workspace(kmer_code.device)
    grad = torch.empty_like(flat_params)
    with torch.cuda.device(kmer_code.device):
        st = _lib.lib().bear_cnn_backward_f64(ws.handle, _ptr(kmer_code), n, int(lag), int(filter_width), CNN_NUM_FILTERS,
                                              CNN_LAYER1_WIDTH, _ptr(flat_params), _ptr(t1), _ptr(prior), _ptr(grad_prior),
                                              _ptr(grad), _stream())
    _lib.check(st, "bear_cnn_backward_f64")
    return grad


LINEAR_MAX_LAG = 21


def linear_supported(lag, alphabet_size):
    return alphabet_size == 4 and 1 <= lag <= LINEAR_MAX_LAG


def _check_mat(mat, lag):
    if not (mat.is_cuda and mat.dtype == torch.float64 and mat.is_contiguous() and tuple(mat.shape) == (lag, 5, 5)):
        raise ValueError("mat must be a contiguous CUDA float64 tensor [lag, 5, 5]")


def linear_forward(kmer_code, mat, lag, ws=None):
    """One launch of ``bear_linear_forward_f64``: the rows softmax(sum_l mat[l, kmer[l], :]) [n, 5] of packed contexts."""
    _check_codes(kmer_code)
    _check_mat(mat, lag)
    n = kmer_code.shape[0]
    ws = ws or default_workspace(kmer_code.device)
    prior = torch.empty((n, 5), dtype=torch.float64, device=kmer_code.device)
    with torch.cuda.device(kmer_code.device):
        st = _lib.lib().bear_linear_forward_f64(ws.handle, _ptr(kmer_code), n, int(lag), _ptr(mat), _ptr(prior), _stream())
    _lib.check(st, "bear_linear_forward_f64")
    return prior


def linear_backward(kmer_code, lag, prior, grad_prior, ws=None):
    """One launch of ``bear_linear_backward_f64``: d L / d mat [lag, 5, 5] from the forward rows and d L / d prior."""
    _check_codes(kmer_code)
    n = kmer_code.shape[0]
    for t in (prior, grad_prior):
        if not (t.is_cuda and t.dtype == torch.float64 and t.is_contiguous() and tuple(t.shape) == (n, 5)):
            raise ValueError("prior and grad_prior must be contiguous CUDA float64 tensors [n, 5]")
    ws = ws or default_workspace(kmer_code.device)
    grad = torch.empty((lag, 5, 5), dtype=torch.float64, device=kmer_code.device)
    with torch.cuda.device(kmer_code.device):
        st = _lib.lib().bear_linear_backward_f64(ws.handle, _ptr(kmer_code), n, int(lag), _ptr(prior), _ptr(grad_prior), _ptr(grad),
                                                 _stream())
    _lib.check(st, "bear_linear_backward_f64")
    return grad


def _check_rows5(n, **tensors):
    for name, t in tensors.items():
        if not (t.is_cuda and t.dtype == torch.float64 and t.is_contiguous() and tuple(t.shape) == (n, 5) and t.data_ptr() % 16 == 0):
            raise ValueError(f"{name} must be a contiguous, 16-byte aligned CUDA float64 tensor [{n}, 5]")


def _check_scalar_param(**tensors):
    for name, t in tensors.items():
        if not (t.is_cuda and t.dtype == torch.float64 and t.numel() == 1):
            raise ValueError(f"{name} must be a CUDA float64 tensor of one element")


def ref_mix_forward(net_rows, ref_rows, tau_signed, net_weight_signed, ws=None):
    """One launch of ``bear_ref_mix_forward_f64``: (nw net_rows + jukes_cantor(ref_rows, tau)) / (nw + 1) (bear_ref.py:63-68)."""
    n = net_rows.shape[0]
    _check_rows5(n, net_rows=net_rows, ref_rows=ref_rows)
    _check_scalar_param(tau_signed=tau_signed, net_weight_signed=net_weight_signed)
    ws = ws or default_workspace(net_rows.device)
    prior = torch.empty_like(net_rows)
    with torch.cuda.device(net_rows.device):
        st = _lib.lib().bear_ref_mix_forward_f64(ws.handle, _ptr(net_rows), _ptr(ref_rows), n, _ptr(tau_signed), _ptr(net_weight_signed),
                                                 _ptr(prior), _stream())
    _lib.check(st, "bear_ref_mix_forward_f64")
    return prior


def ref_mix_backward(net_rows, ref_rows, grad_prior, tau_signed, net_weight_signed, ws=None):
    """One launch of ``bear_ref_mix_backward_f64``: (d L / d net_rows [n, 5], [d L / d tau_signed, d L / d net_weight_signed])."""
    n = net_rows.shape[0]
    _check_rows5(n, net_rows=net_rows, ref_rows=ref_rows, grad_prior=grad_prior)
    _check_scalar_param(tau_signed=tau_signed, net_weight_signed=net_weight_signed)
    ws = ws or default_workspace(net_rows.device)
    grad_rows = torch.empty_like(net_rows)
    scalars = torch.empty(2, dtype=torch.float64, device=net_rows.device)
    with torch.cuda.device(net_rows.device):
        st = _lib.lib().bear_ref_mix_backward_f64(ws.handle, _ptr(net_rows), _ptr(ref_rows), _ptr(grad_prior), n, _ptr(tau_signed),
                                                  _ptr(net_weight_signed), _ptr(grad_rows), _ptr(scalars), _stream())
    _lib.check(st, "bear_ref_mix_backward_f64")
    return grad_rows, scalars


def stream_read(t, ws=None):
    """One launch of ``bear_stream_read`` over tensor ``t`` (measurement helper: a pure HBM read)."""
    ws = ws or default_workspace(t.device)
    with torch.cuda.device(t.device):
        st = _lib.lib().bear_stream_read(ws.handle, _ptr(t), t.numel() * t.element_size(), _stream())
    _lib.check(st, "bear_stream_read")


def encode_kmers(ascii_kmers, alphabet="dna"):
    """One launch of ``bear_encode_kmers_i8``: device uint8 [n, lag] ASCII k-mers -> int8 letter codes [n, lag]
    (the device twin of ``core.encode_kmers``)."""
    if not (ascii_kmers.is_cuda and ascii_kmers.dtype == torch.uint8 and ascii_kmers.dim() == 2 and ascii_kmers.is_contiguous()):
        raise ValueError("ascii_kmers must be a contiguous CUDA uint8 tensor [n, lag]")
    if alphabet not in ("dna", "rna"):
        raise NotImplementedError("device encoding covers the 4-letter alphabets")
    codes = torch.empty(ascii_kmers.shape, dtype=torch.int8, device=ascii_kmers.device)
    with torch.cuda.device(ascii_kmers.device):
        st = _lib.lib().bear_encode_kmers_i8(_ptr(ascii_kmers), ascii_kmers.shape[0], ascii_kmers.shape[1], int(alphabet == "rna"),
                                             _ptr(codes), _stream())
    _lib.check(st, "bear_encode_kmers_i8")
    return codes


def _f64_vec(t, n, name):
    if not (t.is_cuda and t.dtype == torch.float64 and t.is_contiguous() and t.numel() == n):
        raise ValueError(f"{name} must be a contiguous CUDA float64 tensor of {n} elements")


def dm_prior_planned_dev(plan, prior, h_signed_dev, eps=EPSILON, out=None, normalized=False, want_grad=False, train_ar=False):
    """``bear_dm_prior_plan_dev_f64``: dm_prior_planned with h_signed read from a device tensor, so a training step is
    enqueued without the host reading the parameter back.  Returns out, or (out, grad rows) with want_grad."""
    counts = plan.counts
    _check_rows(prior, torch.float64, "prior")
    if prior.data_ptr() % 16 or prior.shape[0] != counts.shape[0] or plan.ncol != 5:
        raise ValueError("prior must be 16-byte aligned with one row per planned context (plan ncol=5)")
    _f64_vec(h_signed_dev, 1, "h_signed_dev")
    if out is None:
        out = torch.empty(2, dtype=torch.float64, device=counts.device)
    grad = torch.empty_like(prior) if want_grad else None
    with torch.cuda.device(counts.device):
        st = _lib.lib().bear_dm_prior_plan_dev_f64(plan.ws.handle, plan._h, _ptr(counts), _ptr(prior), counts.shape[0], _ptr(h_signed_dev),
                                                   float(eps), int(bool(train_ar)), int(bool(normalized)), _ptr(out), _ptr(grad), _stream())
    _lib.check(st, "bear_dm_prior_plan_dev_f64")
    return (out, grad) if want_grad else out


def dm_refmix_planned_dev(plan, net_rows, ref_rows, h_signed_dev, tau_signed_dev, net_weight_signed_dev, eps=EPSILON, out=None,
                          train_ar=False):
    """``bear_dm_refmix_plan_grad_f64``: bear_ref's step for a net function with parameters, the reference mixing inside the DM step.
    Returns (out [4] = sum LL, d/dh_signed, d/dtau_signed, d/dnet_weight_signed; d sum LL / d net_rows [n, 5])."""
    counts = plan.counts
    n = counts.shape[0]
    _check_rows5(n, net_rows=net_rows, ref_rows=ref_rows)
    if plan.ncol != 5:
        raise ValueError("a five-column plan of the training counts is needed")
    _check_scalar_param(h_signed_dev=h_signed_dev, tau_signed_dev=tau_signed_dev, net_weight_signed_dev=net_weight_signed_dev)
    if out is None:
        out = torch.empty(4, dtype=torch.float64, device=counts.device)
    _f64_vec(out, 4, "out")
    grad = torch.empty_like(net_rows)
    with torch.cuda.device(counts.device):
        st = _lib.lib().bear_dm_refmix_plan_grad_f64(plan.ws.handle, plan._h, _ptr(counts), _ptr(net_rows), _ptr(ref_rows), n,
                                                     _ptr(h_signed_dev), _ptr(tau_signed_dev), _ptr(net_weight_signed_dev), float(eps),
                                                     int(bool(train_ar)), _ptr(out), _ptr(grad), _stream())
    _lib.check(st, "bear_dm_refmix_plan_grad_f64")
    return out, grad


def train_apply(theta, packed, adam_m, adam_v, adam_t, learning_rate, scale, loss_buf=None, train_ar=False):
    """Enqueues ``bear_train_apply_f64``: tf.keras Adam on ``theta`` with the gradients ``scale * packed[1:]``;
    ``loss_buf[step] = -scale * packed[0]``.  packed = [sum LL, d/d theta...] (after the all-reduce when rows are sharded)."""
    n = theta.numel()
    for t, k, name in ((theta, n, "theta"), (packed, n + 1, "packed"), (adam_m, n, "adam_m"), (adam_v, n, "adam_v"), (adam_t, 1, "adam_t")):
        _f64_vec(t, k, name)
    with torch.cuda.device(theta.device):
        st = _lib.lib().bear_train_apply_f64(_ptr(theta), n, _ptr(packed), _ptr(adam_m), _ptr(adam_v), _ptr(adam_t), float(learning_rate),
                                             float(scale), int(bool(train_ar)), _ptr(loss_buf),
                                             0 if loss_buf is None else loss_buf.numel(), _stream())
    _lib.check(st, "bear_train_apply_f64")


def ref_train_reduce(plan, ref, theta, packed, eps=EPSILON, train_ar=False):
    """Enqueues ``bear_ref_train_reduce_f64``: this shard's packed = [sum LL, d/dh_s, d/dtau_s, d/dnu_s] with the kernel constants
    derived from the device-resident theta (no host round trip)."""
    train = plan.counts
    _check_rows(ref, torch.int32, "ref")
    _f64_vec(theta, 3, "theta")
    _f64_vec(packed, 4, "packed")
    if ref.data_ptr() % 16 or ref.shape[0] != train.shape[0] or plan.ncol != 4:
        raise ValueError("ref must be 16-byte aligned with one row per planned context (plan ncol=4)")
    with torch.cuda.device(train.device):
        st = _lib.lib().bear_ref_train_reduce_f64(plan.ws.handle, plan._h, _ptr(train), _ptr(ref), train.shape[0], _ptr(theta), float(eps),
                                                  int(bool(train_ar)), _ptr(packed), _stream())
    _lib.check(st, "bear_ref_train_reduce_f64")


def ref_train_step(plan, ref, theta, adam_m, adam_v, adam_t, learning_rate, scale, out, loss_buf=None, eps=EPSILON, train_ar=False):
    """Enqueues one ``bear_ref_train_step_f64`` (constants from theta, planned mode-R kernel, finalize, Adam on theta): no host
    synchronisation, every argument device-resident -- capturable in a HIP graph (``torch.cuda.graph``)."""
    train = plan.counts
    _check_rows(ref, torch.int32, "ref")
    for t, n in ((theta, 3), (adam_m, 3), (adam_v, 3), (adam_t, 1), (out, 4)):
        if not (t.is_cuda and t.dtype == torch.float64 and t.is_contiguous() and t.numel() == n):
            raise ValueError("theta / adam_m / adam_v [3], adam_t [1], out [4] must be contiguous CUDA float64 tensors")
    if ref.data_ptr() % 16 or ref.shape[0] != train.shape[0] or plan.ncol != 4:
        raise ValueError("ref must be 16-byte aligned with one row per planned context (plan ncol=4)")
    with torch.cuda.device(train.device):
        st = _lib.lib().bear_ref_train_step_f64(plan.ws.handle, plan._h, _ptr(train), _ptr(ref), train.shape[0], _ptr(theta), _ptr(adam_m),
                                                _ptr(adam_v), _ptr(adam_t), float(eps), int(bool(train_ar)), float(learning_rate),
                                                float(scale), _ptr(out), _ptr(loss_buf), 0 if loss_buf is None else loss_buf.numel(),
                                                _stream())
    _lib.check(st, "bear_ref_train_step_f64")


def _check_linear_step(plan, kmer_code, lag, theta, packed):
    n = plan.counts.shape[0]
    _f64_vec(theta, 1 + lag * 25, "theta")
    _f64_vec(packed, 2 + lag * 25, "packed")
    if not (kmer_code.is_cuda and kmer_code.dtype == torch.int64 and kmer_code.is_contiguous() and kmer_code.shape == (n,)
            and kmer_code.data_ptr() % 16 == 0):
        raise ValueError("kmer_index must be a contiguous, 16-byte aligned CUDA int64 tensor [n_rows] (linear_index)")
    return n


def net_linear_train_reduce(plan, kmer_code, lag, theta, packed, eps=EPSILON, train_ar=False):
    """Enqueues ``bear_net_linear_train_reduce_f64``: packed = [sum LL, d/dh_s, d/d mat (lag*25)] of this shard, theta = {h_signed, mat}
    device-resident."""
    n = _check_linear_step(plan, kmer_code, lag, theta, packed)
    with torch.cuda.device(theta.device):
        st = _lib.lib().bear_net_linear_train_reduce_f64(plan.ws.handle, plan._h, _ptr(plan.counts), _ptr(kmer_code), int(lag), n, _ptr(theta),
                                                         float(eps), int(bool(train_ar)), _ptr(packed), _stream())
    _lib.check(st, "bear_net_linear_train_reduce_f64")


def net_linear_train_step(plan, kmer_code, lag, theta, adam_m, adam_v, adam_t, packed, learning_rate, scale, loss_buf=None,
                          eps=EPSILON, train_ar=False):
    """Enqueues one ``bear_net_linear_train_step_f64`` (HIP-graph capturable): theta = {h_signed, mat} on the device."""
    n = _check_linear_step(plan, kmer_code, lag, theta, packed)
    size = 1 + lag * 25
    for t, k in ((adam_m, size), (adam_v, size), (adam_t, 1)):
        _f64_vec(t, k, "adam state")
    with torch.cuda.device(theta.device):
        st = _lib.lib().bear_net_linear_train_step_f64(plan.ws.handle, plan._h, _ptr(plan.counts), _ptr(kmer_code), int(lag), n, _ptr(theta),
                                                       _ptr(adam_m), _ptr(adam_v), _ptr(adam_t), _ptr(packed), float(eps),
                                                       int(bool(train_ar)), float(learning_rate), float(scale),
                                                       _ptr(loss_buf), 0 if loss_buf is None else loss_buf.numel(), _stream())
    _lib.check(st, "bear_net_linear_train_step_f64")


def net_cnn_train_reduce(plan, kmer_code, lag, filter_width, theta, bufs, packed, eps=EPSILON, train_ar=False):
    """Enqueues ``bear_net_cnn_train_reduce_f64``: forward, planned DM kernel with gradient rows, backward;
    packed = [sum LL, d/dh_s, d/d params] of this shard.  ``bufs`` = (prior [n,5], t1 [n,16], grad_rows [n,5]) from ``cnn_step_buffers``."""
    n = plan.counts.shape[0]
    prior, t1, grad_rows = bufs
    np_ = cnn_param_count(lag, filter_width)
    _f64_vec(theta, 1 + np_, "theta")
    _f64_vec(packed, 2 + np_, "packed")
    with torch.cuda.device(theta.device):
        st = _lib.lib().bear_net_cnn_train_reduce_f64(plan.ws.handle, plan._h, _ptr(plan.counts), _ptr(kmer_code), n, int(lag), int(filter_width),
                                                      CNN_NUM_FILTERS, CNN_LAYER1_WIDTH, _ptr(theta), _ptr(prior), _ptr(t1), _ptr(grad_rows),
                                                      float(eps), int(bool(train_ar)), _ptr(packed), _stream())
    _lib.check(st, "bear_net_cnn_train_reduce_f64")


def net_cnn_train_step(plan, kmer_code, lag, filter_width, theta, adam_m, adam_v, adam_t, bufs, packed, learning_rate, scale, loss_buf=None,
                       eps=EPSILON, train_ar=False):
    """Enqueues one ``bear_net_cnn_train_step_f64`` (HIP-graph capturable): theta = {h_signed, flat CNN parameters} on the device;
    ``bufs`` = (prior [n,5], t1 [n,16], grad_rows [n,5]) lent by the caller (``cnn_step_buffers``)."""
    n = plan.counts.shape[0]
    prior, t1, grad_rows = bufs
    np_ = cnn_param_count(lag, filter_width)
    _f64_vec(theta, 1 + np_, "theta")
    _f64_vec(packed, 2 + np_, "packed")
    with torch.cuda.device(theta.device):
        st = _lib.lib().bear_net_cnn_train_step_f64(plan.ws.handle, plan._h, _ptr(plan.counts), _ptr(kmer_code), n, int(lag), int(filter_width),
                                                    CNN_NUM_FILTERS, CNN_LAYER1_WIDTH, _ptr(theta), _ptr(adam_m), _ptr(adam_v), _ptr(adam_t),
                                                    _ptr(prior), _ptr(t1), _ptr(grad_rows), _ptr(packed), float(eps), int(bool(train_ar)),
                                                    float(learning_rate), float(scale), _ptr(loss_buf),
                                                    0 if loss_buf is None else loss_buf.numel(), _stream())
    _lib.check(st, "bear_net_cnn_train_step_f64")


def cnn_step_buffers(n_rows, lag, filter_width, device, ws=None):
    """Per-context scratch of the CNN step for (at most) ``n_rows`` contexts + the library-side reservation (``bear_cnn_reserve``):
    (prior [n,5], t1 [n,16], grad_rows [n,5]).  Steps on one stream run one after the other, so one set sized for the largest
    batch serves every batch."""
    ws = ws or default_workspace(device)
    with torch.cuda.device(device):
        _lib.check(_lib.lib().bear_cnn_reserve(ws.handle, int(n_rows), int(lag), int(filter_width), CNN_NUM_FILTERS, CNN_LAYER1_WIDTH),
                   "bear_cnn_reserve")
    # zeros: the step only fills the rows of contexts that hold training counts (nothing reads the others: kept finite)
    return (torch.zeros((n_rows, 5), dtype=torch.float64, device=device), torch.zeros((n_rows, CNN_LAYER1_WIDTH), dtype=torch.float64, device=device),
            torch.zeros((n_rows, 5), dtype=torch.float64, device=device))
