"""Shared machinery of the bear_net / bear_ref drivers: resident shards, Keras-equivalent Adam,
the per-step reduce, and the held-out evaluation built on the DM kernels."""
import math
import os
import warnings

import numpy as np
import torch

from . import _lib, core, dist, kernels
from .dataloader import CountDataset

epsilon = core.epsilon


class KerasAdam:
    """tf.keras.optimizers.Adam defaults (beta 0.9 / 0.999, epsilon 1e-7) with Keras' update
    ``theta -= lr_t * m / (sqrt(v) + eps)``, ``lr_t = lr sqrt(1 - b2^t) / (1 - b1^t)``
    (the reference builds it by name, bear_net.py:264-265)."""

    def __init__(self, params, learning_rate, beta_1=0.9, beta_2=0.999, eps=1e-7):
        self.params = list(params)
        self.lr, self.b1, self.b2, self.eps = float(learning_rate), beta_1, beta_2, eps
        self.m = [torch.zeros_like(p) for p in self.params]
        self.v = [torch.zeros_like(p) for p in self.params]
        self.t = 0

    @torch.no_grad()
    def apply_gradients(self, grads):
        self.t += 1
        lr_t = self.lr * math.sqrt(1.0 - self.b2 ** self.t) / (1.0 - self.b1 ** self.t)
        for p, g, m, v in zip(self.params, grads, self.m, self.v):
            if g is None:
                continue
            g = g.to(p.dtype).to(p.device)
            m.mul_(self.b1).add_(g, alpha=1.0 - self.b1)
            v.mul_(self.b2).addcmul_(g, g, value=1.0 - self.b2)
            p.sub_(lr_t * m / (v.sqrt() + self.eps))


# tf.keras optimizers reachable by name (bear_net.py:264-265 `getattr(tf.keras.optimizers, optimizer_name)(learning_rate)`) whose
# update rule torch.optim reproduces once Keras' defaults are passed explicitly (epsilon 1e-7 everywhere, RMSprop rho 0.9,
# Adagrad initial accumulator 0.1, Adadelta rho 0.95).  Adam has its own class (Keras puts epsilon outside the bias correction).
# RMSprop: with Keras' default momentum = 0 the optimizer_v2 update is `var -= lr g / (sqrt(rms) + epsilon)` (epsilon OUTSIDE the
# root; only the fused kernel taken for momentum > 0 puts it inside) -- torch.optim.RMSprop's rule.  The reference never passes a
# momentum (it builds the optimizer from its name and the learning rate alone), so no other case exists here.
_KERAS_AS_TORCH = {
    "SGD": (torch.optim.SGD, {}),
    "RMSprop": (torch.optim.RMSprop, {"alpha": 0.9, "eps": 1e-7, "momentum": 0.0, "centered": False}),
    "Adagrad": (torch.optim.Adagrad, {"initial_accumulator_value": 0.1, "eps": 1e-7}),
    "Adadelta": (torch.optim.Adadelta, {"rho": 0.95, "eps": 1e-7}),
}


def make_optimizer(name, params, learning_rate):
    if name == "Adam":
        return KerasAdam(params, learning_rate)
    if name not in _KERAS_AS_TORCH:
        raise ValueError(f"optimizer_name {name!r}: supported are Adam, " + ", ".join(sorted(_KERAS_AS_TORCH))
                         + " (tf.keras update rules with Keras defaults)")
    cls, kw = _KERAS_AS_TORCH[name]
    params = list(params)
    opt = cls(params, lr=learning_rate, **kw)

    class _Wrap:
        def apply_gradients(self, grads):
            for p, g in zip(params, grads):
                p.grad = None if g is None else g.to(p.dtype).to(p.device)
            opt.step()
    return _Wrap()


def require_device():
    if not torch.cuda.is_available():
        raise RuntimeError("bear_amd trains on MI355X only (libbear_hip.so has no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


class Uploader:
    """Host arrays -> HBM through a ring of reusable PINNED staging buffers on a side stream (the reference's input pipeline
    keeps `prefetch(10)` batches in flight while a step runs, bear_net.py:268-273).  ``put`` copies a contiguous NumPy block
    into a pinned buffer piece by piece (host memcpy) and enqueues the asynchronous H2D copy of each piece; the host only
    waits when it needs a staging buffer back, so the DMA of piece k overlaps the memcpy of piece k + 1 -- and whatever the
    compute stream is doing (the previous batch's compaction, sort and plan).  ``wait`` orders the compute stream behind
    everything enqueued so far.  Pageable `.to(device)` copies (round 2) ran at a fraction of the PCIe rate and blocked."""

    PIECE = 64 << 20      # bytes per staging buffer (16 MiB pieces: 39 instead of 45 GB/s)
    COPY_THREADS = 4      # host threads that fill a staging buffer (one thread's memcpy, ~20 GB/s, is less than half of what PCIe takes)
    _copiers = None
    _ring = {}            # device index -> (piece bytes, pinned buffers): page-locking 192 MiB takes longer than a small table's
                          # whole run, so the ring is kept for the process and sized by what has been asked for so far

    def __init__(self, device, n_buffers=3, expect_bytes=None):
        self.device = device
        self.consumer = torch.cuda.current_stream(device)     # the stream the uploaded tensors will be used on
        self.stream = torch.cuda.Stream(device)
        piece = self.PIECE if expect_bytes is None else max(1 << 16, min(self.PIECE, -(-int(expect_bytes) // 4096) * 4096))
        key = (device.index, n_buffers)
        have = Uploader._ring.get(key)
        if have is None or have[0] < piece:
            have = (piece, [torch.empty(piece, dtype=torch.uint8, pin_memory=True) for _ in range(n_buffers)])
            Uploader._ring[key] = have
        self.piece = min(have[0], self.PIECE)
        self.bufs = have[1]
        self.views = [b.numpy() for b in self.bufs]
        self.free_at = [None] * n_buffers          # event after which a staging buffer may be overwritten
        self.k = 0
        self.bytes = 0

    def put(self, array, dtype):
        """Device tensor of ``dtype`` with the shape (and bytes) of the C-contiguous NumPy ``array``; valid after ``wait``."""
        array = np.ascontiguousarray(array)
        with torch.cuda.device(self.device), torch.cuda.stream(self.stream):
            dst = torch.empty(array.shape, dtype=dtype, device=self.device)   # the side stream's own block: no wait for the consumer
        dst.record_stream(self.consumer)
        if dst.element_size() != array.dtype.itemsize:
            raise ValueError("Uploader.put: dtype sizes differ")
        nbytes = array.nbytes
        if nbytes == 0:
            return dst
        src = array.reshape(-1).view(np.uint8)
        dst_bytes = dst.view(-1).view(torch.uint8)
        for off in range(0, nbytes, self.piece):
            n = min(self.piece, nbytes - off)
            b = self.k % len(self.bufs)
            if self.free_at[b] is not None:
                self.free_at[b].synchronize()
            self._fill(self.views[b], src, off, n)
            with torch.cuda.device(self.device), torch.cuda.stream(self.stream):
                dst_bytes[off:off + n].copy_(self.bufs[b][:n], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(self.stream)
            self.free_at[b] = ev
            self.k += 1
        self.bytes += nbytes
        return dst

    @classmethod
    def _fill(cls, view, src, off, n):
        """view[:n] = src[off:off + n], large pieces cut over COPY_THREADS host threads (NumPy's copy releases the GIL)."""
        if n < (8 << 20) or cls.COPY_THREADS <= 1:
            np.copyto(view[:n], src[off:off + n])
            return
        if cls._copiers is None:
            import concurrent.futures
            cls._copiers = concurrent.futures.ThreadPoolExecutor(cls.COPY_THREADS)
        step = -(-n // cls.COPY_THREADS)
        step += (-step) % 4096
        jobs = [cls._copiers.submit(np.copyto, view[a:min(a + step, n)], src[off + a:off + min(a + step, n)]) for a in range(0, n, step)]
        for j in jobs:
            j.result()

    def wait(self):
        self.consumer.wait_stream(self.stream)


class _Ready:
    def __init__(self, value):
        self._value = value

    def result(self):
        return self._value


def hbm_budget_check(data, n_columns, want_codes, device, rows=None, per_row_extra=0):
    """Raises before the first byte goes up when this rank's share of the epoch cannot stay resident: every batch is kept in HBM
    for all epochs (counts 20 B per column and context, k-mer letters, packed codes, ~7 B of plan, per-context scratch of a
    fused AR function), and a table that does not fit is a matter of more ranks (rows shard: `python -m torch.distributed.run
    --nproc-per-node N ...`), not of a slower path."""
    rows = data.local_rows if rows is None else rows
    lag = data.lag
    need = rows * (20 * n_columns + (lag + 8 if want_codes else 0) + 8 + per_row_extra)
    # (the k-mer sort's scratch, ~20 B per row of ONE batch at a time, is a torch tensor and fits the 15 % margin below)
    need = int(need * 1.15) + (256 << 20)           # the upload holds a column next to its compacted copy for a moment
    # blocks the caching allocator holds but nobody uses (an earlier train() call's slabs, allocated on another Uploader's side
    # stream and so not reusable from this one's pool) are given back first: mem_get_info does not count them as free
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    free, total = torch.cuda.mem_get_info(device)
    if need > free:
        raise MemoryError(f"this rank's {rows} contexts need about {need / 2**30:.1f} GiB of HBM resident ({n_columns} count column(s)"
                          f"{', k-mer codes' if want_codes else ''}, plans), {free / 2**30:.1f} of {total / 2**30:.1f} GiB are free: "
                          "shard the rows over more GPUs (python -m torch.distributed.run --nproc-per-node N ...)")
    return need


def sort_by_kmer(codes, lag):
    """Permutation (int32 storage) that orders int8 letter codes [n, lag] lexicographically -- first letter most significant,
    unknown letters last, equal k-mers in their given order: a device radix sort of the packed contexts (``bear_kmer_order_u64``)."""
    return kernels.kmer_order(kernels.pack_kmers(codes.contiguous()), lag)


class ResidentBatches:
    """This rank's row shard of every batch of one epoch, uploaded once: per batch the device slabs of the
    requested dataset columns, the k-mer codes, and (lazily) the kernel plans.

    STREAMING (``self.streaming``; the reference's ``tf.data`` pipeline with ``cache=False``, dataloader.py:36-50): when this
    rank's share of the epoch does not fit its HBM -- or BEAR_AMD_STREAM=1 asks for it -- only a window of batches is on the
    device: ``load(k)`` makes batch k resident (upload through the pinned ring, compaction, k-mer order, plans: exactly what a
    resident batch gets), drops every other batch and starts the upload of the batch after it on the side stream, so that a
    step's kernels run under the next batch's PCIe transfer.  ``self.batches[k]`` is then the same dict object for the whole
    run -- meta data only (``global_rows``, ``row0``, the uploaded row count) while the batch is not loaded -- and every
    epoch re-uploads every batch: such a run is bound by PCIe (~45 GB/s), not by the kernels; more ranks are the fast way."""

    def __init__(self, data, columns, device, want_codes=False, drop_empty=None, kmer_order=False, prebuild=(), per_row_extra=0,
                 stream=None):
        """``drop_empty``: name of the column a training run fits, or an evaluation scores.  A context without counts in it adds
        exactly nothing to the ELBO or to any gradient (``D(x, 0) = 0``, core.py:73-74) -- nor to any of the seven sums of
        ``_evaluation_step`` (bear_net.py:323-371: every term carries a held-out count as a factor) -- so its row is left out of the
        resident batch: 30 % of the rows of a typical count table for training, half of them for a held-out column; the loss
        scale keeps the batch's full size (``global_rows``) and ``row_ids`` remembers where each kept row sits in the table
        (the evaluation's tie-breaking noise is keyed by it).  BEAR_AMD_ALL_ROWS=1 keeps every row (tests).

        ``kmer_order``: the rows of every batch are sorted by k-mer (the sums of a step do not depend on the order; the fused
        AR-function kernels share work between neighbouring contexts).  ``prebuild``: plan keys ``(column, ncol, ref_column)``
        cut for every batch as soon as it has landed -- while the next batch is still crossing PCIe (``Uploader``)."""
        if not isinstance(data, CountDataset):
            raise TypeError("train / evaluation expect the CountDataset returned by bear_amd.dataloader")
        self.data, self.device = data, device
        self.batches = []
        # k-mer letter codes: the ASCII bytes go up as they were parsed and are encoded on the device
        fast_codes = want_codes and data.alphabet in ("dna", "rna") and data.lag > 0
        on_dev = getattr(data, "counts_dev", None) is not None       # DeviceCountDataset: the table is in HBM already
        codes = data.codes() if (want_codes and not fast_codes) else None
        rank, world = dist.world()
        pieces = list(zip(data.batch_bounds(), data.rank_pieces(rank, world)))
        self.streaming = False
        self.loads = 0                      # streaming: batches made resident so far
        if not on_dev:
            if stream is None:
                stream = bool(os.environ.get("BEAR_AMD_STREAM")) and os.environ.get("BEAR_AMD_STREAM") != "0"
            largest = max([g1 - g0 for _, (g0, g1, _) in pieces] + [0])
            # 0 = the epoch stays resident, 1 = streamed, 2 = neither works.  The RANKS AGREE on it (one MAX all-reduce): a rank
            # that streams runs an eager loop, a resident one captures a graph and issues a warm-up all-reduce first -- with
            # different decisions the ranks' collectives would pair up wrongly (and the job hang at its end)
            status, why = (1 if stream else 0), None
            if not stream:
                try:
                    hbm_budget_check(data, len(columns), want_codes, device, rows=sum(g1 - g0 for _, (g0, g1, _) in pieces),
                                     per_row_extra=per_row_extra)
                except MemoryError as err:
                    # (a shuffled epoch is permuted on the device as a whole; one batch cannot be streamed around itself)
                    status, why = (2 if data.shuffle_seed is not None or len(pieces) < 2 else 1), str(err)
            if status == 1 and data.shuffle_seed is None:
                # the window: the batch in use, the batch landing, and the blocks of the batch just dropped (free for the side
                # stream only once the steps that read them have run)
                try:
                    hbm_budget_check(data, len(columns), want_codes, device, rows=3 * largest, per_row_extra=per_row_extra)
                except MemoryError as err:
                    status, why = 2, str(err)
            agreed = dist.agree_max(status, device)
            if agreed == 2:
                raise MemoryError(why or "another rank's share of the epoch fits its HBM neither resident nor streamed: every rank stops")
            if agreed == 1:
                if data.shuffle_seed is not None:
                    raise ValueError("a streamed epoch cannot be shuffled on the device (the shuffle permutes the whole shard in HBM)")
                if why is not None:
                    warnings.warn(f"{why}  --  streaming the epoch instead: batches are re-uploaded every epoch (PCIe-bound)")
                elif not stream:
                    warnings.warn("another rank has to stream its share of the epoch: this rank streams too (the ranks run the same loop)")
                self.streaming = True
        LAST_RUN["streaming"] = self.streaming
        up = None if on_dev else Uploader(device, expect_bytes=max(
            [(g1 - g0) * 20 for _, (g0, g1, _) in pieces] + [data.local_rows * 20 if data.shuffle_seed is not None else 0, 1]))
        self.upload_bytes = 0

        def device_column(col, lo, hi):
            if on_dev:
                return data.counts_dev[col, lo:hi].to(device).contiguous().clone()
            return up.put(data.counts[col, lo:hi].view(np.int32), torch.int32)

        def device_codes(lo, hi):
            """-> (tensor, needs_encoding)"""
            if fast_codes:
                if on_dev:
                    return data.kmers_dev[lo:hi].to(device).contiguous(), True
                return up.put(data.kmers[lo:hi], torch.uint8), True
            return up.put(codes[lo:hi], torch.int8), False
        row_index = getattr(data, "row_index", None)      # dataloader.KmerDealtDataset
        shuffled = {}
        if data.shuffle_seed is not None and data.num_rows:
            # whole columns go up once, are permuted by one gather pass each (same seed: columns stay aligned), and the
            # batches below are slices of the permuted slabs
            for name, col in columns.items():
                t = device_column(col, 0, data.local_rows)
                if up:
                    up.wait()
                shuffled[name] = kernels.shuffle_rows(t, data.shuffle_seed)
                del t
            if want_codes:
                t, raw = device_codes(0, data.local_rows)
                if up:
                    up.wait()
                shuffled["codes"] = kernels.shuffle_rows(kernels.encode_kmers(t, data.alphabet) if raw else t, data.shuffle_seed)
                del t
        if shuffled and data.shard is not None:
            raise ValueError("a sharded table cannot be shuffled on the device")
        names = list(columns) + (["codes"] if want_codes else [])

        def enqueue(k):
            """Batch k's slabs: asynchronous uploads on the side stream (or slices of the shuffled columns)."""
            (a, b), (g0, g1, off) = pieces[k]
            lo, hi = off, off + (g1 - g0)           # this rank's piece of the batch inside the dataset's arrays
            entry = {"global_rows": b - a, "rows": hi - lo, "row0": g0, "plans": {}}
            if row_index is not None:           # rows dealt by k-mer range: row i of the piece is table row g0 + row_ids[i]
                entry["row_ids"] = up.put(row_index[lo:hi], torch.int32)
            for name, col in columns.items():
                entry[name] = shuffled[name][lo:hi].clone() if shuffled else device_column(col, lo, hi)
            if want_codes:
                if shuffled:
                    entry["codes"] = shuffled["codes"][lo:hi].clone()
                else:
                    entry["codes"], entry["_raw_codes"] = device_codes(lo, hi)
            return entry

        def finish(entry, slot=None):
            """Everything of a landed batch that runs on the compute stream: encode, drop the empty rows, k-mer order, plans.
            ``slot`` (streaming): the batch's place in ``self.batches`` -- its dict is refilled in place."""
            if entry.pop("_raw_codes", False):
                entry["codes"] = kernels.encode_kmers(entry["codes"], data.alphabet)
            if drop_empty and entry["rows"] and not os.environ.get("BEAR_AMD_ALL_ROWS"):
                keep = (entry[drop_empty] != 0).any(dim=1)
                n_keep = int(keep.sum())
                if n_keep < entry["rows"]:
                    idx = keep.nonzero().squeeze(1)
                    for name in names:
                        entry[name] = entry[name].index_select(0, idx).contiguous()
                    entry["rows"] = n_keep
                    # row i of the compacted batch is row row0 + row_ids[i] of the table: the key of the evaluation's tie noise
                    entry["row_ids"] = (entry["row_ids"].index_select(0, idx) if "row_ids" in entry else idx.to(torch.int32)).contiguous()
                del keep
            if kmer_order and want_codes and entry["rows"] > 1:
                order = sort_by_kmer(entry["codes"], data.lag)
                for name in names + (["row_ids"] if "row_ids" in entry else []):
                    entry[name] = kernels.gather_rows(entry[name], order)
                if "row_ids" not in entry:
                    # no row was dropped, but row i is no longer table row row0 + i: the permutation itself says where each row sits
                    # (the evaluation's tie-breaking noise is keyed by the table row, whatever the order or the sharding)
                    entry["row_ids"] = order.to(torch.int32).contiguous()
                del order
            if slot is None:
                self.batches.append(entry)
                slot = len(self.batches) - 1
            else:
                entry["uploaded_rows"], entry["_loaded"] = self.batches[slot]["uploaded_rows"], True
                self.batches[slot].clear()
                self.batches[slot].update(entry)
            if entry["rows"]:
                for column, ncol, ref_column in prebuild:
                    self.plan(slot, column, ncol, ref_column)
        if self.streaming:
            import concurrent.futures
            self._enqueue, self._finish, self._up = enqueue, finish, up
            self._pool = concurrent.futures.ThreadPoolExecutor(1)
            self._pending = None            # (batch, future of its enqueued upload)
            for (a, b), (g0, g1, _) in pieces:
                self.batches.append({"global_rows": b - a, "rows": g1 - g0, "uploaded_rows": g1 - g0, "row0": g0, "plans": {}, "_loaded": False})
            return

        # one batch in flight: a worker thread feeds batch k + 1 through the staging ring (memcpy and waits release the GIL)
        # while this thread compacts, sorts and plans batch k on the compute stream
        import concurrent.futures
        pool = concurrent.futures.ThreadPoolExecutor(1) if (up and len(pieces) > 1 and not shuffled) else None
        try:
            submit = (lambda k: pool.submit(enqueue, k)) if pool else (lambda k: _Ready(enqueue(k)))
            pending = submit(0) if pieces else None
            for k in range(len(pieces)):
                landed = pending.result()
                if up:
                    up.wait()                              # the compute stream waits for batch k's copies (not the host)
                pending = submit(k + 1) if k + 1 < len(pieces) else None
                finish(landed)
        finally:
            if pool:
                pool.shutdown(wait=True)
        del shuffled
        if up:
            self.upload_bytes = up.bytes
            torch.cuda.current_stream(device).synchronize()    # the staging buffers go away with `up`

    def load(self, k):
        """Batch k's entry with its slabs on the device.  Resident epochs: ``self.batches[k]``.  Streaming: see the class."""
        e = self.batches[k]
        if not self.streaming:
            return e
        stream = torch.cuda.current_stream(self.device)
        if not e["_loaded"]:
            for other in self.batches:              # (plans go with their entry: bear_plan_destroy synchronises the device)
                if other["_loaded"] and other is not e:
                    keep = {key: other[key] for key in ("global_rows", "uploaded_rows", "row0")}
                    other.clear()
                    other.update(keep, rows=keep["uploaded_rows"], plans={}, _loaded=False)
            if self._pending is not None and self._pending[0] != k:
                self._pending[1].result()           # a prefetch nobody asked for (batches taken out of order): dropped
                self._pending = None
            if self._pending is None:
                self._up.stream.wait_stream(stream)
                landed = self._enqueue(k)
            else:
                landed = self._pending[1].result()
            self._pending = None
            self._up.wait()                         # the compute stream waits for the copies (not the host)
            # (starting the next batch's upload HERE, under this batch's compaction / sort / plans, was measured slower: 16.4 -> 23.8 ms
            # per 1e7-context batch -- the upload's host copies and this thread's set-up work get in each other's way)
            self._finish(landed, slot=k)
            self.loads += 1
            self.upload_bytes = self._up.bytes
        self._prefetch(k, stream)
        return e

    def _prefetch(self, k, stream):
        """Starts the upload of the batch after k unless it is resident or on its way.  Its blocks may be those of a batch just
        dropped, so the side stream first waits for everything enqueued so far (the steps that read them)."""
        nxt = (k + 1) % len(self.batches)
        if self._pending is None and nxt != k and not self.batches[nxt]["_loaded"]:
            self._up.stream.wait_stream(stream)
            self._pending = (nxt, self._pool.submit(self._enqueue, nxt))

    def loaded(self):
        """(k, entry) over the batches in order, each resident while it is the current one."""
        for k in range(len(self.batches)):
            yield k, self.load(k)

    def close(self):
        """Streaming: waits for an upload still in flight and drops the staging state (the end of a driver's loop)."""
        if self.streaming and getattr(self, "_pool", None) is not None:
            if self._pending is not None:
                self._pending[1].result()
                self._pending = None
            self._pool.shutdown(wait=True)
            self._pool = None
            torch.cuda.current_stream(self.device).synchronize()    # the staging buffers go away with the uploader

    def eval_plan(self, k, column="test", train_column="train"):
        """Sorted plan of batch k's test column given its conditioning column, if any (built on first use, kept for later
        evaluations of the same shard)."""
        e = self.batches[k]
        key = ("eval", column, train_column)
        if key not in e["plans"]:
            e["plans"][key] = kernels.EvalPlan(e[column], e.get(train_column))
        return e["plans"][key]

    def plan(self, k, column, ncol, ref_column=None):
        """Training plan of batch k (built on first use).  ref_column (ncol = 4): the reference column of bear_ref with the stop net
        function -- the plan then folds the contexts without reference counts into a histogram (``bear_plan_create_ref``).
        ncol = "5 rows if dense": the five-column plan of a step that only calls the mode-N entry points (any torch AR function) --
        a table of large counts gets the plan's dense form (``kernels.Plan(..., rows_if_dense=True)``)."""
        e = self.batches[k]
        key = (column, ncol, ref_column)
        if key not in e["plans"]:
            if ncol == ROWS_IF_DENSE:
                e["plans"][key] = kernels.Plan(e[column], 5, rows_if_dense=True)
            else:
                e["plans"][key] = kernels.Plan(e[column], ncol, ref=None if ref_column is None else e[ref_column])
        return e["plans"][key]


ROWS_IF_DENSE = "5 rows if dense"       # ResidentBatches.plan: a five-column plan for the mode-N entry points only


class StepFns:
    """What a driver hands run_device_steps per batch: ``reduce(packed)`` enqueues this rank's shard reduce; ``step(packed, m, v, t,
    learning_rate, scale, loss_buf)`` (optional) enqueues the WHOLE optimizer step -- reduce and tf.keras Adam in one launch
    (``bear_*_train_step_f64``: the last block of the reduce kernel runs the update) -- which the loop takes when nothing has to
    happen between the two halves (one rank, Adam, no gradient accumulation)."""

    def __init__(self, reduce, step=None):
        self.reduce, self.step = reduce, step


def reducers(res, make):
    """The batches' step functions (``StepFns``; a bare callable = reduce only) from ``make(k)``, which loads batch k and builds what
    its step needs (plans, paired lists, prefix levels).  Resident epochs: all of them now, before anything is captured.  Streamed
    epochs: batch k's is made each time the batch comes up and dropped with it."""
    def wrap(fn):
        return fn if isinstance(fn, StepFns) else StepFns(fn)
    if not res.streaming:
        return [wrap(make(k)) for k in range(len(res.batches))]

    def lazy(k):
        def reduce(packed):
            e = res.load(k)
            fn = e.get("_reduce")
            if fn is None:
                fn = e["_reduce"] = wrap(make(k))
            fn.reduce(packed)
        # (a streamed batch's functions exist only while it is loaded, and its step is bound by PCIe, not by a launch: reduce + update)
        return StepFns(reduce)
    return [lazy(k) for k in range(len(res.batches))]


def run_device_steps(reduce_fns, scales, theta, repeats, learning_rate, optimizer_name, train_ar, acc_steps, device,
                     eager_first_period=False, graph_ok=True):
    """The optimizer loop with every moving quantity in device memory (bear_net.py:292-310 / bear_ref.py:360-381 without a host
    round trip per step).  Per batch k: ``reduce_fns[k](packed)`` enqueues this rank's shard reduce into
    ``packed = [sum LL, d/d theta]``; with several ranks ONE all-reduce of ``packed`` follows on the same stream (bear_net.py:278-290);
    then the update: gradients ``scale_k * packed[1:]`` summed over ``acc_steps`` batches (bear_net.py:193-196: summed, not averaged),
    applied by tf.keras Adam on the device (``bear_train_apply_f64``) or a Keras-equivalent torch optimizer.  With one rank, Adam
    and no accumulation the epoch is captured in a HIP graph and replayed.  ``theta`` is updated in place; returns the logged
    "elbo" of every optimizer step, ``-(sum of the scaled losses) / acc_steps`` (bear_net.py:303-305), read back once at the end.

    ``eager_first_period`` (reduce functions made of torch ops and autograd, ``run_autograd_steps``): the first period of steps is
    enqueued eagerly before anything is captured -- the libraries behind the ops pick their algorithms, the allocator its blocks and
    the plans are cut outside the capture.  ``graph_ok = False``: never capture."""
    reduce_fns = [f if isinstance(f, StepFns) else StepFns(f) for f in reduce_fns]
    n_theta, n_batches = theta.numel(), len(reduce_fns)
    world = dist.world()[1]
    packed = torch.zeros(n_theta + 1, dtype=torch.float64, device=device)
    n_steps = (repeats * n_batches) // acc_steps
    loss_buf = torch.zeros(max(n_steps, 1), dtype=torch.float64, device=device)
    adam = optimizer_name == "Adam"
    if adam:
        m, v = torch.zeros_like(theta), torch.zeros_like(theta)
        t = torch.zeros(1, dtype=torch.float64, device=device)

        def update(vec, scale):
            kernels.train_apply(theta, vec, m, v, t, learning_rate, scale, loss_buf, train_ar=train_ar)
    else:
        h_view, rest_view = theta[0:1], theta[1:]          # leaves sharing theta's storage: the optimizer updates theta in place
        opt = make_optimizer(optimizer_name, [rest_view] if train_ar else [h_view, rest_view], learning_rate)
        done = [0]

        def update(vec, scale):
            g = vec[1:] * scale
            opt.apply_gradients([g[1:]] if train_ar else [g[0:1], g[1:]])      # AR mode: h_signed gets no gradient (bear_net.py:194-196)
            loss_buf[done[0]:done[0] + 1] = -scale * vec[0:1]
            done[0] += 1

    acc = torch.zeros_like(packed) if acc_steps > 1 else None
    # One launch per optimizer step where nothing sits between the reduce and the update: one rank (no all-reduce), Adam, no
    # accumulation -- the reduce kernel's last block runs the update (bear_*_train_step_f64).  BEAR_AMD_TWO_LAUNCH_STEP=1: as before.
    fuse = adam and acc is None and not dist.collective_active() and not os.environ.get("BEAR_AMD_TWO_LAUNCH_STEP")
    fused_steps = [0]

    def one_step(k, step):
        """Batch k as optimizer-loop step number `step` (1-based): reduce -> [one all-reduce] -> update (every acc_steps steps)."""
        if fuse and reduce_fns[k].step is not None:
            reduce_fns[k].step(packed, m, v, t, learning_rate, scales[k], loss_buf)
            fused_steps[0] += 1
            return
        reduce_fns[k].reduce(packed)
        dist.allreduce_sum_(packed)
        if acc is None:
            update(packed, scales[k])
        else:
            acc.add_(packed, alpha=scales[k])
            if step % acc_steps == 0:
                update(acc, 1.0)
                acc.zero_()

    # HIP graph: the reference traces ONE tf.function for every case (bear_net.py:146, :275-290, several replicas and gradient
    # accumulation included).  Here the unit that repeats exactly is `period` = lcm(batches per epoch, acc_steps) steps: it is
    # captured once -- the all-reduce of several ranks too (RCCL collectives are stream-ordered and capturable; gloo stages
    # through the host and is not) -- and replayed; what is left over runs eagerly.  Every rank takes the same decision (it
    # depends on the arguments only), so the ranks' collectives stay paired either way.
    total_steps = repeats * n_batches
    period = math.lcm(n_batches, acc_steps) if n_batches else 0
    graph, replays, done_steps = None, 0, 0
    if (graph_ok and adam and 1 <= period <= GRAPH_MAX_BATCHES and total_steps >= (3 if eager_first_period else 2) * period
            and dist.collective_capturable() and not os.environ.get("BEAR_AMD_NO_GRAPH")):
        if eager_first_period:
            for i in range(period):
                one_step(i % n_batches, i + 1)
            done_steps = period                  # (a multiple of the batches per epoch and of acc_steps: the pattern repeats from here)
        # a replay costs one graph launch (~15 us, scripts/dev/step_latency.py) whatever it holds: short periods are unrolled
        # until a graph carries up to GRAPH_MAX_BATCHES steps, as long as at least four replays remain
        period *= max(1, min(GRAPH_MAX_BATCHES // period, (total_steps - done_steps) // (4 * period)))
        if dist.collective_active():
            dist.allreduce_sum_(torch.zeros_like(packed))     # the communicator comes up outside the capture
        kernels.default_workspace(device, for_capture=True)   # exists before the capture (creating one allocates)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(graph):        # `period` steps in order; nothing runs yet
                for i in range(period):
                    one_step(i % n_batches, i + 1)
        except (_lib.BearError, torch.cuda.OutOfMemoryError):
            raise
        except RuntimeError as err:              # stream capture unavailable: the eager loop below enqueues the same kernels
            warnings.warn(f"HIP-graph capture of the training step failed ({err}); using the eager loop")
            graph = None
    eager_steps = done_steps
    # device time of the steps that follow (HIP events on the launch stream; bench.py's per-config step times): the first quarter
    # of them is left out of `timed_*` so that the clock ramp of a card that idled during set-up is not in the figure
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    loop_steps, timed_from = total_steps - done_steps, None
    ev[0].record()
    if graph is not None:
        replays = (total_steps - done_steps) // period
        for r in range(replays):
            if r == replays // 4:
                ev[1].record()
                timed_from = done_steps + r * period
            graph.replay()
        done_steps += replays * period
    first_eager = done_steps
    for i in range(done_steps, total_steps):
        if timed_from is None and i - first_eager == (total_steps - first_eager) // 4:
            ev[1].record()
            timed_from = i
        one_step(i % n_batches, i + 1)
    ev[2].record()
    torch.cuda.synchronize()
    LAST_RUN.update(graph=graph is not None, replays=replays, period=period, eager_steps=eager_steps + total_steps - done_steps,
                    world=world, collective=dist.collective_active(), loop_steps=loop_steps, one_launch_steps=fuse and fused_steps[0] > 0,
                    loop_ms=ev[0].elapsed_time(ev[2]) if loop_steps else 0.0,
                    timed_steps=total_steps - timed_from if timed_from is not None else 0,
                    timed_ms=ev[1].elapsed_time(ev[2]) if timed_from is not None else 0.0)
    return (loss_buf[:n_steps] / acc_steps).cpu().tolist()


def deterministic():
    """BEAR_AMD_DETERMINISTIC=1 -- or the deterministic BUILD of the library is the one loaded (``libbear_hip_det.so`` through
    BEAR_AMD_LIB, or the variable unset again after import: its kernels take the fixed-point path whatever the variable says
    now): parameter gradients bit-identical from run to run (include/bear_hip.h)."""
    v = os.environ.get("BEAR_AMD_DETERMINISTIC", "")
    if bool(v) and v != "0":
        return True
    return bool(_lib.lib().bear_deterministic_build())


def deterministic_agreed(device=None):
    """``deterministic()`` after checking ONCE per process group that every rank answers alike: the deterministic mode adds two
    all-reduces per batch set-up (the count bound of the fixed-point tables, bear_net.train), so ranks that disagree -- the variable
    exported on some of them only -- would deadlock there, or scale their gradient tables differently without a word."""
    mine = deterministic()
    world = dist.world()[1]
    if world > 1 and _DET_AGREED.get("world") != (world, mine):
        hi = dist.agree_max(1 if mine else 0, device)
        lo = -dist.agree_max(-1 if mine else 0, device)       # min over the ranks
        if hi != lo:
            raise RuntimeError("BEAR_AMD_DETERMINISTIC (or the deterministic build of libbear_hip) is in effect on some ranks only: "
                               "export the variable to every rank of the job")
        _DET_AGREED["world"] = (world, mine)
    return mine


_DET_AGREED = {}


def live_rows(e, *columns, by="train"):
    """Row indices of a resident batch's contexts that hold counts in column ``by`` (cached in the batch entry, together with the
    gathered ``columns`` as ``<name>_live_<by>``), or None when (nearly) all of them do.  An AR function made of torch ops only has
    to produce the prior rows of those contexts: in training the DM kernel reads nobody else's row and their gradient rows are
    zero, in evaluation only contexts with held-out counts enter any sum (``scatter_live`` puts the rows back in place).
    BEAR_AMD_ALL_ROWS=1 switches the gather off (tests)."""
    key = "live_" + by
    if key not in e:
        live = None
        if e["rows"] and not os.environ.get("BEAR_AMD_ALL_ROWS"):
            idx = (e[by] != 0).any(dim=1).nonzero().squeeze(1)
            if idx.numel() < 0.95 * e["rows"]:
                live = idx
        e[key] = live
    if e[key] is not None:
        for c in columns:
            if c + "_" + key not in e:
                e[c + "_" + key] = e[c].index_select(0, e[key]).contiguous()
    return e[key]


def scatter_live(rows_live, live, n_rows):
    """[len(live), W] prior rows -> [n_rows, W] with zero rows for the contexts without counts (differentiable)."""
    full = torch.zeros((n_rows, rows_live.shape[1]), dtype=rows_live.dtype, device=rows_live.device)
    return full.index_copy(0, live, rows_live)


def check_normalized_rows(rows, what):
    """The plugin's ``normalized_rows`` promise, checked ONCE per train() call on the first batch's rows: the kernels that take it
    never form a row's sum (the concentration total is then shared by all contexts), so a wrong flag would give silently wrong
    sums.  Rows of zeros are contexts without counts that ``scatter_live`` filled in: nobody reads them."""
    s = rows.detach().sum(dim=1)
    off = torch.where(s != 0, (s - 1.0).abs(), torch.zeros_like(s)).max() if s.numel() else torch.zeros(())
    if float(off) > 1e-12:
        raise ValueError(f"{what} sets normalized_rows, but its rows do not sum to one (off by up to {float(off):.3g}): "
                         "drop the attribute (the general kernel forms every row's sum) or normalise the rows")


def run_autograd_steps(res, prior_fn, params, h_signed, num_kmers, repeats, learning_rate, optimizer_name, train_ar, acc_steps,
                       normalized, device, ref_mix=None):
    """The optimizer loop for an AR function made of torch ops (any ``ar_funcs`` plugin; bear_net.py:292-310): per batch the
    prior rows come from ``prior_fn(batch entry)`` with autograd, the planned kernel returns sum LL, d/dh and the gradient rows
    (h_signed is read from the parameter vector on the device), ``Tensor.backward`` carries the rows to the parameters, and the
    sums and gradients leave as ONE packed vector ``[sum LL, d/dh, d/d parameters...]`` -- from there on the loop IS
    ``run_device_steps``: one all-reduce of the packed vector over the ranks, tf.keras Adam on the whole parameter vector in one
    launch (``bear_train_apply_f64``; another Keras optimizer through torch), the logged losses on the device, and one period of
    steps captured into a HIP graph and replayed.  For that the parameters become views of one device vector ``theta`` (their
    ``.data`` is re-pointed: they stay the leaves the AR function closes over).

    A small table (configs[0]: 1365 rows) is launch-bound: an eager step was ~220 launches from Python (half of them the
    per-parameter optimizer arithmetic), a replayed one is the AR function's own kernels back to back.  Large batches gain nothing
    from a graph and would pin the AR function's intermediates in its memory pool: eager above BEAR_AMD_GRAPH_MAX_ROWS rows.

    ``ref_mix = (net_fn, ref_fn, tau_signed, net_weight_signed)`` (bear_ref with a net function that has parameters and
    normalised rows): ``net_fn(batch entry)`` are the NET rows (autograd), ``ref_fn(batch entry)`` the reference rows, and the
    mixing of bear_ref.py:63-68 happens inside the DM kernel (``bear_dm_refmix_plan_grad_f64``), which also returns the gradients
    of the two mixing parameters; ``prior_fn`` is then not called."""
    with torch.no_grad():
        theta = torch.cat([p.detach().reshape(-1).to(device=device, dtype=torch.float64) for p in params]).contiguous()
        k = 0
        for p in params:                     # the parameters live in theta from here on (and still do when the caller gets them back)
            p.data = theta[k:k + p.numel()].view(p.shape)
            k += p.numel()
    rest = params[1:]
    out = torch.zeros(2, dtype=torch.float64, device=device)
    out4 = torch.zeros(4, dtype=torch.float64, device=device)
    h_dev = theta[0:1]
    promise_checked = [not (normalized or ref_mix is not None)]

    def reducer(k):
        e = res.batches[k]

        def reduce(packed):
            """packed = [sum LL, d/dh, d/d parameters] of this rank's piece of batch k (unscaled sums).  Tensor ops, autograd and
            kernel launches on the current stream only: capturable."""
            for p in rest:
                p.grad = None
            sums = out
            res.load(k)                            # (streamed epochs: the batch comes up now; resident ones: nothing to do)
            if e["rows"] and ref_mix is not None:
                net_fn, ref_fn, tau_p, nw_p = ref_mix
                net = net_fn(e)
                if not promise_checked[0]:
                    check_normalized_rows(net, "the net function")
                    promise_checked[0] = True
                _, grad_net = kernels.dm_refmix_planned_dev(res.plan(k, "train", 5), net.detach().contiguous(), ref_fn(e), h_dev,
                                                            tau_p.detach().reshape(1), nw_p.detach().reshape(1), out=out4,
                                                            train_ar=train_ar)
                if net.requires_grad:
                    net.backward(grad_net)                         # d sum LL / d net parameters
                tau_p.grad = out4[2].reshape(tau_p.shape)
                nw_p.grad = out4[3].reshape(nw_p.shape)
                sums = out4[:2]
            elif e["rows"]:
                prior = prior_fn(e)
                if not promise_checked[0]:
                    check_normalized_rows(prior, "the AR function")
                    promise_checked[0] = True
                need_rows = prior.requires_grad                    # parameter-free AR function (stop): nothing to feed back
                r = kernels.dm_prior_planned_dev(res.plan(k, "train", ROWS_IF_DENSE), prior.detach(), h_dev, out=out, want_grad=need_rows,
                                                 train_ar=train_ar, normalized=normalized)
                if need_rows:
                    prior.backward(r[1])                           # d sum LL / d AR parameters
            else:
                out.zero_()
            torch.cat([sums] + [(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in rest], out=packed)
            for p in rest:
                p.grad = None                  # (inside a capture these are blocks of the graph's pool: nobody holds them between replays)
        return reduce

    scales = [-(num_kmers / e["global_rows"]) for e in res.batches]       # bear_net.py:190-191 with the global batch
    max_rows = max([e["global_rows"] for e in res.batches] + [0])
    return run_device_steps([reducer(k) for k in range(len(res.batches))], scales, theta, repeats, learning_rate, optimizer_name,
                            train_ar, acc_steps, device, eager_first_period=True,
                            graph_ok=not res.streaming and max_rows <= int(os.environ.get("BEAR_AMD_GRAPH_MAX_ROWS", 1 << 22)))


def compute_dtype(dtype):
    """``precision`` of the reference configs (``config_files/*.cfg``, `models/train_bear_net.py:58`): the HIP kernels compute
    in float64 whatever it says -- float32 runs are accepted and carried out (parameters included) in float64."""
    if dtype in (torch.float64, None):
        return torch.float64
    if dtype == torch.float32:
        import warnings
        warnings.warn("precision = float32: the HIP path computes and keeps its parameters in float64", stacklevel=3)
        return torch.float64
    raise NotImplementedError(f"precision {dtype}: the HIP kernels compute in float64")


def log_losses(losses, writer, loss_save, acc_steps=1):
    """The per-step scalars of bear_net.py:303-307 (TensorBoard 'elbo' at batch step acc_steps, 2 acc_steps, ...; loss_save),
    written once the device loop is done."""
    if loss_save is not None:
        loss_save.extend(losses)
    if writer is not None:
        for i, val in enumerate(losses):
            writer.add_scalar("elbo", val, (i + 1) * acc_steps)


def counts_f64(t):
    """uint32 counts carried in int32 storage -> float64."""
    return torch.where(t < 0, t.to(torch.float64) + 4294967296.0, t.to(torch.float64))


GRAPH_MAX_BATCHES = 64    # a period of at most this many steps is captured as one HIP graph
LAST_RUN = {}             # how the last run_device_steps call ran (tests, logs): graph?, replays, period, eager_steps
MAX_EVAL_MODELS = 64   # EVL_MAX_MODELS of kernels_eval.h: h values + van_reg values per launch


class EvaluationSums:
    """The seven partial sums of ``_evaluation_step`` (bear_net.py:323-371) of any number of batches, kept ON THE DEVICE: ``add``
    enqueues one launch per batch (and per chunk of h values) and returns at once; ``result`` sums the batches' output vectors
    on the device and reads them back -- one host synchronisation per evaluation, not one per batch."""

    def __init__(self, h, van_reg, eps=epsilon, noise_seed=0):
        self.hs = np.atleast_1d(np.asarray(h, dtype=np.float64)).reshape(-1)
        self.van = np.atleast_1d(np.asarray(van_reg, dtype=np.float64)).reshape(-1)
        if self.van.size >= MAX_EVAL_MODELS:
            raise ValueError(f"at most {MAX_EVAL_MODELS - 1} van_reg values")
        self.eps, self.noise_seed = eps, noise_seed
        self.step = MAX_EVAL_MODELS - self.van.size           # h values per launch; more: chunks, noise stream seed + k
        self.outs = [[] for _ in range(0, max(self.hs.size, 1), self.step)]

    def add(self, test, prior, train=None, row_base=0, plan=None, row_ids=None):
        """test / train: uint32 [n,5] device slabs; prior: float64 [n,5] = ar_func rows.  ``row_base`` is the global index of row 0
        (``row_ids``: of a compacted batch, row i is table row ``row_base + row_ids[i]``), so the arg-max noise stream does not
        depend on how the rows are sharded or compacted."""
        if row_ids is not None and plan is None:
            raise ValueError("row_ids go with a planned evaluation (resident batches)")
        for c, k in enumerate(range(0, max(self.hs.size, 1), self.step)):
            hk = self.hs[k:k + self.step]
            first = k == 0
            if plan is not None:      # resident table: the sorted plan of the test column (kernels_evalplan.h)
                out = kernels.evaluate_planned(plan, prior, hk, self.van if first else None, eps=self.eps, with_ar=first,
                                               noise_seed=self.noise_seed + k, row_base=row_base, row_ids=row_ids)
            else:
                out = kernels.evaluate(test, prior, hk, self.van if first else None, train, eps=self.eps, with_ar=first,
                                       noise_seed=self.noise_seed + k, row_base=row_base)
            self.outs[c].append(out)

    def result(self):
        """(ll_ear [H], ll_arm, ll_van [V], cor_ear [H], cor_arm, cor_van [V], total_len) summed over the batches added so far."""
        ll_ear, cor_ear, rest = [], [], None
        for c, k in enumerate(range(0, max(self.hs.size, 1), self.step)):
            H, V = self.hs[k:k + self.step].size, self.van.size if k == 0 else 0
            if self.outs[c]:
                out = (self.outs[c][0] if len(self.outs[c]) == 1 else torch.stack(self.outs[c]).sum(dim=0)).cpu().numpy()
            else:
                out = np.zeros(2 * (H + V) + 3)
            ll_ear.append(out[:H])
            cor_ear.append(out[H + V + 1:2 * H + V + 1])
            if k == 0:
                rest = (out[H], out[H + 1:H + 1 + V], out[2 * H + V + 1], out[2 * H + V + 2:2 * H + 2 * V + 2], out[-1])
        ll_arm, ll_van, cor_arm, cor_van, total_len = rest
        return (np.concatenate(ll_ear), ll_arm, ll_van, np.concatenate(cor_ear), cor_arm, cor_van, total_len)


def evaluation_sums(test, prior, h, van_reg, train=None, eps=epsilon, noise_seed=0, row_base=0, plan=None, row_ids=None):
    """The 7 partial sums of ``_evaluation_step`` (bear_net.py:323-371) for ONE batch of this rank's rows (``EvaluationSums``
    with a single ``add``): h float or 1-D sequence (h_scan, bear_net.py:523)."""
    sums = EvaluationSums(h, van_reg, eps=eps, noise_seed=noise_seed)
    sums.add(test, prior, train, row_base=row_base, plan=plan, row_ids=row_ids)
    return sums.result()


def reduce_evaluation(parts, device, scalar_h):
    """Sums the per-rank partials with one all-reduce and forms the reference's 9-tuple
    (bear_net.py:459-463)."""
    flat = torch.tensor(np.concatenate([np.atleast_1d(np.asarray(p, dtype=np.float64)).reshape(-1) for p in parts]),
                        dtype=torch.float64, device=device if dist.world()[1] > 1 and torch.cuda.is_available() else "cpu")
    dist.allreduce_sum_(flat)
    flat = flat.cpu().numpy()
    sizes = [np.atleast_1d(np.asarray(p)).size for p in parts]
    out, k = [], 0
    for s in sizes:
        out.append(flat[k:k + s])
        k += s
    ll_ear, ll_arm, ll_van, cor_ear, cor_arm, cor_van, total = out
    ll_arm, cor_arm, total = ll_arm[0], cor_arm[0], total[0]
    if scalar_h:
        ll_ear, cor_ear = ll_ear[0], cor_ear[0]
    return (ll_ear, ll_arm, ll_van,
            np.exp(-ll_ear / total), np.exp(-ll_arm / total), np.exp(-ll_van / total),
            cor_ear / total, cor_arm / total, cor_van / total)
