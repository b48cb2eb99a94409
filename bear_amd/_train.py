"""Shared machinery of the bear_net / bear_ref drivers: resident shards, Keras-equivalent Adam,
the per-step reduce, and the held-out evaluation built on the DM kernels."""
import math

import numpy as np
import torch

from . import core, dist, kernels
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


def make_optimizer(name, params, learning_rate):
    if name == "Adam":
        return KerasAdam(params, learning_rate)
    opt = getattr(torch.optim, name)(params, lr=learning_rate)

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


class ResidentBatches:
    """This rank's row shard of every batch of one epoch, uploaded once: per batch the device slabs of the
    requested dataset columns, the k-mer codes, and (lazily) the kernel plans."""

    def __init__(self, data, columns, device, want_codes=False):
        if not isinstance(data, CountDataset):
            raise TypeError("train / evaluation expect the CountDataset returned by bear_amd.dataloader")
        self.data, self.device = data, device
        self.batches = []
        # k-mer letter codes: the ASCII bytes go up as they were parsed and are encoded on the device
        fast_codes = want_codes and data.alphabet in ("dna", "rna") and data.lag > 0
        on_dev = getattr(data, "counts_dev", None) is not None       # DeviceCountDataset: the table is in HBM already
        codes = data.codes() if (want_codes and not fast_codes) else None

        def device_column(col, lo, hi):
            if on_dev:
                return data.counts_dev[col, lo:hi].to(device).contiguous().clone()
            return torch.from_numpy(np.ascontiguousarray(data.counts[col, lo:hi]).view(np.int32)).to(device)

        def device_codes(lo, hi):
            if fast_codes:
                km = data.kmers_dev[lo:hi].to(device).contiguous() if on_dev else \
                    torch.from_numpy(np.ascontiguousarray(data.kmers[lo:hi])).to(device)
                return kernels.encode_kmers(km, data.alphabet)
            return torch.from_numpy(np.ascontiguousarray(codes[lo:hi])).to(device)
        shuffled = {}
        if data.shuffle_seed is not None and data.num_rows:
            # whole columns go up once, are permuted by one gather pass each (same seed: columns stay aligned), and the
            # batches below are slices of the permuted slabs
            for name, col in columns.items():
                up = device_column(col, 0, data.num_rows)
                shuffled[name] = kernels.shuffle_rows(up, data.shuffle_seed)
                del up
            if want_codes:
                shuffled["codes"] = kernels.shuffle_rows(device_codes(0, data.num_rows), data.shuffle_seed)
        for a, b in data.batch_bounds():
            lo, hi = dist.shard_rows(b - a)
            lo, hi = a + lo, a + hi
            entry = {"global_rows": b - a, "rows": hi - lo, "row0": lo}
            for name, col in columns.items():
                if shuffled:
                    entry[name] = shuffled[name][lo:hi].clone()
                else:
                    entry[name] = device_column(col, lo, hi)
            if want_codes:
                entry["codes"] = shuffled["codes"][lo:hi].clone() if shuffled else device_codes(lo, hi)
            entry["plans"] = {}
            self.batches.append(entry)
        del shuffled

    def plan(self, k, column, ncol):
        e = self.batches[k]
        key = (column, ncol)
        if key not in e["plans"]:
            e["plans"][key] = kernels.Plan(e[column], ncol)
        return e["plans"][key]


def counts_f64(t):
    """uint32 counts carried in int32 storage -> float64."""
    return torch.where(t < 0, t.to(torch.float64) + 4294967296.0, t.to(torch.float64))


GRAPH_MAX_BATCHES = 64    # an epoch of at most this many resident batches is captured as one HIP graph
MAX_EVAL_MODELS = 64   # EVL_MAX_MODELS of kernels_eval.h: h values + van_reg values per launch


def evaluation_sums(test, prior, h, van_reg, train=None, eps=epsilon, noise_seed=0, row_base=0):
    """The 7 partial sums of ``_evaluation_step`` (bear_net.py:323-371) for this rank's rows: one launch of
    ``bear_eval_f64`` (all h values, the AR model and all van_reg values in a single pass over the rows).
    test / train: uint32 [n,5] device slabs; prior: float64 [n,5] = ar_func rows; h: float or 1-D sequence
    (h_scan, bear_net.py:523).  ``row_base`` is the global index of row 0, so the arg-max noise stream
    does not depend on how the rows are sharded."""
    hs = np.atleast_1d(np.asarray(h, dtype=np.float64)).reshape(-1)
    van = np.atleast_1d(np.asarray(van_reg, dtype=np.float64)).reshape(-1)
    if van.size >= MAX_EVAL_MODELS:
        raise ValueError(f"at most {MAX_EVAL_MODELS - 1} van_reg values")
    ll_ear, cor_ear = [], []
    rest = None
    step = MAX_EVAL_MODELS - van.size
    for k in range(0, max(hs.size, 1), step):        # more h values than fit one launch: chunks, noise stream seed + k
        hk = hs[k:k + step]
        first = k == 0
        out = kernels.evaluate(test, prior, hk, van if first else None, train, eps=eps, with_ar=first,
                               noise_seed=noise_seed + k, row_base=row_base).cpu().numpy()
        H, V = hk.size, van.size if first else 0
        ll_ear.append(out[:H])
        cor_ear.append(out[H + V + 1:2 * H + V + 1])
        if first:
            rest = (out[H], out[H + 1:H + 1 + V], out[2 * H + V + 1], out[2 * H + V + 2:2 * H + 2 * V + 2], out[-1])
    ll_arm, ll_van, cor_arm, cor_van, total_len = rest
    return (np.concatenate(ll_ear), ll_arm, ll_van, np.concatenate(cor_ear), cor_arm, cor_van, total_len)


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
