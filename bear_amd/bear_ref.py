"""Reference-based BEAR / AR models: host mirror of ``bear_model/bear_ref.py`` over the HIP kernels.

Same entry points and argument meaning as the reference: ``train`` (bear_ref.py:262-389),
``evaluation`` (:453-539), ``change_scope_params`` (:166-204).  With the stop net function
(``ar_funcs.make_ar_func_stop``, the reference's bear_stop_*.cfg configurations) a training step is one
launch of ``bear_dm_ref[_plan]_f64`` per batch shard plus one all-reduce of 4 doubles.
"""
import os

import numpy as np
import torch

from . import _train, ar_funcs as _ar_funcs, core, dist, kernels

epsilon = core.epsilon


def _counts_to_probs(ref_counts, tau, alphabet_size, dtype=torch.float64):
    """bear_ref.py:9-33."""
    norm = ref_counts / ref_counts.abs().sum(-1, keepdim=True)
    shape = torch.tensor(np.r_[np.ones(alphabet_size), 0], dtype=dtype, device=ref_counts.device)
    return (1 / alphabet_size) * shape + torch.exp(-tau) * (norm - (1 / alphabet_size) * shape)


class _FusedRefMix(torch.autograd.Function):
    """``(nw net + jukes_cantor(ref, tau)) / (nw + 1)`` (bear_ref.py:63-68) as ``bear_ref_mix_forward_f64`` /
    ``bear_ref_mix_backward_f64`` behind torch autograd: a dozen passes over [n, 5] temporaries become one launch each way."""

    @staticmethod
    def _rows(t):            # contiguous and 16-byte aligned (a slice of a larger tensor may start at an odd row)
        t = t.detach().contiguous()
        return t if t.data_ptr() % 16 == 0 else t.clone()

    @staticmethod
    def forward(ctx, net_rows, ref_rows, tau_signed, net_weight_signed):
        net_rows, ref_rows = _FusedRefMix._rows(net_rows), _FusedRefMix._rows(ref_rows)
        ctx.save_for_backward(net_rows, ref_rows, tau_signed.detach(), net_weight_signed.detach())
        return kernels.ref_mix_forward(net_rows, ref_rows, tau_signed.detach(), net_weight_signed.detach())

    @staticmethod
    def backward(ctx, grad_prior):
        net_rows, ref_rows, tau_signed, net_weight_signed = ctx.saved_tensors
        grad_rows, scalars = kernels.ref_mix_backward(net_rows, ref_rows, _FusedRefMix._rows(grad_prior), tau_signed, net_weight_signed)
        return grad_rows, None, scalars[0].reshape(tau_signed.shape), scalars[1].reshape(net_weight_signed.shape)


def _make_ref_ar_func(lag, alphabet_size, make_net_func, af_kwargs, dtype=torch.float64, device=None):
    """bear_ref.py:36-69: params = [tau_signed, net_weight_signed] + net params."""
    net_weight_signed = torch.tensor(-np.log(100), dtype=dtype, device=device, requires_grad=True)
    tau_signed = torch.tensor(np.log(1 / 30), dtype=dtype, device=device, requires_grad=True)
    net_func, ar_func_params = make_net_func(lag, alphabet_size, **af_kwargs, dtype=dtype, device=device)

    def ar_func(kmer_seqs, ref_counts):
        net = net_func(kmer_seqs)
        if (alphabet_size == 4 and dtype == torch.float64 and ref_counts.is_cuda and net.is_cuda and tau_signed.is_cuda
                and net.dim() == 2 and net.shape == ref_counts.shape and ref_counts.dtype == torch.float64 and net.shape[0]):
            return _FusedRefMix.apply(net, ref_counts, tau_signed, net_weight_signed)   # one launch per direction (kernels_refmix.h)
        nw = torch.exp(net_weight_signed)
        tau = torch.exp(tau_signed)
        return (nw * net + _counts_to_probs(ref_counts, tau, alphabet_size, dtype=dtype)) / (nw + 1)
    ar_func.net_is_stop = bool(getattr(net_func, "is_stop", False))
    ar_func.net_func, ar_func.tau_signed, ar_func.net_weight_signed = net_func, tau_signed, net_weight_signed
    # the Jukes-Cantor row sums to one, so the mixture does whenever the net function's rows do (every reference AR function)
    ar_func.normalized_rows = bool(getattr(net_func, "normalized_rows", False))
    return ar_func, ([tau_signed, net_weight_signed] + ar_func_params)


def _create_params(lag, alphabet_size, make_ar_func, af_kwargs, dtype=torch.float64, device=None):
    """bear_ref.py:136-163."""
    ar_func, ar_func_params = _make_ref_ar_func(lag, alphabet_size, make_ar_func, af_kwargs, dtype, device)
    h_signed = torch.tensor(0.0, dtype=dtype, device=device, requires_grad=True)
    return [h_signed] + ar_func_params, h_signed, ar_func


def change_scope_params(lag, alphabet_size, make_ar_func, af_kwargs, params, dtype=torch.float64, device=None):
    """bear_ref.py:166-204: rebuilds (params, h_signed, ar_func) from a saved parameter list."""
    new, h_signed, ar_func = _create_params(lag, alphabet_size, make_ar_func, af_kwargs, dtype, device)
    with torch.no_grad():
        for p, q in zip(new, params):
            p.copy_(torch.as_tensor(q, dtype=p.dtype))
    return new, h_signed, ar_func


def _ref_input(ref_slab, dtype=torch.float64):
    """bear_ref.py:332-337: ``(counts[:, ds_loc_ref] + epsilon) * not_stop``."""
    r = _train.counts_f64(ref_slab).to(dtype) + epsilon
    r[:, -1] = 0
    return r


def train(data, num_kmers, epochs, ds_loc, ds_loc_ref, alphabet, lag, make_ar_func, af_kwargs,
          learning_rate, optimizer_name, train_ar, acc_steps=1,
          params_restart=None, writer=None, loss_save=None, dtype=torch.float64):
    """bear_ref.train (bear_ref.py:262-389).  ``data`` is the (repeated) CountDataset of
    ``bear_amd.dataloader``; rows of every batch are sharded over the ranks of an initialised
    ``torch.distributed`` group (one process per GPU).  Returns ``(params, h_signed, ar_func)``."""
    dtype = _train.compute_dtype(dtype)              # float32 configs run in float64 too (the reference recommends float64)
    device = _train.require_device()
    alphabet_size = len(core.alphabets_tf[alphabet]) - 1
    if params_restart is None:
        params, h_signed, ar_func = _create_params(lag, alphabet_size, make_ar_func, af_kwargs, dtype, device)
    else:
        params, h_signed, ar_func = change_scope_params(lag, alphabet_size, make_ar_func, af_kwargs, params_restart, dtype, device)
    dist.broadcast_params(params)                    # mirrored variables: every rank starts from rank 0's values (bear_ref.py:310-321)
    if not ar_func.net_is_stop:
        return _train_general(data, num_kmers, params, h_signed, ar_func, learning_rate, optimizer_name, train_ar, acc_steps, writer,
                              loss_save, ds_loc, ds_loc_ref, device)
    # stop net function: theta = (h_signed, tau_signed, net_weight_signed) lives on the device for the whole run; one step is
    # constants-from-theta -> planned mode-R kernel -> finalize [-> all-reduce of 4 doubles] -> Adam, no host round trip
    res = _train.ResidentBatches(data, {"train": ds_loc, "ref": ds_loc_ref}, device, drop_empty="train",
                                 prebuild=[("train", 4, "ref")])       # plans cut while the next batch is still crossing PCIe
    theta = torch.stack([p.detach().reshape(()) for p in params[:3]]).to(device=device, dtype=torch.float64).contiguous()

    def reducer(k):
        e = res.load(k)
        if e["rows"] == 0:
            return lambda packed: packed.zero_()
        # built here, before any capture (plan creation allocates and synchronises); the reference column is resident too, so the
        # plan folds the contexts without reference counts into a histogram and a step streams only the others' items
        plan = res.plan(k, "train", 4, ref_column="ref")
        return _train.StepFns(
            lambda packed: kernels.ref_train_reduce(plan, e["ref"], theta, packed, train_ar=train_ar),
            lambda packed, m, v, t, lr, scale, loss_buf: kernels.ref_train_step(plan, e["ref"], theta, m, v, t, lr, scale, packed, loss_buf,
                                                                                train_ar=train_ar))
    reduce_fns = _train.reducers(res, reducer)
    scales = [-(num_kmers / e["global_rows"]) for e in res.batches]       # loss = -(num_kmers / B) sum LL, bear_ref.py:252-253
    losses = _train.run_device_steps(reduce_fns, scales, theta, data.repeats, learning_rate, optimizer_name, train_ar, acc_steps, device,
                                     graph_ok=not res.streaming)
    res.close()
    with torch.no_grad():
        for p, val in zip(params[:3], theta):
            p.copy_(val)
    _train.log_losses(losses, writer, loss_save, acc_steps)
    return params, h_signed, ar_func


def _train_general(data, num_kmers, params, h_signed, ar_func, learning_rate, optimizer_name, train_ar, acc_steps, writer, loss_save,
                   ds_loc, ds_loc_ref, device):
    """bear_ref.train with a parametrised net function (linear, cnn; bear_ref.py:63-68).  Per batch: the net function's rows
    (``bear_linear_forward_f64`` / ``bear_cnn_forward_f64`` behind autograd), then ONE launch of ``bear_dm_refmix_plan_grad_f64``
    -- the mixing ``(nw net + jukes_cantor(ref, tau)) / (nw + 1)``, sum LL, d/dh, d/dtau, d/dnet_weight and d/d(net rows) -- and
    the net function's backward launch.  Net rows that are not asserted normalised (a plugin without ``normalized_rows``) take
    three launches instead: ``bear_ref_mix_forward_f64``, the planned kernel with gradient rows, ``bear_ref_mix_backward_f64``
    through autograd -- the same loop as bear_net.train with two more parameters."""
    # a fused net function (linear rows / cnn kernels) shares work between neighbouring contexts: batches are kept in k-mer order
    # (the sums do not depend on the order; cnn forward + backward 70 instead of 137 ms per 1e8 contexts, linear backward 1.45 / 2.0)
    res = _train.ResidentBatches(data, {"train": ds_loc, "ref": ds_loc_ref}, device, want_codes=True, drop_empty="train",
                                 kmer_order=bool(getattr(getattr(ar_func, "net_func", None), "fused", False)),
                                 prebuild=[("train", 5, None)], per_row_extra=120)

    def prior_fn_inputs(e):
        if "ref_in" not in e:
            e["ref_in"] = _ref_input(e["ref"])

    def prior_fn(e):
        prior_fn_inputs(e)
        live = _train.live_rows(e, "codes", "ref_in")    # contexts without training counts need no prior row
        if live is None:
            return ar_func(e["codes"], e["ref_in"]).contiguous()
        return _train.scatter_live(ar_func(e["codes_live_train"], e["ref_in_live_train"]), live, e["rows"])
    # normalised net rows, the reference's own mixing parameters: the mixing runs inside the DM kernel
    # (bear_dm_refmix_plan_grad_f64: one launch instead of mix-forward, gradient rows, mix-backward); BEAR_AMD_UNFUSED_MIX=1 keeps
    # the three launches (tests compare the two)
    ref_mix = None
    if (ar_func.normalized_rows and getattr(ar_func, "net_func", None) is not None and params[1] is ar_func.tau_signed
            and params[2] is ar_func.net_weight_signed and not os.environ.get("BEAR_AMD_UNFUSED_MIX")):
        def net_fn(e):
            prior_fn_inputs(e)
            live = _train.live_rows(e, "codes", "ref_in")
            if live is None:
                return ar_func.net_func(e["codes"])
            return _train.scatter_live(ar_func.net_func(e["codes_live_train"]), live, e["rows"])
        ref_mix = (net_fn, lambda e: e["ref_in"], params[1], params[2])
    losses = _train.run_autograd_steps(res, prior_fn, params, h_signed, num_kmers, data.repeats, learning_rate, optimizer_name, train_ar,
                                       acc_steps, ar_func.normalized_rows, device, ref_mix=ref_mix)
    res.close()
    _ar_funcs.release_ar_func_cache(getattr(ar_func, "net_func", ar_func))
    _train.log_losses(losses, writer, loss_save, acc_steps)
    return params, h_signed, ar_func


def evaluation(data, ds_loc_train, ds_loc_test, ds_loc_ref, alphabet, h, ar_func, van_reg, dtype=torch.float64,
               seed=0):
    """bear_ref.evaluation (bear_ref.py:453-539) -> (ll_ear, ll_arm, ll_van, perp_ear, perp_arm, perp_van,
    acc_ear, acc_arm, acc_van).  ``ds_loc_train = -1``: no conditioning on training counts.  The training
    column is used for conditioning (the reference reads the reference column there by mistake,
    bear_ref.py:397 vs bear_net.py:327)."""
    dtype = _train.compute_dtype(dtype)
    device = _train.require_device()
    use_train = ds_loc_train >= 0
    cols = {"test": ds_loc_test, "ref": ds_loc_ref}
    if use_train:
        cols["train"] = ds_loc_train
    # only the contexts with held-out counts are kept resident: nothing else enters any sum (their table rows travel as row_ids)
    res = _train.ResidentBatches(data, cols, device, want_codes=True, drop_empty="test", per_row_extra=60,   # prior rows + plan
                                 kmer_order=_ar_funcs.wants_kmer_order(getattr(ar_func, "net_func", ar_func)))
    hv = float(torch.as_tensor(h).item()) if np.ndim(torch.as_tensor(h).detach().cpu().numpy()) == 0 else torch.as_tensor(h).detach().cpu().numpy()
    sums = _train.EvaluationSums(hv, van_reg, noise_seed=seed)     # the batches' sums stay on the device until all are enqueued
    with torch.no_grad():
        for k, e in res.loaded():
            if not e["rows"]:
                prior = torch.zeros((0, 5), dtype=dtype, device=device)
            else:                                        # prior rows of the contexts with held-out counts: nothing else enters a sum
                if "ref_in" not in e:
                    e["ref_in"] = _ref_input(e["ref"], dtype)
                live = _train.live_rows(e, "codes", "ref_in", by="test")
                if live is None:
                    prior = ar_func(e["codes"], e["ref_in"]).expand(e["rows"], 5).contiguous()
                else:
                    prior = _train.scatter_live(ar_func(e["codes_live_test"], e["ref_in_live_test"]), live, e["rows"])
            sums.add(e["test"], prior, e.get("train"), row_base=e["row0"], plan=res.eval_plan(k) if e["rows"] else None,
                     row_ids=e.get("row_ids") if e["rows"] else None)
    res.close()
    _ar_funcs.release_ar_func_cache(getattr(ar_func, "net_func", ar_func))
    return _train.reduce_evaluation(sums.result(), device, np.ndim(hv) == 0)
