"""BEAR / AR models with a parametric AR function: host mirror of ``bear_model/bear_net.py``.

``train`` (bear_net.py:200-321), ``evaluation`` (:387-463), ``h_scan`` (:465-531),
``change_scope_params`` (:103-143) with the reference's signatures.  Per batch shard and step the
AR rows come from the plugin (``ar_funcs``, PyTorch-ROCm ops with autograd), and the DM / multinomial
log-likelihood, its gradient w.r.t. ``h_signed`` and w.r.t. the AR rows come from one launch of
``bear_dm_prior_f64``; the row gradient is fed back through ``ar_func`` by ``Tensor.backward``.
"""
import numpy as np
import torch

from . import _train, ar_funcs, core, dist, kernels

epsilon = core.epsilon


def _create_params(lag, alphabet_size, make_ar_func, af_kwargs, dtype=torch.float64, device=None):
    """bear_net.py:73-100."""
    ar_func, ar_func_params = make_ar_func(lag, alphabet_size, **af_kwargs, dtype=dtype, device=device)
    h_signed = torch.tensor(0.0, dtype=dtype, device=device, requires_grad=True)
    return [h_signed] + ar_func_params, h_signed, ar_func


def change_scope_params(lag, alphabet_size, make_ar_func, af_kwargs, params, dtype=torch.float64, device=None):
    """bear_net.py:103-143."""
    new, h_signed, ar_func = _create_params(lag, alphabet_size, make_ar_func, af_kwargs, dtype, device)
    with torch.no_grad():
        for p, q in zip(new, params):
            p.copy_(torch.as_tensor(q, dtype=p.dtype))
    return new, h_signed, ar_func


def train(data, num_kmers, epochs, ds_loc, alphabet, lag, make_ar_func, af_kwargs,
          learning_rate, optimizer_name, train_ar, acc_steps=1,
          params_restart=None, writer=None, loss_save=None, dtype=torch.float64):
    """bear_net.train (bear_net.py:200-321); returns ``(params, h_signed, ar_func)``."""
    dtype = _train.compute_dtype(dtype)
    device = _train.require_device()
    alphabet_size = len(core.alphabets_tf[alphabet]) - 1
    if params_restart is None:
        params, h_signed, ar_func = _create_params(lag, alphabet_size, make_ar_func, af_kwargs, dtype, device)
    else:
        params, h_signed, ar_func = change_scope_params(lag, alphabet_size, make_ar_func, af_kwargs, params_restart, dtype, device)
    dist.broadcast_params(params)                    # mirrored variables: every rank starts from rank 0's values (bear_net.py:246-256)
    ar_params = params[1:]
    # linear AR function on a DNA/RNA-sized alphabet: forward, ELBO and all gradients in one launch per batch
    fused_mat = getattr(ar_func, "linear_mat", None)
    if fused_mat is not None and not (alphabet_size == 4 and lag <= kernels.LINEAR_MAX_LAG and fused_mat is ar_params[0]):
        fused_mat = None
    cnn_ok = (getattr(ar_func, "fused", False) and alphabet_size == 4 and len(ar_params) == 8
              and all(a is b for a, b in zip(getattr(ar_func, "cnn_params", []), ar_params)))
    fused = fused_mat is not None or cnn_ok
    # Fused heads: the sums of a step do not depend on the order of a batch's rows, so every batch is kept sorted by k-mer (first
    # letter most significant) -- consecutive contexts share all but their last letters: the fused linear kernel adds whole waves /
    # quads of them to d/d mat at once instead of one LDS atomic per context, letter and position (kernels_linear.h), the
    # convolutional kernels evaluate a window that all contexts of a wave share once (kernels_cnn.h).  The plans are cut as the
    # batches land, while the next batch is still being uploaded.
    res = _train.ResidentBatches(data, {"train": ds_loc}, device, want_codes=True, drop_empty="train", kmer_order=fused,
                                 prebuild=[("train", 5 if fused else _train.ROWS_IF_DENSE, None)],     # (+ paired lists of the linear head / prefix levels of the cnn step)
                                 per_row_extra=(8 + (208 + 64 if cnn_ok else 4)) if fused else 80)
    scales = [-(num_kmers / e["global_rows"]) for e in res.batches]       # bear_net.py:190-191 with the global batch
    if fused:
        # theta = {h_signed, flattened AR parameters} lives on the device for the whole run: one step is constants-from-theta ->
        # fused kernels [-> all-reduce of the packed vector] -> Adam, no host round trip (_train.run_device_steps)
        theta = torch.cat([h_signed.detach().reshape(1)] + [p.detach().reshape(-1) for p in ar_params]).to(
            device=device, dtype=torch.float64).contiguous()
        def pack_of(e):                 # the batch's k-mers as the fused kernels read them (kept with the batch)
            if "pack" not in e:
                q = kernels.pack_kmers(e["codes"].contiguous())
                e["pack"] = kernels.linear_index(q, lag) if fused_mat is not None else q    # the linear head reads table-row words, not packed letters
            return e["pack"]
        if cnn_ok:
            fw = ar_func.cnn_filter_width
            cnn_ws = kernels.default_workspace(device)     # the workspace the plans below are created on: the step's reservation lives there
            bufs = kernels.cnn_step_buffers(max(max(e["rows"] for e in res.batches), 1), lag, fw, device, ws=cnn_ws)   # one set, largest batch

        def reducer(k):
            e = res.load(k)
            plan = res.plan(k, "train", 5) if e["rows"] else None     # built here, before any capture (plan creation allocates and synchronises)
            if _train.deterministic_agreed(device) and fused_mat is not None:
                # BEAR_AMD_DETERMINISTIC: the fixed-point scale of the linear step's gradient tables follows from the counts of the
                # WHOLE batch -- every rank's piece -- so that d/d mat does not depend on the number of ranks (include/bear_hip.h);
                # a rank whose piece of the batch is empty takes part in the two all-reduces with zeros
                own = plan.count_total()[0] if plan is not None else [0.0, 0.0, 0.0]
                tot, cmax = torch.tensor(own[:2], dtype=torch.float64, device=device), torch.tensor(own[2:], dtype=torch.float64, device=device)
                dist.allreduce_sum_(tot)
                dist.allreduce_max_(cmax)
                if plan is not None:
                    plan.set_count_bound(tot.tolist() + cmax.tolist())
            if plan is None:
                return lambda packed: packed.zero_()
            pack = pack_of(e)
            if fused_mat is not None and pack.data_ptr() % 16 == 0:
                # neighbours of the sorted batch that share all letters but the last three go through the step two at a time
                # (kernels_linear.h, paired lists); declined by the library for tables too sparse to gain from it
                plan.pair_contexts(pack, lag)
            if cnn_ok:
                if pack.data_ptr() % 16 == 0:
                    # a position of the sorted batch once per distinct prefix (kernels_cnn.h, prefix levels); none attached when the
                    # prefixes of the table do not repeat
                    plan.attach_cnn_levels(pack, lag, fw)
                views = tuple(b[:e["rows"]] for b in bufs)
                return lambda packed: kernels.net_cnn_train_reduce(plan, pack, lag, fw, theta, views, packed, train_ar=train_ar)
            return _train.StepFns(
                lambda packed: kernels.net_linear_train_reduce(plan, pack, lag, theta, packed, train_ar=train_ar),
                lambda packed, m, v, t, lr, scale, loss_buf: kernels.net_linear_train_step(plan, pack, lag, theta, m, v, t, packed, lr, scale,
                                                                                           loss_buf, train_ar=train_ar))
        reduce_fns = _train.reducers(res, reducer)
        losses = _train.run_device_steps(reduce_fns, scales, theta, data.repeats, learning_rate, optimizer_name, train_ar, acc_steps, device,
                                         graph_ok=not res.streaming)
        with torch.no_grad():
            k = 0
            for p in params:
                p.copy_(theta[k:k + p.numel()].reshape(p.shape))
                k += p.numel()
    else:
        normalized = bool(getattr(ar_func, "normalized_rows", False))   # every reference AR function ends in a softmax

        def prior_fn(e):
            live = _train.live_rows(e, "codes")          # contexts without training counts need no prior row
            out = ar_func(e["codes"] if live is None else e["codes_live_train"])
            if live is None or out.shape[0] == 1:        # (a parameter-free AR function may return one row for all contexts)
                return out.expand(e["rows"], alphabet_size + 1).contiguous()
            return _train.scatter_live(out, live, e["rows"])
        losses = _train.run_autograd_steps(res, prior_fn, params, h_signed, num_kmers, data.repeats, learning_rate, optimizer_name, train_ar,
                                           acc_steps, normalized, device)
    res.close()
    ar_funcs.release_ar_func_cache(ar_func)
    _train.log_losses(losses, writer, loss_save, acc_steps)
    return params, h_signed, ar_func


def _eval(data, ds_loc_train, ds_loc_test, alphabet, h, ar_func, van_reg, dtype, seed):
    dtype = _train.compute_dtype(dtype)
    device = _train.require_device()
    use_train = ds_loc_train >= 0
    cols = {"test": ds_loc_test}
    if use_train:
        cols["train"] = ds_loc_train
    # only the contexts with held-out counts are kept resident: nothing else enters any sum (their table rows travel as row_ids)
    res = _train.ResidentBatches(data, cols, device, want_codes=True, drop_empty="test", per_row_extra=60,   # prior rows + plan
                                 kmer_order=ar_funcs.wants_kmer_order(ar_func))
    sums = _train.EvaluationSums(h, van_reg, noise_seed=seed)     # the batches' sums stay on the device until all are enqueued
    with torch.no_grad():
        for k, e in res.loaded():
            if not e["rows"]:
                prior = torch.zeros((0, 5), dtype=dtype, device=device)
            else:                                        # prior rows of the contexts with held-out counts: nothing else enters a sum
                live = _train.live_rows(e, "codes", by="test")
                out = ar_func(e["codes"] if live is None else e["codes_live_test"])
                prior = out.expand(e["rows"], 5).contiguous() if live is None or out.shape[0] == 1 else _train.scatter_live(out, live, e["rows"])
            sums.add(e["test"], prior, e.get("train"), row_base=e["row0"], plan=res.eval_plan(k) if e["rows"] else None,
                     row_ids=e.get("row_ids") if e["rows"] else None)
    res.close()
    ar_funcs.release_ar_func_cache(ar_func)
    return sums.result(), device


def evaluation(data, ds_loc_train, ds_loc_test, alphabet, h, ar_func, van_reg, dtype=torch.float64, seed=0):
    """bear_net.evaluation (bear_net.py:387-463) -> the reference's 9-tuple."""
    hv = float(torch.as_tensor(h).detach().cpu().item())
    total, device = _eval(data, ds_loc_train, ds_loc_test, alphabet, hv, ar_func, van_reg, dtype, seed)
    return _train.reduce_evaluation(total, device, True)


def h_scan(data, ds_loc_train, ds_loc_test, alphabet, h, ar_func, dtype=torch.float64, seed=0):
    """bear_net.h_scan (bear_net.py:465-531): BEAR log-likelihood, perplexity and accuracy for a vector of h."""
    hs = torch.as_tensor(h).detach().cpu().numpy().reshape(-1)
    total, device = _eval(data, ds_loc_train, ds_loc_test, alphabet, hs, ar_func, np.ones(1), dtype, seed)
    r = _train.reduce_evaluation(total, device, False)
    return r[0], r[3], r[6]
