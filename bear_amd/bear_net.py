"""BEAR / AR models with a parametric AR function: host mirror of ``bear_model/bear_net.py``.

``train`` (bear_net.py:200-321), ``evaluation`` (:387-463), ``h_scan`` (:465-531),
``change_scope_params`` (:103-143) with the reference's signatures.  Per batch shard and step the
AR rows come from the plugin (``ar_funcs``, PyTorch-ROCm ops with autograd), and the DM / multinomial
log-likelihood, its gradient w.r.t. ``h_signed`` and w.r.t. the AR rows come from one launch of
``bear_dm_prior_f64``; the row gradient is fed back through ``ar_func`` by ``Tensor.backward``.
"""
import os
import warnings

import numpy as np
import torch

from . import _train, core, dist, kernels

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
    if dtype != torch.float64:
        raise NotImplementedError("the HIP kernels compute in float64")
    device = _train.require_device()
    alphabet_size = len(core.alphabets_tf[alphabet]) - 1
    if params_restart is None:
        params, h_signed, ar_func = _create_params(lag, alphabet_size, make_ar_func, af_kwargs, dtype, device)
    else:
        params, h_signed, ar_func = change_scope_params(lag, alphabet_size, make_ar_func, af_kwargs, params_restart, dtype, device)
    ar_params = params[1:]
    optimizer = _train.make_optimizer(optimizer_name, params, learning_rate)
    res = _train.ResidentBatches(data, {"train": ds_loc}, device, want_codes=True)
    n_batches = len(res.batches)
    acc = [torch.zeros_like(p) for p in params]
    loss, step = 0.0, 1
    out = torch.zeros(2, dtype=torch.float64, device=device)
    # linear AR function on a DNA/RNA-sized alphabet: forward, ELBO and all gradients in one launch per batch
    fused_mat = getattr(ar_func, "linear_mat", None)
    if fused_mat is not None and not (alphabet_size == 4 and lag <= kernels.LINEAR_MAX_LAG and fused_mat is ar_params[0]):
        fused_mat = None
    normalized = bool(getattr(ar_func, "normalized_rows", False))   # every reference AR function ends in a softmax
    graphable = (1 <= n_batches <= _train.GRAPH_MAX_BATCHES and acc_steps == 1 and optimizer_name == "Adam"
                 and dist.world()[1] == 1 and all(b["rows"] > 0 for b in res.batches) and data.repeats > 1
                 and not os.environ.get("BEAR_AMD_NO_GRAPH"))
    if fused_mat is not None and graphable:
        try:
            return _train_linear_graph(res, data.repeats, num_kmers, params, h_signed, ar_func, fused_mat, lag, learning_rate, train_ar,
                                       loss_save, device, writer)
        except RuntimeError as err:     # stream capture unavailable: the eager loop below runs the same kernels
            warnings.warn(f"HIP-graph capture of the training step failed ({err}); using the eager loop")
    cnn_ok = (getattr(ar_func, "fused", False) and alphabet_size == 4 and len(ar_params) == 8
              and all(a is b for a, b in zip(getattr(ar_func, "cnn_params", []), ar_params)))
    if cnn_ok and graphable:
        try:
            return _train_cnn_graph(res, data.repeats, num_kmers, params, h_signed, ar_func, lag, learning_rate, train_ar, loss_save, device,
                                    writer)
        except RuntimeError as err:
            warnings.warn(f"HIP-graph capture of the training step failed ({err}); using the eager loop")
    for _ in range(data.repeats):
        for k in range(n_batches):
            e = res.batches[k]
            scale = -(num_kmers / e["global_rows"])                    # bear_net.py:190-191 with the global batch
            for p in ar_params:
                p.grad = None
            if e["rows"] and fused_mat is not None:
                if "packed" not in e:
                    e["packed"] = kernels.pack_kmers(e["codes"].contiguous())
                _, gmat = kernels.dm_linear(res.plan(k, "train", 5), e["packed"], fused_mat.detach(), h_signed.item(),
                                            train_ar=train_ar, out=out)
                fused_mat.grad = scale * gmat
            elif e["rows"]:
                prior = ar_func(e["codes"]).expand(e["rows"], alphabet_size + 1).contiguous()
                need_rows = prior.requires_grad
                if need_rows:     # planned kernel, gradient rows assembled in LDS
                    _, grad_rows = kernels.dm_prior_planned(res.plan(k, "train", 5), prior.detach(), h_signed.item(), out=out,
                                                            want_grad=True, train_ar=train_ar, normalized=normalized)
                else:             # parameter-free AR function (stop): nothing to feed back
                    kernels.dm_prior_planned(res.plan(k, "train", 5), prior.detach(), h_signed.item(), out=out,
                                             train_ar=train_ar, normalized=normalized)
                    grad_rows = None
                if need_rows:
                    prior.backward(scale * grad_rows)                  # d loss / d AR parameters
            else:
                out.zero_()
            flat, unpack = dist.pack([out] + [p.grad if p.grad is not None else torch.zeros_like(p) for p in ar_params])
            dist.allreduce_sum_(flat)                                  # one packed all-reduce: loss, d/dh, AR grads
            parts = unpack(flat)
            loss += scale * parts[0][0].item()
            if not train_ar:
                acc[0] += scale * parts[0][1]                          # AR mode: h_signed gets no gradient (bear_net.py:194-196)
            for a, g in zip(acc[1:], parts[1:]):
                a += g.to(a.dtype)
            if step % acc_steps == 0:
                if writer is not None:
                    writer.add_scalar("elbo", -loss / acc_steps, step)
                if loss_save is not None:
                    loss_save.append(-loss / acc_steps)
                optimizer.apply_gradients([None if train_ar else acc[0]] + acc[1:])
                for a in acc:
                    a.zero_()
                loss = 0.0
            step += 1
    return params, h_signed, ar_func


def _train_linear_graph(res, steps, num_kmers, params, h_signed, ar_func, mat, lag, learning_rate, train_ar, loss_save, device, writer=None):
    """Resident batches, linear AR function, Adam, one GPU: the optimizer step (1/h from the device-resident parameters,
    the fused linear-head kernel, both finalize kernels, Adam on {h_signed, mat}) is captured in a HIP graph once and replayed
    ``steps`` times; parameters and losses come back at the end (see bear_ref._train_stop_graph)."""
    plans = [res.plan(k, "train", 5) for k in range(len(res.batches))]
    packed = [kernels.pack_kmers(e["codes"].contiguous()) for e in res.batches]
    theta = torch.cat([h_signed.detach().reshape(1), mat.detach().reshape(-1)]).to(device=device, dtype=torch.float64).contiguous()
    m, v = torch.zeros_like(theta), torch.zeros_like(theta)
    t = torch.zeros(1, dtype=torch.float64, device=device)
    gmat = torch.zeros(lag * 25, dtype=torch.float64, device=device)
    out = torch.zeros(2, dtype=torch.float64, device=device)
    loss_buf = torch.zeros(steps * len(res.batches), dtype=torch.float64, device=device)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):                                            # one epoch: the batches in order, one optimizer step each
        for e, plan, pk in zip(res.batches, plans, packed):
            kernels.net_linear_train_step(plan, pk, lag, theta, m, v, t, gmat, learning_rate, -(num_kmers / e["global_rows"]), out,
                                          loss_buf, train_ar=train_ar)
    for _ in range(steps):
        graph.replay()
    torch.cuda.synchronize()
    with torch.no_grad():
        h_signed.copy_(theta[0])
        mat.copy_(theta[1:].reshape(mat.shape))
    losses = loss_buf.cpu().tolist()
    if loss_save is not None:
        loss_save.extend(losses)
    if writer is not None:                       # the per-step scalars of bear_net.py:285-287 / bear_ref.py:353-355, written after the replay
        for i, val in enumerate(losses):
            writer.add_scalar("elbo", val, i + 1)
    return params, h_signed, ar_func


def _train_cnn_graph(res, steps, num_kmers, params, h_signed, ar_func, lag, learning_rate, train_ar, loss_save, device, writer=None):
    """Resident batches, convolutional AR function, Adam, one GPU: forward, planned DM kernel with gradient rows, backward
    and Adam on {h_signed, all eight parameter tensors} as one captured HIP graph, replayed ``steps`` times."""
    plans = [res.plan(k, "train", 5) for k in range(len(res.batches))]
    fw = ar_func.cnn_filter_width
    packed = [kernels.pack_kmers(e["codes"].contiguous()) for e in res.batches]
    ar_params = params[1:]
    theta = torch.cat([h_signed.detach().reshape(1)] + [p.detach().reshape(-1) for p in ar_params]).to(device=device, dtype=torch.float64).contiguous()
    m, v = torch.zeros_like(theta), torch.zeros_like(theta)
    t = torch.zeros(1, dtype=torch.float64, device=device)
    out = torch.zeros(2, dtype=torch.float64, device=device)
    loss_buf = torch.zeros(steps * len(res.batches), dtype=torch.float64, device=device)
    bufs = [kernels.cnn_step_buffers(plan, lag, fw) for plan in plans]
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):                                            # one epoch: the batches in order, one optimizer step each
        for e, plan, pk, bf in zip(res.batches, plans, packed, bufs):
            kernels.net_cnn_train_step(plan, pk, lag, fw, theta, m, v, t, bf, learning_rate, -(num_kmers / e["global_rows"]), out,
                                       loss_buf, train_ar=train_ar)
    for _ in range(steps):
        graph.replay()
    torch.cuda.synchronize()
    with torch.no_grad():
        h_signed.copy_(theta[0])
        k = 1
        for p in ar_params:
            p.copy_(theta[k:k + p.numel()].reshape(p.shape))
            k += p.numel()
    losses = loss_buf.cpu().tolist()
    if loss_save is not None:
        loss_save.extend(losses)
    if writer is not None:                       # the per-step scalars of bear_net.py:285-287 / bear_ref.py:353-355, written after the replay
        for i, val in enumerate(losses):
            writer.add_scalar("elbo", val, i + 1)
    return params, h_signed, ar_func


def _eval(data, ds_loc_train, ds_loc_test, alphabet, h, ar_func, van_reg, dtype, seed):
    device = _train.require_device()
    use_train = ds_loc_train >= 0
    cols = {"test": ds_loc_test}
    if use_train:
        cols["train"] = ds_loc_train
    res = _train.ResidentBatches(data, cols, device, want_codes=True)
    total = None
    with torch.no_grad():
        for e in res.batches:
            prior = ar_func(e["codes"]).expand(e["rows"], 5).contiguous() if e["rows"] else torch.zeros((0, 5), dtype=dtype, device=device)
            part = _train.evaluation_sums(e["test"], prior, h, van_reg, e.get("train"), noise_seed=seed, row_base=e["row0"])
            total = part if total is None else tuple(a + b for a, b in zip(total, part))
    return total, device


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
