"""Reference-based BEAR / AR models: host mirror of ``bear_model/bear_ref.py`` over the HIP kernels.

Same entry points and argument meaning as the reference: ``train`` (bear_ref.py:262-389),
``evaluation`` (:453-539), ``change_scope_params`` (:166-204).  With the stop net function
(``ar_funcs.make_ar_func_stop``, the reference's bear_stop_*.cfg configurations) a training step is one
launch of ``bear_dm_ref[_plan]_f64`` per batch shard plus one all-reduce of 4 doubles.
"""
import os
import warnings

import numpy as np
import torch

from . import _train, ar_funcs as _ar_funcs, core, dist, kernels

epsilon = core.epsilon


def _counts_to_probs(ref_counts, tau, alphabet_size, dtype=torch.float64):
    """bear_ref.py:9-33."""
    norm = ref_counts / ref_counts.abs().sum(-1, keepdim=True)
    shape = torch.tensor(np.r_[np.ones(alphabet_size), 0], dtype=dtype, device=ref_counts.device)
    return (1 / alphabet_size) * shape + torch.exp(-tau) * (norm - (1 / alphabet_size) * shape)


def _make_ref_ar_func(lag, alphabet_size, make_net_func, af_kwargs, dtype=torch.float64, device=None):
    """bear_ref.py:36-69: params = [tau_signed, net_weight_signed] + net params."""
    net_weight_signed = torch.tensor(-np.log(100), dtype=dtype, device=device, requires_grad=True)
    tau_signed = torch.tensor(np.log(1 / 30), dtype=dtype, device=device, requires_grad=True)
    net_func, ar_func_params = make_net_func(lag, alphabet_size, **af_kwargs, dtype=dtype, device=device)

    def ar_func(kmer_seqs, ref_counts):
        nw = torch.exp(net_weight_signed)
        tau = torch.exp(tau_signed)
        return (nw * net_func(kmer_seqs) + _counts_to_probs(ref_counts, tau, alphabet_size, dtype=dtype)) / (nw + 1)
    ar_func.net_is_stop = bool(getattr(net_func, "is_stop", False))
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
    if dtype != torch.float64:
        raise NotImplementedError("the HIP kernels compute in float64 (precision = float64 is the reference's recommendation)")
    device = _train.require_device()
    alphabet_size = len(core.alphabets_tf[alphabet]) - 1
    if params_restart is None:
        params, h_signed, ar_func = _create_params(lag, alphabet_size, make_ar_func, af_kwargs, dtype, device)
    else:
        params, h_signed, ar_func = change_scope_params(lag, alphabet_size, make_ar_func, af_kwargs, params_restart, dtype, device)
    tau_signed, nu_signed = params[1], params[2]
    optimizer = _train.make_optimizer(optimizer_name, params, learning_rate)
    if not ar_func.net_is_stop:
        return _train_general(data, num_kmers, params, h_signed, ar_func, optimizer, train_ar, acc_steps, writer, loss_save,
                              ds_loc, ds_loc_ref, device)
    res = _train.ResidentBatches(data, {"train": ds_loc, "ref": ds_loc_ref}, device)
    n_batches = len(res.batches)
    if (1 <= n_batches <= _train.GRAPH_MAX_BATCHES and acc_steps == 1 and optimizer_name == "Adam"
            and dist.world()[1] == 1 and all(b["rows"] > 0 for b in res.batches) and data.repeats > 1
            and not os.environ.get("BEAR_AMD_NO_GRAPH")):
        try:
            return _train_stop_graph(res, data.repeats, num_kmers, params, h_signed, ar_func, learning_rate, train_ar, loss_save, device,
                                     writer)
        except RuntimeError as err:     # stream capture unavailable: the eager loop below runs the same kernels (parameters untouched so far)
            warnings.warn(f"HIP-graph capture of the training step failed ({err}); using the eager loop")
    acc = torch.zeros(3, dtype=torch.float64)
    loss, step = 0.0, 1
    out = torch.zeros(4, dtype=torch.float64, device=device)
    for _ in range(data.repeats):
        for k in range(n_batches):
            e = res.batches[k]
            hs, ts, ns = h_signed.item(), tau_signed.item(), nu_signed.item()
            if e["rows"] == 0:
                out.zero_()
            else:
                kernels.dm_ref_planned(res.plan(k, "train", 4), e["ref"], hs, ts, ns, out=out, train_ar=train_ar)
            dist.allreduce_sum_(out)                                   # replaces bear_ref.py:358 + the grad sum of :346-350
            scaled = (-(num_kmers / e["global_rows"]) * out).cpu()     # loss = -(num_kmers / B) sum LL, bear_ref.py:252-253
            loss += scaled[0].item()
            acc += scaled[1:]
            if step % acc_steps == 0:
                if writer is not None:
                    writer.add_scalar("elbo", -loss / acc_steps, step)
                if loss_save is not None:
                    loss_save.append(-loss / acc_steps)
                grads = [None if train_ar else acc[0].clone(), acc[1].clone(), acc[2].clone()]  # AR mode: h gets no gradient
                optimizer.apply_gradients(grads)
                acc.zero_()
                loss = 0.0
            step += 1
    return params, h_signed, ar_func


def _train_stop_graph(res, steps, num_kmers, params, h_signed, ar_func, learning_rate, train_ar, loss_save, device, writer=None):
    """Resident batches, Adam, one GPU: the whole optimizer step (constants from the parameters, planned kernel,
    finalize, Adam) is enqueued once, captured in a HIP graph and replayed ``steps`` times -- no host round trip per step
    (the reference traces its step with tf.function, bear_ref.py:207; at 1365 contexts the eager loop is launch- and
    synchronisation-bound).  Parameters and optimizer state live in device memory; the losses come back once at the end."""
    plans = [res.plan(k, "train", 4) for k in range(len(res.batches))]       # built before the capture (plan creation allocates)
    theta = torch.stack([p.detach().reshape(()) for p in params[:3]]).to(device=device, dtype=torch.float64).contiguous()
    m, v = torch.zeros(3, dtype=torch.float64, device=device), torch.zeros(3, dtype=torch.float64, device=device)
    t = torch.zeros(1, dtype=torch.float64, device=device)
    out = torch.zeros(4, dtype=torch.float64, device=device)
    loss_buf = torch.zeros(steps * len(res.batches), dtype=torch.float64, device=device)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):                                            # one epoch: the batches in order, one optimizer step each
        for e, plan in zip(res.batches, plans):
            kernels.ref_train_step(plan, e["ref"], theta, m, v, t, learning_rate, -(num_kmers / e["global_rows"]), out, loss_buf,
                                   train_ar=train_ar)
    for _ in range(steps):
        graph.replay()
    torch.cuda.synchronize()
    with torch.no_grad():
        for p, val in zip(params[:3], theta.cpu()):
            p.copy_(val)
    losses = loss_buf.cpu().tolist()
    if loss_save is not None:
        loss_save.extend(losses)
    if writer is not None:                       # the per-step scalars of bear_net.py:285-287 / bear_ref.py:353-355, written after the replay
        for i, val in enumerate(losses):
            writer.add_scalar("elbo", val, i + 1)
    return params, h_signed, ar_func


def _train_general(data, num_kmers, params, h_signed, ar_func, optimizer, train_ar, acc_steps, writer, loss_save,
                   ds_loc, ds_loc_ref, device):
    """bear_ref.train with a parametrised net function (linear, cnn; bear_ref.py:63-68): the mixed prior rows
    ``(nw net(kmers) + jukes_cantor(ref, tau)) / (nw + 1)`` are formed by torch ops, the planned kernel returns
    the ELBO, d/dh and the gradient rows, and autograd carries the rows back to tau, the net weight and the
    net parameters -- the same loop as bear_net.train with two more parameters."""
    rest = params[1:]
    res = _train.ResidentBatches(data, {"train": ds_loc, "ref": ds_loc_ref}, device, want_codes=True)
    acc = [torch.zeros_like(p) for p in params]
    loss, step = 0.0, 1
    out = torch.zeros(2, dtype=torch.float64, device=device)
    for _ in range(data.repeats):
        for k, e in enumerate(res.batches):
            scale = -(num_kmers / e["global_rows"])
            for p in rest:
                p.grad = None
            if e["rows"]:
                if "ref_in" not in e:
                    e["ref_in"] = _ref_input(e["ref"])
                prior = ar_func(e["codes"], e["ref_in"]).contiguous()
                _, grad_rows = kernels.dm_prior_planned(res.plan(k, "train", 5), prior.detach(), h_signed.item(), out=out,
                                                        want_grad=True, train_ar=train_ar)
                prior.backward(scale * grad_rows)
            else:
                out.zero_()
            flat, unpack = dist.pack([out] + [p.grad if p.grad is not None else torch.zeros_like(p) for p in rest])
            dist.allreduce_sum_(flat)
            parts = unpack(flat)
            loss += scale * parts[0][0].item()
            if not train_ar:
                acc[0] += scale * parts[0][1]
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


def evaluation(data, ds_loc_train, ds_loc_test, ds_loc_ref, alphabet, h, ar_func, van_reg, dtype=torch.float64,
               seed=0):
    """bear_ref.evaluation (bear_ref.py:453-539) -> (ll_ear, ll_arm, ll_van, perp_ear, perp_arm, perp_van,
    acc_ear, acc_arm, acc_van).  ``ds_loc_train = -1``: no conditioning on training counts.  The training
    column is used for conditioning (the reference reads the reference column there by mistake,
    bear_ref.py:397 vs bear_net.py:327)."""
    device = _train.require_device()
    use_train = ds_loc_train >= 0
    cols = {"test": ds_loc_test, "ref": ds_loc_ref}
    if use_train:
        cols["train"] = ds_loc_train
    res = _train.ResidentBatches(data, cols, device, want_codes=True)
    hv = float(torch.as_tensor(h).item()) if np.ndim(torch.as_tensor(h).detach().cpu().numpy()) == 0 else torch.as_tensor(h).detach().cpu().numpy()
    total = None
    with torch.no_grad():
        for e in res.batches:
            prior = ar_func(e["codes"], _ref_input(e["ref"], dtype)) if e["rows"] else torch.zeros((0, 5), dtype=dtype, device=device)
            prior = prior.expand(e["rows"], 5).contiguous()
            part = _train.evaluation_sums(e["test"], prior, hv, van_reg, e.get("train"), noise_seed=seed, row_base=e["row0"])
            total = part if total is None else tuple(a + b for a, b in zip(total, part))
    return _train.reduce_evaluation(total, device, np.ndim(hv) == 0)
