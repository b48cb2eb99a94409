"""BASELINE.json configs[1..4], each at the size the config states, through the HOST DRIVERS' own step loop
(bear_ref.train / bear_net.train -> _train.run_device_steps: reduce + Adam per step, one period captured into a HIP graph and
replayed), timed with HIP events around the replayed steps (_train.LAST_RUN: the last three quarters of the loop, behind the
clock ramp), plus configs[4]'s held-out evaluation -- and the dense stress table of SURVEY section 8d (lambda = 1e4 ... 3e5, what the
reference's only real table, data/ysd1_lag_5_file_0_preshuf.tsv, looks like: every item on the Stirling path).
bench.py reports both as also.baseline_configs / also.dense_table; stand-alone:
    python scripts/baseline_configs.py [configs|dense|all]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

SEED = 20211012


def _table(n, lag, dev, cols, row0=0):
    """Host arrays of rows [row0, row0 + n) of the synthetic table (what a parsed count file is): kmers uint8 [n, lag], counts
    uint32 [len(cols), n, 5]."""
    from bear_amd import kernels
    t = kernels.synth_counts(SEED, row0, n, dev, want=cols)
    counts = np.stack([t[k].cpu().numpy().view(np.uint32) for k in cols])
    del t
    # DISTINCT contexts (row index -> a fixed bijection of [0, 4^lag), kernels.synth_kmer_ids) in a scrambled order, as the
    # rows of a pre-shuffled count file are; rounds 1-5 drew them with replacement
    codes = kernels.synth_kmer_codes(SEED, row0, n, lag, dev)
    kmers = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)[codes.long()].cpu().numpy()
    del codes
    torch.cuda.empty_cache()
    return kmers, counts


def _train_config(mod, data, n, lag, make, kw, steps, extra=()):
    """One train() call of `steps` optimizer steps; the loop's own event times."""
    from bear_amd import _train
    loss = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = mod.train(data.repeat(steps), n, steps, 0, *extra, "dna", lag, make, kw, 0.01, "Adam", False, loss_save=loss)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    lr = dict(_train.LAST_RUN)
    ms = lr["timed_ms"] / max(lr["timed_steps"], 1)
    return out, {"rows": n, "lag": lag, "optimizer_steps": steps, "us_per_step": ms * 1e3, "contexts_per_s": n / (ms * 1e-3),
                 "steps_timed": lr["timed_steps"], "captured_in_hip_graph": bool(lr["graph"]), "steps_per_replay": lr["period"],
                 "train_call_wall_s": wall, "elbo_first_step": loss[0], "elbo_last_step": loss[-1]}


def measure_configs(dev, shrink=1, only=None):
    """also.baseline_configs.  `shrink` > 1 divides every size (tests); `only` = 1 ... 4: that config alone (one config per profiled
    process: scripts/profile_round.sh)."""
    from bear_amd import ar_funcs, bear_net, bear_ref, dataloader
    out = {"note": "one optimizer step = this rank's reduce (fused kernels) + Adam (bear_train_apply_f64), enqueued by bear_ref.train / "
                   "bear_net.train themselves (run_device_steps: a period of steps captured into a HIP graph and replayed); us_per_step = "
                   "HIP-event time of the last three quarters of the loop / its steps; one GPU, so configs[3] / [4] run one rank's "
                   "shard of the 8-GPU job (no all-reduce in the figure)"}
    want = lambda k: only is None or only == k
    if want(1) or want(2):
        n1 = 10_000_000 // shrink
        kmers, counts = _table(n1, 13, dev, ("train", "test", "ref"))
        data = dataloader.CountDataset(kmers, counts, "dna", n1)
        if want(1):
            _, out["configs[1] bear_ref, stop prior, k=13, 1e7 contexts"] = _train_config(
                bear_ref, data, n1, 13, ar_funcs.make_ar_func_stop, {}, 2000, extra=(2,))
        if want(2):
            _, out["configs[2] bear_net, linear AR prior, k=13, 1e7 contexts"] = _train_config(
                bear_net, data, n1, 13, ar_funcs.make_ar_func_linear, {}, 600)
        del data, kmers, counts
    if want(4):
        _config4(dev, shrink, out)
    if want(3):
        from bear_amd import ar_funcs as af
        n3 = 125_000_000 // shrink
        kmers, counts = _table(n3, 15, dev, ("train", "ref"), row0=3 * n3)
        data = dataloader.CountDataset(kmers, counts, "dna", n3)
        _, out["configs[3] bear_ref, k=15, rank shard 1.25e8 of 1e9 contexts"] = _train_config(
            bear_ref, data, n3, 15, af.make_ar_func_stop, {}, 400, extra=(1,))
    return out


def _config4(dev, shrink, out):
    from bear_amd import ar_funcs, bear_net, dataloader
    # (a 1e8-context table does not exist at k = 13 -- 4^13 = 6.7e7 -- so "one rank's 1.25e7 rows of the pre-shuffled table" is 1.25e7
    # distinct 13-mers, a random 19 % of all of them: the density such a shard has whatever the table's size)
    n4 = 12_500_000 // shrink
    kmers, counts = _table(n4, 13, dev, ("train", "test"))
    data = dataloader.CountDataset(kmers, counts, "dna", n4)
    (params, h_signed, ar_func), ent = _train_config(bear_net, data, n4, 13, ar_funcs.make_ar_func_cnn, {"filter_width": 8}, 24)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = bear_net.evaluation(data, 0, 1, "dna", torch.exp(h_signed).detach(), ar_func, np.array([0.1, 1.0, 10.0]))
    torch.cuda.synchronize()
    ent.update(heldout_evaluation_wall_s=time.perf_counter() - t0, heldout_perplexity_bear=float(res[3]),
               heldout_perplexity_ar=float(res[4]), heldout_perplexity_bmm=[float(x) for x in np.atleast_1d(res[5])])
    out["configs[4] bear_net, CNN AR prior, k=13, rank shard 1.25e7 of 1e8 contexts, + held-out evaluation"] = ent
    del data, kmers, counts, params, ar_func
    # The same job on a table that exists -- 6.0e7 distinct 13-mers over 8 ranks, 7.5e6 contexts each -- with a rank's piece cut both
    # ways: a contiguous row piece of the pre-shuffled table (a random eighth of its k-mers) and the piece `shard = kmer` deals
    # (dataloader.KmerDealtDataset: the rank's RANGE of the sorted k-mers, here the first eighth).  The synthetic counts do not depend
    # on the contexts, so rows [0, 7.5e6) of the count table serve both.
    from bear_amd import kernels
    n_tab, n_piece = 60_000_000 // shrink, 7_500_000 // shrink
    t = kernels.synth_counts(SEED, 0, n_piece, dev, want=("train", "test"))
    counts = np.stack([t[k].cpu().numpy().view(np.uint32) for k in ("train", "test")])
    del t
    ids = kernels.synth_kmer_ids(SEED, 0, n_tab, 13, dev)
    letters = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    shifts = torch.arange(24, -1, -2, dtype=torch.int64, device=dev)
    for label, mine in (("contiguous row piece", ids[:n_piece]), ("piece dealt by k-mer range", torch.sort(ids).values[:n_piece])):
        kmers = letters[((mine[:, None] >> shifts[None, :]) & 3)].cpu().numpy()
        data = dataloader.CountDataset(kmers, counts, "dna", n_piece)
        _, e2 = _train_config(bear_net, data, n_piece, 13, ar_funcs.make_ar_func_cnn, {"filter_width": 8}, 24)
        out["configs[4] on a 6e7-context table over 8 ranks, 7.5e6 contexts per rank: " + label] = e2
        del data, kmers
        torch.cuda.empty_cache()
    del ids, counts


def measure_dense(dev, n=20_000_000, check_rows=2_000_000):
    """also.dense_table: the planned kernels on synth_counts(dense=True) -- every count in the thousands: every item takes the
    Stirling path and the in-tile large-count lists (PLN_HCAP) overflow to the plan's global lists -- with the ELBO / gradients
    of the timed launches checked against oracle/bear_oracle.c on the first `check_rows` rows (a plan of their own)."""
    from bear_amd import kernels
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import c_oracle as co       # checker only
    t = kernels.synth_counts(SEED, 0, n, dev, dense=True, want=("train", "ref"))
    prior = kernels.synth_prior(SEED, 0, n, dev)
    h_s, tau_s, nu_s = 0.0, float(np.log(1 / 30)), float(-np.log(100))

    def timed(fn, reps=5):
        fn()
        torch.cuda.synchronize()
        best = 1e30
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / reps)
        return best
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    # mode N: the plan a caller of the mode-N entry points gets (bear_plan_create_auto: this table takes the plan's DENSE form -- nothing
    # kept per item, count and prior rows streamed, a context per thread; round 6); the sorted encoding of the same table next to it
    plan_n, plan_r = kernels.Plan(t["train"], 5, rows_if_dense=True), kernels.Plan(t["train"], 4, ref=t["ref"])
    torch.cuda.synchronize()
    build_s = time.perf_counter() - t0
    ms_n = timed(lambda: kernels.dm_prior_planned(plan_n, prior, h_s))
    ms_g = timed(lambda: kernels.dm_prior_planned(plan_n, prior, h_s, want_grad=True, normalized=True))
    plan_s = kernels.Plan(t["train"], 5)
    ms_n_sorted = timed(lambda: kernels.dm_prior_planned(plan_s, prior, h_s))
    ms_g_sorted = timed(lambda: kernels.dm_prior_planned(plan_s, prior, h_s, want_grad=True, normalized=True))
    got_s = kernels.dm_prior_planned(plan_s, prior, h_s).cpu().numpy()
    sorted_bytes = plan_s.nbytes / n
    del plan_s
    ms_r = timed(lambda: kernels.dm_ref_planned(plan_r, t["ref"], h_s, tau_s, nu_s))
    ms_u = timed(lambda: kernels.dm_prior(t["train"], prior, h_s), reps=2)
    # planned == unplanned on the whole table, both == the C oracle on the first check_rows rows
    got_n = kernels.dm_prior_planned(plan_n, prior, h_s).cpu().numpy()
    got_u = kernels.dm_prior(t["train"], prior, h_s)[0].cpu().numpy()
    m = min(n, check_rows)
    tr, rf, pr = (x[:m].contiguous() for x in (t["train"], t["ref"], prior))
    sub_n, grad = kernels.dm_prior_planned(kernels.Plan(tr, 5, rows_if_dense=True), pr, h_s, want_grad=True, normalized=True)
    sub_r = kernels.dm_ref_planned(kernels.Plan(tr, 4, ref=rf), rf, h_s, tau_s, nu_s).cpu().numpy()
    trh, rfh, prh = tr.cpu().numpy().view(np.uint32), rf.cpu().numpy().view(np.uint32), pr.cpu().numpy()
    cores = os.cpu_count() or 1
    want_n, want_g = co.dm_prior(trh, prh, h_s, want_grad=True, nthreads=cores)
    want_r = co.dm_ref(trh, rfh, h_s, tau_s, nu_s, nthreads=cores)
    sub_n = sub_n.cpu().numpy()
    mass_n = float(co.dm_prior_mass(trh, prh, h_s, nthreads=cores))       # L1 masses: the scale a gradient's rounding errors live on
    mass_r = co.dm_ref_mass(trh, rfh, h_s, tau_s, nu_s, nthreads=cores)
    gerr = float(np.abs(grad.cpu().numpy() - want_g).max() / np.abs(want_g).max())
    nbytes = {"net": plan_n.nbytes / n, "ref": plan_r.nbytes / n}
    return {"contexts": n, "distribution": "synth_counts(dense=True): row rate lambda = 1e4 ... 3e5 (SURVEY section 8d), counts up to ~3e5",
            "plan_build_s": build_s, "plan_bytes_per_context": nbytes, "mode_N_plan_is_the_dense_form": bool(plan_n.rowwise),
            "plan_bytes_per_context_sorted_form": sorted_bytes,
            "mode_N": {"kernel_ms": ms_n, "contexts_per_s": n / (ms_n * 1e-3), "credited_GBps_at_60_B": n * 60 / (ms_n * 1e-3) / 1e9,
                       "frac_credited": n * 60 / (ms_n * 1e-3) / 1e9 / 8000.0, "unplanned_kernel_ms": ms_u,
                       "kernel_ms_sorted_form": ms_n_sorted,
                       "dense_vs_sorted_form_elbo_rel": float(abs(got_s[0] - kernels.dm_prior_planned(plan_n, prior, h_s).cpu().numpy()[0]) / abs(got_s[0]))},
            "mode_N_with_gradient_rows": {"kernel_ms": ms_g, "contexts_per_s": n / (ms_g * 1e-3), "kernel_ms_sorted_form": ms_g_sorted},
            "mode_R": {"kernel_ms": ms_r, "contexts_per_s": n / (ms_r * 1e-3)},
            "check": {"planned_vs_unplanned_elbo_rel": float(abs(got_n[0] - got_u[0]) / abs(got_u[0])),
                      "planned_vs_unplanned_dh_rel": float(abs(got_n[1] - got_u[1]) / abs(got_u[1])),
                      "oracle_rows": m, "mode_N_elbo_rel_err": float(abs(sub_n[0] - want_n[0]) / abs(want_n[0])),
                      "mode_N_dh_err_over_l1_mass": float(abs(sub_n[1] - want_n[1]) / mass_n),
                      "gradient_rows_max_err_over_largest": gerr,
                      "mode_R_elbo_rel_err": float(abs(sub_r[0] - want_r[0]) / abs(want_r[0])),
                      "mode_R_grad_max_err_over_l1_mass": float((np.abs(sub_r[1:] - want_r[1:]) / mass_r).max()),
                      "note": "the whole table: planned == unplanned kernels; the first oracle_rows rows: HIP == oracle/bear_oracle.c"}}


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    dev = torch.device("cuda", 0)
    if what in ("configs", "all"):
        print(json.dumps(measure_configs(dev), indent=1))
    if what in ("configs1", "configs2", "configs3", "configs4"):
        print(json.dumps(measure_configs(dev, only=int(what[-1])), indent=1))
    if what in ("dense", "all"):
        print(json.dumps(measure_dense(dev), indent=1))
