"""BASELINE configs[3] and configs[4] at the size of ONE GPU's shard of the 8-GPU job (the 8-rank launch itself is the driver's):
what a rank computes per optimizer step on its rows, checked at that size.  The N>1 reduction (one packed all-reduce of the
per-rank sums) is covered by tests/test_dist_cpu.py and tests/test_dist_gpu.py."""
import os

import numpy as np
import pytest

from oracle import c_oracle as co

pytestmark = pytest.mark.gpu


def test_config3_rank_shard_mode_r():
    """configs[3]: bear_ref (stop prior), 1e9 contexts over 8 GPUs = 1.25e8 contexts per rank.  The rank's packed vector
    {sum LL, d/dh, d/dtau, d/dnw} from the reference-aware plan == oracle/bear_oracle.c over the WHOLE shard, == the streaming
    planned kernel, == the sum of two half-shards (what two ranks of a 16-rank job would add up)."""
    import torch
    from bear_amd import kernels
    dev = torch.device("cuda", 0)
    n = 125_000_000
    t = kernels.synth_counts(20211012, 3 * n, n, dev, want=("train", "ref"))     # rows [3n, 4n): the shard of rank 3
    args = (0.2, float(np.log(1 / 30)), float(-np.log(100)))
    plan = kernels.Plan(t["train"], 4, ref=t["ref"])
    got = kernels.dm_ref_planned(plan, t["ref"], *args).cpu().numpy()
    stream = kernels.dm_ref_planned(kernels.Plan(t["train"], 4), t["ref"], *args).cpu().numpy()
    assert np.allclose(got, stream, rtol=1e-12)
    cut = 62_500_004
    halves = sum(kernels.dm_ref_planned(kernels.Plan(t["train"][lo:hi], 4, ref=t["ref"][lo:hi]), t["ref"][lo:hi], *args).cpu().numpy()
                 for lo, hi in ((0, cut), (cut, n)))
    assert np.allclose(halves, got, rtol=1e-12)
    tr, rf = t["train"].cpu().numpy().view(np.uint32), t["ref"].cpu().numpy().view(np.uint32)
    want = co.dm_ref(tr, rf, *args, nthreads=min(os.cpu_count() or 4, 64))
    assert abs(got[0] - want[0]) <= 1e-10 * abs(want[0]), (got, want)
    assert np.allclose(got[1:], want[1:], rtol=1e-9, atol=1e-9 * np.abs(want).max())


def test_config4_rank_shard_cnn_step_and_heldout():
    """configs[4]: bear_net with the convolutional AR function, 1e8 contexts over 8 GPUs = 1.25e7 per rank, then held-out
    perplexity.  Through the host drivers on an in-memory k=13 table: three optimizer steps (fused CNN forward, planned DM kernel
    with gradient rows, fused backward, Adam) and the planned evaluation.  The first logged ELBO is the sum LL of the
    initial parameters recomputed from the kernels; the held-out sums of the table are those of its two halves."""
    import torch
    from bear_amd import ar_funcs, bear_net, dataloader, kernels
    dev = torch.device("cuda", 0)
    n, lag = 12_500_000, 13
    t = kernels.synth_counts(20211012, 0, n, dev, want=("train", "test"))
    counts = np.stack([t[k].cpu().numpy().view(np.uint32) for k in ("train", "test")])
    gen = torch.Generator(dev).manual_seed(5)
    kmers = np.frombuffer(b"ACGT", dtype=np.uint8)[torch.randint(0, 4, (n, lag), device=dev, generator=gen).cpu().numpy()]
    data = dataloader.CountDataset(kmers, counts, "dna", n)
    torch.manual_seed(11)
    losses = []
    params, h_signed, ar_func = bear_net.train(data.repeat(3), n, 3, 0, "dna", lag, ar_funcs.make_ar_func_cnn, {"filter_width": 8},
                                               0.01, "Adam", False, loss_save=losses)
    assert len(losses) == 3 and all(np.isfinite(losses)) and losses[2] > losses[0]      # the logged scalar is the ELBO (bear_net.py:303-309)
    # the first step's loss from the kernels, with the initial parameters (same seed -> same initialisation)
    torch.manual_seed(11)
    f0, p0 = ar_funcs.make_ar_func_cnn(lag, 4, filter_width=8, device=dev)
    codes = torch.from_numpy(np.searchsorted(np.frombuffer(b"ACGT", dtype=np.uint8), kmers).astype(np.int8)).to(dev)
    flat = torch.cat([q.detach().reshape(-1) for q in p0]).contiguous()
    prior, _ = kernels.cnn_forward(kernels.pack_kmers(codes), flat, lag, 8)
    tr = torch.from_numpy(counts[0].view(np.int32)).to(dev)
    ll = kernels.dm_prior_planned(kernels.Plan(tr, 5), prior, 0.0).cpu().numpy()[0]      # h_signed starts at 0
    assert abs(losses[0] - ll) <= 1e-9 * abs(ll), (losses[0], ll)
    # held-out evaluation: whole table == sum of its halves (a two-rank job adds the halves)
    h = torch.exp(h_signed).detach()
    van = np.array([0.1, 1.0, 10.0])
    whole = bear_net.evaluation(data, 0, 1, "dna", h, ar_func, van)
    assert np.isfinite(float(whole[3])) and 1.0 < float(whole[3]) < 6.0        # BEAR held-out perplexity of a 5-letter model
    cut = 6_000_000
    parts = [dataloader.CountDataset(kmers[lo:hi], np.ascontiguousarray(counts[:, lo:hi]), "dna", hi - lo) for lo, hi in ((0, cut), (cut, n))]
    res = [bear_net.evaluation(p, 0, 1, "dna", h, ar_func, van) for p in parts]
    # log likelihoods add; perplexity is exp(-ll / total held-out count): recombine through the counts
    total = float(counts[1].sum(dtype=np.float64))
    ll_parts = sum(float(r[0]) for r in res)
    assert abs(ll_parts - float(whole[0])) <= 1e-10 * abs(float(whole[0]))
    assert abs(np.exp(-ll_parts / total) - float(whole[3])) <= 1e-9 * float(whole[3])


def test_config3_full_size_on_one_gpu_equals_its_eight_shards():
    """configs[3] at its full size -- 1e9 contexts, 40 GB of count rows -- fits one MI355X: the whole table's packed vector equals
    the sum over the eight rank shards of the 8-GPU job, each through its own reference-aware plan (what the all-reduce adds)."""
    import torch
    from bear_amd import kernels
    dev = torch.device("cuda", 0)
    if torch.cuda.get_device_properties(dev).total_memory < 120e9:
        pytest.skip("needs ~60 GB of HBM")
    n_all, world = 1_000_000_000, 8
    args = (0.2, float(np.log(1 / 30)), float(-np.log(100)))
    t = kernels.synth_counts(20211012, 0, n_all, dev, want=("train", "ref"))
    whole = kernels.dm_ref_planned(kernels.Plan(t["train"], 4, ref=t["ref"]), t["ref"], *args).cpu().numpy()
    n = n_all // world
    parts = np.zeros(4)
    for r in range(world):
        tr, rf = t["train"][r * n:(r + 1) * n], t["ref"][r * n:(r + 1) * n]
        parts += kernels.dm_ref_planned(kernels.Plan(tr, 4, ref=rf), rf, *args).cpu().numpy()
    assert np.all(np.isfinite(whole)) and np.allclose(parts, whole, rtol=1e-12)


def test_dense_full_size():
    """SURVEY section 8d's dense stress distribution (lambda = 1e4 ... 3e5: what the reference's only real table, data/ysd1_*.tsv, looks
    like) at 1e7 rows: every item takes the Stirling path and the in-tile large-count lists overflow to the plan's global lists.
    The planned kernels == the unplanned ones on the whole table (modes N and R, gradient rows both ways), and a chunk from the
    middle of the table == oracle/bear_oracle.c (through a plan of its own)."""
    import torch
    from bear_amd import kernels
    dev = torch.device("cuda", 0)
    n = 10_000_000
    t = kernels.synth_counts(20211012, 0, n, dev, dense=True, want=("train", "ref"))
    assert int(t["train"].max()) > 50_000                      # the large-count branch, not the product path
    prior = kernels.synth_prior(20211012, 0, n, dev)
    args = (-0.3, float(np.log(1 / 30)), float(-np.log(100)))
    plan_n = kernels.Plan(t["train"], 5)
    got, grad = kernels.dm_prior_planned(plan_n, prior, args[0], want_grad=True)
    got_a, grad_a = kernels.dm_prior_planned(plan_n, prior, args[0], want_grad=True, normalized=True)
    plain, grad_u = kernels.dm_prior(t["train"], prior, args[0], want_grad=True)
    for a in (got, got_a):
        assert torch.allclose(a, plain, rtol=1e-11, atol=0), (a, plain)
    scale = float(grad_u.abs().max())
    assert float((grad - grad_u).abs().max()) <= 1e-10 * scale and float((grad_a - grad_u).abs().max()) <= 1e-10 * scale
    # the plan's dense form (bear_plan_create_auto: this table keeps nothing per item) == the sorted form and the unplanned kernels
    plan_d = kernels.Plan(t["train"], 5, rows_if_dense=True)
    assert plan_d.rowwise and plan_d.nbytes < 64 * 1024 and plan_n.nbytes > 50 * n
    got_d, grad_d = kernels.dm_prior_planned(plan_d, prior, args[0], want_grad=True, normalized=True)
    assert torch.allclose(got_d, plain, rtol=1e-11, atol=0) and torch.allclose(kernels.dm_prior_planned(plan_d, prior, args[0]), plain, rtol=1e-11, atol=0)
    assert float((grad_d - grad_u).abs().max()) <= 1e-10 * scale
    del grad, grad_a, grad_u, grad_d, plan_d
    got_r = kernels.dm_ref_planned(kernels.Plan(t["train"], 4, ref=t["ref"]), t["ref"], *args)
    stream_r = kernels.dm_ref_planned(kernels.Plan(t["train"], 4), t["ref"], *args)
    plain_r = kernels.dm_ref(t["train"], t["ref"], *args)
    assert torch.allclose(got_r, plain_r, rtol=1e-10, atol=0) and torch.allclose(stream_r, plain_r, rtol=1e-10, atol=0), (got_r, stream_r, plain_r)
    lo, m = 4_000_004, 1_000_000
    tr, rf, pr = (x[lo:lo + m].contiguous() for x in (t["train"], t["ref"], prior))
    sub, g = kernels.dm_prior_planned(kernels.Plan(tr, 5), pr, args[0], want_grad=True)
    sub_d, g_d = kernels.dm_prior_planned(kernels.Plan(tr, 5, rows_if_dense=True), pr, args[0], want_grad=True)      # (the dense form against the oracle)
    sub_r = kernels.dm_ref_planned(kernels.Plan(tr, 4, ref=rf), rf, *args).cpu().numpy()
    trh, rfh = tr.cpu().numpy().view(np.uint32), rf.cpu().numpy().view(np.uint32)
    want, want_g = co.dm_prior(trh, pr.cpu().numpy(), args[0], want_grad=True, nthreads=min(os.cpu_count() or 4, 64))
    want_r = co.dm_ref(trh, rfh, *args, nthreads=min(os.cpu_count() or 4, 64))
    sub = sub.cpu().numpy()
    mass = float(co.dm_prior_mass(trh, pr.cpu().numpy(), args[0], nthreads=min(os.cpu_count() or 4, 64)))   # L1 mass of d/dh: its error scale
    assert abs(sub[0] - want[0]) <= 1e-10 * abs(want[0]) and abs(sub[1] - want[1]) <= 1e-11 * mass, (sub, want, mass)
    assert np.abs(g.cpu().numpy() - want_g).max() <= 1e-9 * np.abs(want_g).max()
    sub_d = sub_d.cpu().numpy()
    assert abs(sub_d[0] - want[0]) <= 1e-10 * abs(want[0]) and abs(sub_d[1] - want[1]) <= 1e-11 * mass, (sub_d, want, mass)
    assert np.abs(g_d.cpu().numpy() - want_g).max() <= 1e-9 * np.abs(want_g).max()
    assert abs(sub_r[0] - want_r[0]) <= 1e-10 * abs(want_r[0]), (sub_r, want_r)
    mass_r = co.dm_ref_mass(trh, rfh, *args, nthreads=min(os.cpu_count() or 4, 64))
    assert np.all(np.abs(sub_r[1:] - want_r[1:]) <= 1e-11 * mass_r), (sub_r, want_r, mass_r)


def test_baseline_configs_module_small():
    """scripts/baseline_configs.py (bench.py's also.baseline_configs / also.dense_table) at 1/50 of the configs' sizes: every config
    steps through its host driver's own loop, the loop reports its event-timed steps, the dense table's checks hold."""
    import sys
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    import baseline_configs
    dev = torch.device("cuda", 0)
    out = baseline_configs.measure_configs(dev, shrink=50)
    ents = [v for k, v in out.items() if k.startswith("configs[")]
    assert len(ents) == 6             # the four configs + configs[4]'s rank piece cut both ways (contiguous rows / k-mer range)
    for e in ents:
        assert e["us_per_step"] > 0 and e["steps_timed"] > 0 and np.isfinite(e["elbo_last_step"])
        assert e["elbo_last_step"] > e["elbo_first_step"]                 # the logged scalar is the ELBO: the optimizer raises it
        assert abs(e["contexts_per_s"] - e["rows"] / (e["us_per_step"] * 1e-6)) <= 1e-6 * e["contexts_per_s"]
    cnn = [v for k, v in out.items() if "configs[4] bear_net" in k][0]
    assert 1.0 < cnn["heldout_perplexity_bear"] < 6.0
    d = baseline_configs.measure_dense(dev, n=1_000_000, check_rows=200_000)
    c = d["check"]
    assert c["planned_vs_unplanned_elbo_rel"] <= 1e-11 and c["mode_N_elbo_rel_err"] <= 1e-10 and c["mode_R_elbo_rel_err"] <= 1e-10
    assert c["gradient_rows_max_err_over_largest"] <= 1e-9 and c["mode_N_dh_err_over_l1_mass"] <= 1e-11 and c["mode_R_grad_max_err_over_l1_mass"] <= 1e-11
