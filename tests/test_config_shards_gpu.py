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
