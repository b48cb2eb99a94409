"""BEAR_AMD_DETERMINISTIC=1: parameter gradients that are bit-identical from run to run (SURVEY section 5 asked for the option;
include/bear_hip.h says what it does).  Linear step: fixed-point gradient tables -- d/d mat is also identical for any sharding of
the batch that uses the same bound.  Convolutional step: one wave per block."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _sorted_table(n, lag, dev, seed=5, fixed=0):
    import torch
    from bear_amd import kernels
    t = kernels.synth_counts(20211012, 0, n, dev, want=("train",))["train"]
    gen = torch.Generator(dev).manual_seed(seed)
    codes = torch.randint(0, 4, (n, lag), dtype=torch.int8, device=dev, generator=gen)
    codes[:, :fixed] = 1             # (a table dense in k-mer space: prefixes repeat)
    key = torch.zeros(n, dtype=torch.int64, device=dev)
    for l in range(lag):
        key = key * 6 + codes[:, l].to(torch.int64)
    codes = codes[torch.argsort(key)].contiguous()
    return t, codes


@pytest.mark.parametrize("train_ar", [False, True])
def test_linear_step_is_bit_reproducible_and_shard_invariant(train_ar, monkeypatch):
    import torch
    from bear_amd import kernels
    dev = torch.device("cuda", 0)
    n, lag = 3_000_000, 13
    t, codes = _sorted_table(n, lag, dev)
    idx = kernels.linear_index(kernels.pack_kmers(codes), lag)
    mat = 0.3 * torch.randn(lag, 5, 5, dtype=torch.float64, device=dev, generator=torch.Generator(dev).manual_seed(3))
    plan = kernels.Plan(t, 5)
    total, bound = plan.count_total()
    assert total == [float(t.to(torch.int64).sum()), float((t != 0).sum()), float(t.max())] and bound == total
    monkeypatch.delenv("BEAR_AMD_DETERMINISTIC", raising=False)
    ref_out, ref_g = (x.clone() for x in kernels.dm_linear(plan, idx, mat, -0.2, train_ar=train_ar))
    monkeypatch.setenv("BEAR_AMD_DETERMINISTIC", "1")
    runs = []
    for paired in (False, True):
        if paired:
            assert plan.pair_contexts(idx, lag)
        for _ in range(3):
            o, g = kernels.dm_linear(plan, idx, mat, -0.2, train_ar=train_ar)
            runs.append((o.clone(), g.clone()))
    for o, g in runs:
        assert torch.equal(g, runs[0][1])                    # plain and paired lists, every run: the same bits
        assert torch.allclose(o, ref_out, rtol=1e-13, atol=0)
    scale = float(ref_g.abs().max())
    # one rounding to bound 2^-62 per context and letter: far below the rounding of the fp64 sums it is compared with
    assert float((runs[0][1] - ref_g).abs().max()) <= 1e-12 * scale
    assert float(runs[0][1].sum(-1).abs().max()) <= 1e-9 * scale       # a softmax gradient sums to zero over the letters
    # the same batch cut into 2 / 4 / 7 row shards (ragged), every shard on the batch's bound: the shards' INTEGER sums add up to
    # the batch's exactly; each becomes a double on its own (one rounding each), so the doubles agree to the last bits
    for pieces in (2, 4, 7):
        cuts = [0] + [int(n * (k + 1) / pieces) // 4 * 4 + (3 if k == 0 else 0) for k in range(pieces - 1)] + [n]
        tot = torch.zeros_like(ref_g)
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            ts, ix = t[lo:hi].clone(), idx[lo:hi].clone()        # (fresh, 16-byte aligned buffers)
            ps = kernels.Plan(ts, 5)
            ps.set_count_bound(total)
            if pieces != 4:                                       # paired and plain shards mixed: the same integers either way
                ps.pair_contexts(ix, lag)
            tot += kernels.dm_linear(ps, ix, mat, -0.2, train_ar=train_ar)[1]
        assert float((tot - runs[0][1]).abs().max()) <= 4e-15 * scale, pieces
    with pytest.raises(Exception):
        plan.set_count_bound([total[0] / 2, total[1], total[2]])    # a bound below the plan's own total is refused


def test_linear_step_default_mode_unchanged(monkeypatch):
    """Without the switch the tables hold doubles (LDS floating-point atomics): results agree with the deterministic ones to
    rounding, and the switch is read per call."""
    import torch
    from bear_amd import kernels
    dev = torch.device("cuda", 0)
    n, lag = 400_000, 9
    t, codes = _sorted_table(n, lag, dev, seed=9)
    idx = kernels.linear_index(kernels.pack_kmers(codes), lag)
    mat = 0.1 * torch.randn(lag, 5, 5, dtype=torch.float64, device=dev, generator=torch.Generator(dev).manual_seed(4))
    plan = kernels.Plan(t, 5)
    monkeypatch.setenv("BEAR_AMD_DETERMINISTIC", "1")
    _, g1 = (x.clone() for x in kernels.dm_linear(plan, idx, mat, 0.1))
    monkeypatch.setenv("BEAR_AMD_DETERMINISTIC", "0")
    _, g0 = (x.clone() for x in kernels.dm_linear(plan, idx, mat, 0.1))
    assert float((g1 - g0).abs().max()) <= 1e-12 * float(g0.abs().max())


def test_cnn_step_is_bit_reproducible(monkeypatch):
    import torch
    from bear_amd import ar_funcs, kernels
    dev = torch.device("cuda", 0)
    n, lag, fw = 300_000, 13, 8
    t, codes = _sorted_table(n, lag, dev, seed=7, fixed=4)
    keep = (t != 0).any(dim=1).nonzero().squeeze(1)
    t, codes = t.index_select(0, keep).contiguous(), codes.index_select(0, keep).contiguous()
    n = t.shape[0]
    packed = kernels.pack_kmers(codes)
    _, params = ar_funcs.make_ar_func_cnn(lag, 4, filter_width=fw, device=dev, generator=torch.Generator(dev).manual_seed(10))
    theta = torch.cat([torch.zeros(1, dtype=torch.float64, device=dev)] + [q.detach().reshape(-1) for q in params]).contiguous()
    plan = kernels.Plan(t, 5)
    monkeypatch.delenv("BEAR_AMD_DETERMINISTIC", raising=False)
    bufs = kernels.cnn_step_buffers(n, lag, fw, dev, ws=plan.ws)
    pk = torch.zeros(theta.numel() + 1, dtype=torch.float64, device=dev)
    kernels.net_cnn_train_reduce(plan, packed, lag, fw, theta, bufs, pk)
    ref = pk.clone()
    for levels in (False, True):
        if levels:
            assert plan.attach_cnn_levels(packed, lag, fw) >= 1
        monkeypatch.setenv("BEAR_AMD_DETERMINISTIC", "1")
        outs = []
        for _ in range(3):
            pk.zero_()
            kernels.net_cnn_train_reduce(plan, packed, lag, fw, theta, bufs, pk)
            outs.append(pk.clone())
        # the parameter gradients [2:] (sum LL and d/dh follow the draw of the work units in this library: the deterministic
        # build fixes those too, test_deterministic_build_whole_trajectory)
        assert torch.equal(outs[0][2:], outs[1][2:]) and torch.equal(outs[0][2:], outs[2][2:]), levels
        assert abs(float(outs[0][0] - ref[0])) <= 1e-12 * abs(float(ref[0]))
        assert float((outs[0][1:] - ref[1:]).abs().max()) <= 1e-10 * float(ref[1:].abs().max())
        monkeypatch.delenv("BEAR_AMD_DETERMINISTIC", raising=False)


_DET_SCRIPT = r"""
import hashlib, json, os, sys
import numpy as np, torch
sys.path.insert(0, sys.argv[1])
from bear_amd import _lib, ar_funcs, bear_net, bear_ref, dataloader, kernels
assert _lib.lib().bear_deterministic_build() == 1, _lib.LIB_PATH
dev = torch.device("cuda", 0)
out = {}
# kernel level: every step kernel three times, whole outputs
n = 700_000
t = kernels.synth_counts(20211012, 0, n, dev, want=("train", "ref"))
prior = kernels.synth_prior(20211012, 0, n, dev)
plan5, plan4 = kernels.Plan(t["train"], 5), kernels.Plan(t["train"], 4, ref=t["ref"])
def rep(name, fn):
    a = [fn() for _ in range(3)]
    flat = [torch.cat([x.reshape(-1).double() for x in (r if isinstance(r, (tuple, list)) else (r,)) if x is not None]) for r in a]
    out[name] = bool(torch.equal(flat[0], flat[1]) and torch.equal(flat[0], flat[2]))
rep("mode_N", lambda: kernels.dm_prior_planned(plan5, prior, -0.3).clone())
rep("mode_N_grad_rows", lambda: tuple(x.clone() for x in kernels.dm_prior_planned(plan5, prior, -0.3, want_grad=True)))
rep("mode_N_grad_rows_normalized", lambda: tuple(x.clone() for x in kernels.dm_prior_planned(plan5, prior, -0.3, want_grad=True, normalized=True)))
rep("mode_R", lambda: kernels.dm_ref_planned(plan4, t["ref"], 0.1, -3.4, -4.6).clone())
rep("mode_R_streaming", lambda: kernels.dm_ref_planned(kernels.Plan(t["train"], 4), t["ref"], 0.1, -3.4, -4.6).clone())
# the drivers: the bundled ysd1 table, 300 optimizer steps, twice; linear, cnn, reference
path = os.path.join(sys.argv[1], "bear_amd", "data", "ysd1_lag_5_file_0_preshuf.tsv")
data = dataloader.dataloader(path, "dna", 1500, 3)
def digest(params, h):
    m = hashlib.sha256()
    for p in list(params) + [h]:
        m.update(p.detach().cpu().numpy().tobytes())
    return m.hexdigest()
for name, mod, make, kw, extra in (("linear", bear_net, ar_funcs.make_ar_func_linear, {}, ()),
                                   ("cnn", bear_net, ar_funcs.make_ar_func_cnn, {"filter_width": 3}, ()),
                                   ("ref_stop", bear_ref, ar_funcs.make_ar_func_stop, {}, (2,))):
    ds = []
    for _ in range(2):
        torch.manual_seed(10)
        params, h, _ = mod.train(data.repeat(300), 1365, 300, 0, *extra, "dna", 5, make, kw, 0.01, "Adam", False)
        ds.append(digest(params, h))
    out["train_" + name] = ds[0] == ds[1]
    if name != "cnn":
        # ... and the ONE-launch optimizer step (the reduce kernel's last block runs Adam: what the two runs above took) ends in the
        # same bits as reduce + bear_train_apply_f64 in two launches
        from bear_amd import _train
        out["one_launch_" + name] = bool(_train.LAST_RUN.get("one_launch_steps"))
        os.environ["BEAR_AMD_TWO_LAUNCH_STEP"] = "1"
        torch.manual_seed(10)
        params, h, _ = mod.train(data.repeat(300), 1365, 300, 0, *extra, "dna", 5, make, kw, 0.01, "Adam", False)
        del os.environ["BEAR_AMD_TWO_LAUNCH_STEP"]
        out["two_launch_" + name] = not _train.LAST_RUN.get("one_launch_steps")
        out["one_launch_equals_two_launch_" + name] = digest(params, h) == ds[0]
print("RESULT " + json.dumps(out))
"""


def test_deterministic_build_whole_trajectory(tmp_path):
    """BEAR_AMD_DETERMINISTIC=1 at import loads libbear_hip_det.so (work units dealt statically): every output of every step kernel
    is bit-identical from launch to launch, and a 300-step training run of each driver ends in the same bits twice."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "det_run.py"
    script.write_text(_DET_SCRIPT)
    env = dict(os.environ, BEAR_AMD_DETERMINISTIC="1")
    env.pop("BEAR_AMD_LIB", None)
    p = subprocess.run([sys.executable, str(script), root], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    res = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")][0][7:])
    assert res and all(res.values()), res
