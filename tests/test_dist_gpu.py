"""Row (e) on the device: two fresh processes launched by torch.distributed.run train through the product entry points
(bear_ref.train / bear_net.train / evaluation, HIP kernels on row shards, one all-reduce per step) and must reproduce the
single-process run -- and the config driver must work under the launcher (one output folder, rank-0 writes)."""
import configparser
import json
import os
import pickle
import subprocess
import sys

import numpy as np
import pytest
import torch

from bear_amd import ar_funcs, bear_net, bear_ref, dataloader
from conftest import ROOT, YSD1

pytestmark = pytest.mark.gpu

CNN_CFG = {"num_filters": 30, "filter_width": 3, "kmer_layer1_width": 16}


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


# Ranks that share cuda:0 in the many-rank tests.  The GPU boxes of this pool allow six processes on a card at once and the test
# process itself holds one, and so does the launcher, so four ranks is the most that can run here; BEAR_TEST_RANKS=8 on a box without that limit.
MANY = int(os.environ.get("BEAR_TEST_RANKS", "4"))


def _launch(script_args, env_extra, tmp_path, nproc=2):
    env = dict(os.environ, BEAR_ROOT=ROOT, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2", BEAR_AMD_DIST_BACKEND="gloo",
               BEAR_AMD_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0", GLOO_SOCKET_IFNAME="lo", **env_extra)
    port = _free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(port)] + script_args
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-6000:]


def _flat(params):
    return np.concatenate([np.asarray(x, dtype=np.float64).reshape(-1) for x in params])


def test_two_rank_training_matches_single_rank(tmp_path):
    torch.manual_seed(3)
    _, lin = ar_funcs.make_ar_func_linear(5, 4)
    _, cnn = ar_funcs.make_ar_func_cnn(5, 4, **CNN_CFG)
    restart = {"linear": np.array([np.array(0.1)] + [x.detach().numpy() for x in lin], dtype=object),
               "cnn": np.array([np.array(0.1)] + [x.detach().numpy() for x in cnn], dtype=object),
               "ref_linear": np.array([np.array(0.2), np.array(np.log(1 / 30)), np.array(-1.0)] + [x.detach().numpy() for x in lin], dtype=object)}
    np.savez(tmp_path / "restart.npz", **restart)
    out_file = tmp_path / "out.json"
    _launch([os.path.join(ROOT, "tests", "dist_worker.py")], {"BEAR_RESTART": str(tmp_path / "restart.npz"), "BEAR_OUT": str(out_file)}, tmp_path)
    got = json.load(open(out_file))
    assert got["mirrored"] is True

    data = dataloader.dataloader(YSD1, "dna", 500, 3)

    def single(fn, *args, **kw):
        ls = []
        p, _, _ = fn(*args, loss_save=ls, **kw)
        return ls, _flat([x.detach().cpu().numpy() for x in p])

    checks = []
    for train_ar in (False, True):
        key = "ar" if train_ar else "bear"
        checks.append(("ref_stop_" + key, single(bear_ref.train, data.repeat(2), 1365, 2, 0, 2, "dna", 5, ar_funcs.make_ar_func_stop, {}, 0.01,
                                                 "Adam", train_ar)))
        checks.append(("net_linear_" + key, single(bear_net.train, data.repeat(2), 1365, 2, 0, "dna", 5, ar_funcs.make_ar_func_linear, {}, 0.01,
                                                   "Adam", train_ar, params_restart=list(restart["linear"]))))
    checks.append(("net_cnn_bear", single(bear_net.train, data.repeat(2), 1365, 2, 0, "dna", 5, ar_funcs.make_ar_func_cnn, CNN_CFG, 0.01, "Adam",
                                          False, params_restart=list(restart["cnn"]))))
    checks.append(("ref_linear_acc3", single(bear_ref.train, data.repeat(2), 1365, 2, 0, 2, "dna", 5, ar_funcs.make_ar_func_linear, {}, 0.01, "Adam",
                                             False, acc_steps=3, params_restart=list(restart["ref_linear"]))))
    checks.append(("ref_stop_sgd_acc2", single(bear_ref.train, data.repeat(2), 1365, 2, 0, 2, "dna", 5, ar_funcs.make_ar_func_stop, {}, 0.001, "SGD",
                                               False, acc_steps=2)))
    for name, (ls, flat) in checks:
        g = got[name]
        assert len(g["loss"]) == len(ls) and len(ls) > 0, name
        assert np.allclose(g["loss"], ls, rtol=1e-10), name          # the sum over shards in another order: rounding only
        assert np.allclose(_flat(g["params"]), flat, rtol=1e-7, atol=1e-10), name
    torch.manual_seed(1)
    f, _ = ar_funcs.make_ar_func_linear(5, 4, device="cuda")
    r = bear_net.evaluation(data, 0, 1, "dna", torch.tensor(0.37), f, np.array([0.1, 1.0, 10.0]), seed=11)
    for a, b in zip(got["eval"], r):
        assert np.allclose(a, np.asarray(b), rtol=1e-11)
    # accuracies: integer counts of correct rows, the tie-breaking noise is a function of the global row -> identical
    assert got["eval"][8] == np.asarray(r[8]).tolist()


def test_many_ranks_deterministic_mode_with_empty_pieces(tmp_path):
    """BEAR_AMD_DETERMINISTIC=1 with MANY ranks on a table whose last batch leaves most ranks WITHOUT rows: the count bound of the
    fixed-point gradient tables is all-reduced over the ranks when a batch's step is set up (bear_net.train), and a rank with an
    empty piece has to take part (it used to return before the collective: the group aborted).  Losses and parameters equal the
    single-process deterministic run's to rounding."""
    torch.manual_seed(3)
    _, lin = ar_funcs.make_ar_func_linear(5, 4)
    restart = {"linear": np.array([np.array(0.1)] + [x.detach().numpy() for x in lin], dtype=object)}
    np.savez(tmp_path / "restart.npz", **restart)
    small = tmp_path / "small.tsv"
    with open(YSD1) as fh:
        small.write_text("".join(fh.readlines()[:23]))
    out_file = tmp_path / "out.json"
    _launch([os.path.join(ROOT, "tests", "dist_worker_n.py")],
            {"BEAR_RESTART": str(tmp_path / "restart.npz"), "BEAR_OUT": str(out_file), "BEAR_TABLES": json.dumps([["small", str(small), 7]]),
             "BEAR_EXPECT_WORLD": str(MANY), "BEAR_AMD_DETERMINISTIC": "1"}, tmp_path, nproc=MANY)
    got = json.load(open(out_file))["small"]
    assert np.asarray(got["pieces"]).min() == 0
    data = dataloader.dataloader(str(small), "dna", 7, 3)
    ls = []
    os.environ["BEAR_AMD_DETERMINISTIC"] = "1"
    try:
        p, _, _ = bear_net.train(data.repeat(4), data.num_rows, 4, 0, "dna", 5, ar_funcs.make_ar_func_linear, {}, 0.01, "Adam", False,
                                 params_restart=list(restart["linear"]), loss_save=ls)
    finally:
        del os.environ["BEAR_AMD_DETERMINISTIC"]
    assert len(ls) == len(got["net_linear"]["loss"]) and np.allclose(got["net_linear"]["loss"], ls, rtol=1e-10)
    assert np.allclose(_flat(got["net_linear"]["params"]), _flat([x.detach().cpu().numpy() for x in p]), rtol=1e-7, atol=1e-10)


@pytest.mark.parametrize("deterministic", [False, True])
def test_one_rank_streams_every_rank_streams(tmp_path, deterministic):
    """Whether an epoch is streamed is decided from a rank's OWN free HBM (or BEAR_AMD_STREAM): the ranks agree on it with one MAX
    all-reduce before any of them acts (`ResidentBatches`, `dist.agree_max`) -- a streaming rank runs an eager loop, a resident one
    captures a graph behind a warm-up all-reduce, and in deterministic mode the count-bound all-reduces come per batch load on the
    one and up front on the other: the collectives would pair up wrongly.  Rank 1 alone is told to stream here; both stream, and
    the losses and parameters equal the single-process run's."""
    torch.manual_seed(3)
    _, lin = ar_funcs.make_ar_func_linear(5, 4)
    restart = {"linear": np.array([np.array(0.1)] + [x.detach().numpy() for x in lin], dtype=object)}
    np.savez(tmp_path / "restart.npz", **restart)
    out_file = tmp_path / "out.json"
    extra = {"BEAR_AMD_DETERMINISTIC": "1"} if deterministic else {}
    _launch([os.path.join(ROOT, "tests", "dist_worker_n.py")],
            {"BEAR_RESTART": str(tmp_path / "restart.npz"), "BEAR_OUT": str(out_file), "BEAR_TABLES": json.dumps([["ysd1", YSD1, 500]]),
             "BEAR_EXPECT_WORLD": "2", "BEAR_TEST_STREAM_RANK": "1", **extra}, tmp_path, nproc=2)
    got = json.load(open(out_file))["ysd1"]
    assert got["streamed"] == [True, True]
    data = dataloader.dataloader(YSD1, "dna", 500, 3)
    for key, fn, args, kw in (("ref_stop", bear_ref.train, (data.repeat(4), 1365, 4, 0, 2, "dna", 5, ar_funcs.make_ar_func_stop, {}, 0.01, "Adam", False), {}),
                              ("net_linear", bear_net.train, (data.repeat(4), 1365, 4, 0, "dna", 5, ar_funcs.make_ar_func_linear, {}, 0.01, "Adam", False),
                               {"params_restart": list(restart["linear"])})):
        ls = []
        p, _, _ = fn(*args, loss_save=ls, **kw)
        assert len(ls) == len(got[key]["loss"]) and np.allclose(got[key]["loss"], ls, rtol=1e-10), key
        assert np.allclose(_flat(got[key]["params"]), _flat([x.detach().cpu().numpy() for x in p]), rtol=1e-7, atol=1e-10), key


def test_many_ranks_match_single_rank_with_empty_and_uneven_pieces(tmp_path):
    """MANY processes under torch.distributed.run (all on cuda:0, gloo): bear_ref.train, bear_net.train (linear; BEAR mode, and AR
    mode with gradient accumulation) and both evaluations on row shards reproduce the single-process run -- on the bundled table
    (pieces 100 / 73 rows) and on a 23-row table in batches of 7, where a rank's piece of a batch holds 2, 1 or NO rows (the
    last batch has 2 rows for MANY ranks)."""
    torch.manual_seed(3)
    _, lin = ar_funcs.make_ar_func_linear(5, 4)
    restart = {"linear": np.array([np.array(0.1)] + [x.detach().numpy() for x in lin], dtype=object)}
    np.savez(tmp_path / "restart.npz", **restart)
    small = tmp_path / "small.tsv"
    with open(YSD1) as fh:
        small.write_text("".join(fh.readlines()[:23]))
    tables = [["ysd1", YSD1, 500], ["small", str(small), 7]]
    out_file = tmp_path / "out.json"
    _launch([os.path.join(ROOT, "tests", "dist_worker_n.py")],
            {"BEAR_RESTART": str(tmp_path / "restart.npz"), "BEAR_OUT": str(out_file), "BEAR_TABLES": json.dumps(tables),
             "BEAR_EXPECT_WORLD": str(MANY)}, tmp_path, nproc=MANY)
    got = json.load(open(out_file))
    sm = np.asarray(got["small"]["pieces"])                 # [rank][batch]
    assert sm.shape == (MANY, 4) and sm.sum() == 23 and (sm[:, 3] == 0).sum() == MANY - 2 and sm.min() == 0 and sm.max() >= 2
    assert np.asarray(got["ysd1"]["pieces"]).sum() == 1365
    for name, path, batch in tables:
        data = dataloader.dataloader(path, "dna", batch, 3)
        g = got[name]

        def check(key, fn, *args, **kw):
            ls = []
            p, _, _ = fn(*args, loss_save=ls, **kw)
            assert len(g[key]["loss"]) == len(ls) and len(ls) > 0, (name, key)
            assert np.allclose(g[key]["loss"], ls, rtol=1e-10), (name, key)      # the sum over shards in another order: rounding only
            assert np.allclose(_flat(g[key]["params"]), _flat([x.detach().cpu().numpy() for x in p]), rtol=1e-7, atol=1e-10), (name, key)
        check("ref_stop", bear_ref.train, data.repeat(4), data.num_rows, 4, 0, 2, "dna", 5, ar_funcs.make_ar_func_stop, {}, 0.01, "Adam", False)
        check("net_linear", bear_net.train, data.repeat(4), data.num_rows, 4, 0, "dna", 5, ar_funcs.make_ar_func_linear, {}, 0.01, "Adam", False,
              params_restart=list(restart["linear"]))
        check("net_linear_ar_acc2", bear_net.train, data.repeat(4), data.num_rows, 4, 0, "dna", 5, ar_funcs.make_ar_func_linear, {}, 0.01, "Adam",
              True, acc_steps=2, params_restart=list(restart["linear"]))
        torch.manual_seed(1)
        f, _ = ar_funcs.make_ar_func_linear(5, 4, device="cuda")
        r = bear_net.evaluation(data, 0, 1, "dna", torch.tensor(0.37), f, np.array([0.1, 1.0, 10.0]), seed=11)
        fr, _ = bear_ref._make_ref_ar_func(5, 4, ar_funcs.make_ar_func_stop, {}, device="cuda")
        rr = bear_ref.evaluation(data, 0, 1, 2, "dna", torch.tensor(0.21), fr, np.array([0.5, 2.0]), seed=3)
        for key, want in (("eval", r), ("eval_ref", rr)):
            for a, b in zip(g[key], want):
                assert np.allclose(a, np.asarray(b), rtol=1e-11), (name, key)
            # accuracies: integer counts of correct rows over a global-row-keyed noise stream -> identical
            assert g[key][6] == np.asarray(want[6]).tolist() and g[key][8] == np.asarray(want[8]).tolist(), (name, key)


def test_ranks_dealt_by_kmer_range_match_single_rank(tmp_path):
    """`dataloader(..., shard="kmer")`: every batch is dealt to the ranks BY K-MER RANGE (dataloader.KmerDealtDataset: each rank
    parses the table and keeps the rows whose leading letters fall into its range -- its piece of a k-mer-sorted batch keeps the
    batch's density of prefixes and windows) instead of by contiguous row pieces.  Sums do not depend on which rows a rank
    holds: losses and parameters of bear_ref.train / bear_net.train (linear; BEAR mode and AR mode with accumulation) equal the
    single-process run's, and so do both evaluations -- the accuracy COUNTS exactly: the rows' table positions travel with them as
    row_ids, the key of the tie-breaking noise.  Tables: the bundled one and 23 rows in batches of 7 (pieces of 2, 1, 0 rows)."""
    torch.manual_seed(3)
    _, lin = ar_funcs.make_ar_func_linear(5, 4)
    restart = {"linear": np.array([np.array(0.1)] + [x.detach().numpy() for x in lin], dtype=object)}
    np.savez(tmp_path / "restart.npz", **restart)
    small = tmp_path / "small.tsv"
    with open(YSD1) as fh:
        small.write_text("".join(fh.readlines()[:23]))
    tables = [["ysd1", YSD1, 500], ["small", str(small), 7]]
    out_file = tmp_path / "out.json"
    _launch([os.path.join(ROOT, "tests", "dist_worker_n.py")],
            {"BEAR_RESTART": str(tmp_path / "restart.npz"), "BEAR_OUT": str(out_file), "BEAR_TABLES": json.dumps(tables),
             "BEAR_EXPECT_WORLD": str(MANY), "BEAR_TEST_SHARD": "kmer"}, tmp_path, nproc=MANY)
    got = json.load(open(out_file))
    assert np.asarray(got["ysd1"]["pieces"]).sum() == 1365 and np.asarray(got["small"]["pieces"]).sum() == 23
    for name, path, batch in tables:
        data = dataloader.dataloader(path, "dna", batch, 3)
        g = got[name]

        def check(key, fn, *args, **kw):
            ls = []
            p, _, _ = fn(*args, loss_save=ls, **kw)
            assert len(g[key]["loss"]) == len(ls) and len(ls) > 0, (name, key)
            assert np.allclose(g[key]["loss"], ls, rtol=1e-10), (name, key)
            assert np.allclose(_flat(g[key]["params"]), _flat([x.detach().cpu().numpy() for x in p]), rtol=1e-7, atol=1e-10), (name, key)
        check("ref_stop", bear_ref.train, data.repeat(4), data.num_rows, 4, 0, 2, "dna", 5, ar_funcs.make_ar_func_stop, {}, 0.01, "Adam", False)
        check("net_linear", bear_net.train, data.repeat(4), data.num_rows, 4, 0, "dna", 5, ar_funcs.make_ar_func_linear, {}, 0.01, "Adam", False,
              params_restart=list(restart["linear"]))
        check("net_linear_ar_acc2", bear_net.train, data.repeat(4), data.num_rows, 4, 0, "dna", 5, ar_funcs.make_ar_func_linear, {}, 0.01, "Adam",
              True, acc_steps=2, params_restart=list(restart["linear"]))
        torch.manual_seed(1)
        f, _ = ar_funcs.make_ar_func_linear(5, 4, device="cuda")
        r = bear_net.evaluation(data, 0, 1, "dna", torch.tensor(0.37), f, np.array([0.1, 1.0, 10.0]), seed=11)
        fr, _ = bear_ref._make_ref_ar_func(5, 4, ar_funcs.make_ar_func_stop, {}, device="cuda")
        rr = bear_ref.evaluation(data, 0, 1, 2, "dna", torch.tensor(0.21), fr, np.array([0.5, 2.0]), seed=3)
        for key, want in (("eval", r), ("eval_ref", rr)):
            for a, b in zip(g[key], want):
                assert np.allclose(a, np.asarray(b), rtol=1e-11), (name, key)
            assert g[key][6] == np.asarray(want[6]).tolist() and g[key][8] == np.asarray(want[8]).tolist(), (name, key)


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_under_the_launcher_many_ranks(scaling, tmp_path):
    """bench.py --gpus MANY (gloo, all ranks on cuda:0) in both scaling modes: the whole-job value counts every rank's contexts,
    the per-rank entries carry event-timed kernel and all-reduce times, and the reduced ELBO equals the ELBO of the global table
    (strong: ONE table of --contexts rows cut into MANY uneven shards)."""
    from bear_amd import kernels
    n = 1_000_003
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2", BEAR_BENCH_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0",
               GLOO_SOCKET_IFNAME="lo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={MANY}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(MANY), "--steps", "4", "--warmup", "1",
           "--contexts", str(n), "--backend", "gloo", "--no-cpu-baseline", "--scaling", scaling]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-6000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    total = n if scaling == "strong" else n * MANY
    assert d["n_gpus"] == MANY and d["scaling"] == scaling and d["config"]["contexts_total"] == total
    assert abs(d["value"] - total / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    pr = d["per_rank"]
    assert [e["rank"] for e in pr] == list(range(MANY)) and sum(e["contexts"] for e in pr) == total
    assert all(e["kernel_ms"] > 0 and e["allreduce_ms"] > 0 for e in pr)
    if scaling == "strong":
        assert max(e["contexts"] for e in pr) - min(e["contexts"] for e in pr) == 1      # 1 000 003 rows over MANY ranks
    dev = torch.device("cuda", 0)
    t = kernels.synth_counts(20211012, 0, total, dev, want=("train",))["train"]
    want = kernels.dm_prior_planned(kernels.Plan(t, 5), kernels.synth_prior(20211012, 0, total, dev), 0.0).cpu().numpy()
    assert np.allclose(d["result"], want, rtol=1e-12), (d["result"], want)


@pytest.mark.parametrize("kind", ["ref", "net"])
def test_config_driver_under_the_launcher(kind, tmp_path):
    """`python -m torch.distributed.run --nproc-per-node 2 bear_amd/models/train_bear_<kind>.py cfg`: one output folder, written
    by rank 0, same fitted parameters and evaluation as the single-process run of the same config."""
    def make_cfg(out):
        config = configparser.ConfigParser()
        config.read(os.path.join(ROOT, "bear_amd", "models", "config_files", "bear_test.cfg"))
        config["model"]["ar_func_name"] = "stop" if kind == "ref" else "linear"
        config["general"]["out_folder"] = str(out) + "*"
        config["train"]["epochs"] = "6"
        config["train"]["batch_size"] = "400"
        config["train"]["train_ar"] = "False"
        if kind == "ref" and out.name == "two":
            # [data] binary_cache under the launcher: rank 0 builds the missing cache with a plain load, the ranks then read row ranges
            config["data"]["binary_cache"] = str(tmp_path / "cache")
        path = str(out) + ".cfg"
        with open(path, "w") as fh:
            config.write(fh)
        return path
    script = os.path.join(ROOT, "bear_amd", "models", f"train_bear_{kind}.py")
    cfg2 = make_cfg(tmp_path / "two")
    _launch([script, cfg2], {}, tmp_path)
    cfg1 = make_cfg(tmp_path / "one")
    p = subprocess.run([sys.executable, script, cfg1], capture_output=True, text=True, timeout=900, env=dict(os.environ, BEAR_ROOT=ROOT))
    assert p.returncode == 0, p.stderr[-4000:]
    res = []
    for name in ("two", "one"):
        folder = tmp_path / name
        assert sorted(os.listdir(folder))[:1] == ["config.cfg"] and os.path.exists(folder / "results.pickle")
        c = configparser.ConfigParser()
        c.read(folder / "config.cfg")
        with open(folder / "results.pickle", "rb") as fh:
            params = pickle.load(fh)["params"]
        res.append((c["results"], _flat(params)))
    assert np.allclose(res[0][1], res[1][1], rtol=1e-7, atol=1e-10)
    if kind == "ref":
        assert any(f.endswith(".bearcache") for f in os.listdir(tmp_path / "cache"))
    for key in ("h", "heldout_perplex_BEAR", "heldout_perplex_AR", "perplex_BEAR", "heldout_accuracy_BEAR"):
        assert np.allclose(json.loads(res[0][0][key]), json.loads(res[1][0][key]), rtol=1e-9), key


def test_bench_under_the_launcher_two_ranks(tmp_path):
    """bench.py as the driver launches it for N > 1 (torch.distributed.run, one rank per process; here both ranks on cuda:0 over
    gloo): one JSON line from rank 0, whole-job value, weak scaling, and per-rank sums that the all-reduce has added up -- the
    reduced ELBO equals the ELBO of the two row shards computed in this process."""
    from bear_amd import kernels
    n = 2_000_000
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2", BEAR_BENCH_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0",
               GLOO_SOCKET_IFNAME="lo")
    port = _free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
           "--contexts", str(n), "--backend", "gloo", "--no-cpu-baseline"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-6000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["scaling"] == "weak" and d["unit"] == "contexts/s"
    assert abs(d["value"] - 2 * n / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]          # whole job: both ranks' contexts
    assert d["roofline"]["frac"] > 0 and d["cpu_baseline"] is None
    dev = torch.device("cuda", 0)
    want = 0.0
    for r in range(2):           # the shards bench.py gives rank r: rows [r n, (r + 1) n) of the global synthetic table
        t = kernels.synth_counts(20211012, r * n, n, dev, want=("train",))["train"]
        want += kernels.dm_prior_planned(kernels.Plan(t, 5), kernels.synth_prior(20211012, r * n, n, dev), 0.0).cpu().numpy()
    # the bench re-reduces its output buffer at every step: after the last step it holds the all-reduced sums of that step
    assert np.allclose(d["result"], want, rtol=1e-12), (d["result"], want)


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_bare_command_starts_its_own_ranks(scaling):
    """`python3 bench.py --gpus N` with NO launcher and no WORLD_SIZE in the environment -- the shape of the driver's N = 1 command
    with another N: the process starts torch.distributed.run itself as a child, relays rank 0's one line and exits with the
    child's code (here: MANY ranks on cuda:0 over gloo)."""
    from bear_amd import kernels
    n = 2_000_000
    env = dict(os.environ, BEAR_BENCH_DEVICE="0", GLOO_SOCKET_IFNAME="lo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(MANY), "--backend", "gloo", "--contexts", "2e6",
                        "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--scaling", scaling],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-6000:]
    lines = p.stdout.splitlines()
    assert len(lines) == 1 and lines[0].startswith('{"metric"'), p.stdout[-2000:]      # stdout is the one line, nothing else
    d = json.loads(lines[0])
    total = n if scaling == "strong" else n * MANY
    assert d["n_gpus"] == MANY and d["scaling"] == scaling and d["steps"] == 6 and d["config"]["contexts_total"] == total
    assert d["ranks"] == {"world_size": MANY, "backend": "gloo", "launcher": "self", "devices": [0] * MANY,
                          "backend_is_rccl": False, "devices_distinct": False}       # (the line says what it ran on: one card, gloo)
    assert abs(d["value"] - total / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    ss = d["also"].get("strong_scaling")
    if scaling == "weak":        # a weak run of N ranks also times the strong split of ONE --contexts table
        assert ss["scaling"] == "strong" and ss["contexts_total"] == n and len(ss["per_rank"]) == MANY
        assert sum(r["contexts"] for r in ss["per_rank"]) == n
        assert abs(ss["value"] - n / (ss["ms_per_step"] * 1e-3)) <= 1e-6 * ss["value"]
        t1 = kernels.synth_counts(20211012, 0, n, torch.device("cuda", 0), want=("train",))["train"]
        w1 = kernels.dm_prior_planned(kernels.Plan(t1, 5), kernels.synth_prior(20211012, 0, n, torch.device("cuda", 0)), 0.0).cpu().numpy()
        assert np.allclose(ss["result"], w1, rtol=1e-12), (ss["result"], w1)
    else:
        assert ss is None
    dev = torch.device("cuda", 0)
    t = kernels.synth_counts(20211012, 0, total, dev, want=("train",))["train"]
    want = kernels.dm_prior_planned(kernels.Plan(t, 5), kernels.synth_prior(20211012, 0, total, dev), 0.0).cpu().numpy()
    assert np.allclose(d["result"], want, rtol=1e-12), (d["result"], want)


def test_bench_single_rank_line_is_consistent():
    """bench.py as the driver runs it at N = 1 (small table): one JSON line whose roofline object follows from its own
    kernel time, with the 'also' entries of the other kernels in it."""
    n = 3_000_000
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "3", "--contexts", str(n), "--no-cpu-baseline",
                        "--no-baseline-configs"],      # (those run at the configs' own sizes: tests/test_config_shards_gpu.py covers the module)
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"), capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-6000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    r = d["roofline"]
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["unit"] == "contexts/s" and d["dtype"] == "f64" and r["bound"] == "hbm"
    assert abs(d["value"] - n / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    assert abs(r["achieved"] - n * 60 / (r["kernel_ms"] * 1e-3) / 1e9) <= 1e-6 * r["achieved"]           # 60 algorithmic bytes per context
    assert abs(r["frac"] - r["achieved"] / r["peak"]) <= 1e-9 and 0.0 < r["frac"] < 1.5 and r["peak"] == 8000.0
    assert r["kernel_ms"] <= d["ms_per_step"] * 1.05
    also = d["also"]
    for key in ("ref", "net_with_gradient_rows", "linear_head_fused_step", "cnn_head", "heldout_evaluation"):
        assert key in also, key
    c = also["cnn_head"]
    assert abs(c["step_contexts_per_s"] - n / (c["train_step_ms_as_bear_net_train_holds_the_batch"] * 1e-3)) <= 1e-6 * c["step_contexts_per_s"]
    assert 0.5 < c["contexts_with_training_counts"] < 0.9
    # round 4: the gradient-row kernel next to the headline, the convolutional step over prefix levels (same sums as without them;
    # executed flops <= credited flops), the linear head on paired lists (same sums), the own-A path of rows that are not normalised
    g = d["roofline_gradient_rows"]
    assert g["kernel"] == "dm_prior_plan_grad_inplace_kernel" and g["credited_read_B"] == 60 and 0.0 < g["frac_credited"] < g["frac_moved"] < 1.2
    lv = c["prefix_levels"]
    assert lv["equals_step_without_levels"] is True and lv["rows"][0] == round(c["contexts_with_training_counts"] * n)
    if lv["attached"]:
        assert lv["rows"] == sorted(lv["rows"], reverse=True) and 0.0 < lv["position_evaluations_per_context"] < 6.0      # (< 1 since round 5: window tables)
        assert lv["rates_forward"]["executed_fp64_flops"] < lv["rates_forward"]["credited_fp64_flops"]
    assert "frac_of_fp64_peak" not in c["rates_forward_rows_in_kmer_order"]           # (no executed-flop count there: no utilisation claimed)
    lin = also["linear_head_fused_step"]
    assert lin["paired_equals_plain"] is True and isinstance(lin["paired_contexts"], bool)
    assert also["net_rows_not_normalised"]["kernel_ms"] >= r["kernel_ms"] * 0.9
    assert d["ranks"] == {"world_size": 1, "backend": None, "launcher": "none (one process)", "devices": [0],
                          "backend_is_rccl": None, "devices_distinct": True}
    assert d["settle"]["cold_ms_per_step_without_settle"] > 0          # the figure of the same command without the settle launches


def test_bench_rccl_path_on_a_group_of_one():
    """bench.py's N > 1 code path on RCCL itself (--force-collective: process group "nccl", one all-reduce per timed step behind
    the kernel, the MAX / gather collectives of the timing, barrier, teardown) with the one rank a one-GPU box has: the line is
    complete, the all-reduce is event-timed, and the reduced sums are the kernel's."""
    from bear_amd import kernels
    n = 2_000_000
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-collective", "--steps", "6", "--warmup", "2",
                        "--contexts", str(n), "--no-cpu-baseline", "--ingest-rows", "0"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-6000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["per_rank"] is not None and len(d["per_rank"]) == 1
    assert d["per_rank"][0]["contexts"] == n and d["per_rank"][0]["kernel_ms"] > 0 and d["per_rank"][0]["allreduce_ms"] > 0
    dev = torch.device("cuda", 0)
    t = kernels.synth_counts(20211012, 0, n, dev, want=("train",))["train"]
    want = kernels.dm_prior_planned(kernels.Plan(t, 5), kernels.synth_prior(20211012, 0, n, dev), 0.0).cpu().numpy()
    assert np.allclose(d["result"], want, rtol=1e-12)


def test_rccl_group_of_one(tmp_path):
    """RCCL itself on this box: a process group of one rank (two ranks cannot share a card under RCCL) runs the step's
    collectives behind the planned kernel in stream order -- the library loads, a communicator comes up on the device the
    rank is bound to, and a sum over one rank leaves the packed vector as the kernel wrote it."""
    out_file = tmp_path / "rccl.json"
    env = dict(os.environ, BEAR_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0",
               BEAR_OUT=str(out_file))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_worker.py")], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-6000:]
    got = json.load(open(out_file))
    assert got["backend"] == "nccl" and got["world"] == 1
    assert got["same"] is True and np.isfinite(got["out"]).all() and got["out"][0] != 0.0
    assert got["max"] == 1.25 and got["theta"] == [0.0, 1.0, 2.0, 3.0, 4.0]
    # the optimizer loop with the RCCL all-reduce captured into the HIP graph == the eager loop (18 steps, 3 batches per epoch;
    # with acc_steps = 2 the captured period is lcm(3, 2) = 6 steps)
    for name, period in (("ref", 3), ("net_acc2", 6)):
        g, e = got["runs"][name + "_graph"], got["runs"][name + "_eager"]
        assert g["how"]["graph"] is True and g["how"]["collective"] is True and g["how"]["period"] == period, g["how"]
        assert g["how"]["replays"] == 18 // period and g["how"]["eager_steps"] == 0
        assert e["how"]["graph"] is False and e["how"]["eager_steps"] == 18
        assert len(g["loss"]) == len(e["loss"]) == (18 if name == "ref" else 9)
        assert np.allclose(g["loss"], e["loss"], rtol=1e-12) and np.allclose(g["params"], e["params"], rtol=1e-10, atol=1e-13), name
    # ... and the generic loop (an AR function of torch ops): first period eager, the rest replayed with the all-reduce captured
    g, e = got["runs"]["net_generic_graph"], got["runs"]["net_generic_eager"]
    assert g["how"]["graph"] is True and g["how"]["collective"] is True and g["how"]["replays"] >= 1, g["how"]
    assert e["how"]["graph"] is False and e["how"]["eager_steps"] == 18
    assert len(g["loss"]) == len(e["loss"]) == 18
    assert np.allclose(g["loss"], e["loss"], rtol=1e-12) and np.allclose(g["params"], e["params"], rtol=1e-12, atol=1e-14)
