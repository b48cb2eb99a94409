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


def _launch(script_args, env_extra, tmp_path):
    env = dict(os.environ, BEAR_ROOT=ROOT, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2", BEAR_AMD_DIST_BACKEND="gloo",
               BEAR_AMD_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0", GLOO_SOCKET_IFNAME="lo", **env_extra)
    port = _free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
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


def test_bench_single_rank_line_is_consistent():
    """bench.py as the driver runs it at N = 1 (small table): one JSON line whose roofline object follows from its own
    kernel time, with the 'also' entries of the other kernels in it."""
    n = 3_000_000
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "3", "--contexts", str(n), "--no-cpu-baseline"],
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
