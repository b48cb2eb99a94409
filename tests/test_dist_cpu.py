"""World-size-2 and -8 gloo tests of the N > 1 path on CPU: row sharding + the single packed all-reduce of a step.
Per-shard partial sums come from the oracle here (the HIP kernels need a GPU); the reduce, the packing and the
loss scaling are the product code of bear_amd.dist."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["BEAR_ROOT"]); sys.path.insert(0, os.path.join(os.environ["BEAR_ROOT"], "oracle")); sys.path.insert(0, os.path.join(os.environ["BEAR_ROOT"], "tests"))
import numpy as np, torch, torch.distributed as dist
import bear_oracle as o
from bear_amd import dist as bdist
from util import sparse_table, prior_rows
assert "HSA_ENABLE_IPC_MODE_LEGACY" not in os.environ     # (the test's launcher took it out: an external launcher may not export it)
rank, world = bdist.init_from_env()          # the product's own set-up (BEAR_AMD_DIST_BACKEND=gloo here, RCCL on the GPU node)
# dmabuf IPC for RCCL between processes: exported by init_from_env itself, before the process group (and any GPU call) exists
assert os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
W = int(os.environ["BEAR_EXPECT_WORLD"])
assert world == W and bdist.world() == (rank, W) and dist.get_backend() == "gloo"
# mirrored variables: rank 1 draws other initial values, rank 0's win
torch.manual_seed(10 + rank)
ps = [torch.randn(3, 2, dtype=torch.float64), torch.randn((), dtype=torch.float64)]
bdist.broadcast_params(ps)
got = [None] * W
dist.all_gather_object(got, [p.numpy().tolist() for p in ps])
assert all(g == got[0] for g in got)
# input sharding at load time: the two ranks' pieces of every batch tile the table
from bear_amd import dataloader
ysd1 = os.path.join(os.environ["BEAR_ROOT"], "tests", "golden", "ysd1_lag_5_file_0_preshuf.tsv")
part = dataloader.dataloader(ysd1, "dna", 500, 3, shard="auto")
assert part.shard == (rank, W) and part.num_rows == 1365
rows = [None] * W
dist.all_gather_object(rows, [(g0, g1) for g0, g1, _ in part.rank_pieces(rank, W)])
if W == 2:
    assert rows[0] == [(0, 250), (500, 750), (1000, 1183)] and rows[1] == [(250, 500), (750, 1000), (1183, 1365)]
for k, (a, b) in enumerate([(0, 500), (500, 1000), (1000, 1365)]):      # every batch: the ranks' pieces tile it in rank order
    assert rows[0][k][0] == a and rows[W - 1][k][1] == b and all(rows[r][k][1] == rows[r + 1][k][0] for r in range(W - 1))
    sizes = [rows[r][k][1] - rows[r][k][0] for r in range(W)]
    assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)
assert part.local_rows == sum(b - a for a, b in rows[rank])
# a table smaller than the world: some ranks hold nothing of a batch (and still take part in the step's all-reduce)
three = [bdist.shard_rows(3, r, W) for r in range(W)]
assert sum(hi - lo for lo, hi in three) == 3 and all(hi - lo == (1 if r < 3 else 0) for r, (lo, hi) in enumerate(three)) or W == 2
train, _, ref = sparse_table(10007, 3)
f = prior_rows(10007, 4)
args = (0.2, np.log(1 / 30), -np.log(100))
lo, hi = bdist.shard_rows(len(train))
# bear_ref step: packed [sum LL, d/dh, d/dtau, d/dnu]
r = o.bear_ref_step(train[lo:hi], ref[lo:hi], *args)
packed = torch.tensor([r["ll"], r["d_h_signed"], r["d_tau_signed"], r["d_nu_signed"]], dtype=torch.float64)
bdist.allreduce_sum_(packed)
full = o.bear_ref_step(train, ref, *args)
want = np.array([full["ll"], full["d_h_signed"], full["d_tau_signed"], full["d_nu_signed"]])
assert np.allclose(packed.numpy(), want, rtol=1e-12), (packed, want)
# bear_net step: loss + d/dh + a list of AR-parameter gradients in one packed all-reduce
rn = o.bear_net_step(train[lo:hi], f[lo:hi], -0.3)
g_mat = torch.tensor(rn["d_prior"].sum(0)).reshape(1, 5)            # stand-in for a parameter-shaped gradient
flat, unpack = bdist.pack([torch.tensor([rn["ll"], rn["d_h_signed"]]), g_mat, torch.zeros(3, 2)])
bdist.allreduce_sum_(flat)
a, b, c = unpack(flat)
fulln = o.bear_net_step(train, f, -0.3)
assert np.allclose(a.numpy(), [fulln["ll"], fulln["d_h_signed"]], rtol=1e-12)
assert np.allclose(b.numpy().reshape(-1), fulln["d_prior"].sum(0), rtol=1e-10) and c.shape == (3, 2)
# loss scale uses the GLOBAL batch (SURVEY quirk 1): -(num_kmers / B_global) * sum over all ranks
num_kmers, B = 50000, len(train)
assert np.isclose(-(num_kmers / B) * packed[0].item(), -(num_kmers / B) * full["ll"], rtol=1e-12)
if rank == 0:
    print("DIST_OK")
bdist.shutdown()
'''


@pytest.mark.parametrize("world", [2, 8])
def test_gloo_step(tmp_path, world):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    import socket
    env = dict(os.environ, BEAR_ROOT=ROOT, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", BEAR_AMD_DIST_BACKEND="gloo", GLOO_SOCKET_IFNAME="lo",
               BEAR_EXPECT_WORLD=str(world))
    env.pop("HSA_ENABLE_IPC_MODE_LEGACY", None)
    with socket.socket() as sk:         # a free port (a fixed one may still be held by an earlier run)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script)]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert "DIST_OK" in p.stdout


def test_bench_bare_multi_gpu_command_fails_loudly_without_the_devices():
    """`python3 bench.py --gpus N` without a launcher: the parent counts the devices BEFORE it starts any rank (no GPU in this
    container -> exit code 2 and a message, nothing spawned), and with the ranks started (BEAR_BENCH_DEVICE puts them all on one
    card, which does not exist here) a rank that dies makes the whole command fail -- no JSON line, non-zero exit code."""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "BEAR_BENCH_DEVICE"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 2 and "--gpus 2 but this node shows 0 GPU" in p.stderr and p.stdout == ""
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo"], env=dict(env, BEAR_BENCH_DEVICE="0"),
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and p.stdout == "" and "bench.py needs MI355X devices" in p.stderr


def test_driver_path_exports_the_ipc_mode_before_the_process_group(monkeypatch):
    """`torchrun ... train_bear_ref.py cfg` under an external launcher: `dist.init_from_env()` is the drivers' first call
    (models/_driver.py) and exports HSA_ENABLE_IPC_MODE_LEGACY=0 before `init_process_group` -- a value the launcher set wins."""
    import torch.distributed as tdist
    from bear_amd import dist as bdist
    seen = {}

    def fake_init(backend, **kw):
        seen["env"] = os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")
        raise RuntimeError("stop here")
    monkeypatch.setattr(tdist, "init_process_group", fake_init)
    for preset, want in ((None, "0"), ("1", "1")):
        if preset is None:
            monkeypatch.delenv("HSA_ENABLE_IPC_MODE_LEGACY", raising=False)
        else:
            monkeypatch.setenv("HSA_ENABLE_IPC_MODE_LEGACY", preset)
        monkeypatch.setenv("WORLD_SIZE", "2")
        monkeypatch.setenv("RANK", "0")
        monkeypatch.setenv("LOCAL_RANK", "0")
        monkeypatch.setenv("BEAR_AMD_DIST_BACKEND", "gloo")
        with pytest.raises(RuntimeError, match="stop here"):
            bdist.init_from_env()
        assert seen["env"] == want
    # the config driver calls it before it touches anything else
    src = open(os.path.join(ROOT, "bear_amd", "models", "_driver.py")).read()
    body = src[src.index("def _run"):]
    first = [ln.strip() for ln in body.splitlines()[1:] if ln.strip() and not ln.strip().startswith(("#", '"'))][0]
    assert first == "rank, world = dist.init_from_env()", first


def test_batches_dealt_by_kmer_range_tile_every_batch():
    """`CountDataset.deal_by_kmer(rank, world)` (dataloader(..., shard="kmer")): for any world size the ranks' pieces of EVERY batch
    are disjoint and cover it, each rank's rows are one range of the batch's k-mers (in the k-mer order of the device sort), the
    pieces are balanced to a bin of leading letters, `row_index` says where each row sits in its batch, and the dealt dataset
    still describes the whole table (num_rows, batch_bounds)."""
    from bear_amd import dataloader
    ysd1 = os.path.join(ROOT, "tests", "golden", "ysd1_lag_5_file_0_preshuf.tsv")
    whole = dataloader.dataloader(ysd1, "dna", 500, 3)
    assert whole.deal_by_kmer(0, 1) is whole
    for world in (2, 3, 8):
        parts = [whole.deal_by_kmer(r, world) for r in range(world)]
        assert all(p.num_rows == 1365 and p.batch_bounds() == whole.batch_bounds() and p.shard == (r, world) for r, p in enumerate(parts))
        assert sum(p.local_rows for p in parts) == 1365
        for k, (a, b) in enumerate(whole.batch_bounds()):
            rows, keys = [], []
            for r, p in enumerate(parts):
                g0, g1, off = p.rank_pieces(r, world)[k]
                assert g0 == a and g1 - g0 == p.piece_rows[k]
                idx = p.row_index[off:off + g1 - g0]
                assert np.all(np.diff(idx) > 0)                                     # file order inside a piece
                assert np.array_equal(p.kmers[off:off + g1 - g0], whole.kmers[a + idx])
                assert np.array_equal(p.counts[:, off:off + g1 - g0], whole.counts[:, a + idx])
                rows.append(a + idx)
                keys.append(dataloader.kmer_deal_keys(whole.kmers[a + idx], "dna"))
            assert np.array_equal(np.sort(np.concatenate(rows)), np.arange(a, b))       # disjoint, complete
            sizes = [len(x) for x in rows]
            assert max(sizes) - min(sizes) <= 2, sizes                                  # (lag 5 < 6 letters: a bin is one k-mer)
            held = [kk for kk in keys if len(kk)]
            assert all(held[i].max() < held[i + 1].min() for i in range(len(held) - 1))   # rank r's k-mers all precede rank r + 1's
        # the reference-shaped iteration yields this rank's piece of every batch
        got = [km.shape[0] for km, _ in parts[1]]
        assert got == parts[1].piece_rows
        with pytest.raises(ValueError):
            parts[0].rank_pieces(1, world)
        with pytest.raises(ValueError):
            parts[0].shuffle(3)
        assert parts[0].repeat(3).repeats == 3 and parts[0].repeat(3).piece_rows == parts[0].piece_rows
