#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on MI355X: k-mer contexts/sec of the fused
DM-marginal + gradient hot path at k=13, fp64, with the ELBO checked against the CPU oracle.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--contexts C] [--workload net|ref]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over the rank's resident shard of a synthetic k=13 count
table (C contexts per GPU, weak scaling): the planned DM-marginal + d/dh kernel, its fixed-order
finalize kernel and, for N > 1, the single RCCL all-reduce of (ELBO, gradients) -- what
bear_net.py:290 + :278-282 do once per optimizer step.  Inputs are resident in HBM before the timed
region; the plan (count-only sort) is built once per table like the reference's
cached dataset and its build time is reported, not timed.

workload "net" (default, the headline of BASELINE.md section 3): bear_net DM-marginal + d/dh over
count rows + fp64 prior rows, 60 algorithmic bytes per context.  workload "ref": bear_ref with the
stop (flat) AR prior of BASELINE.json configs[1], train + reference count rows, 40 B per context.
The other workload is measured too and reported under "also".

Rank 0 prints ONE JSON line.
"""
import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is the measured copy ceiling
BYTES_PER_CONTEXT = {"net": 60, "ref": 40}
SEED = 20211012
FP64_PEAK_TFLOPS = 78.6    # MI355X fp64 vector (= matrix) peak, MI355X_MICROARCH.md

# ---- useful fp64 flops of the instruction-bound kernels: counted from the ALGORITHM as it is implemented
# (gfx950 has no fp64 transcendental hardware, so a log / exp / reciprocal is the polynomial it takes), fma = 2:
#   product path of an item with count c <= 24:  p, p' by the product rule = (add, mul, fma) per factor          4 c
#   table log 20, reciprocal (v_rcp_f64 + two Newton steps) 9, table exp 20
FLOPS_LOG, FLOPS_RCP, FLOPS_EXP = 20, 9, 20


def _item_stats(counts, ncol):
    """(number of non-zero cells, sum of min(c, 24) over them) of the first ncol columns: the work items of the DM kernels."""
    c = counts[:, :ncol]
    nz = c != 0
    return int(nz.sum()), int(torch.clamp(c, max=24).to(torch.int64).sum())


def flops_ref_items(train, ref):
    """dm_ref_items_kernel: only the items of contexts WITH reference counts are records (the others are a histogram):
    per record  alpha from (r_b, R): rcp + 2 fma = 13;  D, P: 4 c + log + rcp + mul = 4 c + 30;  four accumulators: 10."""
    has_ref = (ref[:, :4] != 0).any(dim=1)
    items, csum = _item_stats(train[has_ref], 4)
    return items * (13 + 30 + 10) + 4 * csum


def flops_linear(train, lag):
    """dm_linear_plan_kernel per LIVE context: softmax numerators as products of ng group rows 4 (ng - 1), normalisation 4 + rcp + 4,
    backward w / s / g = 5 + 5 + 8, its share of the wave reductions ~10;  per item: x = f u + eps 2, D / P 4 c + 30, q, -f q, sums 7."""
    ng = (max(lag - 3, 0) + 1) // 2 + 1
    live = int((train != 0).any(dim=1).sum())
    items, csum = _item_stats(train, 5)
    return live * (4 * (ng - 1) + 8 + FLOPS_RCP + 18 + 10) + items * (2 + 30 + 7) + 4 * csum


def flops_eval(test, n_h=1, n_van=3):
    """eval_plan_kernel<1,4> (1 h + AR + 3 van_reg) per row with held-out counts: AR arg-max 13, BEAR concentrations + arg-max
    10 + 5 + 13, -D(A, n) 4 n + log, vanilla table differences 2 each;  per held-out cell: AR c log(f + eps) log + 2, BEAR
    x + D 3 + 4 c + log, vanilla 2 each.  (The tie noise is integer hashing + fp32 transcendentals: not counted.)"""
    rows = int((test != 0).any(dim=1).sum())
    cells, csum = _item_stats(test, 5)
    nsum = int(torch.clamp(test.to(torch.int64).sum(dim=1), max=24).sum())
    return (rows * (13 + n_h * (28 + FLOPS_LOG) + 2 * n_van) + n_h * 4 * nsum
            + cells * (FLOPS_LOG + 2 + n_h * (3 + FLOPS_LOG) + 2 * n_van) + n_h * 4 * csum)


def flops_cnn(lag, fw, nf=30, l1=16):
    """(forward, backward) per context of make_ar_func_cnn (ar_funcs.py:49-99) at P = lag - fw + 1 positions: conv over a one-hot
    input = fw nf adds, layer norm over nf ~5 nf + rsqrt, elu nf exps, tensordot nf l1 fma;  head: layer norm l1, elu, l1 x 5 fma,
    softmax.  Backward = the gradient of every one of those products (2x) + the recomputed forward of a position (1x)."""
    pos = lag - fw + 1
    per_pos = fw * nf + 5 * nf + FLOPS_RCP + nf * FLOPS_EXP + 2 * nf * l1
    head = 5 * l1 + FLOPS_RCP + l1 * FLOPS_EXP + 2 * l1 * 5 + 5 * FLOPS_EXP + FLOPS_RCP + 5
    fwd = pos * per_pos + head
    return fwd, 3 * fwd


def fp64_roofline(flops, ms, per="launch"):
    tf = flops / (ms * 1e-3) / 1e12
    return {"bound": "fp64 VALU (instruction-bound)", "useful_fp64_flops_per_%s" % per: flops, "achieved_TFLOPs": tf,
            "peak_TFLOPs": FP64_PEAK_TFLOPS, "frac_of_fp64_peak": tf / FP64_PEAK_TFLOPS}


def cnn_rates(credited, executed, ms):
    """The convolutional kernels in k-mer order SKIP work that neighbouring contexts share, so two rates: `credited` = what a
    context-by-context evaluation would execute (flops_cnn per context) over the time -- an algorithmic speed, NOT a utilisation
    and not bounded by the peak -- and `executed` = the flops of the evaluations actually carried out (None where the kernel
    decides per wave what to share and no count exists) over the same time, which is the fraction of the fp64 peak."""
    out = {"credited_fp64_flops": credited, "credited_TFLOPs": credited / (ms * 1e-3) / 1e12, "peak_TFLOPs": FP64_PEAK_TFLOPS,
           "executed_fp64_flops": executed}
    if executed is not None:
        out.update(executed_TFLOPs=executed / (ms * 1e-3) / 1e12, frac_of_fp64_peak=executed / (ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS)
    return out


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: the card needs ~30 launches (25 ms) after idle to ramp its clocks up (scripts/dev/ramp.py: 1.1 -> 0.79 ms per
    # launch of the headline kernel); 100 untimed + 200 timed steps are a quarter of a second
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--contexts", type=float, default=1e8,
                    help="contexts per GPU (--scaling weak) or of the whole table, cut into one contiguous row shard per rank (--scaling strong)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: every rank holds --contexts rows; strong: ONE table of --contexts rows split over the ranks "
                         "(north_star's 10^8 table at 1/2/4/8 GPUs)")
    ap.add_argument("--no-settle", action="store_true", help="skip the clock-settle launches in front of the measurement")
    ap.add_argument("--force-collective", action="store_true",
                    help="run the N > 1 code path (process group, per-step all-reduce, gathers) although WORLD_SIZE is 1: the only "
                         "way to exercise the RCCL path of this file on a one-GPU box (tests/test_dist_gpu.py)")
    ap.add_argument("--workload", choices=["net", "ref"], default="net")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-unnormalised-rows", action="store_true",
                    help="skip also.net_rows_not_normalised: it launches the HEADLINE's kernel instantiation on other data (2.5x slower per "
                         "launch), which a per-kernel-name profile of this command cannot tell apart (scripts/profile_round.sh passes it)")
    ap.add_argument("--no-baseline-configs", action="store_true",
                    help="skip also.baseline_configs / also.dense_table (scripts/baseline_configs.py: ~20 s)")
    ap.add_argument("--no-strong-pass", action="store_true",
                    help="N > 1, weak scaling: skip the additional strong-scaling pass (also.strong_scaling)")
    ap.add_argument("--ingest-rows", type=float, default=None,
                    help="rows of the text table of the also.ingest entry (text -> first step wall time); default: min(--contexts, 1e8); 0 = skip")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for single-GPU smoke tests of the N > 1 path)")
    return ap.parse_args()


def self_launch(n_ranks):
    """`python bench.py --gpus N` without a launcher (the reference needs none for its replicas either: MirroredStrategy() inside
    the process, bear_net.py:246).  One process per GPU is this framework's model, so the bare command starts
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py <same args>`
    as a CHILD process (never an exec: this process stays alive, has not initialised the GPU and does not), relays rank 0's JSON
    line on stdout (anything else the ranks print goes to stderr) and returns the child's exit code -- non-zero when any rank
    dies; no retry, no fallback to fewer ranks."""
    import socket
    import subprocess
    if "BEAR_BENCH_DEVICE" not in os.environ:        # (that override puts every rank on one card: single-GPU tests of this path)
        have = torch.cuda.device_count()             # counting devices does not initialise the GPU
        if have < n_ranks:
            print(f"bench.py: --gpus {n_ranks} but this node shows {have} GPU(s)", file=sys.stderr)
            return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), BEAR_BENCH_LAUNCHER="self")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: RCCL between processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n_ranks) // n_ranks)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for ln in child.stdout:                                  # stderr is inherited: the ranks' diagnostics stream through live
        if ln.startswith('{"metric"'):
            sys.stdout.write(ln)
            sys.stdout.flush()
        else:
            sys.stderr.write(ln)
    return child.wait()


def strong_pass(args, rank, world, dev, dev_index, wl, params):
    """ONE table of --contexts rows cut into `world` contiguous row shards (what `--scaling strong` measures as the headline),
    timed right after the weak measurement of the same launch: barrier + synchronize around K steps, MAX over ranks."""
    from bear_amd import kernels
    h_s, tau_s, nu_s = params
    total = int(args.contexts)
    base, rem = divmod(total, world)
    row0, n = rank * base + min(rank, rem), base + (1 if rank < rem else 0)
    want = ("train",) if wl == "net" else ("train", "ref")
    t = kernels.synth_counts(SEED, row0, n, dev, want=want)
    prior = kernels.synth_prior(SEED, row0, n, dev) if wl == "net" else None
    plan = kernels.Plan(t["train"], 5) if wl == "net" else kernels.Plan(t["train"], 4, ref=t["ref"])
    out = torch.zeros(2 if wl == "net" else 4, dtype=torch.float64, device=dev)

    def step(ev=None):
        if ev is not None:
            ev[0].record()
        if wl == "net":
            kernels.dm_prior_planned(plan, prior, h_s, out=out)
        else:
            kernels.dm_ref_planned(plan, t["ref"], h_s, tau_s, nu_s, out=out)
        if ev is not None:
            ev[1].record()
        dist.all_reduce(out)
        if ev is not None:
            ev[2].record()
    for _ in range(max(args.warmup, 50)):       # (the card is warm from the weak pass; the shard's own first launches are not)
        step()
    evs = [tuple(torch.cuda.Event(enable_timing=True) for _ in range(3)) for _ in range(args.steps)]
    dist.barrier()
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    for k in range(args.steps):
        step(evs[k])
    dist.barrier()
    torch.cuda.synchronize()
    el = torch.tensor([time.perf_counter() - t_start], dtype=torch.float64, device=dev)
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    mine = torch.tensor([float(np.mean([e[0].elapsed_time(e[1]) for e in evs])), float(np.mean([e[1].elapsed_time(e[2]) for e in evs])),
                         float(n), float(dev_index)], dtype=torch.float64, device=dev)
    allr = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allr, mine)
    bpc = BYTES_PER_CONTEXT[wl]
    return {"scaling": "strong", "contexts_total": total, "value": total * args.steps / elapsed, "unit": "contexts/s",
            "ms_per_step": elapsed / args.steps * 1e3, "steps": args.steps,
            "frac_of_n_gpu_hbm_roofline": total * bpc / (elapsed / args.steps) / 1e9 / (HBM_PEAK_GBPS * world),
            "per_rank": [{"rank": r, "device": int(v[3].item()), "contexts": int(v[2].item()), "kernel_ms": float(v[0].item()),
                          "allreduce_ms": float(v[1].item())} for r, v in enumerate(allr)],
            "result": out.cpu().numpy().tolist(),
            "note": "the same step (planned kernel + one packed all-reduce) on ONE table of contexts_total rows cut into one contiguous "
                    "shard per rank; measured after the weak pass of this launch"}


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # bare `python bench.py --gpus N`: this process becomes the launcher (it never touches the GPU) and the ranks are its children
        raise SystemExit(self_launch(args.gpus))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} under a launcher with WORLD_SIZE={world}: the two must agree")
    # an EXTERNAL launcher (`python -m torch.distributed.run ... bench.py --gpus N`) may not have exported it: dmabuf IPC, which
    # RCCL between processes needs on this host driver; read by the HSA runtime at the first GPU call, so set before it
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    assert torch.cuda.is_available(), "bench.py needs MI355X devices"
    dev_index = int(os.environ.get("BEAR_BENCH_DEVICE", local_rank))  # override only for single-GPU smoke tests
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    multi = world > 1 or args.force_collective      # the N > 1 code path
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            if args.backend == "gloo" and os.environ.get("MASTER_ADDR") in ("127.0.0.1", "localhost"):
                os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")   # do not resolve the host name to pick an interface
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    from bear_amd import kernels

    # contiguous row shards of one global table: weak = --contexts rows on every rank, strong = one
    # table of --contexts rows cut into `world` pieces (dist.shard_rows: the first `total % world` ranks hold one more row)
    if args.scaling == "strong":
        total = int(args.contexts)
        base, extra = divmod(total, world)
        row0 = rank * base + min(rank, extra)
        n = base + (1 if rank < extra else 0)
    else:
        n = int(args.contexts)
        row0, total = rank * n, int(args.contexts) * world
    h_s, tau_s, nu_s = 0.0, float(np.log(1 / 30)), float(-np.log(100))  # reference initial values

    t = kernels.synth_counts(SEED, row0, n, dev, want=("train", "ref"))
    prior = kernels.synth_prior(SEED, row0, n, dev)
    torch.cuda.synchronize()
    t0 = time.time()
    # "ref": the product path of bear_ref.train -- a plan that also knows the (equally constant) reference column
    plans = {"net": kernels.Plan(t["train"], 5), "ref": kernels.Plan(t["train"], 4, ref=t["ref"])}
    torch.cuda.synchronize()
    plan_build_s = time.time() - t0
    plan_nbytes = {k: v.nbytes for k, v in plans.items()}     # (before the linear head's paired lists are attached to the net plan below)

    outs = {"net": torch.zeros(2, dtype=torch.float64, device=dev), "ref": torch.zeros(4, dtype=torch.float64, device=dev)}

    def launch(wl):
        if wl == "net":
            kernels.dm_prior_planned(plans["net"], prior, h_s, out=outs["net"])
        else:
            kernels.dm_ref_planned(plans["ref"], t["ref"], h_s, tau_s, nu_s, out=outs["ref"])

    def step(wl, ev=None):
        if ev is not None:
            ev[0].record()
        launch(wl)
        if ev is not None:
            ev[1].record()
        if multi:
            dist.all_reduce(outs[wl])  # one packed RCCL all-reduce of (ELBO, gradients) per step
            if ev is not None:
                ev[2].record()         # the collective is ordered into the launch stream: ev[1] -> ev[2] is its latency there

    def barrier():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    def settle(wl, group=8, tol=0.01, cap_s=1.0, cap_launches=1600):
        """The card ramps its clocks for the first tens of launches after idle (scripts/dev/ramp.py: 1.10 -> 0.79 ms per
        launch of the headline kernel over ~25 ms), and `--warmup 5` ends inside that ramp.  So before the warmup the same
        kernel is launched -- no collective, nothing timed into `value` -- in event-timed groups of 8, two groups in
        flight so the card never idles, until three consecutive group means agree within 1 % (cap: 1 s).  Disclosed in the
        bench line as "settle"; --no-settle skips it."""
        t_host = time.perf_counter()
        means, pending, launches = [], [], 0
        def enqueue():
            nonlocal launches
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(group):
                launch(wl)
            b.record()
            launches += group
            pending.append((a, b))
        enqueue()
        settled = False
        while launches < cap_launches and time.perf_counter() - t_host < cap_s:
            enqueue()
            a, b = pending.pop(0)
            b.synchronize()
            means.append(a.elapsed_time(b) / group)
            if len(means) >= 3 and max(means[-3:]) <= (1.0 + tol) * min(means[-3:]):
                settled = True
                break
        torch.cuda.synchronize()
        for a, b in pending:
            means.append(a.elapsed_time(b) / group)
        return {"launches": launches, "ms": (time.perf_counter() - t_host) * 1e3, "settled": settled,
                "first_group_ms_per_launch": means[0], "last_group_ms_per_launch": means[-1],
                "rule": "groups of %d launches of the timed kernel until 3 consecutive group means agree within %.0f %% (cap %.1f s); "
                        "untimed, before --warmup" % (group, tol * 100, cap_s)}

    def cold_run(wl, steps, warmup):
        """What the same command reads WITHOUT the settle launches (`--no-settle`): `warmup` untimed steps, then `steps` timed ones,
        on a card that idled through the set-up -- the figure of the driver's `--warmup 5` protocol alone, kept in the record."""
        for _ in range(warmup):
            step(wl)
        barrier()
        t_start = time.perf_counter()
        for _ in range(steps):
            step(wl)
        barrier()
        return (time.perf_counter() - t_start) / steps * 1e3

    def measure(wl, steps, warmup, do_settle=False):
        gc.collect()         # now, not between the warm-up and the timed steps: the card must not idle there
        cold_ms = cold_run(wl, min(steps, 20), warmup) if do_settle else None
        settle_info = settle(wl) if do_settle else None
        if settle_info is not None:
            settle_info["cold_ms_per_step_without_settle"] = cold_ms
            settle_info["cold_note"] = ("ms_per_step of min(steps, 20) timed steps behind --warmup alone, taken BEFORE the settle launches "
                                        "on the card as the set-up left it (what --no-settle reports)")
        for _ in range(warmup):
            step(wl)
        evs = [tuple(torch.cuda.Event(enable_timing=True) for _ in range(3 if multi else 2)) for _ in range(steps)]
        gc.disable()         # no collector pause between two launches of the timed region (the GPU queue is only K steps deep)
        barrier()
        t_start = time.perf_counter()
        for k in range(steps):
            step(wl, evs[k])
        barrier()
        elapsed = time.perf_counter() - t_start
        gc.enable()
        # kernel-only duration (main kernel + finalize) of every step, HIP events on the launch stream
        per_step = np.array([e[0].elapsed_time(e[1]) for e in evs])
        k_ms = float(per_step.mean())
        stats = {"kernel_ms_min": float(per_step.min()), "kernel_ms_median": float(np.median(per_step)),
                 "kernel_ms_p90": float(np.percentile(per_step, 90)), "kernel_ms_max": float(per_step.max())}
        ar_ms = float(np.mean([e[1].elapsed_time(e[2]) for e in evs])) if multi else None
        per_rank = None
        if multi:
            el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
            elapsed = float(el.item())
            mine = torch.tensor([k_ms, ar_ms, float(n), float(dev_index)], dtype=torch.float64, device=dev)
            allr = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(allr, mine)
            per_rank = [{"rank": r, "device": int(v[3].item()), "contexts": int(v[2].item()), "kernel_ms": float(v[0].item()),
                         "allreduce_ms": float(v[1].item())} for r, v in enumerate(allr)]
        return elapsed, k_ms, outs[wl].cpu().numpy().copy(), stats, settle_info, per_rank

    primary = args.workload
    other = "ref" if primary == "net" else "net"
    elapsed, k_ms, result, k_stats, settle_info, per_rank = measure(primary, args.steps, args.warmup, do_settle=not args.no_settle)
    o_elapsed, o_k_ms, o_result, _, _, _ = measure(other, max(5, args.steps // 5), 2)
    def timed(fn, reps):
        """Median over five or more event-timed groups of back-to-back calls of `fn` (each group at least `reps` / 5 calls and
        ~1 ms), behind ~20 ms of untimed calls: every entry below follows host-side set-up during which the card clocks down
        (see settle() above), and a one-off host stall inside a group (a GC pause, an allocator call) must not become the
        figure -- one default run of round 3 read 1.65 ms for a 0.375 ms kernel that way."""
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        first = max(e0.elapsed_time(e1), 1e-3)
        for _ in range(min(64, int(20.0 / first))):
            fn()
        per_group = max(1, -(-reps // 5), min(64, int(1.0 / first) + 1))
        n_groups = 5 if first > 5.0 else 7
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_groups)]
        for a, b in evs:
            a.record()
            for _ in range(per_group):
                fn()
            b.record()
        torch.cuda.synchronize()
        return float(np.median([a.elapsed_time(b) / per_group for a, b in evs]))

    norm_ms = None
    if rank == 0:  # the same net kernel with the caller asserting normalised prior rows
        norm_ms = timed(lambda: kernels.dm_prior_planned(plans["net"], prior, h_s, normalized=True), 10)

    stream_gbps = None
    if rank == 0:   # measured read-only stream over the same prior rows (4 GB): the practical ceiling under the spec peak
        stream_gbps = prior.numel() * 8 / (timed(lambda: kernels.stream_read(prior), 10) * 1e-3) / 1e9

    extra = {}
    if rank == 0 and world == 1:  # the rows of SURVEY 8f built on the same kernels: kernel-only times, 1 GPU
        plan_stream = kernels.Plan(t["train"], 4)     # mode R with the reference rows streamed every step (plan from the training counts only)
        ms = timed(lambda: kernels.dm_ref_planned(plan_stream, t["ref"], h_s, tau_s, nu_s), 10)
        extra["ref_streaming_reference_rows"] = {"kernel_ms": ms, "contexts_per_s": n / (ms * 1e-3), "achieved_GBps": n * 40 / (ms * 1e-3) / 1e9,
                                                 "moved_bytes_per_context": 20 + plan_stream.nbytes / n,
                                                 "note": "dm_ref_plan_kernel: 20 B reference row + plan per context; the 'ref' entry above is the "
                                                         "reference-aware plan (bear_plan_create_ref), which streams only the items of contexts with reference counts"}
        del plan_stream
        ms = timed(lambda: kernels.dm_prior_planned(plans["net"], prior, h_s, want_grad=True), 5)
        extra["net_with_gradient_rows"] = {"kernel_ms": ms, "contexts_per_s": n / (ms * 1e-3)}
        ms_n = timed(lambda: kernels.dm_prior_planned(plans["net"], prior, h_s, want_grad=True, normalized=True), 5)
        moved = 40 + plan_nbytes["net"] / n + 40      # prior rows in, plan in, gradient rows out
        extra["net_with_gradient_rows"].update({
            "kernel_ms_rows_asserted_normalized": ms_n, "contexts_per_s_rows_asserted_normalized": n / (ms_n * 1e-3),
            "moved_bytes_per_context": moved, "moved_GBps_rows_asserted_normalized": n * moved / (ms_n * 1e-3) / 1e9,
            "note": "what any torch ar_func trains through (bear_net.train passes the assertion for AR functions that end in a "
                    "softmax: dm_prior_plan_grad_inplace_kernel, double-buffered; without it dm_prior_plan_grad_kernel)"})
        ms = timed(lambda: kernels.dm_prior_planned(plans["net"], prior, h_s, train_ar=True), 5)
        extra["net_multinomial_mode"] = {"kernel_ms": ms, "contexts_per_s": n / (ms * 1e-3)}
        # prior rows that are NOT normalised (a plugin whose rows do not end in a softmax): every context then forms its own
        # A = u sum f + 5 eps, one log and one reciprocal (kernels_plan.h, the branch behind SRT_SUM1_TOL), instead of the table
        # look-up the softmax rows of the headline take.  Rows scaled by 1 + 1e-3 x a hash of the row number.
        scale_rows = 1.0 + 1e-3 * ((torch.arange(n, device=dev, dtype=torch.int64) * 2654435761 % 1000003).to(torch.float64) / 1000003.0)
        prior_un = (prior * scale_rows[:, None]).contiguous()
        del scale_rows
        ms_un = ms_un_g = float("nan")
        if not args.no_unnormalised_rows:
            ms_un = timed(lambda: kernels.dm_prior_planned(plans["net"], prior_un, h_s), 5)
            ms_un_g = timed(lambda: kernels.dm_prior_planned(plans["net"], prior_un, h_s, want_grad=True), 5)
        del prior_un
        extra["net_rows_not_normalised"] = None if args.no_unnormalised_rows else {
            "kernel_ms": ms_un, "contexts_per_s": n / (ms_un * 1e-3), "credited_GBps_at_60_B": n * 60 / (ms_un * 1e-3) / 1e9,
            "frac_credited": n * 60 / (ms_un * 1e-3) / 1e9 / HBM_PEAK_GBPS, "kernel_ms_with_gradient_rows": ms_un_g,
            "note": "dm_prior_plan_kernel<false,false> on rows whose sums differ from 1 by up to 1e-3: the per-context own-A path; "
                    "the headline's rows are softmax outputs (every reference AR function) and take the shared-A table"}
        lag = 13
        mat = 0.05 * torch.randn(lag, 5, 5, dtype=torch.float64, device=dev, generator=torch.Generator(dev).manual_seed(10))
        # ---- the AR-function entries (fused linear head, linear rows, convolutional head) run on a table of DISTINCT contexts: row
        # index -> a fixed bijection of [0, 4^13) (kernels.synth_kmer_ids).  Only 4^13 = 6.7e7 13-mers exist, so the k = 13
        # table of these entries is the first 6.0e7 rows of the synthetic table (every entry states its contexts and a
        # per-1e8-contexts figure); rounds 1-5 drew 1e8 13-mers WITH replacement (28 % of the k-mers with counts occurred twice or
        # more: no count table looks like that, summarize.py:429-449) -- that figure stays one more round as `..._with_replacement`.
        n_all, n = n, min(n, 60_000_000)
        per_1e8 = 1e8 / n
        train_all, ref_all, prior_all, plan_all = t["train"], t["ref"], prior, plans["net"]
        t = dict(t, train=t["train"][:n], ref=t["ref"][:n])
        prior = prior[:n]
        plans = dict(plans, net=plan_all if n == n_all else kernels.Plan(t["train"], 5))
        ids = kernels.synth_kmer_ids(SEED, 0, n, lag, dev)
        shifts = torch.arange(2 * (lag - 1), -1, -2, dtype=torch.int64, device=dev)
        to_codes = lambda v: ((v[:, None] >> shifts[None, :]) & 3).to(torch.int8).contiguous()
        packed_raw = kernels.pack_kmers(to_codes(ids))      # 3 bits per letter, table order: what the convolutional head reads
        packed = kernels.linear_index(packed_raw, lag)      # table-row words: what the linear head reads
        ms_shuffled = timed(lambda: kernels.dm_linear(plans["net"], packed, mat, h_s), 5)
        # bear_net.train sorts the rows of a batch by k-mer at upload; the synthetic counts are independent of the contexts, so
        # sorting the contexts alone gives the same kind of table in that order
        packed_sorted_raw = kernels.pack_kmers(to_codes(torch.sort(ids).values))
        packed = kernels.linear_index(packed_sorted_raw, lag)
        del ids
        ms_plain = timed(lambda: kernels.dm_linear(plans["net"], packed, mat, h_s), 5)
        plain_out = [x.clone() for x in kernels.dm_linear(plans["net"], packed, mat, h_s)]
        # ... and as bear_net.train runs it on a sorted batch: neighbouring contexts that share all letters but the last three
        # taken two at a time (bear_plan_pair_contexts, once per batch; same sums)
        lin_paired = plans["net"].pair_contexts(packed, lag)
        ms = timed(lambda: kernels.dm_linear(plans["net"], packed, mat, h_s), 5)
        paired_out = kernels.dm_linear(plans["net"], packed, mat, h_s)
        lin_same = bool(torch.allclose(paired_out[0], plain_out[0], rtol=1e-12, atol=0)
                        and float((paired_out[1] - plain_out[1]).abs().max()) <= 1e-10 * float(plain_out[1].abs().max()))
        del packed, plain_out, paired_out
        extra["linear_head_fused_step"] = {"lag": lag, "contexts": n, "kmers": "distinct (bijection of the row index)",
                                           "kernel_ms": ms, "kernel_ms_per_1e8_contexts": ms * per_1e8, "contexts_per_s": n / (ms * 1e-3),
                                           "paired_contexts": lin_paired, "paired_equals_plain": lin_same,
                                           "paired_tiles_and_plain_tiles": list(plans["net"].pair_info()),
                                           "kernel_ms_plain_lists": ms_plain,
                                           "kernel_ms_rows_in_random_order": ms_shuffled,
                                           "roofline": fp64_roofline(flops_linear(t["train"], lag), ms),
                                           "note": "forward + ELBO + d/dh + d/dmat from 8-byte context words, rows in k-mer order "
                                                   "(as bear_net.train uploads a batch)"}
        # the linear AR function as rows (evaluation, bear_ref with the linear net function) and bear_ref's mixing of net rows with
        # the reference prior: one bandwidth-bound launch per direction each
        pr_l = kernels.linear_forward(packed_sorted_raw, mat, lag)
        lf_ms = timed(lambda: kernels.linear_forward(packed_sorted_raw, mat, lag), 5)
        _, g_l = kernels.dm_prior_planned(plans["net"], pr_l, h_s, want_grad=True, normalized=True)
        lb_ms = timed(lambda: kernels.linear_backward(packed_sorted_raw, lag, pr_l, g_l), 5)
        lb_ms_random = timed(lambda: kernels.linear_backward(packed_raw, lag, pr_l, g_l), 5)
        ref_in = t["ref"].to(torch.float64) + 1e-7       # bear_ref.py:332-337 (synthetic counts are far below 2^31)
        ref_in[:, -1] = 0
        tau_s_dev = torch.tensor(tau_s, dtype=torch.float64, device=dev)
        nu_s_dev = torch.tensor(nu_s, dtype=torch.float64, device=dev)
        mf_ms = timed(lambda: kernels.ref_mix_forward(pr_l, ref_in, tau_s_dev, nu_s_dev), 5)
        mb_ms = timed(lambda: kernels.ref_mix_backward(pr_l, ref_in, g_l, tau_s_dev, nu_s_dev), 5)
        # ... and bear_ref's whole step on given net rows with the mixing inside the DM kernel (what bear_ref.train runs in BEAR mode)
        h_s_dev = torch.tensor([h_s], dtype=torch.float64, device=dev)
        fused_ms = timed(lambda: kernels.dm_refmix_planned_dev(plans["net"], pr_l, ref_in, h_s_dev, tau_s_dev, nu_s_dev), 5)
        del pr_l, g_l, ref_in
        extra["ar_function_rows"] = {
            "lag": lag,
            "linear_forward_ms": lf_ms, "linear_forward_GBps": n * 48 / (lf_ms * 1e-3) / 1e9,
            "linear_backward_ms": lb_ms, "linear_backward_GBps": n * 88 / (lb_ms * 1e-3) / 1e9,
            "linear_backward_ms_rows_in_random_order": lb_ms_random,
            "ref_mix_forward_ms": mf_ms, "ref_mix_forward_GBps": n * 120 / (mf_ms * 1e-3) / 1e9,
            "ref_mix_backward_ms": mb_ms, "ref_mix_backward_GBps": n * 160 / (mb_ms * 1e-3) / 1e9,
            "ref_mix_dm_step_fused_ms": fused_ms, "ref_mix_dm_step_fused_GBps": n * (120 + plan_nbytes["net"] / n) / (fused_ms * 1e-3) / 1e9,
            "note": "bear_linear_forward / backward_f64 (8 + 40 B; 8 + 40 + 40 B per context, rows in k-mer order) and "
                    "bear_ref_mix_forward / backward_f64 (40 + 40 + 40 B; 3 x 40 + 40 B): what evaluation and bear_ref.train with a "
                    "parametrised net function call; HBM-bound, GB/s on those bytes.  ref_mix_dm_step_fused: bear_dm_refmix_plan_grad_f64 = "
                    "mixing + sum LL + all gradients in one launch (40 + 40 B + plan in, 40 B out), which bear_ref.train uses in BEAR mode "
                    "instead of mix-forward + gradient rows + mix-backward"}
        # BASELINE configs[4]: the convolutional AR function, forward + DM step with gradient rows + backward
        from bear_amd import ar_funcs
        fw = 8
        _, cnn_params = ar_funcs.make_ar_func_cnn(lag, 4, filter_width=fw, device=dev, generator=torch.Generator(dev).manual_seed(10))
        flat = torch.cat([q.detach().reshape(-1) for q in cnn_params]).contiguous()
        packed = packed_raw
        f_ms = timed(lambda: kernels.cnn_forward(packed, flat, lag, fw), 3)
        pr_c, t1_c = kernels.cnn_forward(packed, flat, lag, fw)
        _, g_c = kernels.dm_prior_planned(plans["net"], pr_c, h_s, want_grad=True)
        b_ms = timed(lambda: kernels.cnn_backward(packed, flat, lag, fw, t1_c, pr_c, g_c), 3)
        del pr_c, t1_c, g_c
        # the same two kernels over all rows in k-mer order: a wave's contexts then share their leading letters and both kernels
        # evaluate a window the whole wave / tile shares once (the backward: per distinct window, from the column sums of dT1)
        fs_ms = timed(lambda: kernels.cnn_forward(packed_sorted_raw, flat, lag, fw), 3)
        pr_c, t1_c = kernels.cnn_forward(packed_sorted_raw, flat, lag, fw)
        _, g_c = kernels.dm_prior_planned(plans["net"], pr_c, h_s, want_grad=True)
        bs_ms = timed(lambda: kernels.cnn_backward(packed_sorted_raw, flat, lag, fw, t1_c, pr_c, g_c), 3)
        del pr_c, t1_c, g_c
        # the training step as bear_net.train enqueues it (bear_net_cnn_train_reduce_f64): the same three kernels, forward and
        # backward over the plan's lists of contexts that hold training counts only (the others' gradient rows are zero)
        theta = torch.cat([torch.zeros(1, dtype=torch.float64, device=dev), flat]).contiguous()
        bufs = kernels.cnn_step_buffers(n, lag, fw, dev)
        pk = torch.zeros(2 + flat.numel(), dtype=torch.float64, device=dev)
        s_ms = timed(lambda: kernels.net_cnn_train_reduce(plans["net"], packed_sorted_raw, lag, fw, theta, bufs, pk), 3)
        s_ms_random = timed(lambda: kernels.net_cnn_train_reduce(plans["net"], packed, lag, fw, theta, bufs, pk), 3)
        # ... and as bear_net.train holds a batch since the end of round 2: in k-mer order AND without the contexts that hold no
        # training counts (they add exactly nothing; the loss scale keeps the full batch size) -- a smaller table, all rows live
        keep = (t["train"] != 0).any(dim=1).nonzero().squeeze(1)
        tr_kept = t["train"].index_select(0, keep).contiguous()
        packed_kept = packed_sorted_raw.index_select(0, keep).contiguous()
        plan_kept = kernels.Plan(tr_kept, 5)
        bufs_kept = tuple(b[:keep.numel()] for b in bufs)
        kept_plain_ms = timed(lambda: kernels.net_cnn_train_reduce(plan_kept, packed_kept, lag, fw, theta, bufs_kept, pk), 3)
        pk_plain = pk.clone()
        # ... with prefix levels (bear_plan_attach_cnn_levels, as bear_net.train attaches them to every sorted batch): a position is
        # evaluated once per distinct prefix of the batch, forward and backward
        n_levels = plan_kept.attach_cnn_levels(packed_kept, lag, fw)
        level_rows, level_letters = plan_kept.cnn_level_rows(with_letters=True)
        window_tables = plan_kept.cnn_window_rows()         # [(level, position, distinct windows)]
        kept_ms = timed(lambda: kernels.net_cnn_train_reduce(plan_kept, packed_kept, lag, fw, theta, bufs_kept, pk), 3)
        levels_same = bool(abs(float(pk[0] - pk_plain[0])) <= 1e-12 * abs(float(pk_plain[0]))
                           and float((pk[2:] - pk_plain[2:]).abs().max()) <= 1e-10 * float(pk_plain[2:].abs().max()))
        kept_fwd_ms = timed(lambda: kernels.cnn_forward(packed_kept, flat, lag, fw, plan=plan_kept), 3)
        del pk_plain
        # the gradient-row kernel as bear_net.train runs it for an AR function made of torch ops: on the kept table (every row holds
        # counts, so no gradient row of zeros is written and no prior row of a context without counts is read)
        lin_kept = kernels.linear_index(packed_kept, lag)
        plan_kept.pair_contexts(lin_kept, lag)
        lin_k_ms = timed(lambda: kernels.dm_linear(plan_kept, lin_kept, mat, h_s), 5)
        extra["linear_head_fused_step"]["kernel_ms_as_bear_net_train_holds_the_batch"] = lin_k_ms
        extra["linear_head_fused_step"]["kernel_ms_per_1e8_contexts_as_bear_net_train_holds_the_batch"] = lin_k_ms * per_1e8
        kept_frac = keep.numel() / n
        del keep, tr_kept, packed_kept, plan_kept, bufs_kept, lin_kept
        cnn_f, cnn_b = flops_cnn(lag, fw)
        # flops the level launches execute: level k < K one position per row, the last level the positions that are left; the head
        # (layer 1 onwards) per context
        n_kept = int(kept_frac * n + 0.5)
        pos_f = (cnn_f - flops_cnn(fw, fw)[0]) / (lag - fw)            # one position's share of a context's forward flops
        head_f = cnn_f - (lag - fw + 1) * pos_f
        lv, ll = [n_kept] + level_rows, [lag] + level_letters + [fw - 1]      # level k evaluates the positions p with p + fw in (ll[k + 1], ll[k]]
        # ... minus the positions a level takes from its window tables (evaluated once per distinct window instead)
        n_win = [sum(1 for w in window_tables if w[0] == k) for k in range(len(lv))]
        pos_evals = sum(r * (ll[k] - ll[k + 1] - n_win[k]) for k, r in enumerate(lv)) + sum(w[2] for w in window_tables)
        exec_f = pos_evals * pos_f + n_kept * head_f
        extra["cnn_head"] = {"lag": lag, "filter_width": fw, "contexts": n, "kmers": "distinct (bijection of the row index)",
                             "train_step_ms_per_1e8_contexts_as_bear_net_train_holds_the_batch": kept_ms * per_1e8,
                             "forward_ms": f_ms, "backward_ms": b_ms,
                             "rates_forward_rows_in_kmer_order": cnn_rates(cnn_f * n, None, fs_ms),
                             "rates_backward_rows_in_kmer_order": cnn_rates(cnn_b * n, None, bs_ms),
                             "roofline_forward_rows_in_random_order": fp64_roofline(cnn_f * n, f_ms),
                             "roofline_backward_rows_in_random_order": fp64_roofline(cnn_b * n, b_ms),
                             "forward_ms_rows_in_kmer_order": fs_ms, "backward_ms_rows_in_kmer_order": bs_ms,
                             "all_rows_step_ms": f_ms + b_ms + extra["net_with_gradient_rows"]["kernel_ms"],
                             "train_step_ms": s_ms, "train_step_ms_rows_in_random_order": s_ms_random,
                             "train_step_ms_as_bear_net_train_holds_the_batch": kept_ms, "contexts_with_training_counts": kept_frac,
                             "train_step_ms_without_prefix_levels": kept_plain_ms,
                             "prefix_levels": {"attached": n_levels, "rows": lv, "prefix_letters": ll[:-1], "position_evaluations_per_context": pos_evals / max(n_kept, 1),
                                               "equals_step_without_levels": levels_same,
                                               "window_tables_level_position_windows": [list(w) for w in window_tables],
                                               "forward_ms": kept_fwd_ms,
                                               "forward_ms_scaled_to_all_contexts": kept_fwd_ms / max(kept_frac, 1e-9),
                                               "rates_forward": cnn_rates(cnn_f * n_kept, exec_f, kept_fwd_ms),
                                               "rates_step_forward_plus_backward": cnn_rates((cnn_f + cnn_b) * n_kept, exec_f * (1 + cnn_b / cnn_f),
                                                                                           kept_ms - extra["net_with_gradient_rows"]["kernel_ms_rows_asserted_normalized"] * kept_frac),
                                               "note": "bear_plan_attach_cnn_levels: rows[k] = distinct prefixes of lag - k letters of the sorted batch; level k "
                                                       "evaluates position P - 1 - k once per row (the last level the positions left), forward and backward; "
                                                       "window tables (round 5): a level's position once per DISTINCT filter_width-letter window of the "
                                                       "batch (65 536), its rows gather the window's layer-1 row forward and are summed by window backward"},
                             "step_contexts_per_s": n / (kept_ms * 1e-3),
                             "note": "forward_ms / backward_ms: bear_cnn_forward_f64 / bear_cnn_backward_f64 over all rows in random order (any caller); "
                                     "train_step_ms: bear_net_cnn_train_reduce_f64 = forward + planned DM kernel with gradient rows + "
                                     "backward over the contexts that hold training counts (70 % of this table), rows in k-mer order as "
                                     "bear_net.train uploads a batch (both kernels evaluate a window that a wave's / tile's contexts share once); "
                                     "train_step_ms_as_bear_net_train_holds_the_batch: the same step on the table bear_net.train keeps resident -- "
                                     "k-mer order, the contexts without training counts left out, prefix levels attached (step_contexts_per_s counts all "
                                     "1e8).  rates_*: credited flops / time is an algorithmic speed, not a utilisation (sorted-order kernels skip shared "
                                     "work); frac_of_fp64_peak only where the executed flops are counted"}
        del packed, packed_raw, packed_sorted_raw, bufs, pk, theta
        # ... the fused linear head once more on rounds 1-5's table (1e8 13-mers drawn with replacement, sorted, paired) and on 1e8
        # DISTINCT 14-mers (density 0.37 of all 14-mers: a block of equal leading letters holds 24 of its 64 contexts, not 57)
        if n_all > n:
            for label, lag_x, draw in (("with_replacement", 13, True), ("k14_distinct", 14, False)):
                if draw:
                    c = torch.randint(0, 4, (n_all, lag_x), dtype=torch.int8, device=dev, generator=torch.Generator(dev).manual_seed(SEED))
                    key = torch.zeros(n_all, dtype=torch.int64, device=dev)
                    for l in range(lag_x):
                        key = key * 4 + c[:, l].to(torch.int64)
                    c = c[torch.argsort(key)].contiguous()
                    del key
                else:
                    c = kernels.synth_kmer_codes(SEED, 0, n_all, lag_x, dev, sort=True)
                idx = kernels.linear_index(kernels.pack_kmers(c), lag_x)
                del c
                mat_x = 0.05 * torch.randn(lag_x, 5, 5, dtype=torch.float64, device=dev, generator=torch.Generator(dev).manual_seed(10))
                paired_x = plan_all.pair_contexts(idx, lag_x)
                ms_x = timed(lambda: kernels.dm_linear(plan_all, idx, mat_x, h_s), 5)
                extra["linear_head_fused_step"]["kernel_ms_1e8_contexts_" + label] = ms_x
                extra["linear_head_fused_step"]["paired_" + label] = paired_x
                del idx
            torch.cuda.empty_cache()
        t, prior, plans, n = dict(t, train=train_all, ref=ref_all), prior_all, dict(plans, net=plan_all), n_all
        del train_all, ref_all, prior_all, plan_all
        # the gradient-row kernel as bear_net.train runs it for an AR function made of torch ops: on the kept table (every row holds
        # counts, so no gradient row of zeros is written and no prior row of a context without counts is read) -- of all n rows
        keep = (t["train"] != 0).any(dim=1).nonzero().squeeze(1)
        tr_kept, pr_kept = t["train"].index_select(0, keep).contiguous(), prior.index_select(0, keep).contiguous()
        plan_kept = kernels.Plan(tr_kept, 5)
        extra["net_with_gradient_rows"]["kernel_ms_as_bear_net_train_holds_the_batch"] = timed(
            lambda: kernels.dm_prior_planned(plan_kept, pr_kept, h_s, want_grad=True, normalized=True), 5)
        del keep, tr_kept, pr_kept, plan_kept
        m = min(n, 20_000_000)
        test = kernels.synth_counts(SEED, row0, m, dev, want=("test",))["test"]
        tr_m, pr_m = t["train"][:m], prior[:m]
        eplan = kernels.EvalPlan(test, tr_m)
        ms_all = timed(lambda: kernels.evaluate_planned(eplan, pr_m, [1.0], [0.1, 1.0, 10.0]), 5)
        ms_u = timed(lambda: kernels.evaluate(test, pr_m, [1.0], [0.1, 1.0, 10.0], tr_m), 2)
        # as evaluation() / h_scan hold a batch since round 3: only the contexts with held-out counts (nothing else enters any
        # of the seven sums), their table rows carried as row_ids for the tie noise -- the same sums, accuracies exactly
        keep_t = (test != 0).any(dim=1).nonzero().squeeze(1)
        te_k, tr_k, pr_k = (x.index_select(0, keep_t).contiguous() for x in (test, tr_m, pr_m))
        ids_k = keep_t.to(torch.int32).contiguous()
        eplan_k = kernels.EvalPlan(te_k, tr_k)
        ms = timed(lambda: kernels.evaluate_planned(eplan_k, pr_k, [1.0], [0.1, 1.0, 10.0], row_ids=ids_k), 5)
        r_all = kernels.evaluate_planned(eplan, pr_m, [1.0], [0.1, 1.0, 10.0]).cpu().numpy()
        r_k = kernels.evaluate_planned(eplan_k, pr_k, [1.0], [0.1, 1.0, 10.0], row_ids=ids_k).cpu().numpy()
        same = bool(np.array_equal(r_all[5:], r_k[5:]) and np.allclose(r_all[:5], r_k[:5], rtol=1e-12))
        extra["heldout_evaluation"] = {"contexts": m, "models": "1 h + AR + 3 van_reg", "kernel": "eval_plan_kernel<1,4>", "kernel_ms": ms,
                                       "contexts_per_s": m / (ms * 1e-3), "achieved_GBps": m * 80 / (ms * 1e-3) / 1e9,
                                       "frac_of_hbm_peak": m * 80 / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                       "contexts_with_heldout_counts": keep_t.numel() / m,
                                       "roofline": fp64_roofline(flops_eval(test), ms),
                                       "compacted_equals_all_rows": same,
                                       "kernel_ms_all_rows_resident": ms_all,
                                       "plan_bytes_per_context": eplan_k.nbytes / m,
                                       "unplanned_kernel_ms": ms_u, "unplanned_contexts_per_s": m / (ms_u * 1e-3),
                                       "note": "kernel_ms: the batch as evaluation() keeps it -- the contexts with held-out counts only "
                                               "(row_ids carry their table rows); contexts_per_s counts all contexts of the batch; "
                                               "kernel_ms_all_rows_resident: the same kernel over every row (round 2's figure)"}
        del keep_t, te_k, tr_k, pr_k, ids_k, eplan_k
        # text -> first optimizer step (parse, pinned upload, compaction, plans): wall clock, bounded, 1 GPU only
        ingest_rows = int(min(n, 1e8) if args.ingest_rows is None else args.ingest_rows)
        if ingest_rows > 0:
            sys.path.insert(0, os.path.join(ROOT, "scripts"))
            try:
                import ingest_time
                torch.cuda.empty_cache()
                extra["ingest"] = ingest_time.measure(ingest_rows, dev)
            except Exception as err:    # e.g. no room for the text file on this box: reported, never fatal for the bench line
                extra["ingest"] = {"rows": ingest_rows, "error": f"{type(err).__name__}: {err}"}
        del test, eplan
        # every BASELINE config's own optimizer step at its stated size + the dense stress table (scripts/baseline_configs.py)
        if not args.no_baseline_configs:
            try:
                if os.path.join(ROOT, "scripts") not in sys.path:
                    sys.path.insert(0, os.path.join(ROOT, "scripts"))
                import baseline_configs
                torch.cuda.empty_cache()
                extra["baseline_configs"] = baseline_configs.measure_configs(dev)
                torch.cuda.empty_cache()
                extra["dense_table"] = baseline_configs.measure_dense(dev)
                import stream_time          # an epoch that does not stay on the card (the reference's cache=False pipeline)
                torch.cuda.empty_cache()
                extra["streamed_epochs"] = stream_time.measure(5_000_000, 3, 2, dev)
            except Exception as err:    # reported, never fatal for the bench line
                extra.setdefault("baseline_configs", {"error": f"{type(err).__name__}: {err}"})
                extra.setdefault("dense_table", {"error": f"{type(err).__name__}: {err}"})

    # (7) N > 1 under weak scaling: the strong split of ONE --contexts table too (north_star: the 1e8 table at 1/2/4/8 GPUs)
    strong = None
    if multi and args.scaling == "weak" and not args.no_strong_pass:
        strong = strong_pass(args, rank, world, dev, dev_index, primary, (h_s, tau_s, nu_s))

    value = total * args.steps / elapsed
    ms_per_step = elapsed / args.steps * 1e3
    bpc = BYTES_PER_CONTEXT[primary]
    achieved = n * bpc / (k_ms * 1e-3) / 1e9  # per-GPU algorithmic GB/s of the dominant kernel

    # what the collective actually ran on: the process group's own world size and backend (nccl = RCCL), how the ranks were started,
    # and the two facts a scaling figure rests on -- RCCL carried the all-reduce and every rank had a card of its own
    devices = [e["device"] for e in per_rank] if per_rank else [dev_index]
    ranks_info = {"world_size": dist.get_world_size() if multi else 1, "backend": dist.get_backend() if multi else None,
                  "launcher": os.environ.get("BEAR_BENCH_LAUNCHER", "external") if "WORLD_SIZE" in os.environ else "none (one process)",
                  "devices": devices, "backend_is_rccl": (dist.get_backend() == "nccl") if multi else None,
                  "devices_distinct": len(set(devices)) == len(devices)}
    if world > 1 and "BEAR_BENCH_DEVICE" not in os.environ and args.backend == "nccl":
        # (the override puts every rank on one card: single-GPU tests of this path; gloo: CPU-side smoke tests)
        if not (ranks_info["backend_is_rccl"] and ranks_info["devices_distinct"] and ranks_info["world_size"] == world):
            raise SystemExit(f"bench.py: --gpus {world} did not run as {world} RCCL ranks on {world} cards: {ranks_info}")
    if strong is not None:
        extra["strong_scaling"] = strong

    line = None
    if rank == 0 and extra:
        # vector-ALU utilisation of the instruction-bound kernels as the SQ counters of the committed profile saw it (not this run)
        try:
            prof = json.load(open(os.path.join(ROOT, "profiles", "traffic_latest.json")))
            src = "profiles/traffic_latest.json (%s): SQ_ACTIVE_INST_VALU / (8 SQ_BUSY_CYCLES), rocprofv3 --pmc pass of this command, not this run" % prof.get("tag", "?")
            for entry, key in (("linear_head_fused_step", "linear_head"), ("heldout_evaluation", "heldout_eval")):
                if "valu_busy_frac" in prof.get(key, {}) and "roofline" in extra.get(entry, {}):
                    extra[entry]["roofline"].update(valu_busy_frac_profiled=prof[key]["valu_busy_frac"], valu_source=src)
            for which, key in (("forward", "cnn_forward"), ("backward", "cnn_backward")):
                # per row order (scripts/dev/cnn_order_pmc.py, one order per profiled process, 2e7 contexts)
                for order, entry in (("sorted", "rates_%s_rows_in_kmer_order" % which), ("random", "roofline_%s_rows_in_random_order" % which)):
                    if "valu_busy_frac" in prof.get(key + "_" + order, {}):
                        extra["cnn_head"][entry].update(valu_busy_frac_profiled=prof[key + "_" + order]["valu_busy_frac"], valu_source=src)
        except Exception:
            pass
    if rank == 0:
        # HBM bytes per launch as the PMC counters saw them (FETCH_SIZE x 1024 x 2 + WRITE_SIZE x 1024, separate rocprofv3 --pmc
        # passes of this command, scripts/profile_round.sh): NOT measured in this run -- read from the committed profile
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                ent = tj.get(primary, {})
                traffic = ent.get("bytes_per_launch")
                if traffic is not None:  # measured at ent["contexts_per_launch"]; traffic is linear in the shard size
                    traffic = traffic * n / float(ent.get("contexts_per_launch", n))
                    traffic_source = "profiles/traffic_latest.json (%s): rocprofv3 --pmc passes of this command, not this run" % tj.get("tag", "?")
            except Exception:
                traffic = None
        moved = traffic if traffic is not None else n * (40 if primary == "net" else 0) + plan_nbytes[primary]
        # the other workload: PMC bytes of the committed profile when there are any, else an upper estimate (rows + whole plan)
        other_moved, other_moved_source = n * (40 if other == "net" else 0) + plan_nbytes[other], "estimate: row bytes + plan bytes"
        try:
            ent = json.load(open(tpath)).get(other, {})
            if ent.get("bytes_per_launch") is not None:
                other_moved = ent["bytes_per_launch"] * n / float(ent.get("contexts_per_launch", n))
                other_moved_source = "profiles/traffic_latest.json (rocprofv3 --pmc passes of this command, not this run)"
        except Exception:
            pass
        line = {
            "metric": "k-mer contexts/sec (DM-marginal+grad h), k=13",
            "value": value,
            "unit": "contexts/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": ("bear_net DM-marginal + d/dh over count rows + fp64 prior rows (mode N), k=13, "
                             if primary == "net" else
                             "bear_ref with the stop (flat) AR prior (BASELINE configs[1], mode R), k=13, ")
                            + (f"{n:.3g} synthetic contexts per GPU x {world} GPU" if args.scaling == "weak" else
                               f"ONE table of {total:.3g} synthetic contexts cut into {world} contiguous row shards") + ", planned kernels",
                "contexts_per_gpu": n,
                "contexts_total": total,
                "bytes_per_context": bpc,
                "parallelism": f"rows sharded over {world} GPU, one RCCL all-reduce of (ELBO, grads) per step" if world > 1 else "single GPU",
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS,
                "traffic": traffic,
                "traffic_source": traffic_source,
                # what the kernel actually moves (the plan replaces the 20 B count row by ~5 B): the HBM utilisation proper
                "moved_bytes_per_context": moved / n,
                "moved_bytes_frac": moved / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                "kernel": "dm_prior_plan_kernel" if primary == "net" else "dm_ref_items_kernel",
                "kernel_ms": k_ms,
                **k_stats,
                "measured_stream_read_GBps": stream_gbps,
            },
            # north_star names the gradient w.r.t. the AR-prior logits too: the kernel every parametrised prior trains through,
            # next to the d/dh-only one above (SURVEY 8d credits the 60 B read; the 40 B gradient row written per context is not credited)
            "roofline_gradient_rows": None if "net_with_gradient_rows" not in extra else (lambda g: {
                "kernel": "dm_prior_plan_grad_inplace_kernel", "bound": "hbm", "kernel_ms": g["kernel_ms_rows_asserted_normalized"],
                "credited_read_B": 60, "frac_credited": n * 60 / (g["kernel_ms_rows_asserted_normalized"] * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                "moved_B": g["moved_bytes_per_context"],
                "frac_moved": n * g["moved_bytes_per_context"] / (g["kernel_ms_rows_asserted_normalized"] * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                # bear_net.train keeps only the contexts that hold training counts resident (70 % of this table): the same kernel over
                # that table does the whole batch's work -- no gradient rows of zeros written, no prior rows of empty contexts read
                "kernel_ms_as_bear_net_train_holds_the_batch": g.get("kernel_ms_as_bear_net_train_holds_the_batch"),
                "frac_credited_as_bear_net_train_holds_the_batch": None if not g.get("kernel_ms_as_bear_net_train_holds_the_batch") else
                    n * 60 / (g["kernel_ms_as_bear_net_train_holds_the_batch"] * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                "general_kernel": "dm_prior_plan_grad_kernel", "general_kernel_ms": g["kernel_ms"],
                "general_frac_credited": n * 60 / (g["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                "peak": HBM_PEAK_GBPS, "unit": "GB/s"})(extra["net_with_gradient_rows"]),
            "settle": settle_info,
            # what the collective actually ran on: the process group's own world size and backend (nccl = RCCL), how the ranks were started
            "ranks": ranks_info,
            "per_rank": per_rank,
            "plan_build_s": plan_build_s,
            "plan_bytes_per_context": plan_nbytes[primary] / n,
            "result": result.tolist(),
            "also": {
                other: {
                    "contexts_per_s": total / (o_elapsed / max(5, args.steps // 5)),
                    "kernel_ms": o_k_ms,
                    **({"algorithmic_GBps": n * BYTES_PER_CONTEXT[other] / (o_k_ms * 1e-3) / 1e9,
                        "algorithmic_frac": n * BYTES_PER_CONTEXT[other] / (o_k_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS} if other == "net" else
                       # the reference-aware plan reads no count rows at all: SURVEY 8d's 40 B per context is not what bounds it
                       {"roofline": {**fp64_roofline(flops_ref_items(t["train"], t["ref"]), o_k_ms),
                                     **({"valu_busy_frac_profiled": json.load(open(tpath)).get("ref", {}).get("valu_busy_frac")}
                                        if os.path.exists(tpath) else {})},
                        "not_a_roofline_credited_GBps_at_40_B_per_context": n * 40 / (o_k_ms * 1e-3) / 1e9}),
                    "moved_bytes_per_context": other_moved / n,
                    "moved_GBps": other_moved / (o_k_ms * 1e-3) / 1e9,
                    "moved_bytes_source": other_moved_source,
                    "note": "mode R on the reference-aware plan (bear_plan_create_ref) streams only the item records of contexts WITH "
                            "reference counts (3.3 B per context; the others are a histogram): instruction-bound, so its roofline is the "
                            "fp64 rate on the useful flops of those records, and moved_GBps is the bus rate; SURVEY 8d's 40 B per "
                            "context applies to the streaming form (also.ref_streaming_reference_rows)",
                },
                "net_prior_normalized_asserted": None if norm_ms is None else {
                    "kernel_ms": norm_ms, "contexts_per_s_per_gpu": n / (norm_ms * 1e-3)},
                **extra,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"], line["elbo_rel_err_vs_cpu"], line["elbo_check"] = cpu_baseline(
                t, prior, primary, (h_s, tau_s, nu_s), result)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    if multi:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(t, prior, workload, params, result):
    """The oracle's C restatement (libm lgamma_r + series digamma, OpenMP over all host cores): timed on a bounded sample of the
    same table (the reported baseline), then run over the WHOLE table to check the output of the launch that was timed."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import c_oracle as co  # checker only: never on the measured or shipped path

    cores = os.cpu_count() or 1
    h_s, tau_s, nu_s = params
    n = t["train"].shape[0]
    tr = t["train"].cpu().numpy().view(np.uint32)
    other = prior.cpu().numpy() if workload == "net" else t["ref"].cpu().numpy().view(np.uint32)

    def run(m):
        t0 = time.perf_counter()
        if workload == "net":
            out, _ = co.dm_prior(tr[:m], other[:m], h_s, nthreads=cores)
        else:
            out = co.dm_ref(tr[:m], other[:m], h_s, tau_s, nu_s, nthreads=cores)
        return time.perf_counter() - t0, np.asarray(out)

    run(min(n, 200_000))  # thread start-up
    dt, _ = run(min(n, 2_000_000))
    rate = min(n, 2_000_000) / dt
    m = int(min(n, 30_000_000, max(2_000_000, rate * 12)))  # ~12 s of CPU work
    dt, _ = run(m)
    base = {
        "value": m / dt,
        "unit": "contexts/s",
        "cores": cores,
        "kind": "port",
        "sample": f"first {m} contexts of the same synthetic table, oracle/bear_oracle.c (libm lgamma_r + series digamma, "
                  f"OpenMP x{cores}); TensorFlow is not installable here, so this restatement stands in for the TF-CPU path",
    }
    # the timed launch's own result against the oracle on the full table (bounded: skipped above ~2 minutes of CPU)
    check = {"contexts": 0, "note": "skipped: the full table would take %.0f s on this host" % (n / (m / dt))}
    rel = None
    if n / (m / dt) <= 120.0:
        _, want = run(n)
        k = want.size
        rel = float(abs(result[0] - want[0]) / abs(want[0]))
        # every gradient against ITS OWN L1 mass (the sum of the absolute values of its terms, oracle/bear_oracle.c) and its own value
        mass = np.array([co.dm_prior_mass(tr, other, h_s, nthreads=cores)]) if workload == "net" else \
            co.dm_ref_mass(tr, other, h_s, tau_s, nu_s, nthreads=cores)
        err = np.abs(result[1:k] - want[1:])
        check = {"contexts": n, "elbo_rel_err": rel, "grad_abs_err_over_l1_mass": float((err / mass).max()),
                 "grad_rel_err": float((err / np.abs(want[1:])).max()),
                 "note": "output of the timed launch vs oracle/bear_oracle.c on the whole table; a gradient's L1 mass = the sum of the "
                         "absolute values of the per-row terms it is the sum of"}
    return base, rel, check


if __name__ == "__main__":
    main()
