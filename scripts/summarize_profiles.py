"""Condenses gpurun_out/<tag>_{stats,fetch,write,sq} (rocprofv3 csv) into profiles/<tag>_*.{csv,md,json}."""
import collections, csv, glob, json, os, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
out_md = [f"# rocprofv3 summary, round tag {tag}", "", "Command: `python3 bench.py --steps 100 --warmup 60 --no-cpu-baseline` under",
          "`rocprofv3 --kernel-trace --stats` (kernel stats) and, in separate passes, `--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`,",
          "`--pmc SQ_*` (scripts/profile_round.sh).", ""]
ks = glob.glob(os.path.join(src, f"{tag}_stats", "**", "*_kernel_stats.csv"), recursive=True)
if ks:
    rows = list(csv.DictReader(open(ks[0])))
    with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w") as fh:
        w = csv.DictWriter(fh, fieldnames=rows[0].keys()); w.writeheader(); w.writerows(rows)
    out_md += ["## kernel stats (--kernel-trace --stats)", "", "| kernel | calls | avg ns | total % |", "|---|---|---|---|"]
    for r in rows:
        out_md.append(f"| `{r['Name'][:70]}` | {r['Calls']} | {float(r['AverageNs']):.0f} | {r['Percentage']} |")
    out_md.append("")
    # --stats averages ALL launches of a kernel, including the first ~30 after idle while the card ramps its clocks up
    # (scripts/dev/ramp.py: 1.1 -> 0.79 ms for the headline kernel); the trace has every launch: steady state = the last half
    kt = glob.glob(os.path.join(src, f"{tag}_stats", "**", "*_kernel_trace.csv"), recursive=True)
    if kt:
        per = collections.defaultdict(list)
        for r in csv.DictReader(open(kt[0])):
            per[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        out_md += ["## the same trace, first launches apart (clock ramp after idle)", "",
                   "| kernel | calls | avg ns, first quarter of the launches | avg ns, last half |", "|---|---|---|---|"]
        for name, d in sorted(per.items(), key=lambda kv: -sum(kv[1])):
            if len(d) >= 8 and any(t in name for t in ("dm_", "eval_", "cnn_", "stream_read", "linear_rows", "ref_mix")):
                q, h = d[:len(d) // 4], d[len(d) // 2:]
                out_md.append(f"| `{name[:70]}` | {len(d)} | {sum(q) / len(q):.0f} | {sum(h) / len(h):.0f} |")
        out_md.append("")
traffic = {}
def kernel_key(k):
    """bench.py workload name of a planned kernel instantiation."""
    if "eval_plan_kernel" in k: return "heldout_eval"
    if "eval_sorted_kernel" in k: return "heldout_eval_unplanned"
    if "dm_ref_items" in k: return "ref_ar" if "<true>" in k else "ref"
    if "dm_ref_plan" in k: return "ref_stream_ar" if "<true>" in k else "ref_stream"
    if "cnn_forward" in k: return "cnn_forward"
    if "cnn_backward" in k: return "cnn_backward"
    if "dm_linear" in k: return "linear_head"
    if "dm_refmix_plan" in k: return "ref_mix_dm_step"
    if "plan_grad_inplace" in k: return "net_grad_inplace"
    if "plan_grad_kernel" in k: return "net_grad"
    if "linear_rows_forward" in k: return "linear_rows_forward"
    if "linear_rows_backward" in k: return "linear_rows_backward"
    if "ref_mix_forward" in k: return "ref_mix_forward"
    if "ref_mix_backward" in k: return "ref_mix_backward"
    if "dm_ref" in k: return "ref_ar" if "<true>" in k else "ref"
    if "<true, true>" in k: return "net_ar"
    if "<true, false>" in k: return "net_norm"
    if "<false, false>" in k: return "net"
    return k[:24]
def pmc(kind):
    fs = glob.glob(os.path.join(src, f"{tag}_{kind}", "**", "*_counter_collection.csv"), recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in fs:
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg
for kind in ("fetch", "write", "sq"):
    agg = pmc(kind)
    if not agg: continue
    out_md += [f"## PMC pass: {kind}", "", "| kernel | counter | mean per launch |", "|---|---|---|"]
    for k, v in agg.items():
        if not any(x in k for x in ("plan_kernel", "plan_grad_kernel", "plan_grad_inplace", "sorted", "eval_", "cnn_", "items_kernel", "linear_rows", "ref_mix", "refmix")): continue
        for c, x in v.items():
            out_md.append(f"| `{k[:40]}` | {c} | {sum(x)/len(x):.0f} |")
            name = kernel_key(k)
            if c == "FETCH_SIZE":
                # FETCH_SIZE is in KiB and, on gfx950, counts a wide coalesced stream at half its bytes
                # (MI355X_MICROARCH.md, HBM): bytes = FETCH_SIZE * 1024 * 2
                traffic.setdefault(name, {})["fetch_bytes_per_launch"] = sum(x) / len(x) * 1024 * 2
            if c == "WRITE_SIZE":
                traffic.setdefault(name, {})["write_bytes_per_launch"] = sum(x) / len(x) * 1024
    if kind == "sq":
        # SQ_BUSY_CYCLES sums the 32 shader engines, SQ_ACTIVE_INST_VALU counts quad-cycles over all 1024 SIMDs:
        # fraction of the kernel's cycles a SIMD spends issuing vector-ALU instructions = ACTIVE_VALU * 4 / (BUSY / 32 * 1024)
        for k, v in agg.items():
            if not any(x in k for x in ("plan_kernel", "plan_grad_kernel", "plan_grad_inplace", "eval_", "cnn_", "items_kernel", "linear_rows", "ref_mix", "refmix")): continue
            mean = {c: sum(x) / len(x) for c, x in v.items()}
            if mean.get("SQ_BUSY_CYCLES") and "SQ_ACTIVE_INST_VALU" in mean:
                d = traffic.setdefault(kernel_key(k), {})
                d["valu_busy_frac"] = mean["SQ_ACTIVE_INST_VALU"] / (8.0 * mean["SQ_BUSY_CYCLES"])
                d["valu_wave_instructions_per_launch"] = mean.get("SQ_INSTS_VALU")
    out_md.append("")
# the convolutional kernels by row order (scripts/dev/cnn_order_pmc.py: one order per process, 2e7 contexts)
for order in ("random", "sorted", "levels"):
    agg = pmc(f"sqcnn_{order}")
    if not agg: continue
    out_md += [f"## PMC pass: convolutional kernels, rows in {order} order (2e7 contexts, scripts/dev/cnn_order_pmc.py)", "",
               "| kernel | launches | VALU busy | VALU wave-instructions per launch |", "|---|---|---|---|"]
    for k, v in agg.items():
        if "cnn_" not in k or "finalize" in k: continue
        mean = {c: sum(x) / len(x) for c, x in v.items()}
        if not mean.get("SQ_BUSY_CYCLES"): continue
        busy = mean["SQ_ACTIVE_INST_VALU"] / (8.0 * mean["SQ_BUSY_CYCLES"])
        which = "cnn_forward" if "cnn_forward" in k else ("cnn_level_sum" if "level_sum" in k else "cnn_backward")
        out_md.append(f"| `{k[:40]}` | {len(v['SQ_BUSY_CYCLES'])} | {busy:.3f} | {mean.get('SQ_INSTS_VALU', 0):.0f} |")
        traffic.setdefault(f"{which}_{order}", {}).update(valu_busy_frac=busy, valu_wave_instructions_per_launch=mean.get("SQ_INSTS_VALU"),
                                                        contexts_per_launch=20000000, launches_averaged=len(v["SQ_BUSY_CYCLES"]))
    out_md.append("")
for name, d in list(traffic.items()):
    if "fetch_bytes_per_launch" in d or "write_bytes_per_launch" in d:
        d["bytes_per_launch"] = d.get("fetch_bytes_per_launch", 0) + d.get("write_bytes_per_launch", 0)
    # bench.py defaults (--contexts 1e8; the evaluation extra runs on the first 2e7); bench.py scales linearly for other sizes
    d.setdefault("contexts_per_launch", 20000000 if name.startswith("heldout_eval") else 100000000)
if traffic:
    traffic["tag"] = tag
    json.dump(traffic, open(os.path.join(dst, f"{tag}_traffic.json"), "w"), indent=1)
    json.dump(traffic, open(os.path.join(dst, "traffic_latest.json"), "w"), indent=1)
    out_md += ["## HBM traffic per launch (corrected as MI355X_MICROARCH.md prescribes)", "", "```", json.dumps(traffic, indent=1), "```", ""]
log = os.path.join(src, f"{tag}_stats.log")
if os.path.exists(log):
    last = [l for l in open(log).read().splitlines() if l.startswith("{")]
    if last:
        out_md += ["## bench line of the profiled run (profiled runs clock lower than un-profiled ones)", "", "```", last[-1], "```", ""]
open(os.path.join(dst, f"{tag}_summary.md"), "w").write("\n".join(out_md))
print("\n".join(out_md))
