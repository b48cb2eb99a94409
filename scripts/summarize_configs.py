"""gpurun_out/<tag>_cfgK (scripts/profile_configs.sh: rocprofv3 --kernel-trace of each BASELINE config's optimizer loop) ->
profiles/<tag>_configs_summary.md: per config what ONE optimizer step is made of (kernels, launches per step, mean duration of a
launch in the steady half of the loop, the gap between consecutive launches) next to the loop's own HIP-event figure."""
import collections, csv, glob, json, os, re, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
md = [f"# BASELINE configs, one optimizer step each: rocprofv3 kernel trace, round tag {tag}", "",
      "`scripts/profile_configs.sh`: `rocprofv3 --kernel-trace --stats -- python3 scripts/baseline_configs.py configsK`, one config per",
      "process.  `us/step (events)` is the loop's own HIP-event figure (`_train.LAST_RUN`, what bench.py reports as",
      "`also.baseline_configs`); the table is the trace of the SAME run: launches of the second half of the loop only (steady clocks),",
      "`gap` = idle time on the device between the end of one launch of the step and the start of the next one.", ""]
out_json = {}
for k in (1, 2, 3, 4):
    logs = glob.glob(os.path.join(src, f"{tag}_cfg{k}.log"))
    kt = glob.glob(os.path.join(src, f"{tag}_cfg{k}", "**", "*_kernel_trace.csv"), recursive=True)
    if not logs or not kt:
        continue
    txt = open(logs[0]).read()
    m = re.search(r"\{.*\}", txt, re.S)
    ent = json.loads(m.group(0)) if m else {}
    # (a process that trains several times -- configs[4]: the record's shard, then a rank's piece cut both ways -- : the trace's
    # last `optimizer_steps` step markers belong to the LAST train() call, so that entry is the one reported)
    name, cfg = ([(n, v) for n, v in ent.items() if isinstance(v, dict)] or [("configs[%d]" % k, {})])[-1]
    rows = sorted(csv.DictReader(open(kt[0])), key=lambda r: int(r["Start_Timestamp"]))
    steps = int(cfg.get("optimizer_steps", 0))
    # the optimizer loop = the last stretch of launches whose kernels repeat; take the launches of the second half of the trace's
    # training kernels (names that occur >= steps / 2 times)
    counts = collections.Counter(r["Kernel_Name"] for r in rows)
    loop_names = {n for n, c in counts.items() if steps and c >= steps // 2}
    loop = [r for r in rows if r["Kernel_Name"] in loop_names]
    # the step's marker: the kernel that is launched exactly once per step (the Adam update, or the one fused step kernel); the
    # steady half of the loop = from the marker's launch number steps / 2 to its last one
    marker = next((n for n in ("adam_vec_kernel", "dm_ref_items_kernel", "dm_linear_plan_kernel") for m in loop_names if n in m), None)
    marks = [i for i, r in enumerate(loop) if marker and marker in r["Kernel_Name"]]
    marks = marks[-steps:] if steps and len(marks) >= steps else marks
    if len(marks) >= 4:
        loop = loop[marks[len(marks) // 2] + 1:marks[-1] + 1]
        n_steps_seen = len(marks) - 1 - len(marks) // 2
    else:
        loop, n_steps_seen = loop[len(loop) // 2:], 0
    per = collections.defaultdict(list)
    gaps = []
    for a, b in zip(loop[:-1], loop[1:]):
        gaps.append(int(b["Start_Timestamp"]) - int(a["End_Timestamp"]))
    for r in loop:
        per[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    n_steps_seen = n_steps_seen or (max(1, min(len(v) for v in per.values())) if per else 1)
    span = (int(loop[-1]["End_Timestamp"]) - int(loop[0]["Start_Timestamp"])) / 1e3 if loop else 0.0
    md += [f"## {name}", "",
           f"us/step (events): **{cfg.get('us_per_step', float('nan')):.1f}**; captured in a HIP graph: {cfg.get('captured_in_hip_graph')}; "
           f"trace: {len(loop)} launches over {span:.0f} us in the second half of the loop = "
           f"**{span / max(1, len(loop)) * (len(loop) / n_steps_seen):.1f} us per step**, mean gap between launches {sum(gaps) / max(1, len(gaps)) / 1e3:.2f} us", "",
           "| kernel | launches per step | mean us per launch | us per step |", "|---|---|---|---|"]
    tot = 0.0
    for n, d in sorted(per.items(), key=lambda kv: -sum(kv[1])):
        lps = len(d) / n_steps_seen
        us = sum(d) / len(d) / 1e3
        tot += us * lps
        md.append(f"| `{n[:90]}` | {lps:.2f} | {us:.2f} | {us * lps:.2f} |")
    md += [f"| (sum of the kernels) | | | {tot:.2f} |", ""]
    out_json[name] = {"us_per_step_events": cfg.get("us_per_step"), "us_per_step_trace": span / max(1, n_steps_seen),
                      "kernel_us_per_step": tot, "mean_gap_us": sum(gaps) / max(1, len(gaps)) / 1e3}
os.makedirs(dst, exist_ok=True)
open(os.path.join(dst, f"{tag}_configs_summary.md"), "w").write("\n".join(md) + "\n")
json.dump(out_json, open(os.path.join(dst, f"{tag}_configs_summary.json"), "w"), indent=1)
print("\n".join(md))
