"""A whole config-driven run (models/_driver.py = bear_model/models/train_bear_{net,ref}.py) on a synthetic k=13 table, wall
clock per stage: count the rows, parse, train (resident upload + plans + optimizer steps), held-out evaluation, train-set
evaluation.  The stages are timed by wrapping the driver's own calls; nothing is skipped.
    python scripts/driver_time.py [kind=net|ref] [ar_func=linear|cnn|stop] [rows=1e7] [steps=50]"""
import configparser
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "net"
    name = sys.argv[2] if len(sys.argv) > 2 else "linear"
    n_rows = int(float(sys.argv[3])) if len(sys.argv) > 3 else 10_000_000
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 50
    from bear_amd import bear_net, bear_ref, dataloader, kernels
    from bear_amd.models import _driver
    dev = torch.device("cuda", 0)
    lag, seed = 13, 20211012
    t = kernels.synth_counts(seed, 0, n_rows, dev)
    codes = torch.randint(0, 4, (n_rows, lag), dtype=torch.int64, device=dev, generator=torch.Generator(dev).manual_seed(seed))
    kmers = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)[codes].cpu().numpy()
    counts = np.stack([t[k].cpu().numpy().view(np.uint32) for k in ("train", "test", "ref")])
    del codes, t
    torch.cuda.empty_cache()
    tmp = tempfile.mkdtemp(prefix="bear_driver_")
    path = os.path.join(tmp, "synth_lag_13_file_0.tsv")
    dataloader.write_counts_tsv(path, kmers, counts)
    del kmers, counts
    cfg = configparser.ConfigParser()
    cfg.read(os.path.join(os.path.dirname(_driver.__file__), "config_files", "bear_lin_bear.cfg"))
    cfg["general"]["out_folder"] = os.path.join(tmp, "out") + "*"
    cfg["data"].update(files_path=tmp, start_token="synth")
    cfg["hyperp"]["lag"] = str(lag)
    cfg["train"].update(epochs=f"{steps}s", batch_size="1.0")
    cfg["model"].update(ar_func_name=name, af_kwargs=json.dumps({"filter_width": 8} if name == "cnn" else {}))
    stages = {}

    def wrap(obj, attr, label):
        fn = getattr(obj, attr)

        def timed(*a, **k):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r = fn(*a, **k)
            torch.cuda.synchronize()
            stages.setdefault(label, []).append(time.perf_counter() - t0)
            return r
        setattr(obj, attr, timed)
    mod = bear_net if kind == "net" else bear_ref
    wrap(_driver, "_count_lines", "count rows")
    wrap(dataloader, "dataloader", "parse")
    wrap(mod, "train", f"train ({steps} steps incl. upload, plans)")
    wrap(mod, "evaluation", "evaluation")
    t0 = time.perf_counter()
    _driver.main(cfg, kind)
    total = time.perf_counter() - t0
    print(f"{kind} + {name}, {n_rows:.0e} rows ({os.path.getsize(path) / 1e9:.2f} GB of text), {steps} steps: total {total:.2f} s")
    for k, v in stages.items():
        print(f"  {k:45s} " + "  ".join(f"{x:.3f} s" for x in v))
    print("  results:", {k: cfg["results"][k] for k in ("h", "heldout_perplex_BEAR", "heldout_perplex_AR", "perplex_BEAR") if k in cfg["results"]})
    os.remove(path)


if __name__ == "__main__":
    main()
