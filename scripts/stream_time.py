"""A streamed epoch against a resident one (ResidentBatches.streaming: the reference's tf.data pipeline with cache=False): the same
bear_net.train call over `batches` batches of `rows` synthetic k=13 contexts, linear AR function, once with every batch resident and
once with BEAR_AMD_STREAM=1 (every batch re-uploaded, compacted, sorted and planned every epoch, the next one crossing PCIe while a
step runs).  Prints ms per optimizer step and the host->device rate of the streamed run.
    python scripts/stream_time.py [rows per batch] [batches] [epochs]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from baseline_configs import _table


def measure(rows=10_000_000, batches=4, epochs=3, dev=None):
    """also.streamed_epochs of the bench line / this script's output."""
    from bear_amd import _train, ar_funcs, bear_net, dataloader
    dev = dev or torch.device("cuda", 0)
    n = rows * batches
    kmers, counts = _table(n, 13, dev, ("train",))
    data = dataloader.CountDataset(kmers, counts, "dna", rows)
    out = {"rows_per_batch": rows, "batches_per_epoch": batches, "epochs": epochs, "host_bytes_per_context": 20 + 13,
           "note": "bear_net.train, linear AR function, the same call twice: every batch resident (HIP-graph replay) and "
                   "BEAR_AMD_STREAM=1 (ResidentBatches.streaming: every batch re-uploaded through the pinned ring, compacted, sorted, "
                   "planned and paired every epoch, the next one crossing PCIe under the step; what happens by itself when an "
                   "epoch does not fit the card)"}
    keep = os.environ.get("BEAR_AMD_STREAM")
    try:
        for mode in ("resident", "streamed"):
            if mode == "streamed":
                os.environ["BEAR_AMD_STREAM"] = "1"
            else:
                os.environ.pop("BEAR_AMD_STREAM", None)
            torch.manual_seed(1)
            loss = []
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            bear_net.train(data.repeat(epochs), n, epochs, 0, "dna", 13, ar_funcs.make_ar_func_linear, {}, 0.01, "Adam", False, loss_save=loss)
            torch.cuda.synchronize()
            wall = time.perf_counter() - t0
            lr = dict(_train.LAST_RUN)
            out[mode] = {"train_call_wall_s": wall, "loop_ms_per_step": lr["loop_ms"] / max(lr["loop_steps"], 1), "graph": lr["graph"],
                         "elbo_last": loss[-1]}
            if mode == "streamed":
                per_step = lr["loop_ms"] / max(lr["loop_steps"], 1) * 1e-3
                out[mode]["host_to_device_GBps_over_the_loop"] = rows * 33 / per_step / 1e9
                out[mode]["contexts_per_s"] = rows / per_step
    finally:
        if keep is None:
            os.environ.pop("BEAR_AMD_STREAM", None)
        else:
            os.environ["BEAR_AMD_STREAM"] = keep
    out["elbo_rel_diff"] = abs(out["streamed"]["elbo_last"] - out["resident"]["elbo_last"]) / abs(out["resident"]["elbo_last"])
    return out


def main():
    rows = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
    batches = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    epochs = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    print(json.dumps(measure(rows, batches, epochs), indent=1))


if __name__ == "__main__":
    main()
