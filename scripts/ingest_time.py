"""Text -> first optimizer step, wall clock (the reference's path: CsvDataset + JSON decode per batch, dataloader.py:35-46, then
`.cache()` in host RAM, bear_net.py:268-273): a synthetic k=13 count table of N rows (three dataset columns) is written in the
summarize.py row format, then timed:
    parse      bear_amd.dataloader.dataloader  (threaded mmap parser -> planar uint32 host arrays)
    resident   _train.ResidentBatches          (pinned staging + async H2D on a side stream, compaction to the contexts with
                                                training counts, plans cut as the batches land)
    first step one bear_ref reduce (planned mode-R kernel) + synchronize
Writing the file is preparation, not ingestion.  The file is in the page cache when it is read (it was just written).
    python scripts/ingest_time.py [rows] [batches]         # 1e8 rows = 5.3 GB of text; bench.py reports the same as also.ingest"""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def measure(n_rows, dev, batches=4, tmpdir=None, seed=20211012):
    from bear_amd import _train, dataloader, kernels
    t = kernels.synth_counts(seed, 0, n_rows, dev)
    lag = 13
    codes = torch.randint(0, 4, (n_rows, lag), dtype=torch.int64, device=dev, generator=torch.Generator(dev).manual_seed(seed))
    letters = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    kmers = letters[codes].cpu().numpy()
    del codes
    counts = np.stack([t[k].cpu().numpy().view(np.uint32) for k in ("train", "test", "ref")])
    del t
    torch.cuda.empty_cache()
    fd, path = tempfile.mkstemp(suffix=".tsv", prefix="bear_ingest_", dir=tmpdir)
    os.close(fd)
    out = {"rows": n_rows, "columns": 3, "lag": lag, "batches": batches}
    try:
        t0 = time.perf_counter()
        dataloader.write_counts_tsv(path, kmers, counts)
        out["write_s_not_counted"] = time.perf_counter() - t0
        out["text_bytes"] = os.path.getsize(path)
        del kmers
        batch = (n_rows + batches - 1) // batches
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        data = dataloader.dataloader(path, "dna", batch, 3)
        t1 = time.perf_counter()
        res = _train.ResidentBatches(data, {"train": 0, "ref": 2}, dev, drop_empty="train", prebuild=[("train", 4, "ref")])
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        theta = torch.tensor([0.0, np.log(1 / 30), -np.log(100)], dtype=torch.float64, device=dev)
        packed = torch.zeros(4, dtype=torch.float64, device=dev)
        e = res.batches[0]
        kernels.ref_train_reduce(res.plan(0, "train", 4, ref_column="ref"), e["ref"], theta, packed)
        first = packed.cpu().numpy()
        t3 = time.perf_counter()
        out["resident_rows"] = int(sum(b["rows"] for b in res.batches))
        out["upload_bytes"] = res.upload_bytes
        assert np.array_equal(data.counts, counts), "the parsed table differs from the one written"
        # the bare host -> HBM rate of the staging ring (one 2 GB column, nothing else running): what a caller that hands over
        # HOST buffers every step would pay per byte
        del res
        up = _train.Uploader(dev)
        torch.cuda.synchronize()
        t4 = time.perf_counter()
        col = up.put(data.counts[0].view(np.int32), torch.int32)
        up.wait()
        torch.cuda.synchronize()
        out["h2d_pinned_ring_GBps"] = col.numel() * 4 / (time.perf_counter() - t4) / 1e9
        del col, up
        out.update(parse_s=t1 - t0, resident_s=t2 - t1, first_step_s=t3 - t2, text_to_first_step_s=t3 - t0,
                   parse_GBps=out["text_bytes"] / (t1 - t0) / 1e9,
                   upload_and_plan_GBps=out["upload_bytes"] / (t2 - t1) / 1e9,
                   first_step_sum_ll=float(first[0]),
                   host_threads=os.cpu_count(),
                   note="parse: threaded mmap reader; resident: pinned staging ring + async H2D on a side stream, compaction to the "
                        "contexts with training counts, reference-aware plans cut per batch as it lands; the file was just written "
                        "(page cache)")
    finally:
        try:
            os.remove(path)
        except OSError:
            pass
    return out


if __name__ == "__main__":
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
    nb = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    import json
    print(json.dumps(measure(n, torch.device("cuda", 0), nb), indent=1))
