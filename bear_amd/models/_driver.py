"""Config-driven training / evaluation driver shared by train_bear_net.py and train_bear_ref.py
(host mirror of bear_model/models/train_bear_{net,ref}.py:31-204; same .cfg keys and semantics)."""
import datetime
import json
import os
import pickle

import numpy as np
import torch

from bear_amd import ar_funcs, bear_net, bear_ref, core, dataloader, dist

PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _count_lines(path):
    return dataloader.count_newlines(path)      # `wc -l` on all host threads (a Python block loop took 5x the parse)


def _writer(out_folder):
    try:
        from torch.utils.tensorboard import SummaryWriter
        return SummaryWriter(out_folder)
    except Exception:
        return None


def main(config, kind):
    """kind: 'net' or 'ref'.  Returns what the reference's main() returns (1, or (1, ll_van, perp_van)
    when train_test is on, models/train_bear_net.py:197-200).

    Multi-GPU (replaces the MirroredStrategy of bear_net.py:246 / bear_ref.py:310): launched as one process per GPU --
    ``python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 train_bear_ref.py cfg`` -- every
    process binds to its GPU and joins the RCCL group here, before any GPU work; each rank loads only its rows of every batch,
    rank 0 alone writes the output files.  A process group this call created is destroyed again when it returns or raises (the
    exception propagates: the process exits non-zero and the launcher tears the job down)."""
    owned = not (torch.distributed.is_available() and torch.distributed.is_initialized())
    try:
        return _run(config, kind)
    finally:
        if owned:
            dist.shutdown(barrier=False)


def _run(config, kind):
    rank, world = dist.init_from_env()
    mod = bear_net if kind == "net" else bear_ref
    time_stamp = datetime.datetime.now().strftime("%Y%m%d-%H%M%S")
    of = config["general"]["out_folder"]
    if of == "TEST":
        out_folder = os.path.join(PKG, "models", "out_data", "logs", time_stamp)
    elif of[-1] == "*":
        out_folder = of[:-1]
    else:
        out_folder = os.path.join(of, "logs", time_stamp)
    if world > 1:       # one folder for the job: rank 0's time stamp
        names = [out_folder]
        torch.distributed.broadcast_object_list(names, src=0)
        out_folder = names[0]
    if rank == 0:
        os.makedirs(out_folder, exist_ok=True)
    dist.barrier()
    torch.manual_seed(int(config["general"]["seed"]))
    dtype = getattr(torch, config["general"]["precision"])
    writer = _writer(out_folder) if rank == 0 else None

    if config["data"]["files_path"] == "TEST":
        files = [os.path.join(PKG, "data", "ysd1_lag_5_file_0_preshuf.tsv")]
    else:
        fp = config["data"]["files_path"]
        files = sorted(os.path.join(fp, f) for f in os.listdir(fp) if f.startswith(config["data"]["start_token"]))
    sparse = config["data"]["sparse"] == "True"
    num_kmers = sum(_count_lines(f) for f in files)          # `wc -l`, models/train_bear_net.py:52-55 (a sparse file's header included)
    kmer_batch_size = float(config["train"]["batch_size"])
    kmer_batch_size = int(num_kmers * kmer_batch_size) if kmer_batch_size <= 1 else int(kmer_batch_size)
    epochs = config["train"]["epochs"]
    epochs = int(epochs[:-1]) // (1 + num_kmers // kmer_batch_size) + 1 if epochs[-1] == "s" else int(epochs)
    num_ds = int(config["data"]["num_ds"])
    load = dataloader.sparse_dataloader if sparse else dataloader.dataloader
    # two optional keys beyond the reference's: [data] binary_cache (True or a directory: parsed tables are kept on disk,
    # dense format only) and [data] shuffle_seed (rows are permuted on the device at upload instead of `shuf`-ing the file)
    extra_kw = {}
    if config["data"].get("binary_cache") and load is dataloader.dataloader:
        bc = config["data"]["binary_cache"]
        extra_kw["binary_cache"] = True if bc == "True" else bc
    shuffle_seed = config["data"].get("shuffle_seed")
    # a third one: [data] shard = kmer deals every batch to the ranks BY K-MER RANGE instead of by contiguous row pieces (or
    # BEAR_AMD_SHARD=kmer): every rank parses the whole table and keeps its range.  The sums of a step do not depend on which rows a
    # rank holds; a rank's piece then keeps the table's density of distinct prefixes / windows, which is what the fused linear and
    # convolutional steps live on (a contiguous piece of a pre-shuffled table is a random 1 / world of its k-mers).
    deal_kmer = (config["data"].get("shard") or os.environ.get("BEAR_AMD_SHARD", "")) == "kmer"
    if world > 1 and deal_kmer and load is dataloader.dataloader and not shuffle_seed:
        parts = [load(f, config["data"]["alphabet"], kmer_batch_size, num_ds, cache=config["train"]["cache"] == "True", dtype=dtype,
                      **extra_kw) for f in files]
        whole = parts[0] if len(parts) == 1 else dataloader.concatenate(parts)
        parts = [whole.deal_by_kmer(rank, world)]
        del whole
    elif world > 1 and load is dataloader.dataloader and not shuffle_seed:
        # every rank decodes and holds only its pieces of the batches (the device shuffle needs whole columns: then, as for the
        # small sparse format, each rank loads the table and slices its rows at upload)
        if extra_kw.get("binary_cache"):
            # a sharded load reads a binary cache but cannot write one (no rank holds the whole table): rank 0 builds the missing
            # ones with a plain load first, the others wait, then every rank reads its row ranges
            if rank == 0:
                for f in files:
                    if dataloader._cache_meta(dataloader.cache_path_for(f, extra_kw["binary_cache"]), f, num_ds) is None:
                        load(f, config["data"]["alphabet"], kmer_batch_size, num_ds, dtype=dtype, **extra_kw)
            dist.barrier()
        file_rows = [dataloader.count_rows(f) for f in files]
        total, base, parts = sum(file_rows), 0, []
        for f, n in zip(files, file_rows):
            parts.append(load(f, config["data"]["alphabet"], kmer_batch_size, num_ds, cache=config["train"]["cache"] == "True", dtype=dtype,
                              shard=(rank, world), row_base=base, total_rows=total, **extra_kw))
            base += n
    else:
        parts = [load(f, config["data"]["alphabet"], kmer_batch_size, num_ds, cache=config["train"]["cache"] == "True", dtype=dtype,
                      **extra_kw) for f in files]
    data = parts[0] if len(parts) == 1 and isinstance(parts[0], dataloader.CountDataset) else dataloader.concatenate(parts)
    if shuffle_seed:
        data = data.shuffle(int(shuffle_seed))
    data_train = data.repeat(epochs)

    result_file = os.path.join(out_folder, "results.pickle")
    config["results"]["out_folder"] = out_folder
    config["results"]["file"] = result_file

    def save_config():
        if rank == 0:
            with open(os.path.join(out_folder, "config.cfg"), "w") as cw:
                config.write(cw)
    save_config()

    ds_loc = int(config["data"]["train_column"])
    ds_loc_ref = int(config["data"]["reference_column"])
    alphabet = config["data"]["alphabet"]
    alphabet_size = len(core.alphabets_tf[alphabet]) - 1
    lag = int(config["hyperp"]["lag"])
    make_ar_func = getattr(ar_funcs, "make_ar_func_" + config["model"]["ar_func_name"])
    af_kwargs = json.loads(config["model"]["af_kwargs"])
    learning_rate = float(config["train"]["learning_rate"])
    optimizer_name = config["train"]["optimizer_name"]
    train_ar = config["train"]["train_ar"] == "True"
    acc_steps = int(config["train"]["accumulation_steps"])

    params_restart = None
    if config["train"]["restart"] == "True":
        with open(os.path.join(config["train"]["restart_path"], "results.pickle"), "rb") as fr:
            params_restart = pickle.load(fr)["params"]

    extra = (ds_loc_ref,) if kind == "ref" else ()
    if config["train"]["train"] == "True":
        loss_save = []
        params, h_signed, ar_func = mod.train(
            data_train, num_kmers, epochs, ds_loc, *extra, alphabet, lag, make_ar_func, af_kwargs,
            learning_rate, optimizer_name, train_ar=train_ar, acc_steps=acc_steps,
            params_restart=params_restart, writer=writer, loss_save=loss_save, dtype=dtype)
        if rank == 0:
            try:
                import matplotlib
                matplotlib.use("Agg")
                from matplotlib import pyplot as plt
                plt.figure(figsize=[10, 10])
                plt.xlabel("steps", fontsize=30)
                plt.ylabel("loss", fontsize=30)
                plt.plot(loss_save)
                plt.tight_layout()
                plt.savefig(os.path.join(out_folder, "loss.png"), dpi=100)
                plt.close()
            except Exception:
                pass
    else:
        assert config["train"]["restart"] == "True"
        params, h_signed, ar_func = mod.change_scope_params(lag, alphabet_size, make_ar_func, af_kwargs, params_restart, dtype=dtype,
                                                            device=torch.device("cuda", torch.cuda.current_device()))

    config["results"]["h"] = str(torch.exp(h_signed).item())
    if kind == "ref":
        tau = torch.exp(params[1]).item()
        config["results"]["error_rate"] = str(1 - np.exp(-tau))
        nw = torch.exp(params[2]).item()
        config["results"]["stop_rate"] = str(1 / (nw / (1 + nw)))
    save_config()
    if rank == 0:
        with open(result_file, "wb") as rw:
            pickle.dump({"params": [p.detach().cpu().numpy() for p in params]}, rw)

    h = torch.exp(h_signed).detach()
    ret = 1
    if config["test"]["test"] == "True":
        ds_loc_test = int(config["data"]["test_column"])
        van_reg = np.array(json.loads(config["test"]["van_reg"]))
        r = mod.evaluation(data, ds_loc, ds_loc_test, *extra, alphabet, h, ar_func, van_reg)
        for name, v in zip(["heldout_loglikelihood_BEAR", "heldout_loglikelihood_AR", "heldout_loglikelihood_BMM",
                            "heldout_perplex_BEAR", "heldout_perplex_AR", "heldout_perplex_BMM",
                            "heldout_accuracy_BEAR", "heldout_accuracy_AR", "heldout_accuracy_BMM"], r):
            config["results"][name] = json.dumps(np.asarray(v).tolist()) if np.ndim(v) else str(float(v))
        save_config()
    if config["test"]["train_test"] == "True":
        van_reg = np.array(json.loads(config["test"]["van_reg"]))
        r = mod.evaluation(data, -1, ds_loc, *extra, alphabet, h, ar_func, van_reg)
        for name, v in zip(["loglikelihood_BEAR", "loglikelihood_AR", "loglikelihood_BMM",
                            "perplex_BEAR", "perplex_AR", "perplex_BMM",
                            "accuracy_BEAR", "accuracy_AR", "accuracy_BMM"], r):
            config["results"][name] = json.dumps(np.asarray(v).tolist()) if np.ndim(v) else str(float(v))
        save_config()
        ret = (1, np.asarray(r[2]), np.asarray(r[5]))
    dist.barrier()
    return ret
