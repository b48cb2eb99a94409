"""Worker of tests/test_dist_gpu.py: one rank of a 2-process run of the product training entry points on row shards.
Launched by ``python -m torch.distributed.run``; both ranks share cuda:0 (BEAR_AMD_DEVICE=0) and reduce over gloo
(BEAR_AMD_DIST_BACKEND=gloo) -- on a multi-GPU node the same code runs on RCCL with one GPU per rank."""
import json
import os
import sys

ROOT = os.environ["BEAR_ROOT"]
sys.path.insert(0, ROOT)

import numpy as np
import torch

from bear_amd import ar_funcs, bear_net, bear_ref, dataloader, dist

YSD1 = os.path.join(ROOT, "tests", "golden", "ysd1_lag_5_file_0_preshuf.tsv")
CNN_CFG = {"num_filters": 30, "filter_width": 3, "kmer_layer1_width": 16}


def main():
    rank, world = dist.init_from_env()
    assert world == 2 and torch.cuda.current_device() == 0
    restart = np.load(os.environ["BEAR_RESTART"], allow_pickle=True)
    out = {}
    data = dataloader.dataloader(YSD1, "dna", 500, 3, shard="auto")          # 3 batches, the last one short (365 rows)
    assert data.shard == (rank, 2) and data.local_rows < data.num_rows == 1365
    for train_ar in (False, True):
        key = "ar" if train_ar else "bear"
        ls = []
        p, _, _ = bear_ref.train(data.repeat(2), 1365, 2, 0, 2, "dna", 5, ar_funcs.make_ar_func_stop, {}, 0.01, "Adam", train_ar, loss_save=ls)
        out["ref_stop_" + key] = {"loss": ls, "params": [x.detach().cpu().numpy().tolist() for x in p]}
        ls = []
        p, _, _ = bear_net.train(data.repeat(2), 1365, 2, 0, "dna", 5, ar_funcs.make_ar_func_linear, {}, 0.01, "Adam", train_ar,
                                 params_restart=list(restart["linear"]), loss_save=ls)
        out["net_linear_" + key] = {"loss": ls, "params": [x.detach().cpu().numpy().tolist() for x in p]}
    ls = []
    p, _, _ = bear_net.train(data.repeat(2), 1365, 2, 0, "dna", 5, ar_funcs.make_ar_func_cnn, CNN_CFG, 0.01, "Adam", False,
                             params_restart=list(restart["cnn"]), loss_save=ls)
    out["net_cnn_bear"] = {"loss": ls, "params": [x.detach().cpu().numpy().tolist() for x in p]}
    # gradient accumulation + a torch-op AR function (bear_ref with a parametrised net function) + another optimizer
    ls = []
    p, _, _ = bear_ref.train(data.repeat(2), 1365, 2, 0, 2, "dna", 5, ar_funcs.make_ar_func_linear, {}, 0.01, "Adam", False, acc_steps=3,
                             params_restart=list(restart["ref_linear"]), loss_save=ls)
    out["ref_linear_acc3"] = {"loss": ls, "params": [x.detach().cpu().numpy().tolist() for x in p]}
    ls = []
    p, _, _ = bear_ref.train(data.repeat(2), 1365, 2, 0, 2, "dna", 5, ar_funcs.make_ar_func_stop, {}, 0.001, "SGD", False, acc_steps=2, loss_save=ls)
    out["ref_stop_sgd_acc2"] = {"loss": ls, "params": [x.detach().cpu().numpy().tolist() for x in p]}
    # mirrored variables: the ranks draw different initial values, rank 0's are broadcast
    torch.manual_seed(100 + rank)
    p, _, _ = bear_net.train(data.repeat(1), 1365, 1, 0, "dna", 5, ar_funcs.make_ar_func_linear, {}, 0.01, "Adam", False)
    mine = torch.cat([x.detach().reshape(-1).cpu() for x in p])
    both = [None, None]
    torch.distributed.all_gather_object(both, mine.numpy().tolist())
    out["mirrored"] = bool(np.array_equal(both[0], both[1]))
    # held-out evaluation on shards: one all-reduce of the partial sums at the end
    torch.manual_seed(1)
    f, _ = ar_funcs.make_ar_func_linear(5, 4, device="cuda")
    r = bear_net.evaluation(data, 0, 1, "dna", torch.tensor(0.37), f, np.array([0.1, 1.0, 10.0]), seed=11)
    out["eval"] = [np.asarray(v).tolist() for v in r]
    if rank == 0:
        with open(os.environ["BEAR_OUT"], "w") as fh:
            json.dump(out, fh)
    dist.shutdown()


if __name__ == "__main__":
    main()
