"""Worker of tests/test_dist_gpu.py::test_many_ranks_*: one rank of an N-process run (any N) of the product entry points on row
shards.  Launched by ``python -m torch.distributed.run``; all ranks share cuda:0 (BEAR_AMD_DEVICE=0) and reduce over gloo
(BEAR_AMD_DIST_BACKEND=gloo) -- on a multi-GPU node the same code runs on RCCL with one GPU per rank.  BEAR_TABLES is a JSON
list of [name, path, batch_size]: small batch sizes leave some ranks' pieces of a batch EMPTY and the others uneven."""
import json
import os
import sys

ROOT = os.environ["BEAR_ROOT"]
sys.path.insert(0, ROOT)

import numpy as np
import torch

from bear_amd import _train, ar_funcs, bear_net, bear_ref, dataloader, dist


def main():
    rank, world = dist.init_from_env()
    assert world == int(os.environ["BEAR_EXPECT_WORLD"]) and torch.cuda.current_device() == 0
    restart = np.load(os.environ["BEAR_RESTART"], allow_pickle=True)
    out = {}
    if os.environ.get("BEAR_TEST_STREAM_RANK", "") == str(rank):
        os.environ["BEAR_AMD_STREAM"] = "1"          # ONE rank is told to stream its epochs: the ranks have to agree on it
    for name, path, batch in json.loads(os.environ["BEAR_TABLES"]):
        data = dataloader.dataloader(path, "dna", batch, 3, shard=os.environ.get("BEAR_TEST_SHARD", "auto"))    # "kmer": dealt by k-mer range
        pieces = [g1 - g0 for g0, g1, _ in data.rank_pieces(rank, world)]
        every = [None] * world
        torch.distributed.all_gather_object(every, pieces)
        res = {"pieces": every}
        ls = []
        p, _, _ = bear_ref.train(data.repeat(4), data.num_rows, 4, 0, 2, "dna", 5, ar_funcs.make_ar_func_stop, {}, 0.01, "Adam", False, loss_save=ls)
        res["ref_stop"] = {"loss": ls, "params": [x.detach().cpu().numpy().tolist() for x in p]}
        ls = []
        p, _, _ = bear_net.train(data.repeat(4), data.num_rows, 4, 0, "dna", 5, ar_funcs.make_ar_func_linear, {}, 0.01, "Adam", False,
                                 params_restart=list(restart["linear"]), loss_save=ls)
        res["net_linear"] = {"loss": ls, "params": [x.detach().cpu().numpy().tolist() for x in p]}
        streamed = [None] * world
        torch.distributed.all_gather_object(streamed, bool(_train.LAST_RUN.get("streaming")))
        res["streamed"] = streamed
        ls = []
        p, _, _ = bear_net.train(data.repeat(4), data.num_rows, 4, 0, "dna", 5, ar_funcs.make_ar_func_linear, {}, 0.01, "Adam", True, acc_steps=2,
                                 params_restart=list(restart["linear"]), loss_save=ls)
        res["net_linear_ar_acc2"] = {"loss": ls, "params": [x.detach().cpu().numpy().tolist() for x in p]}
        torch.manual_seed(1)
        f, _ = ar_funcs.make_ar_func_linear(5, 4, device="cuda")
        r = bear_net.evaluation(data, 0, 1, "dna", torch.tensor(0.37), f, np.array([0.1, 1.0, 10.0]), seed=11)
        res["eval"] = [np.asarray(v).tolist() for v in r]
        fr, _ = bear_ref._make_ref_ar_func(5, 4, ar_funcs.make_ar_func_stop, {}, device="cuda")
        r = bear_ref.evaluation(data, 0, 1, 2, "dna", torch.tensor(0.21), fr, np.array([0.5, 2.0]), seed=3)
        res["eval_ref"] = [np.asarray(v).tolist() for v in r]
        out[name] = res
    if rank == 0:
        with open(os.environ["BEAR_OUT"], "w") as fh:
            json.dump(out, fh)
    dist.shutdown()


if __name__ == "__main__":
    main()
