"""BASELINE configs[1], [2], [4] through the HOST DRIVERS (bear_ref.train / bear_net.train / evaluation) on a synthetic k=13
table held in memory: seconds per optimizer step and per held-out evaluation, including every host-side cost."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bear_amd import ar_funcs, bear_net, bear_ref, dataloader, kernels

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
lag = 13
dev = torch.device("cuda", 0)
t = kernels.synth_counts(20211012, 0, n, dev)
counts = np.stack([t[k].cpu().numpy().view(np.uint32) for k in ("train", "test", "ref")])
kmers = np.frombuffer(b"ACGT", dtype=np.uint8)[torch.randint(0, 4, (n, lag), device=dev).cpu().numpy()]
del t
data = dataloader.CountDataset(kmers, counts, "dna", n).shuffle(1)       # one batch per epoch, rows shuffled on the device
print(f"table: {n:.2e} contexts, lag {lag}, 3 groups; host arrays {counts.nbytes / 1e9:.2f} + {kmers.nbytes / 1e9:.2f} GB", flush=True)

def run(name, fn):
    torch.cuda.synchronize(); t0 = time.time(); out = fn(); torch.cuda.synchronize()
    return out, time.time() - t0

for name, make, kw, mod in (("bear_ref / stop (configs[1])", ar_funcs.make_ar_func_stop, {}, bear_ref),
                            ("bear_net / linear (configs[2])", ar_funcs.make_ar_func_linear, {}, bear_net),
                            ("bear_net / cnn (configs[4])", ar_funcs.make_ar_func_cnn, {"filter_width": 8}, bear_net)):
    extra = (2,) if mod is bear_ref else ()
    loss = []
    train = lambda k, ls=None: mod.train(data.repeat(k), n, k, 0, *extra, "dna", lag, make, kw, 0.01, "Adam", False, loss_save=ls)
    run(name, lambda: train(2))                                    # warm-up: lazy initialisation, allocator
    # per-step time = (time of 2 + K steps - time of 2 steps) / K with K grown until the K steps themselves take >= 0.5 s:
    # set-up (upload, sort, plans) is ~0.2 s with +-10 ms of noise, a step of configs[1] is 0.15 ms -- at K = 5 the
    # difference was noise (round 4 printed a negative step time from it)
    t1 = min(run(name, lambda: train(2))[1] for _ in range(3))
    k = max(steps, 50)
    while True:
        loss = []
        (params, h_signed, ar_func), tk = run(name, lambda: train(2 + k, loss))
        if tk - t1 >= 0.5 or k >= 20000:
            break
        k *= 4
    per_step = (tk - t1) / k
    h = torch.exp(h_signed).detach()
    res, te = run(name, lambda: mod.evaluation(data, 0, 1, *extra, "dna", h, ar_func, np.array([0.1, 1.0, 10.0])))
    rate = f"{per_step * 1e3:.3f} ms per further step ({k} steps) = {n / per_step / 1e9:.2f} Gctx/s" if per_step > 0 else \
        f"per-step time below the resolution of this measurement ({k} steps in {tk - t1:+.3f} s)"
    print(f"{name}: setup+2 steps {t1:.2f} s; {rate}; "
          f"evaluation {te:.2f} s; ELBO {loss[0]:.6e} -> {loss[-1]:.6e}; held-out perplexity BEAR {float(res[3]):.4f}", flush=True)
