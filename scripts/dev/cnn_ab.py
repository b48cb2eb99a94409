"""Dev helper: cnn backward, both tile forms (BEAR_CNN_BACKWARD=1: 64-context tiles, one wave per SIMD), results compared."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if len(sys.argv) > 1:
    import torch
    from bear_amd import kernels, ar_funcs
    n, lag, fw = int(float(sys.argv[1])), 13, 8
    dev = torch.device("cuda", 0)
    gen = torch.Generator(dev).manual_seed(3)
    t = kernels.synth_counts(20211012, 0, n, dev, want=("train",))["train"]
    codes = torch.randint(0, 4, (n, lag), dtype=torch.int8, device=dev, generator=gen)
    if os.environ.get("CNN_AB_START"): codes[torch.rand(n, lag, device=dev, generator=gen) < 0.01] = 4
    packed = kernels.pack_kmers(codes)
    torch.manual_seed(5)
    _, params = ar_funcs.make_ar_func_cnn(lag, 4, filter_width=fw, device=dev)
    flat = torch.cat([q.detach().reshape(-1) for q in params]).contiguous()
    prior, t1 = kernels.cnn_forward(packed, flat, lag, fw)
    _, g = kernels.dm_prior_planned(kernels.Plan(t, 5), prior, 0.0, want_grad=True)
    out = kernels.cnn_backward(packed, flat, lag, fw, t1, prior, g); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): out = kernels.cnn_backward(packed, flat, lag, fw, t1, prior, g)
    e1.record(); torch.cuda.synchronize()
    print("form %s: %.2f ms per %.0e contexts" % (os.environ.get("BEAR_CNN_BACKWARD", "2"), e0.elapsed_time(e1) / 3, n))
    torch.save(out.cpu(), sys.argv[2])
else:
    import torch
    for form in ("1", "2"):
        subprocess.run([sys.executable, __file__, "1e7", "/tmp/cnn_ab_%s.pt" % form], env=dict(os.environ, BEAR_CNN_BACKWARD=form), check=True)
    a = torch.load("/tmp/cnn_ab_1.pt")
    for form in ("2",):
        b = torch.load("/tmp/cnn_ab_%s.pt" % form)
        print("form %s vs 1: max |diff| / max |grad| = %.3e" % (form, float((a - b).abs().max() / a.abs().max())))
