"""AR-function plugin surface: host mirror of ``bear_model/ar_funcs.py``.

``make_ar_func_<name>(lag, alphabet_size, **af_kwargs, dtype) -> (ar_func, params)`` exactly as the
reference (selected with ``getattr(ar_funcs, 'make_ar_func_' + name)``, models/train_bear_net.py:103).
``ar_func`` maps contexts to transition-probability rows ``[..., alphabet_size + 1]`` (the "prior rows"
of the DM kernels).  Contexts are accepted in the reference's one-hot form ``[..., lag, A+1]`` or, to
avoid materialising 5.2 GB of one-hot at 1e7 contexts, as integer codes ``[..., lag]``
(-1 = unknown letter = all-zero one-hot row).  Parameters are torch tensors with ``requires_grad``.  On
integer codes bear_net.train fuses the linear function into the DM step (``bear_dm_linear_f64``); called on its own
(evaluation, bear_ref.train, get_var_probs) it runs as ``bear_linear_forward_f64`` / ``bear_linear_backward_f64`` and the
convolutional one as ``bear_cnn_forward_f64`` / ``bear_cnn_backward_f64``, both behind torch autograd; one-hot input (and
shapes the kernels do not cover) takes the PyTorch-ROCm formulation, the DM kernels consume the rows either way.
"""
import numpy as np
import torch
import torch.nn.functional as F


def _is_codes(x):
    return not x.dtype.is_floating_point


class _FusedCnn(torch.autograd.Function):
    """The convolutional AR function on integer context codes as one HIP launch per direction
    (``bear_cnn_forward_f64`` / ``bear_cnn_backward_f64``, kernels_cnn.h) behind torch autograd, so that
    ``prior = ar_func(codes); prior.backward(grad_rows)`` in bear_net.train / bear_ref.train runs fused."""

    @staticmethod
    def forward(ctx, codes, lag, filter_width, *params):
        from . import kernels
        lead = codes.shape[:-1]
        packed = kernels.pack_kmers(codes.reshape(-1, lag).to(torch.int8).contiguous())
        flat = torch.cat([p.detach().reshape(-1) for p in params]).contiguous()
        need = any(p.requires_grad for p in params) and torch.is_grad_enabled()
        prior, t1 = kernels.cnn_forward(packed, flat, lag, filter_width, save=True)
        ctx.lag, ctx.fw, ctx.shapes = lag, filter_width, [p.shape for p in params]
        ctx.save_for_backward(packed, flat, t1, prior)
        return prior.reshape(lead + (5,))

    @staticmethod
    def backward(ctx, grad_rows):
        from . import kernels
        packed, flat, t1, prior = ctx.saved_tensors
        g = kernels.cnn_backward(packed, flat, ctx.lag, ctx.fw, t1, prior, grad_rows.reshape(-1, 5).to(torch.float64).contiguous())
        out, k = [], 0
        for shp in ctx.shapes:
            n = int(np.prod(shp))
            out.append(g[k:k + n].reshape(shp))
            k += n
        return (None, None, None) + tuple(out)


class _FusedLinear(torch.autograd.Function):
    """The linear AR function on integer context codes as one HIP launch per direction (``bear_linear_forward_f64`` /
    ``bear_linear_backward_f64``, kernels_linrows.h) behind torch autograd: evaluation and bear_ref.train get the rows without
    the [n, lag] int64 index matrix and the scatter-add of an embedding-bag (121 ms per 1e7 contexts forward + backward)."""

    @staticmethod
    def forward(ctx, codes, lag, mat):
        from . import kernels
        packed = kernels.pack_kmers(codes.reshape(-1, lag).to(torch.int8).contiguous())
        prior = kernels.linear_forward(packed, mat.detach().contiguous(), lag)
        ctx.lag = lag
        ctx.save_for_backward(packed, prior)
        return prior.reshape(codes.shape[:-1] + (5,))

    @staticmethod
    def backward(ctx, grad_rows):
        from . import kernels
        packed, prior = ctx.saved_tensors
        return None, None, kernels.linear_backward(packed, ctx.lag, prior, grad_rows.reshape(-1, 5).to(torch.float64).contiguous())


def wants_kmer_order(ar_func):
    """One forward pass (an evaluation) pays for sorting the batch by k-mer only where the forward kernel shares work between
    neighbouring contexts: the fused convolutional function (18 instead of 33 ms per 1e8 contexts).  The sums of an evaluation do
    not depend on the order (its tie-breaking noise is keyed by table row)."""
    return bool(getattr(ar_func, "fused", False) and getattr(ar_func, "cnn_params", None) is not None)


def _l2_normalize(x, dims):
    return x / torch.sqrt(torch.clamp((x * x).sum(dim=dims, keepdim=True), min=1e-12))


def _normalize_layer(layer, reduce_dims=(-1,)):
    """ar_funcs.py:5-20 (tf.nn.moments: biased variance)."""
    if tuple(reduce_dims) == (-1,) and layer.is_cuda:
        # the same formula as ONE kernel per direction (forward, backward) instead of six and ten: an AR function of torch ops on a
        # small table is bound by its launches (_train.run_autograd_steps)
        return F.layer_norm(layer, layer.shape[-1:], eps=1e-5)
    mean = layer.mean(dim=reduce_dims, keepdim=True)
    var = layer.var(dim=reduce_dims, unbiased=False, keepdim=True)
    return (layer - mean) / torch.sqrt(var + 1e-5)


def make_ar_func_linear(lag, alphabet_size, dtype=torch.float64, device=None, generator=None):
    """ar_funcs.py:23-46: softmax(einsum('...jk,jkl->...l', kmers, mat)); mat = 0.05 * l2-normalised N(0,1)."""
    mat = torch.randn(lag, alphabet_size + 1, alphabet_size + 1, dtype=dtype, device=device, generator=generator)
    mat = (0.05 * _l2_normalize(mat, (1,))).requires_grad_(True)

    from . import kernels
    fused_ok = dtype == torch.float64 and kernels.linear_supported(lag, alphabet_size)

    def ar_func(kmers):
        if fused_ok and _is_codes(kmers) and kmers.is_cuda and mat.is_cuda and kmers.shape[-1] == lag and kmers.numel():
            return _FusedLinear.apply(kmers, lag, mat)
        if _is_codes(kmers):
            # sum_l mat[l, a_l]: an embedding-bag over the flattened [lag * (A+1), A+1] table (unknown letters
            # carry weight 0) -- no [.., lag, A+1] intermediate, and a dense scatter-add backward
            idx = kmers.long()
            flat = idx.clamp(min=0) + (alphabet_size + 1) * torch.arange(lag, device=idx.device)
            z = F.embedding_bag(flat.reshape(-1, lag), mat.reshape(lag * (alphabet_size + 1), alphabet_size + 1), mode="sum",
                                per_sample_weights=(idx >= 0).to(mat.dtype).reshape(-1, lag))
            z = z.reshape(idx.shape[:-1] + (alphabet_size + 1,))
        else:
            z = torch.einsum("...jk,jkl->...l", kmers, mat)
        return torch.softmax(z, dim=-1)
    ar_func.linear_mat = mat      # bear_net.train: whole step fused in one kernel (bear_dm_linear_f64)
    ar_func.fused = fused_ok      # integer codes on the device take bear_linear_forward / backward_f64; one-hot input the torch ops
    ar_func.normalized_rows = True   # softmax output: the DM kernels may take the shared concentration total (prior_normalized)
    return ar_func, [mat]


def make_ar_func_cnn(lag, alphabet_size, filter_width=8, num_filters=30, kmer_layer1_width=16,
                     dtype=torch.float64, device=None, generator=None):
    """ar_funcs.py:49-99.  Returns params in the reference's order (ar_funcs.py:98-99):
    filters, intercept0, weights1, intercept1, weights2, intercept2, scale0, scale1."""
    filter_width, num_filters, kmer_layer1_width = int(filter_width), int(num_filters), int(kmer_layer1_width)
    small_start = 0.05
    A1 = alphabet_size + 1
    P = lag - filter_width + 1
    kw = dict(dtype=dtype, device=device)
    filters = _l2_normalize(torch.randn(filter_width, A1, num_filters, generator=generator, **kw), (0, 1)).requires_grad_(True)
    kmer_intercept0 = torch.ones(P, num_filters, **kw).requires_grad_(True)
    kmer_scale0 = torch.ones(P, num_filters, **kw).requires_grad_(True)
    kmer_weights1 = _l2_normalize(torch.randn(P, num_filters, kmer_layer1_width, generator=generator, **kw), (0,)).requires_grad_(True)
    kmer_intercept1 = torch.ones(kmer_layer1_width, **kw).requires_grad_(True)
    kmer_scale1 = torch.ones(kmer_layer1_width, **kw).requires_grad_(True)
    kmer_weights2 = (small_start * _l2_normalize(torch.randn(kmer_layer1_width, A1, generator=generator, **kw), (0,))).requires_grad_(True)
    kmer_intercept2 = torch.zeros(A1, **kw).requires_grad_(True)

    windows = {}        # the index form of the last code tensor seen: a training loop hands over the same resident batch every step

    def conv(data):
        if _is_codes(data):
            # conv1d VALID over a one-hot input = for every output position the sum of filter_width rows of the
            # flattened [filter_width * (A+1), nf] filter table: one embedding-bag (dense scatter-add backward)
            key = (data.data_ptr(), tuple(data.shape), data.dtype, data._version, str(data.device))
            if windows.get("key") != key:
                idx = data.long()
                win = idx.unfold(-1, filter_width, 1)                                  # [..., P, fw] windows
                flat = (win.clamp(min=0) + A1 * torch.arange(filter_width, device=idx.device)).reshape(-1, filter_width)
                windows.update(key=key, keep=data, flat=flat, weights=(win >= 0).to(filters.dtype).reshape(-1, filter_width))
            out = F.embedding_bag(windows["flat"], filters.reshape(filter_width * A1, num_filters), mode="sum",
                                  per_sample_weights=windows["weights"])
            return out.reshape(data.shape[:-1] + (P, num_filters))                # [..., P, nf]
        x = data.reshape((-1,) + data.shape[-2:]).transpose(1, 2)          # [B, A1, lag]
        y = F.conv1d(x, filters.permute(2, 1, 0))                          # [B, nf, P]
        return y.transpose(1, 2).reshape(data.shape[:-2] + (P, num_filters))

    params = [filters, kmer_intercept0, kmer_weights1, kmer_intercept1, kmer_weights2, kmer_intercept2,
              kmer_scale0, kmer_scale1]
    from . import kernels
    fused_ok = dtype == torch.float64 and kernels.cnn_supported(lag, alphabet_size, filter_width, num_filters, kmer_layer1_width)

    def ar_func(data):
        if fused_ok and _is_codes(data) and data.is_cuda and data.shape[-1] == lag:
            if torch.is_grad_enabled() and any(p.requires_grad for p in params):
                return _FusedCnn.apply(data, lag, filter_width, *params)
            flat = torch.cat([p.detach().reshape(-1) for p in params]).contiguous()
            packed = kernels.pack_kmers(data.reshape(-1, lag).to(torch.int8).contiguous())
            return kernels.cnn_forward(packed, flat, lag, filter_width, save=False)[0].reshape(data.shape[:-1] + (A1,))
        nn0 = kmer_scale0 * _normalize_layer(conv(data)) + kmer_intercept0
        t1 = torch.tensordot(F.elu(nn0), kmer_weights1, dims=([-2, -1], [0, 1]))
        nn1 = kmer_scale1 * _normalize_layer(t1) + kmer_intercept1
        nn2 = torch.tensordot(F.elu(nn1), kmer_weights2, dims=([-1], [0])) + kmer_intercept2
        return torch.softmax(nn2, dim=-1)
    ar_func.fused = fused_ok       # integer codes on the device take the fused kernels; one-hot input the torch ops
    # the un-fused path caches the index form of the last code tensor (~16 P fw bytes per row): the loops release it when they
    # are done (release_ar_func_cache).  The key is (data_ptr, shape, _version): a code buffer REFILLED in place by a bear kernel
    # does not bump _version -- code tensors handed to this function must not be rewritten in place while it is in use.
    ar_func.clear_cache = windows.clear
    ar_func.cnn_params, ar_func.cnn_filter_width = params, filter_width
    ar_func.normalized_rows = True
    return ar_func, params


def release_ar_func_cache(ar_func):
    """Drops what an AR function cached for the batches of a finished loop (make_ar_func_cnn's window indices pin the memory
    pool of a captured graph otherwise)."""
    clear = getattr(ar_func, "clear_cache", None)
    if clear is not None:
        clear()


def make_ar_func_stop(lag, alphabet_size, dtype=torch.float64, device=None, generator=None):
    """ar_funcs.py:102-127: always predicts a stop; no parameters (for the reference AR model)."""
    stop = torch.zeros(alphabet_size + 1, dtype=dtype, device=device)
    stop[-1] = 1

    def ar_func(y):
        return stop
    ar_func.is_stop = True  # lets bear_ref pick the fused reference-prior kernels
    ar_func.normalized_rows = True
    return ar_func, []
