"""ctypes binding of libbear_hip.so (C ABI: include/bear_hip.h).

Fails loudly: a missing library raises ImportError with the build command, and every
non-zero status from the library raises BearError.  No compute path exists outside the
library.
"""
import ctypes
import os

# torch ships its own libamdhip64 (SONAME libamdhip64.so.7); importing it first makes the
# dynamic loader bind libbear_hip.so to that same runtime instance, so device pointers and
# streams are shared with torch.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
def _deterministic_requested():
    v = os.environ.get("BEAR_AMD_DETERMINISTIC", "")
    return bool(v) and v != "0"


# BEAR_AMD_DETERMINISTIC set when the package is first used: the deterministic build of the same sources (libbear_hip_det.so:
# every sum of a launch bit-identical from run to run, include/bear_hip.h).  Set later, the regular library still switches its
# parameter gradients over per call; BEAR_AMD_LIB: developer A/B builds.
LIB_PATH = os.environ.get("BEAR_AMD_LIB") or os.path.join(_HERE, "libbear_hip_det.so" if _deterministic_requested() else "libbear_hip.so")

ABI_VERSION = 6   # BEAR_ABI_VERSION of include/bear_hip.h the argtypes below were written against

SYMBOLS = [
    "bear_abi_version", "bear_strerror", "bear_last_hip_error", "bear_ws_create", "bear_ws_destroy",
    "bear_dm_prior_f64", "bear_dm_ref_f64", "bear_dm_items_f64", "bear_eval_f64", "bear_bmm_f64", "bear_pack_kmers_u64", "bear_linear_index_u64", "bear_parse_sparse_counts", "bear_plan_tile_count", "bear_plan_tile_info", "bear_dm_linear_f64",
    "bear_plan_create", "bear_plan_create_ref", "bear_plan_destroy", "bear_plan_bytes", "bear_dm_prior_plan_f64", "bear_dm_prior_plan_grad_f64", "bear_dm_ref_plan_f64", "bear_synth_counts_u32", "bear_synth_prior_f64",
    "bear_count_rows", "bear_count_newlines", "bear_parse_counts_tsv", "bear_log_gamma_f64", "bear_logdir_sample_f64",
    "bear_stat_source", "bear_cache_write", "bear_cache_info", "bear_cache_read", "bear_shuffle_rows", "bear_shuffle_source_row",
    "bear_stream_read", "bear_encode_kmers_i8", "bear_ref_train_step_f64", "bear_net_linear_train_step_f64", "bear_cnn_reserve", "bear_net_cnn_train_step_f64", "bear_cnn_param_count", "bear_cnn_forward_f64", "bear_cnn_backward_f64", "bear_linear_forward_f64", "bear_linear_backward_f64", "bear_ref_mix_forward_f64", "bear_ref_mix_backward_f64", "bear_dm_refmix_plan_grad_f64",
    "bear_dm_prior_plan_dev_f64", "bear_train_apply_f64", "bear_ref_train_reduce_f64", "bear_net_linear_train_reduce_f64", "bear_net_cnn_train_reduce_f64",
    "bear_eval_plan_create", "bear_eval_plan_destroy", "bear_eval_plan_bytes", "bear_eval_plan_f64",
    "bear_shard_rows_count", "bear_parse_counts_tsv_shard",
    "bear_kmer_sort_create", "bear_kmer_sort_reduce", "bear_kmer_sort_destroy", "bear_count_last_hip_error", "bear_write_counts_tsv", "bear_fastx_size", "bear_fastx_encode",
    "bear_kmer_order_u64", "bear_gather_rows", "bear_plan_pair_contexts", "bear_plan_pair_info", "bear_plan_attach_cnn_levels", "bear_plan_cnn_level_rows", "bear_cnn_forward_plan_f64",
    "bear_plan_count_total", "bear_plan_set_count_bound", "bear_deterministic_build", "bear_plan_cnn_window_rows",
    "bear_plan_create_auto",
]


ERR_NOMEM = -5   # BEAR_ERR_NOMEM (include/bear_hip.h)


class BearError(RuntimeError):
    def __init__(self, status, where):
        self.status = status
        msg = _lib.bear_strerror(status).decode() if _lib is not None else "?"
        hip = _lib.bear_last_hip_error() if (_lib is not None and status == -4) else 0
        super().__init__(f"{where}: {msg} (status {status}" + (f", hipError {hip})" if hip else ")"))


_lib = None


def _load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make -C {os.path.join(_HERE, 'csrc')}` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). bear_amd has no CPU fallback.")
    L = ctypes.CDLL(LIB_PATH)
    vp, u64, dbl, cint = ctypes.c_void_p, ctypes.c_uint64, ctypes.c_double, ctypes.c_int
    L.bear_abi_version.restype = cint
    if L.bear_abi_version() != ABI_VERSION:
        # a library from another tree may export every symbol and still take its arguments in another order
        raise ImportError(f"{LIB_PATH} speaks ABI version {L.bear_abi_version()}, this binding was written for {ABI_VERSION}: "
                          f"rebuild it with `make -C {os.path.join(_HERE, 'csrc')}`")
    L.bear_strerror.restype = ctypes.c_char_p
    L.bear_strerror.argtypes = [cint]
    L.bear_last_hip_error.restype = cint
    L.bear_ws_create.argtypes = [cint, ctypes.POINTER(vp)]
    L.bear_ws_destroy.argtypes = [vp]
    L.bear_dm_prior_f64.argtypes = [vp, vp, vp, u64, dbl, dbl, cint, vp, vp, vp]
    L.bear_dm_ref_f64.argtypes = [vp, vp, vp, u64, dbl, dbl, dbl, dbl, cint, vp, vp]
    L.bear_plan_create.argtypes = [vp, vp, u64, cint, ctypes.POINTER(vp)]
    L.bear_plan_create_ref.argtypes = [vp, vp, vp, u64, ctypes.POINTER(vp)]
    L.bear_plan_destroy.argtypes = [vp]
    L.bear_plan_bytes.argtypes = [vp]
    L.bear_plan_bytes.restype = u64
    L.bear_dm_prior_plan_f64.argtypes = [vp, vp, vp, vp, u64, dbl, dbl, cint, cint, vp, vp]
    L.bear_dm_prior_plan_grad_f64.argtypes = [vp, vp, vp, vp, u64, dbl, dbl, cint, cint, vp, vp, vp]
    L.bear_plan_create_auto.argtypes = [vp, vp, u64, ctypes.POINTER(cint), ctypes.POINTER(vp)]
    L.bear_eval_f64.argtypes = [vp, vp, vp, vp, u64, vp, cint, cint, vp, cint, dbl, u64, u64, vp, vp]
    L.bear_pack_kmers_u64.argtypes = [vp, u64, cint, vp, vp]
    L.bear_linear_index_u64.argtypes = [vp, u64, cint, vp, vp]
    L.bear_plan_tile_count.argtypes = [vp]
    L.bear_plan_tile_count.restype = u64
    L.bear_plan_tile_info.argtypes = [vp, u64, u64, vp, vp, vp, vp]
    L.bear_parse_sparse_counts.argtypes = [ctypes.c_char_p, cint, cint, cint, u64, u64, vp, vp, ctypes.POINTER(u64)]
    L.bear_dm_linear_f64.argtypes = [vp, vp, vp, vp, vp, cint, u64, dbl, dbl, cint, vp, vp, vp]
    L.bear_bmm_f64.argtypes = [vp, vp, u64, vp, cint, vp, vp]
    L.bear_dm_ref_plan_f64.argtypes = [vp, vp, vp, vp, u64, dbl, dbl, dbl, dbl, cint, vp, vp]
    L.bear_dm_items_f64.argtypes = [vp, vp, vp, u64, cint, vp, vp, vp]
    L.bear_synth_counts_u32.argtypes = [u64, u64, u64, cint, vp, vp, vp, vp]
    L.bear_synth_prior_f64.argtypes = [u64, u64, u64, vp, vp]
    L.bear_log_gamma_f64.argtypes = [vp, u64, u64, u64, vp, vp]
    L.bear_logdir_sample_f64.argtypes = [vp, vp, u64, vp, cint, cint, vp, cint, cint, cint, u64, u64, vp, vp]
    i64 = ctypes.c_int64
    L.bear_stat_source.argtypes = [ctypes.c_char_p, ctypes.POINTER(u64), ctypes.POINTER(i64)]
    L.bear_cache_write.argtypes = [ctypes.c_char_p, vp, vp, u64, cint, cint, u64, i64]
    L.bear_cache_info.argtypes = [ctypes.c_char_p, ctypes.POINTER(u64), ctypes.POINTER(cint), ctypes.POINTER(cint),
                                  ctypes.POINTER(u64), ctypes.POINTER(i64)]
    L.bear_cache_read.argtypes = [ctypes.c_char_p, u64, u64, vp, vp]
    L.bear_shuffle_rows.argtypes = [vp, vp, u64, ctypes.c_uint32, u64, vp]
    L.bear_shuffle_source_row.argtypes = [u64, u64, u64]
    L.bear_shuffle_source_row.restype = u64
    L.bear_cnn_param_count.argtypes = [cint, cint, cint, cint]
    L.bear_cnn_forward_f64.argtypes = [vp, vp, u64, cint, cint, cint, cint, vp, vp, vp, vp]
    L.bear_cnn_backward_f64.argtypes = [vp, vp, u64, cint, cint, cint, cint, vp, vp, vp, vp, vp, vp]
    L.bear_linear_forward_f64.argtypes = [vp, vp, u64, cint, vp, vp, vp]
    L.bear_linear_backward_f64.argtypes = [vp, vp, u64, cint, vp, vp, vp, vp]
    L.bear_ref_mix_forward_f64.argtypes = [vp, vp, vp, u64, vp, vp, vp, vp]
    L.bear_ref_mix_backward_f64.argtypes = [vp, vp, vp, vp, u64, vp, vp, vp, vp, vp]
    L.bear_dm_refmix_plan_grad_f64.argtypes = [vp, vp, vp, vp, vp, u64, vp, vp, vp, dbl, cint, vp, vp, vp]
    L.bear_kmer_sort_create.argtypes = [vp, vp, u64, cint, ctypes.POINTER(vp), ctypes.POINTER(u64), vp]
    L.bear_kmer_sort_reduce.argtypes = [vp, cint, vp, vp, vp, vp]
    L.bear_kmer_sort_destroy.argtypes = [vp]
    L.bear_write_counts_tsv.argtypes = [ctypes.c_char_p, vp, vp, u64, cint, cint, u64, u64, cint]
    L.bear_stream_read.argtypes = [vp, vp, u64, vp]
    L.bear_encode_kmers_i8.argtypes = [vp, u64, cint, cint, vp, vp]
    L.bear_fastx_size.argtypes = [ctypes.c_char_p, cint, cint, ctypes.POINTER(u64), ctypes.POINTER(u64)]
    L.bear_fastx_encode.argtypes = [ctypes.c_char_p, cint, cint, cint, u64, vp, vp, ctypes.POINTER(u64)]
    L.bear_ref_train_step_f64.argtypes = [vp, vp, vp, vp, u64, vp, vp, vp, vp, dbl, cint, dbl, dbl, vp, vp, u64, vp]
    L.bear_net_linear_train_step_f64.argtypes = [vp, vp, vp, vp, cint, u64, vp, vp, vp, vp, vp, dbl, cint, dbl, dbl, vp, u64, vp]
    L.bear_cnn_reserve.argtypes = [vp, u64, cint, cint, cint, cint]
    L.bear_net_cnn_train_step_f64.argtypes = [vp, vp, vp, vp, u64, cint, cint, cint, cint, vp, vp, vp, vp, vp, vp, vp, vp, dbl, cint, dbl,
                                              dbl, vp, u64, vp]
    L.bear_dm_prior_plan_dev_f64.argtypes = [vp, vp, vp, vp, u64, vp, dbl, cint, cint, vp, vp, vp]
    L.bear_train_apply_f64.argtypes = [vp, cint, vp, vp, vp, vp, dbl, dbl, cint, vp, u64, vp]
    L.bear_ref_train_reduce_f64.argtypes = [vp, vp, vp, vp, u64, vp, dbl, cint, vp, vp]
    L.bear_net_linear_train_reduce_f64.argtypes = [vp, vp, vp, vp, cint, u64, vp, dbl, cint, vp, vp]
    L.bear_net_cnn_train_reduce_f64.argtypes = [vp, vp, vp, vp, u64, cint, cint, cint, cint, vp, vp, vp, vp, dbl, cint, vp, vp]
    L.bear_shard_rows_count.argtypes = [u64, u64, u64, u64, cint, cint, ctypes.POINTER(u64)]
    L.bear_parse_counts_tsv_shard.argtypes = [ctypes.c_char_p, cint, cint, u64, u64, u64, u64, cint, cint, u64, vp, vp, ctypes.POINTER(u64),
                                              ctypes.POINTER(u64)]
    L.bear_eval_plan_create.argtypes = [vp, vp, vp, u64, ctypes.POINTER(vp), vp]
    L.bear_eval_plan_destroy.argtypes = [vp]
    L.bear_eval_plan_bytes.argtypes = [vp]
    L.bear_eval_plan_bytes.restype = u64
    L.bear_plan_cnn_level_rows.argtypes = [vp, vp, vp, cint]
    L.bear_plan_cnn_window_rows.argtypes = [vp, vp, vp, vp, cint]
    L.bear_cnn_forward_plan_f64.argtypes = [vp, vp, vp, u64, cint, cint, cint, cint, vp, vp, vp, vp]
    L.bear_plan_attach_cnn_levels.argtypes = [vp, vp, cint, cint, ctypes.POINTER(cint), vp]
    L.bear_plan_pair_info.argtypes = [vp, ctypes.POINTER(u64), ctypes.POINTER(u64)]
    L.bear_plan_count_total.argtypes = [vp, ctypes.POINTER(dbl), ctypes.POINTER(dbl)]
    L.bear_plan_set_count_bound.argtypes = [vp, ctypes.POINTER(dbl)]
    L.bear_plan_pair_contexts.argtypes = [vp, vp, cint, ctypes.POINTER(cint), vp]
    L.bear_kmer_order_u64.argtypes = [vp, u64, cint, vp, vp, ctypes.POINTER(u64), vp]
    L.bear_gather_rows.argtypes = [vp, vp, vp, u64, ctypes.c_uint32, vp]
    L.bear_eval_plan_f64.argtypes = [vp, vp, vp, vp, vp, u64, vp, cint, cint, vp, cint, dbl, u64, u64, vp, vp, vp]
    L.bear_count_rows.argtypes = [ctypes.c_char_p, ctypes.POINTER(u64)]
    L.bear_count_newlines.argtypes = [ctypes.c_char_p, ctypes.POINTER(u64)]
    L.bear_parse_counts_tsv.argtypes = [ctypes.c_char_p, cint, cint, u64, vp, vp, ctypes.POINTER(u64)]
    for name in SYMBOLS:
        fn = getattr(L, name)
        if name in ("bear_plan_bytes", "bear_shuffle_source_row", "bear_eval_plan_bytes"):
            continue
        if fn.restype is ctypes.c_int and name not in ("bear_abi_version", "bear_last_hip_error"):
            fn.restype = cint
    _lib = L
    return L


def lib():
    return _load()


def check(status, where):
    if status != 0:
        raise BearError(status, where)
