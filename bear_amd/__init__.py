"""bear_amd -- MI355X-native implementation of BEAR's empirical-Bayes training hot path.

Host side mirrors the reference's Python surface (ar_funcs / bear_net / bear_ref /
dataloader / core); the arithmetic runs in hand-written HIP kernels behind the C ABI in
``include/bear_hip.h`` (``bear_amd/libbear_hip.so``).  There is no CPU fallback.
"""
__version__ = "0.1.0"
