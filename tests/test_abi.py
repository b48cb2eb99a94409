"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/bear_hip.h declares, fails loudly without a device, and its host-side count-table
parser reproduces the reference's golden first batch."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT, YSD1


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "bear_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(bear_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from bear_amd import _lib
    L = _lib.lib()
    names = _declared_symbols()
    assert len(names) >= 10
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/bear_hip.h but not exported"
    assert sorted(_lib.SYMBOLS) == names
    assert L.bear_abi_version() == 1
    assert L.bear_strerror(0) == b"ok"


def test_no_device_is_a_loud_error():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a device is present")
    from bear_amd import _lib, kernels
    h = ctypes.c_void_p()
    st = _lib.lib().bear_ws_create(0, ctypes.byref(h))
    assert st == -2  # BEAR_ERR_NO_DEVICE
    with pytest.raises(RuntimeError):
        kernels.Workspace()


def test_parser_matches_reference_golden(ysd1):
    from bear_amd import _lib
    L = _lib.lib()
    n = ctypes.c_uint64()
    assert L.bear_count_rows(YSD1.encode(), ctypes.byref(n)) == 0
    assert n.value == 1365
    kmers = np.zeros((1365, 5), dtype=np.uint8)
    counts = np.zeros((3, 1365, 5), dtype=np.uint32)
    got = ctypes.c_uint64()
    st = L.bear_parse_counts_tsv(YSD1.encode(), 3, 5, 1365, kmers.ctypes.data, counts.ctypes.data, ctypes.byref(got))
    assert st == 0 and got.value == 1365
    ref_kmers, ref_counts = ysd1
    assert [bytes(k).decode() for k in kmers] == ref_kmers
    assert np.array_equal(counts.transpose(1, 0, 2).astype(np.float64), ref_counts)  # bit-exact counts
    # tests/test_dataloader.py:25-30
    assert bytes(kmers[0]) == b"TAATC" and counts[0, 0].tolist() == [14837, 15127, 22260, 16279, 446]
    assert counts[2, 1].tolist() == [69, 0, 45, 39, 0]


def test_parser_errors(tmp_path):
    from bear_amd import _lib
    L = _lib.lib()
    n = ctypes.c_uint64()
    assert L.bear_count_rows(b"/nonexistent/file.tsv", ctypes.byref(n)) == -6
    bad = tmp_path / "bad.tsv"
    bad.write_text("ACG\t[[1,2,3,4,5],[1,2,3,4]]\n")
    counts = np.zeros((2, 4, 5), dtype=np.uint32)
    st = L.bear_parse_counts_tsv(str(bad).encode(), 2, 3, 4, None, counts.ctypes.data, ctypes.byref(n))
    assert st == -7
    ok = tmp_path / "ok.tsv"
    ok.write_text("[[A\t[[1,2,3,4,5],[0,0,0,0,4000000000]]\n\nAC[\t[[0,0,0,0,0],[7,8,9,10,11]]\n")
    kmers = np.zeros((4, 3), dtype=np.uint8)
    st = L.bear_parse_counts_tsv(str(ok).encode(), 2, 3, 4, kmers.ctypes.data, counts.ctypes.data, ctypes.byref(n))
    assert st == 0 and n.value == 2
    assert bytes(kmers[0]) == b"[[A" and counts[1, 0, 4] == 4000000000 and counts[1, 1].tolist() == [7, 8, 9, 10, 11]
    # wrong lag
    st = L.bear_parse_counts_tsv(str(ok).encode(), 2, 4, 4, None, counts.ctypes.data, ctypes.byref(n))
    assert st == -7
