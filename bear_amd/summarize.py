"""k-mer transition count tables from sequence files: host mirror of ``bear_model/summarize.py``.

``main(args)`` / ``run(args)`` keep the reference's arguments (summarize.py:650-696: ``file`` -- a csv of
``FILE, GROUP, TYPE`` rows with TYPE ``fa`` / ``fq`` --, ``out_prefix``, ``l`` max lag, ``nf``, ``r``, ``mf``; the KMC
options ``mk``, ``p``, ``t``, ``pr``, ``s12``, ``s3`` are accepted and ignored) and write the same files,
``<out_prefix>_lag_<L>_file_<b>.tsv`` with rows ``kmer \\t [[A,C,G,T,$ of group 0],[group 1],...]``, '[' padded
(summarize.py:429-449, 472-473).  Where the reference writes prefix / suffix / full FASTQ files, runs the external KMC
counter and heap-merges its dumps (stages 1-3), this build counts on the device: the sequences are uploaded once as a
code text and every lag is one emit + radix sort + run-length reduce (``bear_kmer_sort_*``, bear_count.hip).  What is
counted is what the reference's test defines (tests/test_summarize.py:88-115).  ``count_tables`` returns the tables as
``CountDataset`` objects without writing text at all.

Rows come out sorted by packed k-mer code and are dealt round-robin to the output bins (the reference assigns rows to
random bins, summarize.py:439,447, and states that the order carries no meaning, :72); shuffle before training
(``CountDataset.shuffle`` or ``shuf``) exactly as with the reference's files.
"""
import csv
import ctypes
import datetime

import numpy as np
import torch

from . import _lib
from .dataloader import CountDataset, DeviceCountDataset

alphabet = {"A": 0, "C": 1, "G": 2, "T": 3, "]": 4}     # summarize.py:380

_START, _STOP, _OTHER = 5, 4, 6
_LUT = np.full(256, _OTHER, dtype=np.uint8)
for _ch, _v in (("A", 0), ("C", 1), ("G", 2), ("T", 3)):
    _LUT[ord(_ch)] = _v
_COMP = np.array([3, 2, 1, 0, 4, 5, 6], dtype=np.uint8)    # reverse complement on codes


def load_input(in_file, file_type):
    """summarize.py:96-100 (Biopython's SimpleFastaParser / FastqGeneralIterator): yields ``(name, seq)``."""
    if file_type == "fa":
        name, parts = None, []
        for line in in_file:
            line = line.rstrip("\n\r")
            if line.startswith(">"):
                if name is not None:
                    yield name, "".join(parts)
                name, parts = line[1:], []
            elif name is not None:
                parts.append(line.strip())
        if name is not None:
            yield name, "".join(parts)
    elif file_type == "fq":
        while True:
            head = in_file.readline()
            if not head:
                return
            if not head.strip():
                continue
            seq = in_file.readline().rstrip("\n\r")
            in_file.readline()
            in_file.readline()
            yield head[1:].rstrip("\n\r"), seq
    else:
        raise ValueError("file type must be 'fa' or 'fq'")


def read_file_list(seq_list_file):
    """summarize.py:252-256: rows ``FILE, GROUP, TYPE``."""
    rows = []
    with open(seq_list_file, newline="") as fh:
        for row in csv.reader(fh):
            if row:
                rows.append((row[0].strip(), int(row[1]), row[2].strip()))
    return rows


def encode_sequences(seqs, groups, reverse=False):
    """Sequences (str) with their group ids -> the device text of bear_kmer_sort_create: per sequence a start marker,
    the letter codes and the stop code; with ``reverse`` every sequence is followed by its reverse complement
    (summarize.py:202-207).  Returns ``(text uint8 [n_pos], group uint8 [n_pos])``."""
    parts, gparts = [], []
    for seq, g in zip(seqs, groups):
        if not 0 <= int(g) <= 254:
            raise ValueError("group ids must lie in [0, 254]")
        codes = _LUT[np.frombuffer(seq.upper().encode("ascii", "replace"), dtype=np.uint8)]
        for c in ((codes, _COMP[codes[::-1]]) if reverse else (codes,)):
            parts.append(np.concatenate([[_START], c, [_STOP]]).astype(np.uint8))
            gparts.append(np.full(c.size + 2, int(g), dtype=np.uint8))
    if not parts:
        return np.zeros(0, dtype=np.uint8), np.zeros(0, dtype=np.uint8)
    return np.concatenate(parts), np.concatenate(gparts)


def count_transitions(text, group, lag, n_groups, device=None, on_device=False):
    """One lag on the device.  text / group: uint8 arrays or CUDA tensors.  Returns
    ``(kmers uint8 [n_rows, lag] ASCII, counts uint32 [n_groups, n_rows, 5])`` as numpy arrays, or with ``on_device`` as
    CUDA tensors (counts in int32 storage)."""
    if not torch.cuda.is_available():
        raise RuntimeError("bear_amd counts on an MI355X only (libbear_hip.so has no CPU fallback)")
    device = torch.device(device or "cuda")
    t = text if isinstance(text, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(text)).to(device)
    g = group if isinstance(group, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(group)).to(device)
    if t.dtype != torch.uint8 or g.dtype != torch.uint8 or t.shape != g.shape or t.dim() != 1:
        raise ValueError("text and group must be uint8 vectors of the same length")
    L = _lib.lib()
    h, n_rows = ctypes.c_void_p(), ctypes.c_uint64()
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    with torch.cuda.device(t.device):
        _lib.check(L.bear_kmer_sort_create(t.data_ptr(), g.data_ptr(), t.numel(), int(lag), ctypes.byref(h), ctypes.byref(n_rows),
                                           stream), "bear_kmer_sort_create")
        try:
            n = n_rows.value
            kmers = torch.empty((n, lag), dtype=torch.uint8, device=t.device)
            counts = torch.empty((n_groups, n, 5), dtype=torch.int32, device=t.device)
            _lib.check(L.bear_kmer_sort_reduce(h, int(n_groups), kmers.data_ptr(), None, counts.data_ptr(), stream),
                       "bear_kmer_sort_reduce")
            torch.cuda.current_stream().synchronize()
        finally:
            L.bear_kmer_sort_destroy(h)
    if on_device:
        return kmers, counts
    return kmers.cpu().numpy(), counts.cpu().numpy().view(np.uint32)


def _load_sequences(seq_list_file):
    seqs, groups = [], []
    for path, group, ftype in read_file_list(seq_list_file):
        with open(path) as fh:
            for _, seq in load_input(fh, ftype):
                seqs.append(seq)
                groups.append(group)
    return seqs, groups


def load_text(seq_list_file, reverse=False):
    """Every file of the list as the device code text (C++ readers, ``bear_fastx_encode``): ``(text, group, n_groups)``."""
    L = _lib.lib()
    parts, gparts, groups = [], [], []
    for path, group, ftype in read_file_list(seq_list_file):
        if ftype not in ("fa", "fq"):
            raise ValueError("file type must be 'fa' or 'fq'")
        n = ctypes.c_uint64()
        _lib.check(L.bear_fastx_size(path.encode(), int(ftype == "fq"), int(bool(reverse)), ctypes.byref(n), None), "bear_fastx_size")
        text = np.empty(n.value, dtype=np.uint8)
        grp = np.empty(n.value, dtype=np.uint8)
        got = ctypes.c_uint64()
        _lib.check(L.bear_fastx_encode(path.encode(), int(ftype == "fq"), int(bool(reverse)), int(group), n.value, text.ctypes.data,
                                       grp.ctypes.data, ctypes.byref(got)), "bear_fastx_encode")
        if got.value != n.value:      # the file changed between the sizing pass and the encoding pass
            raise RuntimeError(f"{path}: {got.value} positions encoded, {n.value} counted")
        parts.append(text)
        gparts.append(grp)
        groups.append(group)
    if not parts:
        return np.zeros(0, dtype=np.uint8), np.zeros(0, dtype=np.uint8), 1
    return np.concatenate(parts), np.concatenate(gparts), max(groups) + 1


def count_tables(seq_list_file, max_lag, reverse=False, batch_size=1 << 30, device=None, on_device=False):
    """The tables of every lag 1..max_lag as ``CountDataset`` objects (index L-1), never written as text; with
    ``on_device`` as ``DeviceCountDataset`` objects that never leave HBM (count -> shuffle -> plan -> train)."""
    text, grp, n_groups = load_text(seq_list_file, reverse)
    device = torch.device(device or "cuda")
    t, g = torch.from_numpy(text).to(device), torch.from_numpy(grp).to(device)
    out = []
    for lag in range(1, max_lag + 1):
        kmers, counts = count_transitions(t, g, lag, n_groups, on_device=on_device)
        out.append(DeviceCountDataset(kmers, counts, "dna", batch_size) if on_device else CountDataset(kmers, counts, "dna", batch_size))
    return out


def compute_n_bin_bits(total_size, n_groups, mf):
    """summarize.py:594-598."""
    if total_size <= 0:
        return 0
    return int(max([np.ceil(np.log(total_size * n_groups / (mf * 1e9)) / np.log(2)), 0]))


def write_tables(tables, out_prefix, n_bins):
    """``<out_prefix>_lag_<L>_file_<b>.tsv`` (summarize.py:472-473, 529-530), rows dealt round-robin to the bins."""
    L = _lib.lib()
    for li, d in enumerate(tables):
        km = np.ascontiguousarray(d.kmers)
        cn = np.ascontiguousarray(d.counts)
        for b in range(n_bins):
            path = "{}_lag_{}_file_{}.tsv".format(out_prefix, li + 1, b)
            _lib.check(L.bear_write_counts_tsv(path.encode(), km.ctypes.data, cn.ctypes.data, d.num_rows, li + 1, d.num_ds,
                                               b, n_bins, 0), "bear_write_counts_tsv")


def run(args):
    """summarize.py:622-645: all stages for one direction."""
    print("Start: counting on the device...", datetime.datetime.now())
    tables = count_tables(args.file, args.l, reverse=bool(args.r))
    n_groups = tables[0].num_ds if tables else 1
    # the reference sizes the bins from the KMC dump sizes (kmer \\t count \\n per distinct k+1-mer); same formula
    total_size = sum(int((d.counts > 0).sum()) * (li + 2 + 4) for li, d in enumerate(tables))
    n_bins = 2 ** compute_n_bin_bits(total_size, n_groups, float(getattr(args, "mf", None) or 0.1))
    write_tables(tables, args.out_prefix, n_bins)
    print("Finished.", datetime.datetime.now())
    return n_bins


def main(args):
    """summarize.py:648-665: forward tables under ``out_prefix``, with ``-r`` forward + reverse-complement tables under
    ``out_prefix + '_rev'``.  Returns ``(n_bins, n_bins_rev)``."""
    store_r, prefix = bool(getattr(args, "r", False)), args.out_prefix
    n_bins = n_bins_rev = None
    if not getattr(args, "nf", False):
        args.r = False
        n_bins = run(args)
    if store_r:
        args.r = True
        args.out_prefix = prefix + "_rev"
        n_bins_rev = run(args)
    args.r, args.out_prefix = store_r, prefix
    return n_bins, n_bins_rev


if __name__ == "__main__":
    import argparse
    parser = argparse.ArgumentParser(description="Count k-mer transitions for BEAR training (device build).")
    parser.add_argument("file")
    parser.add_argument("out_prefix")
    parser.add_argument("-l", default=10, type=int)
    parser.add_argument("-mk", default=12, type=float)
    parser.add_argument("-mf", default=0.1, type=float)
    parser.add_argument("-p", default="")
    parser.add_argument("-nf", action="store_true", default=False)
    parser.add_argument("-r", action="store_true", default=False)
    parser.add_argument("-pr", action="store_true", default=False)
    parser.add_argument("-t", default="tmp/")
    parser.add_argument("-s12", action="store_true", default=False)
    parser.add_argument("-s3", action="store_true", default=False)
    main(parser.parse_args())
