"""Posterior predictive scores of variants and sequences: host mirror of ``bear_model/get_var_probs.py``.

Same functions and argument meaning as the reference: ``get_pdf`` (get_var_probs.py:91-194),
``get_bear_probs`` (:330-452), ``get_bear_probs_seqs`` (:498-631), ``parse_var`` (:323-328), ``load_bear``
(:58-82), ``load_ds`` (:36-56).  The numerical part of ``get_pdf`` -- concentrations of every model, the
log-Gamma draws of ``log_gamma.log_gamma`` and their normalisation, or the MAP table -- is one launch of
``bear_logdir_sample_f64`` over the batch of k-mers (``kernels_sample.h``); the exact marginal term
(``get_marg``) is one launch of ``bear_eval_f64``.  The k+1-mer bookkeeping stays on the host as in the
reference (string windows, a lookup table indexed by k+1-mer).

Differences a user can see: draws come from a counter-based stream (``seed`` argument; ``None`` takes a fresh
seed from numpy's global generator) instead of numpy's Mersenne twister, and the KMC database look-up
(``make_kmc_genome_counter``, :213-290, external ``py_kmc_api``) is replaced by ``make_sequence_counter``: the same counts
from the sequence files themselves, counted on the device and passed as ``counter=``.
"""
import configparser
import json
import os
import pickle

import numpy as np
import torch

from . import ar_funcs, bear_net, core, dataloader, kernels
from .log_gamma import _next_seed

epsilon = core.epsilon


def cross_str_arrays(array1, array2, exch="X"):
    """get_var_probs.py:23-34: every string of array1 followed by every string of array2, array1-major."""
    a1 = np.asarray(array1).astype(str)
    a2 = np.asarray(array2).astype(str)
    return np.char.add(a1[:, None], a2[None, :]).reshape(-1)


def load_ds(files_path, start_token, kmer_batch_size, sparse, alphabet, num_ds, dtype=torch.float64):
    """get_var_probs.py:36-56."""
    files = sorted(os.path.join(files_path, f) for f in os.listdir(files_path) if f.startswith(start_token))
    load = dataloader.sparse_dataloader if sparse else dataloader.dataloader
    parts = [load(f, alphabet, kmer_batch_size, num_ds, cache=False, dtype=dtype) for f in files]
    return parts[0] if len(parts) == 1 else dataloader.concatenate(parts)


def load_bear(path):
    """get_var_probs.py:58-82: a trained model folder (config.cfg + results.pickle) ->
    ``(lag, alphabet, h, ar_func, data)``; ``ar_func`` is the reference's ``softmax(ar_func(.)) + epsilon``
    wrapper (:79-81)."""
    config = configparser.ConfigParser()
    config.read(os.path.join(path, "config.cfg"))
    dtype = getattr(torch, config["general"]["precision"])
    lag = int(config["hyperp"]["lag"])
    alphabet = config["data"]["alphabet"]
    alphabet_size = len(core.alphabets_tf[alphabet]) - 1
    make_ar_func = getattr(ar_funcs, "make_ar_func_" + config["model"]["ar_func_name"])
    af_kwargs = json.loads(config["model"]["af_kwargs"])
    with open(os.path.join(path, "results.pickle"), "rb") as fr:
        params_restart = pickle.load(fr)["params"]
    device = torch.device("cuda", torch.cuda.current_device())
    params, h_signed, ar_func = bear_net.change_scope_params(lag, alphabet_size, make_ar_func, af_kwargs, params_restart,
                                                             dtype=dtype, device=device)
    h = float(np.exp(h_signed.item()))
    files_path = config["data"]["files_path"]
    if files_path == "TEST":
        data = dataloader.dataloader(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data",
                                                  "ysd1_lag_5_file_0_preshuf.tsv"), alphabet,
                                     int(float(config["train"]["batch_size"])) or 1, int(config["data"]["num_ds"]))
    else:
        data = load_ds(files_path, config["data"]["start_token"], int(float(config["train"]["batch_size"])),
                       config["data"]["sparse"] == "True", alphabet, int(config["data"]["num_ds"]))

    def ar_func_tf(kmers):
        with torch.no_grad():
            return torch.softmax(ar_func(kmers), dim=-1) + epsilon
    return lag, alphabet, h, ar_func_tf, data


class _Table:
    """The reference's DataFrame indexed by k+1-mer (get_var_probs.py:185-194) as a dense array
    ``values[kmer, letter] -> [num_models, mc_samples]`` plus the k-mer index."""

    def __init__(self, kmers, letters, values):
        self.index = {k: i for i, k in enumerate(kmers)}
        self.letter = {b: j for j, b in enumerate(letters)}
        self.values = values          # numpy [K, A+1, M, mc]
        self.lag = len(kmers[0]) if len(kmers) else 0

    def rows(self, kp1mers):
        k = np.fromiter((self.index[s[:self.lag]] for s in kp1mers), dtype=np.int64, count=len(kp1mers))
        b = np.fromiter((self.letter[s[self.lag:]] for s in kp1mers), dtype=np.int64, count=len(kp1mers))
        return self.values[k, b]      # [n, M, mc]


def df_to_func(table, num_models, mc_samples, summed=True):
    """get_var_probs.py:84-89."""
    if summed:
        return lambda kp1mers_ex: np.sum(table.rows(kp1mers_ex).reshape([-1, num_models, mc_samples]), axis=0)
    return lambda kp1mers_ex: table.rows(kp1mers_ex).reshape([-1, num_models, mc_samples])


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("bear_amd samples on an MI355X only (libbear_hip.so has no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


def get_pdf(kmers, counts, h, ar_func, mc_samples, vans, train_col, alphabet_name, get_map,
            get_marg=False, summed=True, output="func", seed=None, row_base=0):
    """get_var_probs.get_pdf (get_var_probs.py:91-194).  ``kmers``: strings; ``counts``: [K, num_ds, A+1];
    ``h``: sequence of h values (used when ``ar_func`` is given); ``vans``: BMM pseudo-counts.  Returns, by
    ``output``: 'func' a function of k+1-mers -> [num_models, mc_samples] (summed) or [n, num_models, mc_samples];
    'df' a pandas DataFrame indexed by k+1-mer; 'numpy' an array [k-mer, letter, model, mc_sample].
    With ``get_marg`` a function ``(kmers, counts) -> [num_models]`` (exact DM marginal, :155-173)."""
    assert not (get_marg and get_map), "pick marg or map"
    assert not (get_marg and output != "func"), "not implemented"
    alphabet = core.alphabets_en[alphabet_name]
    A1 = len(alphabet)
    if A1 != 5:
        raise NotImplementedError("the HIP kernels are built for 4-letter alphabets (+ stop): dna / rna")
    if get_map or get_marg:
        mc_samples = 1
    device = _device()
    kmers = [k.decode() if isinstance(k, bytes) else str(k) for k in np.asarray(kmers).reshape(-1).tolist()]
    K = len(kmers)
    counts = torch.as_tensor(np.asarray(counts.cpu() if isinstance(counts, torch.Tensor) else counts))
    col = counts.reshape(K, -1, A1)[:, train_col, :]
    if col.dtype.is_floating_point and not torch.equal(col, col.round()):
        raise ValueError("transition counts must be integers")
    col_dev = (col.to(torch.int64) & 0xFFFFFFFF).to(torch.int32).contiguous().to(device) if K else None
    vans = np.atleast_1d(np.asarray(vans, dtype=np.float64)).reshape(-1)
    num_models = len(vans)
    prior, hs = None, np.zeros(0)
    if ar_func is not None:
        hs = np.atleast_1d(np.asarray(h, dtype=np.float64)).reshape(-1)
        num_models += len(hs)
        codes = torch.from_numpy(core.encode_kmers(kmers, alphabet_name)).to(device)
        with torch.no_grad():
            prior = ar_func(codes).to(torch.float64).expand(K, A1).contiguous()
    with_ar = ar_func is not None and get_map
    num_models += int(with_ar)

    if get_marg:
        table_index = {k: i for i, k in enumerate(kmers)}

        def prob_func(q_kmers, q_counts):
            idx = torch.as_tensor([table_index[str(k)] for k in q_kmers], dtype=torch.long, device=device)
            q = torch.as_tensor(np.asarray(q_counts), dtype=torch.int64).reshape(len(idx), A1).to(torch.int32).to(device)
            out = kernels.evaluate(q.contiguous(), prior[idx].contiguous() if prior is not None else None,
                                   hs if hs.size else None, vans if vans.size else None, col_dev[idx].contiguous(),
                                   eps=0.0, with_ar=False).cpu().numpy()
            return np.concatenate([out[:hs.size], out[hs.size + 1:hs.size + 1 + vans.size]])
        return prob_func

    if K:
        log_probs = kernels.logdir_sample(col_dev, prior, hs, vans, mc_samples, get_map=get_map, with_ar=with_ar,
                                          seed=_next_seed() if seed is None else seed, row_base=row_base).cpu().numpy()
    else:
        log_probs = np.zeros((0, A1, num_models, mc_samples))
    if output == "numpy":
        return log_probs
    table = _Table(kmers, list(alphabet), log_probs)
    if output == "df":
        import pandas as pd
        df = pd.DataFrame(log_probs.reshape(K * A1, num_models * mc_samples), index=cross_str_arrays(kmers, alphabet))
        df.columns = np.arange(len(df.columns))
        return df
    return df_to_func(table, num_models, mc_samples, summed=summed)


def make_kmc_genome_counter(path, lag, reverse=True, no_end=False):
    """get_var_probs.py:213-290 looks counts up in KMC databases through ``py_kmc_api`` (an external binary
    module, imported lazily by the reference too).  This build has no KMC: use ``make_sequence_counter`` (the same
    counts from the sequence files, counted on the device) and pass it as ``counter=``."""
    try:
        import py_kmc_api  # noqa: F401
    except ImportError as e:
        raise ImportError("make_kmc_genome_counter needs the external py_kmc_api module (KMC); build the counter "
                          "with make_sequence_counter(...) and pass it as counter=, or use the count-table path (data=...)") from e
    raise NotImplementedError("KMC database look-ups are outside this build; use make_sequence_counter")


def make_sequence_counter(seqs, lag, reverse=True, no_end=False, device=None):
    """The counter of ``make_kmc_genome_counter`` (get_var_probs.py:213-290: k-mer strings -> transition counts
    ``[..., alphabet_size + 1]``) without KMC: the lag-``lag`` transition table of the sequences, counted on the device
    (``bear_amd.summarize``), as a look-up.  ``seqs``: sequences (list of str) or a summarize-style csv of
    ``FILE, GROUP, TYPE`` rows (all groups are pooled, as one KMC database is).  ``reverse`` adds the counts of the reverse
    complements (:246-248, 271-278); ``no_end`` drops the stop counts and serves only full-length contexts (:240-252)."""
    from . import summarize
    if isinstance(seqs, str):
        seqs, _ = summarize._load_sequences(seqs)
    seqs = [str(s) for s in seqs]
    text, grp = summarize.encode_sequences(seqs, [0] * len(seqs), reverse=reverse)
    kmers, counts = summarize.count_transitions(text, grp, lag, 1, device=device)
    table = {bytes(k).decode(): counts[0, i].astype(np.float64) for i, k in enumerate(kmers)}
    A1 = counts.shape[-1]

    def counter(kmers):
        kmers = np.asarray(kmers)
        out = np.zeros((kmers.size, A1))
        for i, k in enumerate(kmers.reshape(-1).tolist()):
            k = k.decode() if isinstance(k, bytes) else str(k)
            if no_end and "[" in k:
                continue
            row = table.get(k)
            if row is not None:
                out[i] = row
        if no_end:
            out[:, -1] = 0
        return out.reshape(kmers.shape + (A1,))
    return counter


# ------------------------------------------------------------------------------------------- variants
def _add_kmer_probs_vars(vars_, scores, wt_seq, pdf, lag, seen_kmers):
    """get_var_probs.py:294-306."""
    seen = set(np.asarray(seen_kmers).astype(str).tolist())
    for i, (wt_aa, mt_aa, pos) in enumerate(vars_):
        pos = pos + lag
        assert wt_aa == wt_seq[pos:pos + len(wt_aa)]
        wt_win = wt_seq[pos - lag:pos + lag + len(wt_aa)]
        wt_kmers = [wt_win[j:j + lag + 1] for j in range(len(wt_win) - lag) if wt_win[j:j + lag] in seen]
        mt_win = wt_seq[pos - lag:pos] + mt_aa + wt_seq[pos + len(wt_aa):pos + lag + len(wt_aa)]
        mt_kmers = [mt_win[j:j + lag + 1] for j in range(len(mt_win) - lag) if mt_win[j:j + lag] in seen]
        scores[i, :, :] += pdf(mt_kmers) - pdf(wt_kmers)
    return scores


def _get_all_kmers_vars(vars_, wt_seq, lag):
    """get_var_probs.py:309-337."""
    all_kmers = []
    for (wt_aa, mt_aa, pos) in vars_:
        pos = pos + lag
        assert wt_aa == wt_seq[pos:pos + len(wt_aa)]
        wt_win = wt_seq[pos - lag:pos + lag + len(wt_aa)]
        mt_win = wt_seq[pos - lag:pos] + mt_aa + wt_seq[pos + len(wt_aa):pos + lag + len(wt_aa)]
        all_kmers += [wt_win[j:j + lag] for j in range(len(wt_win) - lag)]
        all_kmers += [mt_win[j:j + lag] for j in range(len(mt_win) - lag)]
    return np.array(sorted(set(all_kmers))).astype(str)


def parse_var(var):
    """'AAG23CC' -> ('AAG', 'CC', 23) (get_var_probs.py:339-344)."""
    is_int = [ch.isnumeric() for ch in var]
    pos_num = int(np.min(np.argwhere(is_int)))
    len_num = int(np.sum(is_int))
    return (var[:pos_num], var[pos_num + len_num:], int(var[pos_num:pos_num + len_num]))


def _scan(data, all_kmers, add, make_pdf, alphabet_size, train_col):
    """The batch loop shared by get_bear_probs / get_bear_probs_seqs (get_var_probs.py:421-447, 598-626):
    every batch contributes the k-mers it holds; k-mers never seen contribute through the prior alone."""
    seen_all = np.zeros(len(all_kmers))
    row0 = 0
    for kmers, counts in iter(data):
        kmers = np.asarray(kmers).astype(str)
        in_kmers = np.isin(kmers, all_kmers)
        if np.sum(in_kmers) > 0:
            seen_kmers = kmers[in_kmers]
            seen_counts = np.asarray(counts)[in_kmers]
            seen_all += np.isin(all_kmers, seen_kmers)
            add(make_pdf(seen_kmers, seen_counts, row0), seen_kmers)
        row0 += len(kmers)
    unseen = seen_all == 0
    if np.sum(unseen) > 0:
        add(make_pdf(all_kmers[unseen], np.zeros([int(np.sum(unseen)), train_col + 1, alphabet_size + 1]), row0),
            all_kmers[unseen].astype(str))


def _setup(bear_path, lag, alphabet_name, h, data, vans, kmc_path, counter=None):
    if bear_path is not None:
        lag, alphabet_name, h_bear, ar_func, data = load_bear(bear_path)
        if h is None:
            h = np.array([h_bear])
        len_h = len(h)
    else:
        assert ((lag is not None and alphabet_name is not None)
                and ((data is not None or kmc_path is not None or counter is not None) and len(vans) > 0))
        len_h, ar_func = 0, None
    if kmc_path is not None and counter is None:
        make_kmc_genome_counter(kmc_path, lag)
    return lag, alphabet_name, h, ar_func, data, len_h


def get_bear_probs(bear_path, wt_seq, vars_, train_col, mc_samples=41, vans=[0.1, 1, 10], get_map=False,
                   lag=None, alphabet_name=None, h=None, data=None, kmc_path=None, kmc_reverse=False,
                   kmc_no_end=False, seed=None, counter=None):
    """get_var_probs.get_bear_probs (get_var_probs.py:346-452): posterior predictive log-probability ratios of
    variants -> [num variants, num models, mc_samples] ([num variants, num models] with ``get_map``)."""
    lag, alphabet_name, h, ar_func, data, len_h = _setup(bear_path, lag, alphabet_name, h, data, vans, kmc_path, counter)
    alphabet_size = len(core.alphabets_en[alphabet_name]) - 1
    wt_seq = lag * "[" + wt_seq + "]"
    vars_ = [parse_var(v) for v in vars_]
    all_kmers = _get_all_kmers_vars(vars_, wt_seq, lag)
    if get_map:
        mc_samples = 1
    num_models = (ar_func is not None) * (len_h + get_map) + len(vans)
    scores = np.zeros([len(vars_), num_models, mc_samples])
    seed = _next_seed() if seed is None else seed

    def make_pdf(kmers, counts, row0):
        return get_pdf(kmers, counts, h, ar_func, mc_samples, vans, train_col, alphabet_name, get_map, seed=seed, row_base=row0)

    def add(pdf, seen):
        _add_kmer_probs_vars(vars_, scores, wt_seq, pdf, lag, seen)
    if counter is not None:     # get_var_probs.py:411-420: counts looked up per k-mer instead of scanning the table
        add(make_pdf(all_kmers, counter(all_kmers)[:, None, :], 0), all_kmers)
    else:
        _scan(data, all_kmers, add, make_pdf, alphabet_size, train_col)
    return scores[..., 0] if get_map else scores


# ------------------------------------------------------------------------------------------- whole sequences
def _add_kmer_probs_seqs(seqs, scores, pdf, lag, seen_kmers, get_marg, alphabet):
    """get_var_probs.py:455-482."""
    seen = set(np.asarray(seen_kmers).astype(str).tolist())
    if get_marg:
        for i, seq in enumerate(seqs):
            kp1 = [(seq[l:l + lag], alphabet == seq[l + lag]) for l in range(len(seq) - lag) if seq[l:l + lag] in seen]
            if kp1:
                kmers_red = np.array([k for k, _ in kp1])
                counts_red = np.array([b for _, b in kp1]).astype(int)
                kmers, maps = np.unique(kmers_red, return_inverse=True)
                counts = np.zeros((len(kmers), len(alphabet)), dtype=int)
                np.add.at(counts, maps, counts_red)          # transitions of a repeated k-mer add up (:471-473)
                scores[i, :, 0] += pdf(kmers, counts)
    else:
        for i, seq in enumerate(seqs):
            kp1mers = [seq[l:l + lag + 1] for l in range(len(seq) - lag) if seq[l:l + lag] in seen]
            scores[i, :, :] += pdf(kp1mers)
    return scores


def _get_all_kmers_seqs(seqs, lag):
    """get_var_probs.py:484-508."""
    all_kmers = set()
    for seq in seqs:
        assert len(seq.replace("[", "").replace("]", "")) >= lag
        all_kmers.update(seq[j:j + lag] for j in range(len(seq) - lag))
    return np.array(sorted(all_kmers)).astype(str)


def get_bear_probs_seqs(bear_path, seqs, train_col, mc_samples=41, vans=[0.1, 1, 10], get_map=False, get_marg=False,
                        lag=None, alphabet_name=None, h=None, data=None, kmc_path=None, kmc_reverse=False,
                        no_ends=False, seed=None, counter=None):
    """get_var_probs.get_bear_probs_seqs (get_var_probs.py:511-631): log-probabilities of whole sequences ->
    [num sequences, num models, mc_samples] (last axis dropped with ``get_map`` / ``get_marg``)."""
    lag, alphabet_name, h, ar_func, data, len_h = _setup(bear_path, lag, alphabet_name, h, data, vans, kmc_path, counter)
    alphabet = core.alphabets_en[alphabet_name]
    alphabet_size = len(alphabet) - 1
    if not no_ends:
        seqs = [lag * "[" + s + "]" for s in seqs]
    all_kmers = _get_all_kmers_seqs(seqs, lag)
    if get_map or get_marg:
        mc_samples = 1
    num_models = (ar_func is not None) * (len_h + get_map) + len(vans)
    scores = np.zeros([len(seqs), num_models, mc_samples])
    seed = _next_seed() if seed is None else seed

    def make_pdf(kmers, counts, row0):
        return get_pdf(kmers, counts, h, ar_func, mc_samples, vans, train_col, alphabet_name, get_map, get_marg,
                       seed=seed, row_base=row0)

    def add(pdf, seen):
        _add_kmer_probs_seqs(seqs, scores, pdf, lag, seen, get_marg, alphabet)
    if counter is not None:     # get_var_probs.py:585-597
        add(make_pdf(all_kmers, counter(all_kmers)[:, None, :], 0), all_kmers)
    else:
        _scan(data, all_kmers, add, make_pdf, alphabet_size, train_col)
    return scores[..., 0] if (get_map or get_marg) else scores
