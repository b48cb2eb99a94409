"""Sequence generation from seeds under a BEAR / BMM model: host mirror of ``bear_model/assemble.py``.

``assemble_no_ends`` (assemble.py:21-184) with the reference's arguments: for every seed and replicate a flank is grown
to the right of the seed and (through the reverse complement) to its left, one letter at a time; the transition
log-probabilities of the current end k-mer come from ``get_var_probs.get_pdf`` (posterior samples -- one AR model per
generated sequence -- or the MAP table), the letter from a Gumbel-max draw (:121).  The sampling of the tables runs on
the device (``bear_logdir_sample_f64``); the per-letter bookkeeping stays on the host as in the reference.

Differences: transition counts come from ``counter`` (e.g. ``get_var_probs.make_sequence_counter(...)``, the device-built
table look-up) instead of a KMC database (``kmc_path`` still works where ``py_kmc_api`` is installed); a BMM run
(``van=...``) does not need a trained-model folder; and a new k-mer's table row is stored under the index the k-mer
was given (the reference indexes k-mers in order of appearance but stores their rows in sorted order, assemble.py:96-99,
so that rows can be swapped between k-mers that first appear in the same step).
"""
import os

import numpy as np
from scipy.special import xlogy

from . import core, get_var_probs, summarize

_COMP = str.maketrans("ACGTUacgtu", "TGCAAtgcaa")


def reverse_complement(seq):
    return seq.translate(_COMP)[::-1]


def assemble_no_ends(seqs_fa_file, lengths_to_gen, num_to_gen, bear_path, kmc_path,
                     h=None, reverse=True, save_folder=None, batch_size=100,
                     van=None, lag=None, alphabet_name=None, get_map=False, counter=None, seed=None):
    """assemble.assemble_no_ends (assemble.py:21-184) -> ``(gen_seqs [len(seqs), num_to_gen], sw_ent)``."""
    if bear_path is not None:
        lag, alphabet_name, h_bear, ar_func, _ = get_var_probs.load_bear(bear_path)
        if h is None:
            h = h_bear
    else:
        assert van is not None, "without a trained model folder only a BMM (van=...) can generate"
        ar_func = None
    h = np.array([h]) if h is not None else None
    if van is not None:
        assert lag is not None and alphabet_name is not None
        vans, ar_func, h = van * np.ones(1), None, None
    else:
        vans = []
    train_col = 0
    alphabet = core.alphabets_en[alphabet_name][:-1]
    alphabet_size = len(alphabet)
    if counter is None:
        counter = get_var_probs.make_kmc_genome_counter(kmc_path, lag, reverse=reverse, no_end=True)
    rng = np.random if seed is None else np.random.RandomState(seed)

    with open(seqs_fa_file) as fh:
        fwd_seqs = np.array([s for _, s in summarize.load_input(fh, "fa")])
    n_seeds = len(fwd_seqs)
    fwd_seqs = np.repeat(fwd_seqs, num_to_gen)
    lengths = np.repeat(np.asarray(lengths_to_gen).reshape(n_seeds, 2), num_to_gen, axis=0)     # [:, 0] left, [:, 1] right
    rev_seqs = np.array([reverse_complement(s) for s in fwd_seqs])

    flanks = []
    for seqs_all_b, length_all_b in zip([rev_seqs, fwd_seqs], [lengths[:, 0], lengths[:, 1]]):
        all_batch_new = []
        for lo in range(0, len(seqs_all_b), batch_size):
            seqs = seqs_all_b[lo:lo + batch_size]
            length_to_gen = length_all_b[lo:lo + batch_size]
            new_seq = len(seqs) * [""]
            new_len = np.zeros(len(seqs), dtype=int)
            inds = np.nonzero(new_len < length_to_gen)[0]
            end_kmers = np.array([s[-lag:] for s in seqs[inds]])
            kmer_row, all_pdf = {}, None                      # k-mer -> row of all_pdf [rows, letters, (live sequences)]
            step = 0
            while len(inds) > 0:
                new_kmers = np.unique([k for k in end_kmers if k not in kmer_row])
                if len(new_kmers) > 0:
                    counts = counter(new_kmers)[:, None, :]
                    for k in new_kmers:
                        kmer_row[k] = len(kmer_row)
                    pdf = get_var_probs.get_pdf(new_kmers, counts, h, ar_func, len(end_kmers), vans, train_col, alphabet_name,
                                                get_map, output="numpy", seed=None if seed is None else seed + 7919 * step,
                                                row_base=len(kmer_row))[:, :-1, 0, :]
                    all_pdf = pdf if all_pdf is None else np.concatenate([all_pdf, pdf])
                rows = np.array([kmer_row[k] for k in end_kmers])
                tlp = all_pdf[rows]                               # [live, letters, samples]
                tlp = tlp[:, :, 0] if get_map else tlp[np.arange(len(rows)), :, np.arange(len(rows))]
                new_letters = alphabet[np.argmax(rng.gumbel(size=tlp.shape) + tlp, axis=-1)]      # assemble.py:121
                keep = new_len[inds] + 1 < length_to_gen[inds]
                for j, ind in enumerate(inds):
                    new_seq[ind] += new_letters[j]
                    new_len[ind] += 1
                end_kmers = np.array([e[1:] + l for e, l, k in zip(end_kmers, new_letters, keep) if k])
                if not get_map:
                    all_pdf = all_pdf[:, :, keep]                 # one posterior sample per live sequence (assemble.py:126)
                inds = inds[keep]
                step += 1
            all_batch_new.append(new_seq)
        flanks.append(np.concatenate(all_batch_new) if all_batch_new else np.array([]))

    gen_seqs = [reverse_complement(left) + seed_seq + right for left, right, seed_seq in zip(flanks[0], flanks[1], fwd_seqs)]
    gen_seqs = np.array(gen_seqs).reshape([-1, num_to_gen])
    sw_ent = []
    for seqs in gen_seqs:
        probs = np.average(core.tf_one_hot(list(seqs), alphabet_name).cpu().numpy(), axis=0)
        sw_ent.append(-np.sum(xlogy(probs, probs), axis=-1))
    if save_folder is not None:
        os.makedirs(save_folder, exist_ok=True)
        with open(os.path.join(save_folder, "seqs.fa"), "w") as f:
            for i, seqs in enumerate(gen_seqs):
                for j, seq in enumerate(seqs):
                    f.write(">seq{}_rep{}\n{}\n".format(i, j, seq))
        try:
            import matplotlib
            matplotlib.use("Agg")
            from matplotlib import pyplot as plt
            plt.figure(figsize=[10, 5])
            plt.xlabel("entropy", fontsize=15)
            plt.ylabel("position", fontsize=15)
            for ent, ltg in zip(sw_ent, np.asarray(lengths_to_gen).reshape(n_seeds, 2)):
                plt.plot(np.arange(len(ent)) - ltg[0], ent, color="blue", linewidth=1, alpha=0.1)
            plt.savefig(os.path.join(save_folder, "entropy.png"), dpi=200)
            plt.close()
        except Exception:
            pass
    return gen_seqs, sw_ent
