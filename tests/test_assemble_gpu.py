"""GPU test of the assemble mirror (bear_model/assemble.py has no test in the reference): structure of the output, a
deterministic table (one overwhelmingly likely continuation) and the letter frequencies of a BMM against its posterior
mean."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_assemble_bmm(tmp_path):
    from bear_amd import assemble, get_var_probs
    # training sequences: ACGT repeated -> after ACG comes T, after CGT comes A, ... with overwhelming counts
    train = ["ACGT" * 50] * 40
    counter = get_var_probs.make_sequence_counter(train, 3, reverse=False, no_end=True)
    seeds = tmp_path / "seeds.fa"
    seeds.write_text(">s0\nACGTACG\n>s1\nGTACGTA\n")
    gen, ent = assemble.assemble_no_ends(str(seeds), [[0, 12], [4, 8]], 5, None, None, van=0.01, lag=3, alphabet_name="dna",
                                          counter=counter, seed=3, save_folder=str(tmp_path / "out"))
    assert gen.shape == (2, 5) and len(ent) == 2
    for s in gen[0]:
        assert s == "ACGTACG" + "TACGTACGTACG"                      # the periodic continuation, probability ~1 - 1e-4 per letter
    for s in gen[1]:
        assert len(s) == 4 + 7 + 8 and s[4:11] == "GTACGTA" and s[11:] == "CGTACGTA"
    assert (tmp_path / "out" / "seqs.fa").read_text().count(">") == 10
    assert ent[0].shape == (19,) and ent[1].shape == (19,) and np.all(ent[0] == 0.0)      # all replicates agree: zero site-wise entropy
    # uninformative table: letters follow the flat posterior mean
    flat = lambda kmers: np.zeros(np.shape(kmers) + (5,))
    gen, _ = assemble.assemble_no_ends(str(seeds), [[0, 300], [0, 300]], 3, None, None, van=1.0, lag=3, alphabet_name="dna",
                                       counter=flat, seed=4, get_map=True)
    letters = np.array(list("".join(s[7:] for s in gen.reshape(-1))))
    freq = np.array([(letters == a).mean() for a in "ACGT"])
    assert np.all(np.abs(freq - 0.25) < 0.05)
