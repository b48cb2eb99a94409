"""The tie-breaking noise of the arg-max accuracies ON THE HIP PATH, held to the reference's own criterion.

The reference breaks ties of `ml_output` with `tf.random.normal` (core.py:69-71 `100 epsilon N(0,1)` on the Dirichlet-multinomial
concentration, core.py:134-136 `epsilon N(0,1)` on the multinomial's probabilities) and tests it statistically
(tests/test_core.py:29-39, 63-73): every arg-max is one of the tied letters, and `|sum(x - mu)| / sqrt(n) < norm.ppf(0.9995)` over
repeated draws.  The HIP kernels draw a counter-based stream keyed by (seed, model, table row, letter) instead; `oracle.eval_noise`
restates that stream bit for bit, so comparing the two says nothing about the stream's QUALITY.  Here the kernels' own outputs --
`bear_eval_plan_f64` (resident tables) and `bear_eval_f64` -- are put through the reference's construction: >= 1e5 rows whose model
concentrations tie on 2 or on all 5 letters, for the BEAR, AR and vanilla models; the accuracy sums then COUNT how often each letter
won (the held-out row is one count on the letter in question), and every count must pass the reference's z criterion, per letter
and per model, for several seeds -- and no row may ever pick a letter outside the tie."""
import numpy as np
import pytest
import scipy.stats as st
import torch

from bear_amd import kernels

pytestmark = pytest.mark.gpu

N = 200_000
Z_MAX = st.norm.ppf(0.9995)          # the reference's threshold (tests/test_core.py:39, :73)
H, VAN = [0.7, 3.0], [0.1, 1.0, 10.0]


def _counts(n, row):
    return torch.tensor(row, dtype=torch.int32, device="cuda").repeat(n, 1).contiguous()


def _wins(which, test_row, train, prior, seed, row_base=0, row_ids=None):
    """Accuracy sums {BEAR per h, AR, vanilla per van_reg} of n rows whose held-out row is `test_row`."""
    n = train.shape[0]
    test = _counts(n, test_row)
    if which == "planned":
        out = kernels.evaluate_planned(kernels.EvalPlan(test, train), prior, H, VAN, noise_seed=seed, row_base=row_base, row_ids=row_ids)
    else:
        out = kernels.evaluate(test, prior, H, VAN, train, noise_seed=seed, row_base=row_base)
    out = out.cpu().numpy()
    nh, nv = len(H), len(VAN)
    assert out[-1] == n * sum(test_row)
    return out[nh + nv + 1:2 * (nh + nv) + 2]        # cor_ear[H], cor_arm, cor_van[V]


@pytest.mark.parametrize("which", ["planned", "unplanned"])
@pytest.mark.parametrize("seed", [0, 11, 20211012])
def test_two_way_ties_follow_the_reference_criterion(which, seed):
    """The reference's construction (concentration [1, 0.5, 1]: letters 0 and 2 tie) on five letters, for all three models at
    once: the AR rows tie on letters 0 and 2, the training counts tie on the same letters (so the vanilla concentrations
    `train + van_reg + eps` and the BEAR ones `f / h + train + eps` do too)."""
    n = N
    prior = torch.tensor([0.4, 0.1, 0.4, 0.05, 0.05], dtype=torch.float64, device="cuda").repeat(n, 1).contiguous()
    train = _counts(n, [3, 1, 3, 0, 0])
    in_tie = _wins(which, [1, 0, 1, 0, 0], train, prior, seed)
    assert np.array_equal(in_tie, np.full(in_tie.shape, float(n))), in_tie       # EVERY arg-max is a tied letter ...
    outside = _wins(which, [0, 1, 0, 1, 1], train, prior, seed)
    assert not outside.any(), outside                                             # ... and never another one
    first = _wins(which, [1, 0, 0, 0, 0], train, prior, seed)
    third = _wins(which, [0, 0, 1, 0, 0], train, prior, seed)
    assert np.array_equal(first + third, np.full(first.shape, float(n)))
    # the reference's statistic: x in {0, 2}, mu = 1: sum(x - 1) = #third - #first
    z = np.abs(third - first) / np.sqrt(n)
    assert (z < Z_MAX).all(), (z, first)
    # the models draw DIFFERENT noise: their counts are not one number repeated
    assert len(set(first.tolist())) > 1


@pytest.mark.parametrize("which", ["planned", "unplanned"])
@pytest.mark.parametrize("seed", [3, 77])
def test_five_way_ties_every_letter_wins_its_share(which, seed):
    """Contexts without any training count under a flat AR row: all five letters tie in all three models (over half of the tied
    rows of a k = 13 table look like this).  Each letter must win n / 5 of the rows within the reference's z threshold
    (binomial standard deviation), for every model."""
    n = N
    prior = torch.full((n, 5), 0.2, dtype=torch.float64, device="cuda")
    train = _counts(n, [0, 0, 0, 0, 0])
    total = 0.0
    for b in range(5):
        row = [0] * 5
        row[b] = 1
        wins = _wins(which, row, train, prior, seed)
        z = np.abs(wins - n / 5.0) / np.sqrt(n * 0.2 * 0.8)
        assert (z < Z_MAX).all(), (b, z, wins)
        total = total + wins
    assert np.array_equal(total, np.full(total.shape, float(n)))


def test_planned_and_unplanned_kernels_draw_the_same_rows_noise():
    """Same (seed, model, table row, letter) -> same draw in both kernels, whatever the sharding or compaction: the planned kernel
    over a compacted, permuted half of the rows (row_ids) counts exactly what the unplanned one counts over those table rows."""
    n = 50_000
    prior = torch.tensor([0.4, 0.1, 0.4, 0.05, 0.05], dtype=torch.float64, device="cuda").repeat(n, 1).contiguous()
    train = _counts(n, [3, 1, 3, 0, 0])
    full = _wins("unplanned", [1, 0, 0, 0, 0], train, prior, 5, row_base=1000)
    again = _wins("planned", [1, 0, 0, 0, 0], train, prior, 5, row_base=1000)
    assert np.array_equal(full, again)
    ids = torch.randperm(n, device="cuda", generator=torch.Generator("cuda").manual_seed(1))[:n // 2].to(torch.int32).contiguous()
    part = _wins("planned", [1, 0, 0, 0, 0], train[:n // 2].contiguous(), prior[:n // 2].contiguous(), 5, row_base=1000, row_ids=ids)
    rest_mask = torch.ones(n, dtype=torch.bool, device="cuda")
    rest_mask[ids.long()] = False
    rest = rest_mask.nonzero().squeeze(1).to(torch.int32).contiguous()
    other = _wins("planned", [1, 0, 0, 0, 0], train[:n - n // 2].contiguous(), prior[:n - n // 2].contiguous(), 5, row_base=1000, row_ids=rest)
    assert np.array_equal(part + other, full)
