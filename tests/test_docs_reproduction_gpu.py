"""End-to-end anchor on the reference's published results (docs/usage.rst:236-265): the six example configs, trained for
the documented 10 000 steps on the bundled YSD1 lag-5 table through the config driver, must land on the documented
held-out perplexity / accuracy and on the documented fitted h (which the reference quotes to three digits).

The linear and reference configs do not depend on the initialisation (h = 0.0432576 ... 0.0432609 over 14 seeds).  The CNN
configs do, and the docs quote ONE run of the reference: profiles/r05_docs_cnn_seeds.jsonl holds 14 seeds of each in the
deterministic build (every run bit-reproducible: seed 10 twice gives the same h to the last digit), scripts/seed_sweep.py:
  bear_cnn_bear  h          0.01071 ... 0.01290, mean 0.01203, sd 0.00065   docs 0.0119  (0.2 sd below the mean)
  bear_cnn_ar    perplexity 3.8500 ... 3.8788,  median 3.8527, sd 0.0069    docs 3.85
                 accuracy   35.21 ... 35.73 %,  median 35.62,  sd 0.12      docs 35.8    (1.8 sd above the mean)
Round 6, 70 seeds of bear_cnn_ar (profiles/r06_docs_cnn_seeds.jsonl): accuracy 34.99 ... 35.79 %, mean 35.612, sd 0.153; TEN of the
70 runs print as the documented "35.8" (>= 35.75): the docs' run is an ordinary draw, +0.9 ... +1.6 sd, not a systematic gap (round
5's 14 seeds happened to stop at 35.73).  Perplexity 3.8458 ... 3.8714, mean 3.8536, sd 0.0052 (right-skewed); 71 % of the runs
print as "3.85".
The documented values are draws from those distributions; the tolerances of the CNN rows below are mean +- 3 sd of them (perplexity:
4 sd, the skewed side) -- for the run AND for the documented value (the regular build adds summation-order noise of +-0.0003 in h at
a fixed seed: two runs of seed 10 gave 0.01206 and 0.01244)."""
import configparser
import json
import os

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

DOCS = {  # config: (kind, which, perplexity, accuracy %, h, perplexity tol, accuracy tol)   docs/usage.rst:258-264
    "bear_lin_ar": ("net", "AR", 3.99, 32.9, None, 0.01, 0.15),
    "bear_cnn_ar": ("net", "AR", 3.85, 35.8, None, 0.035, 0.7),     # (seed distribution: CNN_AR below)
    "bear_stop_ar": ("ref", "AR", 3.84, 36.5, None, 0.01, 0.15),
    "bear_lin_bear": ("net", "BEAR", 3.79, 36.8, 0.0433, 0.006, 0.1),
    "bear_cnn_bear": ("net", "BEAR", 3.79, 36.8, 0.0119, 0.006, 0.1),
    "bear_stop_bear": ("ref", "BEAR", 3.79, 36.8, 0.0142, 0.006, 0.1),
}


CNN_AR = {"perplexity": (3.8536, 4 * 0.0052), "accuracy": (35.612, 3 * 0.153)}     # (mean, tolerance) over 70 seeds


@pytest.mark.parametrize("name", list(DOCS))
def test_example_config_reproduces_documented_results(name, tmp_path):
    from bear_amd.models import _driver
    kind, which, perp, acc, h, ptol, atol = DOCS[name]
    config = configparser.ConfigParser()
    config.read(os.path.join(ROOT, "bear_amd", "models", "config_files", name + ".cfg"))
    config["train"]["epochs"] = "10000"          # the reference's configs: 10000 epochs of one 1365-row batch
    config["train"]["batch_size"] = "1500"
    config["general"]["out_folder"] = str(tmp_path) + "*"
    _driver.main(config, kind)
    r = config["results"]
    assert abs(float(r["heldout_perplex_" + which]) - perp) < ptol
    assert abs(100 * float(r["heldout_accuracy_" + which]) - acc) < atol
    if name == "bear_cnn_ar":      # this run and the documented run are both draws of the seed distribution
        for got, docs, (mean, tol) in ((float(r["heldout_perplex_AR"]), perp, CNN_AR["perplexity"]),
                                       (100 * float(r["heldout_accuracy_AR"]), acc, CNN_AR["accuracy"])):
            assert abs(got - mean) < tol and abs(docs - mean) < tol, (got, docs, mean, tol)
    bmm = json.loads(r["heldout_perplex_BMM"])
    assert all(abs(v - 3.79) < 0.006 for v in bmm)                                  # docs/usage.rst:261
    if h is not None:
        if "cnn" in name:     # mean +- 3 sd of the seed distribution (module docstring); the documented 0.0119 lies inside it
            assert abs(float(r["h"]) - 0.01203) < 3 * 0.00065 and abs(h - 0.01203) < 3 * 0.00065
        else:                 # stable to three digits
            assert abs(float(r["h"]) - h) / h < 0.015
