"""End-to-end anchor on the reference's published results (docs/usage.rst:236-265): the six example configs, trained for
the documented 10 000 steps on the bundled YSD1 lag-5 table through the config driver, must land on the documented
held-out perplexity / accuracy and on the documented fitted h (which the reference quotes to three digits).  The
values depend on the random initialisation only weakly except for the CNN AR model, whose tolerance is wider."""
import configparser
import json
import os

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

DOCS = {  # config: (kind, which, perplexity, accuracy %, h, perplexity tol, accuracy tol)   docs/usage.rst:258-264
    "bear_lin_ar": ("net", "AR", 3.99, 32.9, None, 0.01, 0.15),
    "bear_cnn_ar": ("net", "AR", 3.85, 35.8, None, 0.04, 1.5),
    "bear_stop_ar": ("ref", "AR", 3.84, 36.5, None, 0.01, 0.15),
    "bear_lin_bear": ("net", "BEAR", 3.79, 36.8, 0.0433, 0.006, 0.1),
    "bear_cnn_bear": ("net", "BEAR", 3.79, 36.8, 0.0119, 0.006, 0.1),
    "bear_stop_bear": ("ref", "BEAR", 3.79, 36.8, 0.0142, 0.006, 0.1),
}


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
    bmm = json.loads(r["heldout_perplex_BMM"])
    assert all(abs(v - 3.79) < 0.006 for v in bmm)                                  # docs/usage.rst:261
    if h is not None:
        # the CNN's fitted h moves with the summation order of its gradient atomics (0.0120 ... 0.0132 over runs of this build,
        # chaotic over 10 000 Adam steps; the docs quote one run of the reference); the others are stable to three digits
        assert abs(float(r["h"]) - h) / h < (0.25 if "cnn" in name else 0.015)
