"""Runs the six example configs the reference documents (docs/usage.rst:236-265: 10 000 steps on the bundled YSD1 lag-5
table) and prints held-out perplexity / accuracy and h next to the documented values."""
import configparser, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bear_amd.models import _driver

DOCS = {  # docs/usage.rst:258-264: (perplexity, accuracy %, h)
    "bear_lin_ar": ("AR", 3.99, 32.9, None), "bear_cnn_ar": ("AR", 3.85, 35.8, None), "bear_stop_ar": ("AR", 3.84, 36.5, None),
    "bear_lin_bear": ("BEAR", 3.79, 36.8, 0.0433), "bear_cnn_bear": ("BEAR", 3.79, 36.8, 0.0119), "bear_stop_bear": ("BEAR", 3.79, 36.8, 0.0142),
}
steps = sys.argv[1] if len(sys.argv) > 1 else "10000"
rows = []
for name, (which, perp, acc, h) in DOCS.items():
    config = configparser.ConfigParser()
    config.read(os.path.join(ROOT, "bear_amd", "models", "config_files", name + ".cfg"))
    config["train"]["epochs"] = steps            # the reference configs: epochs = 10000, one 1365-row batch per epoch
    config["train"]["batch_size"] = "1500"
    t0 = time.time()
    _driver.main(config, "ref" if "stop" in name else "net")
    r = config["results"]
    rows.append({"config": name, "seconds": round(time.time() - t0, 1),
                 "perplexity": float(r["heldout_perplex_" + which]), "docs_perplexity": perp,
                 "accuracy_pct": 100 * float(r["heldout_accuracy_" + which]), "docs_accuracy_pct": acc,
                 "h": float(r["h"]), "docs_h": h,
                 "perplexity_BMM": json.loads(r["heldout_perplex_BMM"]), "docs_perplexity_BMM": 3.79})
    print(json.dumps(rows[-1]), flush=True)
