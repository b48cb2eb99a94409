"""Train and evaluate reference-based AR or BEAR models on MI355X (mirror of
bear_model/models/train_bear_ref.py).  Usage:  python train_bear_ref.py config.cfg
Example configs: bear_amd/models/config_files/."""
import argparse
import configparser
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bear_amd.models import _driver


def main(config):
    return _driver.main(config, "ref")


if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument("configPath")
    args = parser.parse_args()
    config = configparser.ConfigParser()
    config.read(args.configPath)
    main(config)
