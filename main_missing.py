#!/usr/bin/env python3
"""Entry point with the reference's name and behaviour (`src/main_missing.py`): reads `config.yaml` (or the file named
first on the command line, then key=value overrides), trains or evaluates on the MI355X hot path.

    python main_missing.py                       # config.yaml in the working directory, as the reference
    python main_missing.py cfg.yaml epochs=2 data_source=synthetic ckpt_root=/tmp/ckpt
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import mrdis  # noqa: E402

if __name__ == '__main__':
    mrdis.train.main()
