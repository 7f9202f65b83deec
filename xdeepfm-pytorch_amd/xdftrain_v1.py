#!/usr/bin/env python3
"""Stand-in for the reference's xdftrain_v1.py on the MI355X path: same command line (`--test_size`, `--val_size`,
`--patience`; a held-out test split reported at the end, early stopping on val_auc always on), implemented in
xdftrain_amd.py with `--model xdeepfm` and that script's defaults preselected."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import xdftrain_amd  # noqa: E402

if __name__ == "__main__":
    xdftrain_amd.main(model="xdeepfm", script="v1")
