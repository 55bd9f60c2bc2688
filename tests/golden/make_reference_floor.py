#!/usr/bin/env python
"""Generate tests/golden/reference_floor.npz: what R's callers see of the dispersion-floor regime
(tests/floor_regime.py) from the REFERENCE'S OWN src/DESeq2.cpp -- the binary128 build (`ref`) and the libm-double
build (`ref_fast`) of oracle/Makefile, so that the reference's self-disagreement there is on record next to the
engine's.  Run in the development container (needs /root/reference):  python tests/golden/make_reference_floor.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import reference as R                                  # noqa: E402
from tests.floor_regime import SEEDS, floor_case, visible_chain    # noqa: E402

out = {}
for seed in SEEDS:
    d = floor_case(seed)
    for which, fast in (("ref", False), ("ref_fast", True)):
        R.use_fast(fast)
        for k, v in visible_chain(R, d).items():
            out["%s/seed%d/%s" % (which, seed, k)] = np.asarray(v)
    R.use_fast(False)
path = os.path.join(ROOT, "tests", "golden", "reference_floor.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")
