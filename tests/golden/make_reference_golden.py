#!/usr/bin/env python
"""Generate tests/golden/reference_golden.npz from the REFERENCE'S OWN src/DESeq2.cpp, compiled by
oracle/Makefile (target `ref`) against the stand-in headers of oracle/shim/.  Run in the development
container (needs /root/reference):   python tests/golden/make_reference_golden.py
The inputs are regenerated from seeds by tests/test_oracle_vs_reference.py::golden_cases(), so only the
reference's OUTPUTS are stored."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import reference as R                                  # noqa: E402
from tests.test_oracle_vs_reference import golden_cases, run_all   # noqa: E402

out = {}
for name, case in golden_cases().items():
    res = run_all(R, case)
    for fn, d in res.items():
        for k, v in d.items():
            out["%s/%s/%s" % (name, fn, k)] = np.asarray(v)
path = os.path.join(ROOT, "tests", "golden", "reference_golden.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")
