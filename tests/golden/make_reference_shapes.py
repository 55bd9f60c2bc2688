#!/usr/bin/env python
"""Generate tests/golden/reference_shapes.npz: outputs of the REFERENCE'S OWN src/DESeq2.cpp (compiled by
oracle/Makefile `ref` against oracle/shim/, special functions in binary128) at the SHAPES of BASELINE.json
configs C2..C5 (tests/test_oracle_vs_reference.py::shape_cases()).  Run in the development container
(needs /root/reference; about five minutes of CPU):   python tests/golden/make_reference_shapes.py
Inputs are regenerated from seeds, only outputs are stored; of the n x m hat matrix the first HEAD genes."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import reference as R                                  # noqa: E402
from tests.test_oracle_vs_reference import shape_cases, run_all, HEAD    # noqa: E402

out = {}
for name, case in shape_cases().items():
    t0 = time.time()
    res = run_all(R, case)
    for fn, d in res.items():
        for k, v in d.items():
            v = np.asarray(v)
            if v.ndim == 2 and v.shape[1] == case["counts"].shape[1]:   # n x m: keep the head rows
                v = v[:HEAD]
            out["%s/%s/%s" % (name, fn, k)] = v
    print(name, case["counts"].shape, "%.0f s" % (time.time() - t0), flush=True)
path = os.path.join(ROOT, "tests", "golden", "reference_shapes.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")
