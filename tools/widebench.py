#!/usr/bin/env python
"""times the wide-design (generic, scratch-resident) kernels next to a register-resident width"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (loads the HIP runtime torch bundles before the engine library)
from deseq2_amd import core, simulate
from deseq2_amd.engine import DeviceEngine
E = DeviceEngine("cuda:0")
def paired(patients):
    """~ patient + treatment: p = patients + 1, 2 * patients cells (the general per-sample kernels)"""
    mm = 2 * patients
    pat = np.repeat(np.arange(patients), 2)
    return np.column_stack([np.ones(mm)] + [(pat == k).astype(float) for k in range(1, patients)] + [np.tile([0.0, 1.0], patients)])


def main(cases):
    for levels, n, m, xx in cases:
        run(levels, n, m, xx)


CASES = [(10, 20000, 200, None), (12, 20000, 200, None), (16, 20000, 200, None), (20, 20000, 200, None), (24, 20000, 240, None),
         (32, 20000, 256, None), (40, 20000, 240, None), (48, 20000, 288, None),          # round 5: the 32- and 48-column builds
         (31, 20000, 60, paired(30)), (46, 20000, 90, paired(45)),
         (56, 20000, 110, paired(55)), (64, 20000, 256, None)]                              # round 6: the 64-column build
def run(levels, n, m, xx):
    x = simulate.design_factor(m, levels) if xx is None else xx
    d = simulate.make_counts(n, x, seed=3)
    dds = core.DESeqDataSet(d["counts"], x, engine=E)
    E.record = []
    core.DESeq(dds, minReplicatesForReplace=np.inf)
    rec, E.record = E.record, None
    big = {}
    for name, g, ms in rec:
        if g > n // 2:
            big.setdefault(name, []).append(ms)
    print("p=%2d%s n=%d m=%d: " % (levels, " (paired)" if xx is not None else "", dds.n, m) + "  ".join("%s %.1f ms" % (k, np.mean(v)) for k, v in big.items()), flush=True)


if __name__ == "__main__":
    sel = sys.argv[1:]
    main([c for c in CASES if not sel or str(c[0]) in sel])
