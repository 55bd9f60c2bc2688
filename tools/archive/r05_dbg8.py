import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from tests.test_gpu_deseq_host import CASES, _host_entry
counts, x, sf, kw = CASES["bc_outliers"]
res = {}
for sh in (0, 2, 3, 4, 5, 8):
    if sh: os.environ["DSQ_HOST_SHARDS"] = str(sh)
    else: os.environ.pop("DSQ_HOST_SHARDS", None)
    res[sh] = _host_entry(counts, x, sf, kw, assays=("mu", "cooks"))
one = res[0]
n = counts.shape[0]
for sh in (2, 3, 4, 5, 8):
    r = res[sh]
    bad = {}
    for k in sorted(one):
        if isinstance(one[k], np.ndarray):
            a, b = np.asarray(r[k], float), np.asarray(one[k], float)
            ne = ~((a == b) | (np.isnan(a) & np.isnan(b)))
            if ne.any():
                rows = np.unique(np.argwhere(ne)[:, 0])
                bad[k] = rows.tolist()[:20]
    print("shards", sh, "differences:", bad)
    if bad:
        rows = sorted(set(sum(bad.values(), [])))
        print("   replace flags of those rows (1 range):", np.asarray(one["replace"])[rows][:20], "range size", n / sh)
