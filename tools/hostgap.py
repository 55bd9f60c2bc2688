import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import bench
from deseq2_amd import core, fused, simulate
from deseq2_amd.engine import DeviceEngine
cfg = dict(bench.CONFIGS["C3"]); m = cfg["samples"]; x = bench.make_design(cfg["design"], m)
d = simulate.make_counts(int(sys.argv[1]) if len(sys.argv) > 1 else cfg["genes"], x, seed=1); counts = d["counts"]; n = counts.shape[0]
dev = torch.device("cuda", 0); E = DeviceEngine(dev)
counts_r = torch.as_tensor(np.ascontiguousarray(counts.T), device=dev)
nf_r = torch.ones((m, n), dtype=torch.float64, device=dev)
T = {}
def mark(k, t0): T[k] = T.get(k, 0.0) + (time.perf_counter() - t0)
orig_init = fused._Run.__init__; orig_launch = fused._Run.launch; orig_read = fused._Run.read_all; orig_sup = fused.supported
def w(name, f):
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); mark(name, t0); return r
    return g
fused._design_facts = w("design_facts", fused._design_facts)
E._design_qr_dev = w("design_qr_dev", E._design_qr_dev); E.design = w("design", E.design)
fused._Run.start_read = w("start_read", fused._Run.start_read)
fused._Run.__init__ = w("run_init", orig_init); fused._Run.launch = w("launch", orig_launch); fused._Run.read_all = w("read_all(sync)", orig_read); fused.supported = w("supported", orig_sup)
core.DESeqDataSet.from_device = classmethod(w("from_device", core.DESeqDataSet.from_device.__func__))
def step():
    dds = core.DESeqDataSet.from_device(E, counts_r, nf_r, x, sizeFactors=np.ones(m))
    fused.DESeq(dds); return dds
for _ in range(3): step()
T.clear(); torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): step()
torch.cuda.synchronize(); tot = (time.perf_counter() - t0) / 20 * 1e3
print("ms/step %.3f" % tot, {k: round(v / 20 * 1e3, 3) for k, v in T.items()})
