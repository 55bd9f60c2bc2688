"""-m gpu: the whole DESeq() chain (host mirror of the R callers + the three HIP routines).
  * HostEngine(native) vs HostEngine(oracle): identical host glue, so every column of the
    result must be IDENTICAL (flags, iteration counts, dispersions, beta, SE, Wald, p).
  * DeviceEngine (HBM-resident, gene-major, torch glue for the O(n m) pre-steps) vs the
    oracle chain: the start values differ in the last bits (torch vs numpy QR), so the bar
    is north_star's: values within 1e-6 relative, flags equal."""
import os
import numpy as np
import pytest

from deseq2_amd import core, simulate
from deseq2_amd.engine import DeviceEngine, HostEngine
from tests.helpers import assert_same

pytestmark = pytest.mark.gpu

COLS = ["dispGeneEst", "dispGeneIter", "dispFit", "dispMAP", "dispIter", "dispOutlier", "dispersion", "beta",
        "betaSE", "WaldStatistic", "WaldPvalue", "betaConv", "betaIter", "deviance"]


def _run(engine, d, x, weights=None):
    dds = core.DESeqDataSet(d["counts"], x, sizeFactors=d["size_factors"], weights=weights, engine=engine)
    return core.DESeq(dds)


@pytest.mark.parametrize("n,m,design", [(1000, 6, "two"), (400, 60, "bc")])
def test_chain_host_engine_identical_to_oracle(oracle, n, m, design):
    x = simulate.design_two_group(m) if design == "two" else simulate.design_batch_condition(m)
    d = simulate.make_counts(n, x, seed=21)
    a = _run(HostEngine(), d, x)
    b = _run(HostEngine(oracle), d, x)
    for k in COLS:
        assert_same(a.mcols[k], b.mcols[k], "DESeq()$" + k)
    assert a.dispersionFunction["coefficients"][0] == b.dispersionFunction["coefficients"][0]


def test_chain_with_weights_identical_to_oracle(oracle):
    m = 40
    x = simulate.design_two_group(m)
    d = simulate.make_counts(300, x, seed=22)
    rng = np.random.default_rng(3)
    w = rng.uniform(0.05, 1.0, d["counts"].shape)
    w[rng.uniform(size=w.shape) < 0.02] = 0.0
    a = _run(HostEngine(), d, x, weights=w)
    b = _run(HostEngine(oracle), d, x, weights=w)
    for k in COLS:
        assert_same(a.mcols[k], b.mcols[k], "DESeq(weights)$" + k)


def test_chain_device_resident_matches_oracle(oracle):
    """HBM-resident chain: every O(n m) step is now a HIP kernel with the oracle's arithmetic, so
    the whole result is identical -- only the Wald p-values go through a different erfc."""
    m = 100
    x = simulate.design_batch_condition(m)
    d = simulate.make_counts(600, x, seed=23)
    a = _run(DeviceEngine("cuda:0"), d, x)
    b = _run(HostEngine(oracle), d, x)
    for k in COLS:
        if k == "WaldPvalue":
            np.testing.assert_allclose(a.mcols[k], b.mcols[k], rtol=1e-10, atol=1e-300)
        else:
            assert_same(a.mcols[k], b.mcols[k], "device DESeq()$" + k)


def test_chain_beta_prior_weights_identical_to_oracle(oracle):
    """BASELINE configs[4] shape family: weights + betaPrior on the expanded design (p = 3,
    rank deficient, lambda = 1/sigma^2) -- every column identical to the oracle chain."""
    factors = {"condition": np.repeat([0, 1], 20)}
    x, _ = core.standard_model_matrix(factors)
    d = simulate.make_counts(300, x, seed=24)
    w = np.random.default_rng(4).uniform(0.05, 1.0, d["counts"].shape)
    w[np.random.default_rng(5).uniform(size=w.shape) < 0.02] = 0.0
    res = []
    for eng in (HostEngine(), HostEngine(oracle)):
        dds = core.DESeqDataSet(d["counts"], x, sizeFactors=d["size_factors"], weights=w, engine=eng)
        core.estimateDispersions(dds)
        core.nbinomWaldTest(dds, betaPrior=True, factors=factors)
        res.append(dds)
    for k in ("dispersion", "beta", "betaSE", "WaldStatistic", "betaIter", "betaConv", "MLE_beta"):
        assert_same(res[0].mcols[k], res[1].mcols[k], "betaPrior DESeq()$" + k)
    assert_same(res[0].attrs["betaPriorVar"], res[1].attrs["betaPriorVar"], "betaPriorVar")


def test_chain_lrt_identical_to_oracle(oracle):
    """BASELINE configs[3] family: nbinomLRT, full (6-level factor) vs a 2-column reduced model"""
    m = 36
    x = simulate.design_factor(m, 6)
    d = simulate.make_counts(250, x, seed=25)
    red = np.column_stack([np.ones(m), (np.arange(m) >= m // 2).astype(float)])
    res = []
    for eng in (HostEngine(), HostEngine(oracle)):
        dds = core.DESeqDataSet(d["counts"], x, sizeFactors=d["size_factors"], engine=eng)
        core.DESeq(dds, test="LRT", reduced=red)
        res.append(dds)
    for k in ("dispersion", "beta", "betaSE", "LRTStatistic", "LRTPvalue", "betaIter"):
        assert_same(res[0].mcols[k], res[1].mcols[k], "LRT DESeq()$" + k)


@pytest.mark.parametrize("k", [2, 3])
def test_pipelined_chunks_equal_serial(k):
    """DESeq() as k chunk threads on k HIP streams (parallel.Pipeline) == the serial device chain: per-gene
    fits are independent and the all-gene steps see the same vectors in the same order."""
    from deseq2_amd import parallel
    m = 60
    x = simulate.design_batch_condition(m)
    d = simulate.make_counts(900, x, seed=41)
    c = d["counts"]
    c[::45, 2] = 120000                      # some outliers: the per-chunk refit path runs too
    E = DeviceEngine("cuda:0")
    serial = core.DESeq(core.DESeqDataSet(c, x, sizeFactors=d["size_factors"], engine=E))
    pipe = parallel.Pipeline(E, n_chunks=k)
    for _ in range(2):                       # streams / workspaces are reused across calls
        shards = pipe.run(lambda lo, hi: core.DESeqDataSet(c[lo:hi], x, sizeFactors=d["size_factors"], engine=E),
                          c.shape[0])
    got = parallel.concat_mcols(shards, COLS + ["maxCooks", "replace"])
    for kk in COLS + ["maxCooks", "replace"]:
        assert_same(got[kk], serial.mcols[kk], "pipelined DESeq()$" + kk)
    assert serial.mcols["replace"].sum() >= 10
    assert shards[0].dispersionFunction["coefficients"][0] == serial.dispersionFunction["coefficients"][0]


def test_device_chain_weights_beta_prior_matches_oracle(oracle):
    """BASELINE configs[4] family on the HBM-resident engine: observation weights (some zero) + betaPrior on
    the expanded design -- every column identical to the oracle chain."""
    factors = {"condition": np.repeat([0, 1], 25)}
    x, _ = core.standard_model_matrix(factors)
    d = simulate.make_counts(400, x, seed=26)
    w = np.random.default_rng(6).uniform(0.05, 1.0, d["counts"].shape)
    w[np.random.default_rng(7).uniform(size=w.shape) < 0.02] = 0.0
    res = []
    for eng in (DeviceEngine("cuda:0"), HostEngine(oracle)):
        dds = core.DESeqDataSet(d["counts"], x, sizeFactors=d["size_factors"], weights=w, engine=eng)
        core.estimateDispersions(dds)
        core.nbinomWaldTest(dds, betaPrior=True, factors=factors)
        res.append(dds)
    for k in ("dispGeneEst", "dispersion", "beta", "betaSE", "WaldStatistic", "betaIter", "betaConv", "MLE_beta",
              "maxCooks", "deviance"):
        assert_same(res[0].mcols[k], res[1].mcols[k], "device betaPrior+weights DESeq()$" + k)
    assert_same(res[0].attrs["betaPriorVar"], res[1].attrs["betaPriorVar"], "betaPriorVar")


def test_device_chain_lrt_ten_levels_matches_oracle(oracle):
    """BASELINE configs[3] family: 10-level factor (p = 10), nbinomLRT full vs intercept-only reduced"""
    m = 120
    x = simulate.design_factor(m, 10)
    d = simulate.make_counts(250, x, seed=27)
    a = core.DESeq(core.DESeqDataSet(d["counts"], x, engine=DeviceEngine("cuda:0")), test="LRT",
                   reduced=np.ones((m, 1)))
    b = core.DESeq(core.DESeqDataSet(d["counts"], x, engine=HostEngine(oracle)), test="LRT", reduced=np.ones((m, 1)))
    for k in ("dispGeneEst", "dispersion", "beta", "betaSE", "LRTStatistic", "LRTPvalue", "fullBetaConv", "betaIter",
              "deviance", "maxCooks"):
        assert_same(a.mcols[k], b.mcols[k], "device LRT DESeq()$" + k)


def test_full_size_properties(oracle):
    """BASELINE configs[2] at FULL size (50 000 genes x 500 samples, p = 4) on the device engine, through
    size-independent properties: (1) gene-permutation equivariance of every per-gene column of DESeq()
    (per-gene fits are independent; the all-gene steps -- trend fit, prior variance -- are order-dependent
    only through summation order, so they are compared at 1e-9 and the per-gene columns under a FIXED
    dispersion function bit for bit); (2) a random sample of genes equals the oracle on the same rows."""
    import torch
    m, n = 500, 50000
    x = simulate.design_batch_condition(m)
    d = simulate.make_counts(n, x, seed=51)
    c = d["counts"]
    n = c.shape[0]
    E = DeviceEngine("cuda:0")
    perm = np.random.default_rng(8).permutation(n)

    def gene_est(counts):
        dds = core.DESeqDataSet(counts, x, sizeFactors=d["size_factors"], engine=E)
        core.estimateDispersionsGeneEst(dds)
        return dds
    a, b = gene_est(c), gene_est(c[perm])
    for k in ("baseMean", "baseVar", "dispGeneEst", "dispGeneIter"):
        assert_same(b.mcols[k], a.mcols[k][perm], "permuted " + k)
    # all-gene steps: same trend up to summation order
    core.estimateDispersionsFit(a); core.estimateDispersionsFit(b)
    np.testing.assert_allclose(b.dispersionFunction["coefficients"], a.dispersionFunction["coefficients"], rtol=1e-9)
    # per-gene steps under the SAME dispersion function and prior: bit for bit
    b.dispersionFunction = dict(a.dispersionFunction)
    b.mcols["dispFit"] = a.mcols["dispFit"][perm]
    pv = core.estimateDispersionsPriorVar(a)
    for dds in (a, b):
        core.estimateDispersionsMAP(dds, dispPriorVar=pv)
        core.nbinomWaldTest(dds)
    for k in ("dispMAP", "dispIter", "dispersion", "beta", "betaSE", "WaldStatistic", "betaIter", "deviance",
              "maxCooks"):
        assert_same(b.mcols[k], a.mcols[k][perm], "permuted " + k)
    assert a.mcols["betaConv"].mean() > 0.999
    # a random sample of rows against the oracle (gene-wise steps and the final fit under a's dispersions)
    rows = np.sort(np.random.default_rng(9).choice(n, 192, replace=False))
    o = core.DESeqDataSet(c[rows], x, sizeFactors=d["size_factors"], engine=HostEngine(oracle))
    core.estimateDispersionsGeneEst(o)
    for k in ("baseMean", "dispGeneEst", "dispGeneIter"):
        assert_same(o.mcols[k], a.mcols[k][rows], "sample vs oracle " + k)
    o.dispersionFunction = dict(a.dispersionFunction)
    o.mcols["dispFit"] = a.mcols["dispFit"][rows]
    core.estimateDispersionsMAP(o, dispPriorVar=pv)
    core.nbinomWaldTest(o)
    for k in ("dispMAP", "dispIter", "dispersion", "beta", "betaSE", "WaldStatistic", "betaIter", "deviance", "maxCooks"):
        assert_same(o.mcols[k], a.mcols[k][rows], "sample vs oracle " + k)
    del a, b
    torch.cuda.empty_cache()


def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus 2` outside torchrun starts two ranks itself (both on cuda:0 here, n-vector exchange
    over gloo), reports n_gpus = 2, strong scaling over the config's genes in total plus the weak-scaling figure,
    and the sharded result equals the serial one (tests/testthat/test_parallel.R:27-37)"""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DSQ_BENCH_ONE_DEVICE="1")
    common = ["--steps", "1", "--warmup", "0", "--genes", "1500", "--no-cpu-baseline", "--no-hostpath"]
    r2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"] + common, env=env,
                        capture_output=True, text=True, timeout=600)
    assert r2.returncode == 0, r2.stderr[-2000:]
    j2 = json.loads(r2.stdout.strip().splitlines()[-1])
    r1 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1"] + common, env=env,
                        capture_output=True, text=True, timeout=600)
    assert r1.returncode == 0, r1.stderr[-2000:]
    j1 = json.loads(r1.stdout.strip().splitlines()[-1])
    assert j2["n_gpus"] == 2 and j1["n_gpus"] == 1
    assert j2["scaling"] == "strong" and j2["config"]["genes_total"] == j1["config"]["genes_total"]
    assert j2["weak"]["genes_total"] > j2["config"]["genes_total"]
    assert j2["config"]["genes_this_gpu"] * 2 - j2["config"]["genes_total"] in (0, 1)
    assert j2["result_digest"] == j1["result_digest"]          # sharded == serial, every per-gene result column


def test_bench_eight_ranks_on_one_device_equal_serial():
    """the N = 8 path of bench.py (what the driver launches for SCALE): eight ranks (all on cuda:0 here, the n-vector
    exchanges over gloo), shards cut as R/parallel.R:10 -- the digest over every per-gene result column equals the serial
    run's (tests/testthat/test_parallel.R:27-37 at 8 workers), and the parity block reports the oracle's bits"""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DSQ_BENCH_ONE_DEVICE="1")
    common = ["--steps", "1", "--warmup", "0", "--genes", "4100", "--no-cpu-baseline", "--no-hostpath", "--no-weak"]
    r8 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8"] + common, env=env,
                        capture_output=True, text=True, timeout=900)
    assert r8.returncode == 0, r8.stderr[-2000:]
    j8 = json.loads(r8.stdout.strip().splitlines()[-1])
    r1 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1"] + common, env=env,
                        capture_output=True, text=True, timeout=600)
    assert r1.returncode == 0, r1.stderr[-2000:]
    j1 = json.loads(r1.stdout.strip().splitlines()[-1])
    assert j8["n_gpus"] == 8 and j8["config"]["genes_total"] == j1["config"]["genes_total"]
    assert abs(j8["config"]["genes_this_gpu"] * 8 - j8["config"]["genes_total"]) < 8
    assert j8["config"]["chain"].startswith("fused")
    assert j8["result_digest"] == j1["result_digest"]
    for j in (j1, j8):
        assert j["parity"]["iter_equal"] == 1.0 and j["parity"]["max_rel"] == 0.0, j["parity"]


def test_global_refit_count_over_rccl_single_rank():
    """fused._global_refit_count with a DEVICE communicator (RCCL, backend "nccl"): the device all-reduce of the N_REFIT
    counter, its one-time check against the host exchange and the ranks' consensus -- exercised on a 1-rank group (the
    box has one GPU; the N > 1 path itself runs with ranks sharing the device over gloo, test above)"""
    import os
    import socket
    import subprocess
    import sys
    code = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["REPO"])
from deseq2_amd import fused, _lib as L
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
try:
    dist.init_process_group(backend="nccl", world_size=1, rank=0, device_id=dev)
    probe = torch.ones(1, device=dev); dist.all_reduce(probe); torch.cuda.synchronize()
except Exception as e:
    print("SKIP", repr(e)); sys.exit(0)
class Stub:
    pass
run = Stub(); run.E = Stub(); run.E.device = dev
run.status = torch.zeros(L.DSQ_ST_COUNT, dtype=torch.int32, device=dev)
run.status[L.DSQ_ST["N_REFIT"]] = 37
run.read_status = lambda: ({k: int(run.status[i].item()) for k, i in L.DSQ_ST.items()}, None)
for rep in range(2):
    tot = fused._global_refit_count(run, dev, torch)
    assert tot.dtype == torch.int32 and tot.device.type == "cuda" and int(tot.item()) == 37, tot
assert fused._DEVICE_REDUCE_OK is True
# the other two exchanges of the N > 1 chain on the same communicator: shard sizes, the trend's two n-vectors
from deseq2_amd import parallel
assert parallel.allgather_sizes(123, dev) == [123]
a = torch.arange(5, dtype=torch.float64, device=dev); b = a * 2
ga, gb = parallel.allgather_device_pairs(a, b, 8, dev, torch)
assert ga.numel() == 8 and torch.equal(ga[:5], a) and torch.isnan(ga[5:]).all() and torch.equal(gb[:5], b)
print("OK device route")
dist.destroy_process_group()
'''
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, REPO=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    try:
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=150)
    except subprocess.TimeoutExpired:
        pytest.skip("the 1-rank RCCL group did not come up within 150 s on this box")
    assert r.returncode == 0, r.stdout + r.stderr
    if "SKIP" in r.stdout:
        pytest.skip("RCCL did not come up on this box: " + r.stdout.strip())
    assert "OK device route" in r.stdout, r.stdout + r.stderr


def _fused_rank(rank, world, port, outdir, betaPrior):
    import os
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from deseq2_amd import core, fused, parallel, simulate
    from deseq2_amd.engine import DeviceEngine
    torch.cuda.set_device(0)                                       # (the test box has one GPU: the ranks share it)
    E = DeviceEngine("cuda:0")
    m = 48
    x = simulate.design_factor(m, 3)                               # cells of 16: outlier replacement + refit on every shard
    factors = {"group": (np.arange(m) * 3) // m}
    d = simulate.make_counts(900, x, seed=41)
    counts = d["counts"].copy()
    rng = np.random.default_rng(2)
    for r in rng.choice(counts.shape[0], 8, replace=False):
        counts[r, rng.integers(m)] = int(counts[r].max() * 40 + 1000)
    idx = parallel.shard_ranges(counts.shape[0], world)[rank]
    dds = core.DESeqDataSet(counts[idx], x, sizeFactors=d["size_factors"], engine=E)
    kw = dict(betaPrior=True, factors=factors) if betaPrior else {}
    assert fused.supported(dds, **kw)
    fused.DESeq(dds, comm_device=None, **kw)
    assert dds.attrs.get("fused")
    cols = ["dispGeneEst", "dispersion", "beta", "betaSE", "WaldStatistic", "WaldPvalue", "maxCooks", "replace"] + (["MLE_beta"] if betaPrior else [])
    np.savez(os.path.join(outdir, "fr%d.npz" % rank), idx=idx, n_refit=dds.attrs["status"]["N_REFIT"],
             bpv=np.asarray(dds.attrs.get("betaPriorVar", [0.0])), **{k: np.asarray(dds.mcols[k], np.float64) for k in cols})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("betaPrior", [False, True])
def test_fused_chain_on_two_gene_shards_equals_the_serial_chain(tmp_path, betaPrior):
    """the MULTI-RANK branch of the fused device chain itself (deseq2_amd/fused.py: the trend exchange, defer_finish, the
    global count of refitted rows; with betaPrior the MLE exchange of R/parallel.R:34-40) as two processes over gloo --
    against the one-process chain, every column bit for bit (tests/testthat/test_parallel.R:27-37)"""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_fused_rank, args=(2, port, str(tmp_path), betaPrior), nprocs=2, join=True)
    m = 48
    x = simulate.design_factor(m, 3)
    factors = {"group": (np.arange(m) * 3) // m}
    d = simulate.make_counts(900, x, seed=41)
    counts = d["counts"].copy()
    rng = np.random.default_rng(2)
    for r in rng.choice(counts.shape[0], 8, replace=False):
        counts[r, rng.integers(m)] = int(counts[r].max() * 40 + 1000)
    serial = core.DESeqDataSet(counts, x, sizeFactors=d["size_factors"], engine=DeviceEngine("cuda:0"))
    from deseq2_amd import fused
    fused.DESeq(serial, **(dict(betaPrior=True, factors=factors) if betaPrior else {}))
    parts = [np.load(os.path.join(str(tmp_path), "fr%d.npz" % r)) for r in range(2)]
    assert sum(int(p["n_refit"]) for p in parts) == serial.attrs["status"]["N_REFIT"] >= 2
    for k in parts[0].files:
        if k in ("idx", "n_refit", "bpv"):
            continue
        got = np.concatenate([p[k] for p in parts])
        assert_same(got, np.asarray(serial.mcols[k], np.float64), "2 shards vs serial: " + k)
    if betaPrior:
        assert_same(parts[0]["bpv"], np.asarray(serial.attrs["betaPriorVar"]), "betaPriorVar")
        assert_same(parts[1]["bpv"], parts[0]["bpv"], "betaPriorVar, rank 1")


def test_bench_two_ranks_over_rccl_when_two_devices_are_visible():
    """one process per GPU over RCCL (what the driver's multi-GPU bench launches): only where the box has >= 2 devices"""
    import json, os, subprocess, sys
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one visible device: the N > 1 path runs with ranks sharing it (tests above)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--steps", "2", "--warmup", "1", "--genes", "3000", "--no-cpu-baseline", "--no-hostpath", "--no-weak"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("DSQ_BENCH_ONE_DEVICE", None)
    r2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"] + common, env=env, capture_output=True, text=True, timeout=600)
    assert r2.returncode == 0, r2.stderr[-2000:]
    assert "RCCL unavailable" not in r2.stderr, r2.stderr[-2000:]
    r1 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1"] + common, env=env, capture_output=True, text=True, timeout=600)
    assert r1.returncode == 0, r1.stderr[-2000:]
    j2, j1 = json.loads(r2.stdout.strip().splitlines()[-1]), json.loads(r1.stdout.strip().splitlines()[-1])
    assert j2["n_gpus"] == 2 and j2["result_digest"] == j1["result_digest"]
