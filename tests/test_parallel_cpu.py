"""CPU: the N > 1 path (gene shards, one process each, all-gather of the two n-vectors the
global dispersion trend needs) over gloo with world_size 2, run on the CPU oracle engine.
Mirrors tests/testthat/test_parallel.R:2-37: sharded == serial for every column."""
import os
import socket

import numpy as np
import pytest

from deseq2_amd import parallel, simulate

COLS = ["dispGeneEst", "dispFit", "dispMAP", "dispersion", "dispIter", "beta", "betaSE", "WaldStatistic",
        "WaldPvalue", "betaIter"]


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n, m, seed, outdir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from deseq2_amd import core
    from deseq2_amd.engine import HostEngine
    from oracle import oracle as O
    x = simulate.design_two_group(m)
    d = simulate.make_counts(n, x, seed=seed)
    idx = parallel.shard_ranges(d["counts"].shape[0], world)[rank]
    dds = core.DESeqDataSet(d["counts"][idx], x, sizeFactors=d["size_factors"], engine=HostEngine(O))
    parallel.DESeqParallel(dds)
    np.savez(os.path.join(outdir, "shard%d.npz" % rank), idx=idx,
             prior=dds.dispersionFunction["dispPriorVar"], **{k: dds.mcols[k] for k in COLS})
    dist.barrier()
    dist.destroy_process_group()


def test_shard_ranges_match_reference_rule():
    r = parallel.shard_ranges(10, 4)         # sort(rep(1:4, length.out = 10)) -> 3,3,2,2
    assert [len(a) for a in r] == [3, 3, 2, 2]
    assert np.concatenate(r).tolist() == list(range(10))


@pytest.mark.timeout(300)
def test_two_shards_equal_serial(tmp_path, oracle):
    import torch.multiprocessing as mp
    n, m, seed, world = 400, 12, 31, 2
    mp.spawn(_worker, args=(world, _free_port(), n, m, seed, str(tmp_path)), nprocs=world, join=True)
    from deseq2_amd import core
    from deseq2_amd.engine import HostEngine
    x = simulate.design_two_group(m)
    d = simulate.make_counts(n, x, seed=seed)
    serial = core.DESeq(core.DESeqDataSet(d["counts"], x, sizeFactors=d["size_factors"], engine=HostEngine(oracle)))
    parts = [np.load(os.path.join(str(tmp_path), "shard%d.npz" % r)) for r in range(world)]
    assert np.concatenate([p["idx"] for p in parts]).tolist() == list(range(d["counts"].shape[0]))
    for k in COLS:
        got = np.concatenate([p[k] for p in parts])
        np.testing.assert_array_equal(got, serial.mcols[k], err_msg=k)
    assert float(parts[0]["prior"]) == float(parts[1]["prior"]) == serial.dispersionFunction["dispPriorVar"]


def _run_chunks(dd, x, k, comm_device=None, O=None):
    """k chunk threads of one process over a LocalGroup (what parallel.Pipeline does on HIP streams)"""
    import threading
    from deseq2_amd import core
    from deseq2_amd.engine import HostEngine
    counts = dd["counts"]
    ranges = parallel.shard_ranges(counts.shape[0], k)
    group = parallel.LocalGroup(k, comm_device)
    out, errs = [None] * k, []

    def work(c):
        try:
            dds = core.DESeqDataSet(counts[ranges[c]], x, sizeFactors=dd["size_factors"], engine=HostEngine(O))
            out[c] = parallel.DESeqParallel(dds, group=group, chunk=c)
        except BaseException as e:      # noqa: BLE001
            errs.append(e)
            group.barrier.abort()
    th = [threading.Thread(target=work, args=(c,)) for c in range(k)]
    [t.start() for t in th]
    [t.join() for t in th]
    if errs:
        raise errs[0]
    return out


def test_chunk_threads_equal_serial(oracle):
    """in-process chunks (LocalGroup): 3 threads over contiguous gene ranges == serial, with outlier refits"""
    from deseq2_amd import core
    from deseq2_amd.engine import HostEngine
    x = simulate.design_two_group(16)
    d = simulate.make_counts(300, x, seed=33)
    d["counts"][::40, 1] = 90000
    serial = core.DESeq(core.DESeqDataSet(d["counts"], x, sizeFactors=d["size_factors"], engine=HostEngine(oracle)))
    shards = _run_chunks(d, x, 3, O=oracle)
    got = parallel.concat_mcols(shards, COLS + ["maxCooks", "replace"])
    for k in COLS + ["maxCooks", "replace"]:
        np.testing.assert_array_equal(got[k], serial.mcols[k], err_msg=k)
    assert serial.mcols["replace"].sum() >= 5


def _worker_chunks(rank, world, port, n, m, seed, outdir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    x = simulate.design_two_group(m)
    d = simulate.make_counts(n, x, seed=seed)
    idx = parallel.shard_ranges(d["counts"].shape[0], world)[rank]
    sub = {"counts": d["counts"][idx], "size_factors": d["size_factors"]}
    shards = _run_chunks(sub, x, 2, O=O)
    np.savez(os.path.join(outdir, "cshard%d.npz" % rank), idx=idx, **parallel.concat_mcols(shards, COLS))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_ranks_times_two_chunks_equal_serial(tmp_path, oracle):
    """ranks x chunks: the global gene order of the gathered vectors is (rank, chunk)"""
    import torch.multiprocessing as mp
    n, m, seed, world = 400, 12, 35, 2
    mp.spawn(_worker_chunks, args=(world, _free_port(), n, m, seed, str(tmp_path)), nprocs=world, join=True)
    from deseq2_amd import core
    from deseq2_amd.engine import HostEngine
    x = simulate.design_two_group(m)
    d = simulate.make_counts(n, x, seed=seed)
    serial = core.DESeq(core.DESeqDataSet(d["counts"], x, sizeFactors=d["size_factors"], engine=HostEngine(oracle)))
    parts = [np.load(os.path.join(str(tmp_path), "cshard%d.npz" % r)) for r in range(world)]
    for k in COLS:
        np.testing.assert_array_equal(np.concatenate([p[k] for p in parts]), serial.mcols[k], err_msg=k)


def test_cooperative_chunks_with_baton(oracle):
    """the token-passing form (parallel.Baton): chunks take turns, release the token while waiting for each
    other at the all-gene step, and a finished chunk retires -- same results, no deadlock"""
    import threading
    from deseq2_amd import core
    from deseq2_amd.engine import HostEngine
    x = simulate.design_two_group(10)
    d = simulate.make_counts(240, x, seed=36)
    k = 3
    ranges = parallel.shard_ranges(d["counts"].shape[0], k)
    baton = parallel.Baton(k)
    group = parallel.LocalGroup(k, None, baton)
    out, errs = [None] * k, []

    def work(c):
        try:
            baton.acquire(c)
            dds = core.DESeqDataSet(d["counts"][ranges[c]], x, sizeFactors=d["size_factors"], engine=HostEngine(oracle))
            baton.handoff(c)                       # a point where a device chunk would wait for its kernels
            out[c] = parallel.DESeqParallel(dds, group=group, chunk=c)
            baton.retire(c)
        except BaseException as e:      # noqa: BLE001
            errs.append(e)
            baton.retire(c)
            group.barrier.abort()
    th = [threading.Thread(target=work, args=(c,)) for c in range(k)]
    [t.start() for t in th]
    [t.join(timeout=120) for t in th]
    assert not any(t.is_alive() for t in th), "deadlock"
    if errs:
        raise errs[0]
    serial = core.DESeq(core.DESeqDataSet(d["counts"], x, sizeFactors=d["size_factors"], engine=HostEngine(oracle)))
    got = parallel.concat_mcols(out, COLS)
    for kk in COLS:
        np.testing.assert_array_equal(got[kk], serial.mcols[kk], err_msg=kk)


def _worker_bp(rank, world, port, n, m, seed, outdir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from deseq2_amd import core
    from deseq2_amd.engine import HostEngine
    from oracle import oracle as O
    x = simulate.design_factor(m, 3)
    factors = {"group": (np.arange(m) * 3) // m}
    d = simulate.make_counts(n, x, seed=seed)
    idx = parallel.shard_ranges(d["counts"].shape[0], world)[rank]
    dds = core.DESeqDataSet(d["counts"][idx], x, sizeFactors=d["size_factors"], engine=HostEngine(O))
    parallel.DESeqParallel(dds, betaPrior=True, factors=factors)
    np.savez(os.path.join(outdir, "bp%d.npz" % rank), idx=idx, bpv=dds.attrs["betaPriorVar"],
             **{k: dds.mcols[k] for k in COLS + ["MLE_beta"]})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_shards_equal_serial_with_the_beta_prior(tmp_path, oracle):
    """R/parallel.R:30-48: MAP + MLE coefficients per shard, estimateBetaPriorVar over all rows, the prior fit per shard"""
    import torch.multiprocessing as mp
    n, m, seed, world = 300, 12, 33, 2
    mp.spawn(_worker_bp, args=(world, _free_port(), n, m, seed, str(tmp_path)), nprocs=world, join=True)
    from deseq2_amd import core
    from deseq2_amd.engine import HostEngine
    x = simulate.design_factor(m, 3)
    factors = {"group": (np.arange(m) * 3) // m}
    d = simulate.make_counts(n, x, seed=seed)
    serial = core.DESeq(core.DESeqDataSet(d["counts"], x, sizeFactors=d["size_factors"], engine=HostEngine(oracle)),
                        betaPrior=True, factors=factors)
    parts = [np.load(os.path.join(str(tmp_path), "bp%d.npz" % r)) for r in range(world)]
    np.testing.assert_array_equal(parts[0]["bpv"], parts[1]["bpv"])
    np.testing.assert_array_equal(parts[0]["bpv"], serial.attrs["betaPriorVar"])
    assert serial.mcols["beta"].shape[1] == 4                     # expanded model matrix: intercept + 3 levels
    for k in COLS + ["MLE_beta"]:
        got = np.concatenate([p[k] for p in parts])
        np.testing.assert_array_equal(got, serial.mcols[k], err_msg=k)


# ---- a caller's trend function (fitType = callable, the 'custom' dispersion function) ------------------
def _loglinear_trend(means, disps):
    """a stand-in for R's fitType = "local": least squares of log disp on log mean, returned as a function of the mean"""
    ok = (disps > 1e-6) & (means > 0)
    b, a = np.polyfit(np.log(means[ok]), np.log(disps[ok]), 1)
    return lambda bm: np.exp(a + b * np.log(bm))


def _worker_custom(rank, world, port, n, m, seed, outdir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from deseq2_amd import core
    from deseq2_amd.engine import HostEngine
    from oracle import oracle as O
    x = simulate.design_two_group(m)
    d = simulate.make_counts(n, x, seed=seed)
    idx = parallel.shard_ranges(d["counts"].shape[0], world)[rank]
    dds = core.DESeqDataSet(d["counts"][idx], x, sizeFactors=d["size_factors"], engine=HostEngine(O))
    parallel.DESeqParallel(dds, fitType=_loglinear_trend)
    assert dds.dispersionFunction["fitType"] == "custom"
    np.savez(os.path.join(outdir, "cu%d.npz" % rank), idx=idx, **{k: dds.mcols[k] for k in COLS})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_shards_with_a_callers_trend_function(tmp_path, oracle):
    """DESeqParallel with fitType = a function: the shard evaluates the global function on its own means (it used to
    build an object array out of the function and fail in estimateDispersionsMAP)"""
    import torch.multiprocessing as mp
    n, m, seed, world = 300, 12, 37, 2
    mp.spawn(_worker_custom, args=(world, _free_port(), n, m, seed, str(tmp_path)), nprocs=world, join=True)
    from deseq2_amd import core
    from deseq2_amd.engine import HostEngine
    x = simulate.design_two_group(m)
    d = simulate.make_counts(n, x, seed=seed)
    serial = core.DESeq(core.DESeqDataSet(d["counts"], x, sizeFactors=d["size_factors"], engine=HostEngine(oracle)),
                        fitType=_loglinear_trend)
    parts = [np.load(os.path.join(str(tmp_path), "cu%d.npz" % r)) for r in range(world)]
    for k in COLS:
        got = np.concatenate([p[k] for p in parts])
        np.testing.assert_array_equal(got, serial.mcols[k], err_msg=k)
