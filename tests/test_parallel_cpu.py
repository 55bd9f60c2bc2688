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
