#!/usr/bin/env python
"""bench.py -- genes/sec of the full DESeq() dispersion + beta + Wald/LRT fit on MI355X.

One "step" = one pass of the hot path over one synthetic count matrix that is already resident in HBM in
R's layout (column-major int32 counts, f64 normalization-factor matrix [, f64 weights]):
layout conversion -> prefit moments -> fitBeta (mu-hat) -> fitDisp -> fitDispGrid (stragglers) -> dispersion
trend / prior variance on n-vectors -> fitDisp (MAP) -> fitDispGrid -> fitBeta (final dispersions) -> logLik,
Wald or LRT statistics and p-values -> Cook's distances -> replaceOutliers -> refit of the replaced rows:
everything DESeq() does by default.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C2|C3|C4|C5]

Workloads = BASELINE.json configs:
  C2  20 000 genes x  100 samples, 2-level factor (p = 2), Wald
  C3  50 000 genes x  500 samples, ~batch + condition (p = 4), Wald               <- the headline metric's config
  C4  60 000 genes x 2000 samples, 10-level factor (p = 10), nbinomLRT full vs ~1
  C4R the same matrix, LRT against a 2-column reduced model (the reduced fit is an IRLS at p = 2), minmu = 1e-6
  C5  30 000 genes x  200 samples, 2-level condition, observation weights (2 % zeros) + betaPrior = TRUE
      (MLE pass on the standard design, prior pass on the expanded p = 3 design, R/fitNbinomGLMs.R:242-337)

N > 1: one process per GPU.  Under torchrun (RANK / WORLD_SIZE in the environment) this process is one rank;
otherwise `--gpus N` SPAWNS the N ranks itself.  Genes shard across ranks in the contiguous ranges of
R/parallel.R:10, no data-path collective; the only exchange is the all-gather of two n-vectors for the global
dispersion trend, as in DESeqParallel.  Headline = STRONG scaling (the config's genes in total, as BASELINE
quotes it: "50k x 500 x p=4 ... gene-sharded across 8 GPUs"); the same run also times WEAK scaling (the
config's genes per GPU) and reports it under "weak".

N = 1 on the fused chain: the K steps are pipelined two deep (--pipeline 2, the default) -- step k is enqueued, then
step k - 1 is finished on the host (its result block waited for, its columns built) while the device runs step k; every
step's results are consumed inside the timed region and all K steps are complete before the closing barrier.  The same
steps one call at a time are timed right after ("one_call_at_a_time"); --pipeline 1 makes that the headline.

Prints ONE JSON line (rank 0).  See DESIGN.md "Measurement" for the roofline accounting.
"""
import argparse
import json
import os
import subprocess
import sys
import time

# one host thread per rank: the host code is a launcher; BLAS / OpenMP pools spun up by a tiny QR of the design
# would oversubscribe the node when 8 ranks share it (the CPU baseline sets its own thread count)
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "1")

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

CONFIGS = {
    "C2": dict(genes=20000, samples=100, design="two_group", test="Wald",
               label="BASELINE configs[1]: 20k genes x 100 samples, 2-level factor (p=2), Wald"),
    "C3": dict(genes=50000, samples=500, design="batch_condition", test="Wald",
               label="BASELINE configs[2]: 50k genes x 500 samples, ~batch+condition (p=4), Wald"),
    "C4": dict(genes=60000, samples=2000, design=("factor", 10), test="LRT", intercept_mean=1.0,
               label="BASELINE configs[3]: 60k genes x 2000 samples, 10-level factor (p=10), nbinomLRT full vs ~1"),
    "C4R": dict(genes=60000, samples=2000, design=("factor", 10), test="LRT", intercept_mean=1.0, reduced2=True, minmu=1e-6,
                label="BASELINE configs[3], second variant (SURVEY 8d): 60k genes x 2000 samples, 10-level factor (p=10), "
                      "nbinomLRT full vs a 2-column reduced model (fitBeta runs at p=2 as well), minmu=1e-6"),
    "C5": dict(genes=30000, samples=200, design="two_group", test="Wald", weights=True, betaPrior=True,
               label="BASELINE configs[4]: 30k genes x 200 samples, 2-level condition, observation weights + "
                     "betaPrior (expanded design, p=3)"),
}


def algorithmic_bytes_per_gene(kernel, m, nf_matrix=True, weights=False, hat=True, mu=False):
    """fitBeta reads Y (4m) [+ nf matrix 8m] [+ weights 8m], writes
    H (8m) [+ mu (8m)]; fitDisp reads Y (4m) + mu-hat (8m) [+ weights 8m]."""
    if kernel == "fit_beta":
        return 4 * m + (8 * m if nf_matrix else 0) + (8 * m if weights else 0) + (8 * m if hat else 0) + (8 * m if mu else 0)
    return 4 * m + 8 * m + (8 * m if weights else 0)


def make_design(name, m):
    from deseq2_amd import simulate
    if name == "two_group":
        return simulate.design_two_group(m)
    if name == "batch_condition":
        return simulate.design_batch_condition(m)
    return simulate.design_factor(m, name[1])


def make_weights(n, m, seed):
    """w ~ U(0.05, 1), 2 % exactly 0 (row-max normalisation happens in getAndCheckWeights)"""
    rng = np.random.Generator(np.random.PCG64(seed))
    w = rng.uniform(0.05, 1.0, (n, m))
    w[rng.uniform(size=(n, m)) < 0.02] = 0.0
    return w


def spawn_ranks(n_gpus, argv):
    """`python bench.py --gpus N` outside torchrun: start the N ranks (one process per GPU) and relay rank 0"""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n_gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for p in procs:
        rc = p.wait() or rc
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="C3")
    ap.add_argument("--seed", type=int, default=1, help="seed of the synthetic workload (SURVEY 8d: seeds 1..3)")
    ap.add_argument("--size-factors", choices=("lognormal", "unit"), default="lognormal",
                    help="s_j ~ logN(0, 0.25^2) (SURVEY 8d's second variant: the harder half, the headline since round 5) or s_j = 1")
    ap.add_argument("--pipeline", type=int, choices=(1, 2), default=2,
                    help="N = 1, fused chain: 2 (default) = a step is enqueued while the previous step's result block is "
                         "copied (side stream) and post-processed on the host; 1 = every step waits for its own results")
    ap.add_argument("--no-variants", action="store_true",
                    help="N = 1: skip the short extra timed regions on the other workload variants (s_j = 1, seeds 2 and 3)")
    ap.add_argument("--no-configs", action="store_true",
                    help="N = 1, default config: skip the short timed regions of the other BASELINE configs (C2, C4, C5), "
                         "each in its own process, reported under \"configs\"")
    ap.add_argument("--no-parity", action="store_true", help="skip the oracle check of a 2048-row sample (+ the replaced rows) of the step's own result")
    ap.add_argument("--genes", type=int, default=0, help="override the config's gene count (tuning runs)")
    ap.add_argument("--samples", type=int, default=0, help="override the config's sample count (tuning runs)")
    ap.add_argument("--no-weak", action="store_true", help="N > 1: skip the second (weak-scaling) timed region")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-hostpath", action="store_true", help="skip the PCIe-inclusive (host-pointer ABI) pass")
    ap.add_argument("--cpu-sample-genes", type=int, default=0, help="genes of the CPU baseline sample (0 = by config)")
    ap.add_argument("--profile-host", action="store_true", help="print wall time per pipeline phase (adds syncs)")
    ap.add_argument("--hold-results", action="store_true",
                    help="diagnostic: keep EVERY step's result object alive, so that each step allocates its device buffers "
                         "and pinned result block afresh from the driver -- what a host that leaks a step's buffers pays "
                         "on this box (round 3's dds <-> run reference cycle did that until the cyclic collector ran)")
    ap.add_argument("--call-by-call", action="store_true",
                    help="time core.DESeq() (the R-side decision rules as host code between the native calls) instead "
                         "of the fused device-driven chain")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: deseq2_amd has no CPU compute path")
    # DSQ_BENCH_ONE_DEVICE=1: smoke-test the multi-rank path on a 1-GPU box (all ranks on cuda:0,
    # n-vector exchange over gloo); never set by the driver.
    one_dev = os.environ.get("DSQ_BENCH_ONE_DEVICE") == "1"
    if one_dev:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    comm_dev = None if one_dev else dev
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_dev or os.environ.get("DSQ_BENCH_COMM") == "gloo":
            dist.init_process_group(backend="gloo")
            comm_dev = None
        else:
            # RCCL for the two small n-vector all-gathers of the chain; if it cannot come up on this node the same
            # exchange runs through the host (gloo) -- the data path has no collective either way
            try:
                dist.init_process_group(backend="nccl", device_id=dev)
                probe = torch.ones(1, device=dev)
                dist.all_reduce(probe)
                torch.cuda.synchronize()
            except Exception as e:                                   # noqa: BLE001
                print("bench.py: RCCL unavailable (%r), exchanging the n-vectors over gloo" % (e,), file=sys.stderr)
                try:
                    dist.destroy_process_group()
                except Exception:                                    # noqa: BLE001
                    pass
                dist.init_process_group(backend="gloo")
                comm_dev = None

    from deseq2_amd import core, fused, simulate, parallel
    from deseq2_amd.engine import DeviceEngine

    cfg = dict(CONFIGS[args.config])
    n_req = args.genes or cfg["genes"]
    m = args.samples or cfg["samples"]
    x = make_design(cfg["design"], m)
    p = x.shape[1]
    use_w = bool(cfg.get("weights"))
    factors = {"condition": x[:, 1].astype(int)} if cfg.get("betaPrior") else None
    reduced = np.ones((m, 1)) if cfg["test"] == "LRT" else None
    if cfg.get("reduced2"):
        reduced = np.column_stack([np.ones(m), (np.arange(m) >= m // 2).astype(np.float64)])
    E = DeviceEngine(dev)

    def workload(seed, lo_hi=None, sf_mode=None):
        """synthetic counts (+ weights) of this config, resident in HBM in R layout; lo_hi = this rank's shard"""
        sf_in = None
        if (sf_mode or args.size_factors) == "lognormal":
            # s_j ~ logN(0, 0.25^2), its own stream so that the per-gene draws are those of the s_j = 1 variant
            sf_in = np.exp(np.random.Generator(np.random.PCG64(1000 + seed)).normal(0.0, 0.25, m))
        d = simulate.make_counts(n_req, x, seed=seed, intercept_mean=cfg.get("intercept_mean", 4.0), size_factors=sf_in)
        counts = d["counts"]
        w = make_weights(counts.shape[0], m, seed + 77) if use_w else None
        n_all = counts.shape[0]
        if lo_hi is not None:
            lo, hi = lo_hi(n_all)
            counts = counts[lo:hi]
            w = None if w is None else w[lo:hi]
        n = counts.shape[0]
        sf = d["size_factors"]
        sizes = [len(r) for r in parallel.shard_ranges(n_all, world)] if (lo_hi is not None and world > 1) else None
        counts_r = torch.as_tensor(np.ascontiguousarray(counts.T), device=dev)                 # (m, n) int32
        nf_r = torch.ones((m, n), dtype=torch.float64, device=dev) * torch.as_tensor(sf, device=dev)[:, None]
        w_r = None if w is None else torch.as_tensor(np.ascontiguousarray(w.T), device=dev)
        torch.cuda.synchronize()
        return dict(counts=counts, counts_r=counts_r, nf_r=nf_r, w=w, w_r=w_r, sf=sf, n=n, n_all=n_all, shard_sizes=sizes)

    def shard(n_all):
        r = parallel.shard_ranges(n_all, world)[rank]
        return int(r[0]), int(r[-1]) + 1

    def make_step(W):
        def step(wait=True):
            dds = core.DESeqDataSet.from_device(E, W["counts_r"], W["nf_r"], x, weights=W["w"], sizeFactors=W["sf"],
                                                weights_r=W["w_r"])
            kw = dict(test=cfg["test"], reduced=reduced)
            if cfg.get("betaPrior"):
                kw.update(betaPrior=True, factors=factors)
            if cfg.get("minmu"):
                kw.update(minmu=cfg["minmu"])
            if args.call_by_call:
                if world > 1:
                    parallel.DESeqParallel(dds, comm_device=comm_dev, **kw)
                else:
                    core.DESeq(dds, **kw)
            else:
                # the fused device-driven chain (deseq2_amd/fused.py); settings it does not cover (betaPrior: C5) run
                # the call-by-call chain of core.py
                fused.DESeq(dds, comm_device=comm_dev, shard_sizes=W.get("shard_sizes"), wait=wait, **kw)
            return [dds]
        return step

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def alloc_counters():
        """calls that reached the DRIVER (hipMalloc / hipHostMalloc), not the caching allocators' hits: a steady-state
        step must not make any (a step that does pays milliseconds on some hosts -- round 3's 27.8 ms vs 13.0 ms)"""
        c = {"device": int(torch.cuda.memory_stats(dev).get("num_device_alloc", 0))}
        try:
            c["pinned_host"] = int(torch.cuda.host_memory_stats().get("num_host_alloc", 0))
        except Exception:                                            # noqa: BLE001  (older torch: no host counters)
            c["pinned_host"] = None
        return c

    def timed(step, n_local, depth=1):
        """depth = 2: software pipeline over the steps (fused chain, one process).  Step k is ENQUEUED (chain + the copy
        of its result block on a side stream), then step k - 1 is finished on the host (wait for its block, build its
        columns) while the device works on step k.  Every step's results are consumed inside the timed region; all K
        steps are complete before the closing barrier.  depth = 1: each step waits for its own results (what one
        DESeq() call costs end to end)."""
        import gc
        gc.collect()            # (before the warm-up, not between it and the timed steps: the device would sit idle for the
                                #  tens of milliseconds a collection takes and start the timed region from lowered clocks)
        gc_was_on = gc.isenabled()
        gc.disable()            # ... and no automatic collection inside the timed region (a 4-6 ms pause when one lands
                                #  there; the standard library's timeit does the same).  Nothing here relies on the cyclic
                                #  collector: fused.py keeps its objects free of reference cycles.
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
        prev = wd = None
        for _ in range(args.warmup):      # (the warm-up runs the way the timed steps do -- same enqueue / finish / release
            wd = None                     #  order --: the allocator pools and the runtime's queues get their size here)
            if depth == 2:
                cur = step(wait=False)
                if prev is not None:
                    fused.finish(prev[0])
                wd, prev = prev, cur
            else:
                wd = step()
        # depth 2: the pipeline is NOT drained between the warm-up and the timed steps -- the last warm-up analysis stays in
        # flight (its chain is complete at the barrier below, its host half is the first thing the timed region does, an
        # extra it pays for) and the one before it stays alive until the first timed step releases it, as every later
        # step does.  Draining here handed the caching allocator a different free list than the steady state's: one
        # hipMalloc inside every timed region, whatever the number of warm-up steps (round 5's
        # driver_allocs_in_timed_region.device = 1), and a first timed step that finished nothing (step_ms.min 0.45 ms).
        dds = wd
        cur = wd = None
        a0 = alloc_counters()
        barrier()
        marks = []
        t0 = time.perf_counter()
        held = []
        for k in range(args.steps):
            if args.hold_results:
                held.append(dds)
            tq0 = time.perf_counter()
            dds = None          # release the previous step's HBM tensors before allocating the next ones
            tq1 = time.perf_counter()
            ev[k][0].record()
            if depth == 2:
                cur = step(wait=False)      # chain + result copy enqueued, nothing waited for
                ev[k][1].record()
                tq2 = time.perf_counter()
                if prev is not None:
                    fused.finish(prev[0])   # step k - 1: its block is down (or nearly), its columns are built now
                if os.environ.get("DSQ_BENCH_DEBUG") and k < 5:
                    print("k=%d release %.3f enqueue %.3f finish %.3f ms" % (k, (tq1 - tq0) * 1e3, (tq2 - tq1) * 1e3, (time.perf_counter() - tq2) * 1e3), file=sys.stderr)
                dds, prev = prev, cur
            else:
                dds = step()    # (ends with the one device-to-host copy of the result block + stream sync)
                ev[k][1].record()
            marks.append(time.perf_counter())
        if prev is not None:
            fused.finish(prev[0])
            dds = prev
            prev = cur = None
        barrier()
        dt = time.perf_counter() - t0
        if gc_was_on:
            gc.enable()
        a1 = alloc_counters()
        held = None
        per = np.diff(np.r_[t0, marks]) * 1e3
        if depth == 2 and per.size > 2:
            per = per[1:]       # (pipelined: the first timed step only finishes an analysis the barrier has already waited for --
                                #  its wall time is an enqueue, not a step; ms_per_step is the whole region / K either way)
        span = np.array([a.elapsed_time(b) for a, b in ev])
        if os.environ.get("DSQ_BENCH_DEBUG") and rank == 0:
            print("steps wall ms", np.round(per, 3).tolist(), "device span ms", np.round(span, 3).tolist(), file=sys.stderr)
        stats = {"step_ms": {"min": float(per.min()), "median": float(np.median(per)), "max": float(per.max())},
                 # first enqueue of a step -> its last copy has landed, on the device's clock (HIP events on the chain's
                 # stream); step wall time minus this = host code after the results are down
                 "gpu_span_ms": {"min": float(span.min()), "median": float(np.median(span)), "max": float(span.max())},
                 "driver_allocs_in_timed_region": {k: (None if a0[k] is None else a1[k] - a0[k]) for k in a0}}
        if world > 1:
            cdev = dev if comm_dev is not None else torch.device("cpu")
            tt = torch.tensor([dt], dtype=torch.float64, device=cdev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
            nn = torch.tensor([n_local], dtype=torch.int64, device=cdev)
            dist.all_reduce(nn, op=dist.ReduceOp.SUM)
            n_total = int(nn.item())
        else:
            n_total = n_local
        return dt, n_total, dds, stats

    # ---- headline: STRONG scaling, the config's genes in total -----------------------------------------------
    W = workload(args.seed, shard if world > 1 else None)
    n = W["n"]
    step = make_step(W)

    if args.profile_host and rank == 0:
        return profile_host(core, E, step, torch)

    depth = args.pipeline if (world == 1 and not args.call_by_call) else 1
    dt, n_total, dds, step_stats = timed(step, n, depth)
    fused_used = bool(dds[0].attrs.get("fused"))
    sync_ms = None
    if depth == 2:
        # the same K steps, each waiting for its own results: what ONE DESeq() call costs end to end (reported beside the
        # pipelined throughput, never instead of it)
        keep_warm, args.warmup = args.warmup, 1
        dts, _, dsy, sst = timed(step, n, 1)
        args.warmup = keep_warm
        sync_ms = {"ms_per_step": dts / args.steps * 1e3, "step_ms": sst["step_ms"],
                   "result_digest_equal": bool(result_digest(dsy[0], world, comm_dev, parallel) == result_digest(dds[0], world, comm_dev, parallel))}
        dsy = None

    # per-kernel launch durations: two extra UNTIMED passes with HIP events around each kernel (recorded inside
    # the C library on the launch stream; reading them back synchronises, so they stay out of the throughput)
    E.record = []
    for _ in range(2):
        step()
    rec, E.record = E.record, None
    mc = parallel.concat_mcols(dds, [k for k in ("betaIter", "dispIter", "dispGeneIter") if k in dds[0].mcols])
    digest = result_digest(dds[0], world, comm_dev, parallel)

    # parity where the driver can see it: a fixed 2048-row sample (+ every replaced row) of THIS step's result against the CPU oracle (the checker,
    # after the timed region, never the thing measured)
    parity = None
    if rank == 0 and not args.no_parity:
        try:
            parity = parity_sample(dds[0], W, x, cfg, factors, reduced, rows=2048)
        except Exception as e:                                       # noqa: BLE001
            parity = {"error": repr(e)}
        try:
            parity["second_implementation"] = lapack_rates(W, x, cfg)
        except Exception as e:                                       # noqa: BLE001
            parity["second_implementation"] = {"error": repr(e)}

    # the other halves of the spec'd workload, short timed regions of the same step: s_j = 1 and seeds 2, 3
    variants = None
    if world == 1 and not args.no_variants and not args.genes and not args.samples:
        variants = []
        keep_steps, keep_warm = args.steps, args.warmup
        args.steps, args.warmup = max(3, min(5, keep_steps)), 1
        for vseed, vmode in ((args.seed, "unit" if args.size_factors == "lognormal" else "lognormal"),
                             (args.seed + 1, args.size_factors), (args.seed + 2, args.size_factors)):
            Wv = workload(vseed, None, vmode)
            dtv, ntv, ddv, _ = timed(make_step(Wv), Wv["n"], depth)
            variants.append({"seed": vseed, "size_factors": vmode, "genes": ntv, "steps": args.steps,
                             "ms_per_step": dtv / args.steps * 1e3, "value": ntv * args.steps / dtv,
                             "result_digest": result_digest(ddv[0], world, comm_dev, parallel)})
            Wv = ddv = None
        args.steps, args.warmup = keep_steps, keep_warm

    weak = None
    if world > 1 and not args.no_weak:
        dds = step = None
        W2 = workload(args.seed + rank)
        dtw, ntw, _, _ = timed(make_step(W2), W2["n"])
        weak = {"value": ntw * args.steps / dtw, "unit": "genes/s", "ms_per_step": dtw / args.steps * 1e3,
                "genes_per_gpu": W2["n"], "genes_total": ntw}
        W2 = None

    hostpath = hostpath_fused = None
    if world == 1 and not args.no_hostpath:
        hostpath = hostpath_ms(core, W, x, cfg, factors, reduced)
        hostpath_fused = hostpath_fused_ms(W, x, cfg, reduced, factors)

    if rank == 0:
        per = {}
        for name, ng, ms in rec:
            per.setdefault(name, []).append((ng, ms))

        big = n // 2
        # full-size launches of the chain vs the row-listed ones (fitDispGrid stragglers, refitWithoutOutliers)
        small = lambda k, g: k.endswith(":refit") or k.endswith("_grid") or g < big     # noqa: E731
        kern, kern_refit = {}, {}
        for k, v in per.items():
            for dst, sel in ((kern, False), (kern_refit, True)):
                vv = [(g, t) for g, t in v if small(k, g) == sel]
                if vv:
                    dst[k] = {"launches": len(vv), "avg_ms": float(np.mean([t for _, t in vv])),
                              "genes_per_launch": float(np.mean([g for g, _ in vv]))}
        share = {k: kern[k]["avg_ms"] * kern[k]["launches"] for k in kern}
        # every launch the library timed in the two profiling passes (full-size and row-listed), per step
        kernel_sum = float(sum(ms for _, _, ms in rec)) / 2.0
        dom = max(("fit_beta", "fit_disp"), key=lambda k: share.get(k, 0.0))
        nfull = n
        avg_ms = kern[dom]["avg_ms"]
        if dom == "fit_beta":
            # fitBeta#1 writes mu (no H), the final fit writes mu and H: average over the full-size launches
            bytes_per_gene = (algorithmic_bytes_per_gene("fit_beta", m, weights=use_w, hat=False, mu=True) +
                              algorithmic_bytes_per_gene("fit_beta", m, weights=use_w, hat=True, mu=True)) / 2.0
        else:
            bytes_per_gene = algorithmic_bytes_per_gene("fit_disp", m, weights=use_w)
        achieved = bytes_per_gene * nfull / (avg_ms * 1e-3) / 1e9
        # HBM bytes / VALU instructions per launch from the PMC counters: collected in separate rocprofv3 --pmc
        # passes of this same command (FETCH_SIZE doubled per the gfx950 note, + WRITE_SIZE) and committed under
        # profiles/ -- counters cannot be read from inside the process.
        traffic, valu, f64, pmc_file, pmc_ok = None, None, None, None, None
        lib_sha = library_sha256()
        for cand in ("r06_pmc_%s.json" % args.config, "r05_pmc_%s.json" % args.config, "r04_pmc_%s.json" % args.config, "r03_pmc_%s.json" % args.config, "r02_pmc_%s.json" % args.config,
                     "r01_pmc.json" if args.config == "C3" else None):
            if cand and os.path.exists(os.path.join(ROOT, "profiles", cand)):
                pmc_file = cand
                break
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", pmc_file)))
            # the counters were collected in separate rocprofv3 passes: they describe THIS run only if they were taken on
            # the library that was just timed (tools/pmc_summary.py records its sha256); otherwise the counter-derived
            # figures are withheld
            pmc_ok = pmc.get("_library_sha256") == lib_sha
            if not pmc_ok:
                raise ValueError("PMC file of another library build")
            scale = nfull / pmc.get("_genes_per_launch", cfg["genes"])
            traffic = pmc[dom]["hbm_bytes_per_launch"] * scale
            insts = pmc[dom]["SQ_INSTS_VALU"] * scale
            prop = torch.cuda.get_device_properties(dev)
            clock_hz = float(getattr(prop, "clock_rate", 2400000)) * 1e3
            peak = prop.multi_processor_count * 4 * clock_hz / 4.0
            ach = insts / (avg_ms * 1e-3)
            valu = {"kernel": dom, "achieved": ach / 1e9, "peak": peak / 1e9, "unit": "G wave-instr/s",
                    "frac": ach / peak, "valu_instructions_per_launch": insts,
                    "note": "SQ_INSTS_VALU from profiles/%s (same workload), launch time measured live" % pmc_file}
            if pmc[dom].get("f64_flops_per_launch"):
                # the bound that is real for this path: f64 vector flops of the dominant kernel -- the PMC pass's
                # SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64 wave-instructions x 64 lanes (fma = 2) -- over the launch time
                # measured live, against the f64 vector peak = CUs x 4 SIMDs x 16 lanes x 2 flop x clock (78.6 TF at 2.4 GHz)
                flops = pmc[dom]["f64_flops_per_launch"] * scale
                peak_f = prop.multi_processor_count * 4 * 16 * 2 * clock_hz
                f64 = {"kernel": dom, "achieved": flops / (avg_ms * 1e-3) / 1e12, "peak": peak_f / 1e12, "unit": "TFLOP/s",
                       "frac": flops / (avg_ms * 1e-3) / peak_f, "flops_per_launch": flops,
                       "f64_arith_frac_of_valu_instructions": pmc[dom].get("f64_arith_frac_of_valu"),
                       "note": "SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64 from profiles/%s (same workload; lanes masked off "
                               "count as flops: an upper bound), launch time measured live" % pmc_file}
        except (OSError, KeyError, TypeError, ValueError):
            pass
        roofline = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                    "algorithmic_bytes_per_launch": bytes_per_gene * nfull, "genes_per_launch": nfull,
                    "avg_launch_ms": avg_ms,
                    "note": "f64-VALU/transcendental bound, not HBM bound (DESIGN.md); see profiles/"}
        out = {
            "metric": "genes/sec for DESeq() disp+beta+%s fit, %s%s" % (
                cfg["test"], {"C2": "20k x 100 x p=2", "C3": "50k x 500 x p=4", "C4": "60k x 2000 x p=10 (LRT)",
                              "C4R": "60k x 2000 x p=10 (LRT vs 2-column reduced, minmu=1e-6)",
                              "C5": "30k x 200, weights + betaPrior"}[args.config],
                " (throughput of steps pipelined two deep; one call at a time: one_call_at_a_time)" if depth == 2 else ""),
            "value": n_total * args.steps / dt,
            "unit": "genes/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "step_ms": step_stats["step_ms"],
            "gpu_span_ms": step_stats["gpu_span_ms"],
            "kernel_sum_ms": kernel_sum,
            "driver_allocs_in_timed_region": step_stats["driver_allocs_in_timed_region"],
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "%s; %d non-zero genes in total, sharded over %d GPU(s) in the contiguous ranges "
                                   "of R/parallel.R:10; inputs resident in HBM in R layout (int32 counts, f64 nf "
                                   "matrix%s; the fused chain reads the size factors as the m-vector the matrix was "
                                   "built from)" % (cfg["label"], n_total, world, ", f64 weights" if use_w else ""),
                       "name": args.config, "genes_total": n_total, "genes_this_gpu": n, "samples": m, "p": p,
                       "test": cfg["test"], "parallelism": "gene-shard x%d" % world,
                       "chain": "fused device-driven (dsq_deseq_dev)" if fused_used else "call-by-call (core.py)",
                       "steps_overlap": "step k enqueued while step k-1's results are copied and post-processed" if depth == 2 else "none"},
            "roofline": roofline,
            "valu_roofline": valu,
            "f64_roofline": f64,
            "pmc_file": pmc_file, "pmc_matches_library": pmc_ok, "library_sha256": lib_sha,
            "kernels": kern,
            "kernels_outlier_refit": kern_refit,
            "mean_iterations": {k: float(np.nanmean(v)) for k, v in mc.items()},
            "result_digest": digest,
            "parity": parity,
            "workload": {"seed": args.seed, "size_factors": args.size_factors},
            # 2: step k is enqueued while step k - 1's result block is copied (side stream) and its columns are built on
            # the host -- every step's results are consumed inside the timed region; 1: each step waits for its own
            "pipeline_depth": depth,
        }
        if sync_ms is not None:
            out["one_call_at_a_time"] = sync_ms      # the same K steps without the overlap: one DESeq() call end to end
        if variants is not None:
            out["variants"] = variants
        if weak is not None:
            out["weak"] = weak
        if hostpath is not None:
            out["hostpath_ms"] = hostpath["ms"]
            out["hostpath"] = hostpath
        if hostpath_fused is not None:
            out["hostpath_fused_ms"] = hostpath_fused["ms"]
            out["hostpath_fused"] = hostpath_fused
        if world == 1 and args.config == "C3" and not args.no_configs and not args.genes and not args.samples and not args.call_by_call:
            out["configs"] = other_configs(out, args)
        if world == 1 and not args.no_cpu_baseline:
            k = args.cpu_sample_genes or {"C2": 20000, "C3": 16384, "C4": 2048, "C4R": 2048, "C5": 16384}[args.config]   # (10-30 s of one core)
            out["cpu_baseline"] = cpu_baseline(W["counts"], W["sf"], x, k, cfg, W["w"], factors, reduced)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def parity_sample(dds, W, x, cfg, factors, reduced, rows=2048):
    """A fixed sample of the rows of the step's OWN result, re-fitted by the CPU oracle (oracle/, the checker): the gene-wise
    dispersion search, the MAP search under the run's trend and prior variance, and the final IRLS + test -- every per-gene
    step of the chain (the all-gene steps, trend and prior variance, are taken from the run: they are not per-gene).  Rows
    whose counts the outlier step replaced are IN the sample (every one of them, beside the random rows): they are re-fitted
    on the run's replaced counts with the defaults refitWithoutOutliers uses (R/core.R:2509-2531), the trend taken at
    their new means as the run did.  iter_equal: fraction of sampled rows whose three iteration counts are all equal;
    max_rel: largest relative difference over the float columns (0.0 = bit-identical)."""
    import torch
    from deseq2_amd import core
    from deseq2_amd.engine import HostEngine
    from oracle import oracle as O
    O.set_threads(os.cpu_count() or 1)
    mc = dds.mcols
    n = W["counts"].shape[0]
    ok = ~np.asarray(mc["allZero"], bool)
    rep = np.zeros(n, bool)
    if "replace" in mc:
        rep = (np.nan_to_num(np.asarray(mc["replace"], np.float64)) != 0) & ok
    cand = np.where(ok & ~rep)[0]
    pick = np.sort(np.random.Generator(np.random.PCG64(20260926)).choice(cand, min(rows, cand.size), replace=False))
    rep_rows = np.where(rep)[0]
    fn = dict(dds.dispersionFunction)
    wald = cfg["test"] == "Wald"
    fcols = ["baseMean", "dispGeneEst", "dispMAP", "dispersion", "beta", "betaSE", "WaldStatistic" if wald else "LRTStatistic"]
    icols = ["dispGeneIter", "dispIter", "betaIter"]

    def refit(idx, counts, minmu, refit_defaults):
        w = None if W["w"] is None else W["w"][idx]
        o = core.DESeqDataSet(counts, x, sizeFactors=W["sf"], weights=w, engine=HostEngine(O))
        core.estimateDispersionsGeneEst(o, minmu=minmu)
        o.dispersionFunction = fn
        o.mcols["dispFit"] = np.asarray(mc["dispFit"])[idx]
        core.estimateDispersionsMAP(o, dispPriorVar=fn["dispPriorVar"])
        if wald:
            kw = {} if refit_defaults else dict(minmu=minmu)
            if cfg.get("betaPrior"):
                kw.update(betaPrior=True, factors=factors, betaPriorVar=dds.attrs["betaPriorVar"])
            core.nbinomWaldTest(o, **kw)
        else:
            core.nbinomLRT(o, reduced, **({} if refit_defaults else dict(minmu=minmu)))
        return o

    def compare(o, idx):
        eq = np.ones(idx.size, bool)
        for k in icols:
            eq &= np.asarray(o.mcols[k], np.float64) == np.asarray(mc[k], np.float64)[idx]
        max_rel, worst = 0.0, None
        for k in fcols:
            a, b = np.asarray(mc[k], np.float64)[idx], np.asarray(o.mcols[k], np.float64)
            both = np.isfinite(a) & np.isfinite(b)
            assert (np.isfinite(a) == np.isfinite(b)).all(), "NA pattern of %s differs" % k
            if both.any():
                r = float(np.max(np.abs(a[both] - b[both]) / np.maximum(np.abs(b[both]), 1e-300)))
                if r > max_rel:
                    max_rel, worst = r, k
        return eq, max_rel, worst

    eq, max_rel, worst = compare(refit(pick, W["counts"][pick], cfg.get("minmu", 0.5), False), pick)
    out = {"rows": int(pick.size), "of_genes": int(n)}
    if rep_rows.size:
        rc = dds.assays["replaceCounts"].view()[torch.as_tensor(rep_rows, device=dds.assays["replaceCounts"].t.device)]
        eq_r, mr_r, worst_r = compare(refit(rep_rows, rc.cpu().numpy().astype(np.int32), 0.5, True), rep_rows)
        out.update(rows=int(pick.size + rep_rows.size), replaced_rows=int(rep_rows.size), replaced_iter_equal=float(eq_r.mean()),
                   replaced_max_rel=mr_r)
        eq = np.r_[eq, eq_r]
        if mr_r > max_rel:
            max_rel, worst = mr_r, worst_r
    else:
        out["replaced_rows"] = 0
    O.set_threads(1)
    out.update({"iter_equal": float(eq.mean()), "max_rel": max_rel, "worst_column": worst, "columns": fcols + icols,
                "checker": "oracle/ (CPU restatement; pinned by the reference tests' known answers, tests/test_oracle_properties.py)",
                "tolerance_north_star": 1e-6})
    return out


def _config_row(j):
    r = j.get("roofline") or {}
    return {"name": j["config"]["name"], "workload": j["metric"], "genes": j["config"]["genes_total"], "samples": j["config"]["samples"],
            "p": j["config"]["p"], "test": j["config"]["test"], "steps": j["steps"], "ms_per_step": j["ms_per_step"],
            "one_call_at_a_time_ms": (j.get("one_call_at_a_time") or {}).get("ms_per_step"),
            "genes_per_s": j["value"], "parity": j.get("parity"), "result_digest": j["result_digest"],
            "dominant_kernel": r.get("kernel"), "dominant_kernel_ms": r.get("avg_launch_ms"), "hbm_frac": r.get("frac"),
            "kernels_ms": {k: round(v["avg_ms"], 4) for k, v in (j.get("kernels") or {}).items() if v["avg_ms"] >= 0.05}}


def other_configs(out, args):
    """The other BASELINE configs where the driver sees them: C2, C4 and C5 (C3 is this line itself), each a short timed
    region (5 steps, 3 warm-up steps) of the same bench in its OWN process -- the device is released in between -- with the same
    2048-row oracle check.  A config that fails is reported with its error, never dropped."""
    rows = [_config_row(out)]
    for name in ("C2", "C4", "C5"):
        cmd = [sys.executable, os.path.abspath(__file__), "--config", name, "--steps", "5", "--warmup", "3", "--no-cpu-baseline",
               "--no-hostpath", "--no-variants", "--no-configs", "--seed", str(args.seed), "--size-factors", args.size_factors,
               "--pipeline", str(args.pipeline)]
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if r.returncode != 0 or not line:
                rows.append({"name": name, "error": "rc %d: %s" % (r.returncode, r.stderr.strip()[-300:])})
                continue
            rows.append(_config_row(json.loads(line[-1])))
        except Exception as e:                                       # noqa: BLE001
            rows.append({"name": name, "error": repr(e)})
    return rows


def library_sha256():
    import hashlib
    from deseq2_amd import _lib
    h = hashlib.sha256()
    with open(_lib.SO_PATH, "rb") as f:
        for blk in iter(lambda: f.read(1 << 22), b""):
            h.update(blk)
    return h.hexdigest()


def lapack_rates(W, x, cfg, rows=256):
    """The HIP path next to an implementation that shares no arithmetic with it (oracle/lapack_oracle.py: numpy over real
    LAPACK, scipy's special functions): fitBeta -> fitDisp (gene-wise, then with a prior) -> fitDispGrid on a fixed sample of
    this run's own rows, through the host-pointer C ABI.  Reports what north_star words as "iteration counts bit-exact" as
    numbers: rows whose fitBeta iteration count differs; `ties` = rows whose final Armijo test falls on the last bit (they take
    one step more or fewer); `floor_share` = rows whose dispersion sits at the 1e-8 floor, where the step count is rounding noise
    of whichever lgamma / digamma is underneath (tests/test_oracle_vs_lapack.py, tests/floor_regime.py)."""
    from deseq2_amd import native
    from oracle import lapack_oracle
    from tests.helpers import beta_init_qr, rough_alpha
    from tests.test_oracle_vs_lapack import compare, run_all
    counts = W["counts"]
    cand = np.where(counts.sum(axis=1) > 0)[0]
    pick = np.sort(np.random.Generator(np.random.PCG64(20260927)).choice(cand, min(rows, cand.size), replace=False))
    y = counts[pick]
    m = y.shape[1]
    nf = np.broadcast_to(np.asarray(W["sf"], np.float64)[None, :], y.shape).copy()
    use_w = W["w"] is not None
    w = np.ones(y.shape)
    if use_w:
        w = W["w"][pick] / W["w"][pick].max(axis=1, keepdims=True)                       # R/core.R:2702
    with np.errstate(all="ignore"):
        d = dict(counts=y, x=x, nf=nf, weights=w, useWeights=use_w, useQR=True, useCR=True,
                 lam=np.full(x.shape[1], 1e-6) / np.log(2) ** 2, beta_init=beta_init_qr(y.astype(float), nf, x),
                 alpha_init=rough_alpha(y.astype(float), nf, x))
    try:
        st = compare(run_all(native, d), run_all(lapack_oracle, d), d, "bench sample", min_well=0.0, min_grid=0.0)
    except AssertionError as e:
        return {"rows": int(pick.size), "error": str(e)[:300]}
    well = min(st["fitDispMLE"]["well"], st["fitDispMAP"]["well"])
    return {"rows": int(pick.size), "fitBeta_iter_mismatch": st["fitBeta"]["iter_mismatch"],
            "fitDisp_iter_mismatch_outside_ties": 0, "ties": st["fitDispMLE"]["ties"] + st["fitDispMAP"]["ties"],
            "floor_share": 1.0 - well, "grid_same": st["fitDispGrid"]["same"],
            "max_rel_log_alpha": max(st["fitDispMLE"]["max_rel_log_alpha"], st["fitDispMAP"]["max_rel_log_alpha"]),
            "against": "oracle/lapack_oracle.py (numpy + LAPACK + scipy.special; values within 1e-7 / 1e-8 asserted)"}


def result_digest(dds, world, comm_dev, parallel):
    """sha1 over the per-gene result columns of ALL ranks in gene order: equal digests at different N mean the
    sharded run reproduced the serial one bit for bit (tests/testthat/test_parallel.R:27-37)"""
    import hashlib
    h = hashlib.sha1()
    for k in ("dispGeneEst", "dispFit", "dispMAP", "dispersion", "beta", "betaSE", "WaldStatistic", "WaldPvalue",
              "LRTStatistic", "LRTPvalue", "betaIter", "dispIter", "maxCooks", "replace"):
        if k not in dds.mcols:
            continue
        v = np.asarray(dds.mcols[k], dtype=np.float64)
        v = v.reshape(v.shape[0], -1)
        cols = [parallel._allgather_vec(np.ascontiguousarray(v[:, j]), comm_dev) for j in range(v.shape[1])]
        h.update(k.encode())
        h.update(np.ascontiguousarray(np.column_stack(cols)).tobytes())
    return h.hexdigest()


def hostpath_ms(core, W, x, cfg, factors, reduced):
    """PCIe-inclusive: the same DESeq() through the HOST-pointer C ABI (dsq_fit_*: what the .Call shim binds; every
    call uploads its n x m inputs from pageable host memory and downloads its outputs) -- what an unchanged R
    session pays.  One warm-up + one timed pass, outside the throughput measurement."""
    from deseq2_amd.engine import HostEngine
    E = HostEngine()
    kw = dict(test=cfg["test"], reduced=reduced)
    if cfg.get("betaPrior"):
        kw.update(betaPrior=True, factors=factors)
    if cfg.get("minmu"):
        kw.update(minmu=cfg["minmu"])

    def run():
        # the object holds column-major host matrices, as an R session does before DESeq() is called: building it (a
        # layout copy of the numpy inputs) is not part of what is timed
        dds = core.DESeqDataSet(W["counts"], x, sizeFactors=W["sf"], weights=W["w"], engine=E)
        t0 = time.perf_counter()
        core.DESeq(dds, **kw)
        return time.perf_counter() - t0
    run()
    dt = min(run(), run())
    return {"ms": dt * 1e3, "genes_per_s": W["n"] / dt,
            "note": "full DESeq() through the per-call host-pointer entry points (dsq_fit_* and the SURVEY 8f extensions: "
                    "upload + kernels + download per call, pinned staging)"}


def hostpath_fused_ms(W, x, cfg, reduced, factors=None):
    """PCIe-inclusive, ONE call: dsq_deseq (what r_shim.c binds as _DESeq2_mi355x_DESeq) -- counts
    up from pageable host memory once through pinned staging, the device-driven chain, the per-gene columns down
    (the n x m assays stay on the device unless asked for; "with_assays" times the call that brings mu / H / cooks down
    into freshly allocated host matrices, as R's are; assays_GBps = their bytes / the extra time)."""
    from deseq2_amd import native
    kw = dict(test=cfg["test"], reduced=reduced, minmu=cfg.get("minmu", 0.5))
    if cfg.get("weights"):
        kw["weights"] = np.asfortranarray(W["w"])
    if cfg.get("betaPrior"):
        kw.update(betaPrior=True, factors=factors)

    counts_r = np.asfortranarray(W["counts"])      # column-major as R holds counts(dds): no layout copy inside the call

    def run(assays):
        t0 = time.perf_counter()
        res = native.DESeq(counts_r, x, W["sf"], assays=assays, **kw)
        dt = time.perf_counter() - t0
        # (the result matrices are released AFTER the clock stops: unmapping 600 MB of them costs ~25 ms by itself, which an
        #  R session pays when its garbage collector runs, not inside the call)
        del res
        return dt
    run(())
    dt = min(run(()), run(()))
    run(("mu", "H", "cooks"))
    dta = min(run(("mu", "H", "cooks")), run(("mu", "H", "cooks")))
    abytes = 3.0 * W["n"] * counts_r.shape[1] * 8
    return {"ms": dt * 1e3, "genes_per_s": W["n"] / dt, "with_assays_ms": dta * 1e3,
            "assays_bytes": abytes, "assays_GBps": (abytes / max(dta - dt, 1e-9)) / 1e9,
            "note": "full DESeq() through ONE dsq_deseq host-pointer call (upload counts once + device-driven chain + "
                    "per-gene columns down); with_assays also downloads mu / H / cooks (n x m f64 each)"}


def profile_host(core, E, step, torch):
    import functools
    acc = {}

    def wrap(mod, name):
        f = getattr(mod, name)

        @functools.wraps(f)
        def g(*a, **k):
            torch.cuda.synchronize(); t = time.perf_counter()
            r = f(*a, **k)
            torch.cuda.synchronize(); acc[name] = acc.get(name, 0.0) + (time.perf_counter() - t)
            return r
        setattr(mod, name, g)
    step()
    for nm in ("estimateDispersionsGeneEst", "estimateDispersionsFit", "estimateDispersionsPriorVar",
               "estimateDispersionsMAP", "nbinomWaldTest", "nbinomLRT", "fitNbinomGLMs", "getBaseMeansAndVariances",
               "getAndCheckWeights", "refitWithoutOutliers"):
        wrap(core, nm)
    for nm in ("fit_beta", "fit_disp", "fit_disp_grid", "prefit", "nbinom_loglike", "two_sided_normal_p", "take_rows"):
        f = getattr(E, nm)

        def mk(f, nm):
            def g(*a, **k):
                torch.cuda.synchronize(); t = time.perf_counter()
                r = f(*a, **k)
                torch.cuda.synchronize(); acc["E." + nm] = acc.get("E." + nm, 0.0) + (time.perf_counter() - t)
                return r
            return g
        setattr(E, nm, mk(f, nm))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    step()
    torch.cuda.synchronize(); tot = time.perf_counter() - t0
    print("HOSTPROFILE total %.2f ms" % (tot * 1e3))
    for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
        print("HOSTPROFILE %-32s %8.2f ms" % (k, v * 1e3))


# ---------------------------------------------------------------------------------------------- CPU baseline
def cpu_baseline(counts, sf, x, k, cfg, weights, factors, reduced):
    """CPU baseline on the first k genes of the same workload through the same host code, on the GPU box's host
    cores: 1 thread (what the reference is) AND all cores (OpenMP over genes).  kind "port" = the C oracle: the reference's
    src/DESeq2.cpp needs R, Rcpp and RcppArmadillo, which this image does not have, so it is not built here."""
    from deseq2_amd import core
    from deseq2_amd.engine import HostEngine
    from oracle import oracle as O
    ncores = os.cpu_count() or 1
    sub = counts[:k]
    keep = sub.sum(axis=1) > 0
    sub = sub[keep]
    w = None if weights is None else weights[:k][keep]
    kw = dict(test=cfg["test"], reduced=reduced)
    if cfg.get("betaPrior"):
        kw.update(betaPrior=True, factors=factors)
    if cfg.get("minmu"):
        kw.update(minmu=cfg["minmu"])

    # the all-core legs take a larger sample (4 x, up to the whole matrix): with a few thousand genes on a few hundred workers the time is the
    # worker start-up, not the fits
    sub_all = counts[: 4 * k]
    keep_all = sub_all.sum(axis=1) > 0
    sub_all = sub_all[keep_all]
    w_all = None if weights is None else weights[: 4 * k][keep_all]

    def run(fns, big=False):
        yy, ww = (sub_all, w_all) if big else (sub, w)
        t0 = time.perf_counter()
        core.DESeq(core.DESeqDataSet(yy, x, sizeFactors=sf, weights=ww, engine=HostEngine(fns)), **kw)
        return time.perf_counter() - t0
    O.set_threads(1)
    dt_port = run(O)
    O.set_threads(ncores)
    dt_port_all = run(O, big=True)
    what = "first %d genes of the same %d-sample matrix, full DESeq() chain" % (sub.shape[0], counts.shape[1])
    out = {"value": sub.shape[0] / dt_port, "unit": "genes/s", "cores": 1, "kind": "port",
           "sample": "%s over the C oracle, %.1f s" % (what, dt_port),
           "all_cores": {"value": sub_all.shape[0] / dt_port_all, "cores": ncores, "kind": "port",
                         "sample": "first %d genes, OpenMP over genes, %.1f s" % (sub_all.shape[0], dt_port_all)}}
    O.set_threads(1)
    return out


if __name__ == "__main__":
    main()
