#!/usr/bin/env python
"""bench.py -- genes/sec of the full DESeq() dispersion + beta + Wald fit on MI355X.

One "step" = one pass of the hot path over one synthetic count matrix that is already
resident in HBM in R's layout (column-major int32 counts, f64 normalization-factor matrix):
layout conversion -> prefit moments -> fitBeta (mu-hat) -> fitDisp -> fitDispGrid (stragglers) -> dispersion
trend (device kernel) / prior variance on n-vectors -> fitDisp (MAP) -> fitDispGrid -> fitBeta (final
dispersions) -> logLik, Wald statistics and p-values -> Cook's distances -> replaceOutliers -> refit of the
replaced rows: everything DESeq() does by default.  Workload at every N: BASELINE.json configs[2]
(50k genes x 500 samples, ~batch+condition, p = 4) PER GPU (weak scaling: genes shard
across ranks, no data-path collective; the only exchange is the all-gather of two n-vectors
for the global dispersion trend, as in DESeqParallel).

Prints ONE JSON line (rank 0).  See DESIGN.md "Measurement" for the roofline accounting.
"""
import argparse
import json
import os
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
F64_VALU_PEAK_TFLOPS = 78.6    # public MI355X spec (vector f64); reported for context only


def algorithmic_bytes_per_gene(kernel, m, nf_matrix=True, weights=False, hat=True, mu=False):
    """fitBeta reads Y (4m) [+ nf matrix 8m] [+ weights 8m], writes
    H (8m) [+ mu (8m)]; fitDisp reads Y (4m) + mu-hat (8m) [+ weights 8m]."""
    if kernel == "fit_beta":
        return 4 * m + (8 * m if nf_matrix else 0) + (8 * m if weights else 0) + (8 * m if hat else 0) + (8 * m if mu else 0)
    return 4 * m + 8 * m + (8 * m if weights else 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--genes", type=int, default=50000)
    ap.add_argument("--samples", type=int, default=500)
    ap.add_argument("--chunks", type=int, default=int(os.environ.get("DSQ_BENCH_CHUNKS", "1")),
                    help="gene chunks per GPU, each on its own HIP stream + host thread (1 = serial DESeq(), the "
                         "default: measured on MI355X, 2-4 chunk threads are 5-35 %% SLOWER -- the per-call host "
                         "cost is fixed, not per gene, and the interpreter serialises it; see DESIGN.md)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-genes", type=int, default=4096)
    ap.add_argument("--profile-host", action="store_true", help="print wall time per pipeline phase (adds syncs)")
    args = ap.parse_args()

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
        if one_dev:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)

    from deseq2_amd import core, simulate, parallel
    from deseq2_amd.engine import DeviceEngine

    n_req, m = args.genes, args.samples
    x = simulate.design_batch_condition(m)              # ~batch(3) + condition(2): p = 4
    p = x.shape[1]
    d = simulate.make_counts(n_req, x, seed=1 + rank)
    counts = d["counts"]
    n = counts.shape[0]
    E = DeviceEngine(dev)
    # inputs resident in HBM in R layout before the timed region
    counts_r = torch.as_tensor(np.ascontiguousarray(counts.T), device=dev)                 # (m, n) int32
    nf_r = torch.ones((m, n), dtype=torch.float64, device=dev) * torch.as_tensor(d["size_factors"], device=dev)[:, None]
    torch.cuda.synchronize()

    pipe = parallel.Pipeline(E, n_chunks=args.chunks, comm_device=comm_dev) if args.chunks > 1 else None

    def make_dds(lo, hi):
        return core.DESeqDataSet.from_device(E, counts_r[:, lo:hi].contiguous(), nf_r[:, lo:hi].contiguous(), x,
                                             sizeFactors=d["size_factors"])

    def step():
        """one DESeq() over this rank's genes; returns the chunk results (one DESeqDataSet per chunk)"""
        if pipe is not None:
            return pipe.run(make_dds, n)
        dds = core.DESeqDataSet.from_device(E, counts_r, nf_r, x, sizeFactors=d["size_factors"])
        if world > 1:
            parallel.DESeqParallel(dds, comm_device=comm_dev)
        else:
            core.DESeq(dds)
        return [dds]

    for _ in range(args.warmup):
        step()

    if args.profile_host and rank == 0:
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
        for nm in ("estimateDispersionsGeneEst", "estimateDispersionsFit", "estimateDispersionsPriorVar",
                   "estimateDispersionsMAP", "nbinomWaldTest", "fitNbinomGLMs", "getBaseMeansAndVariances",
                   "parametricDispersionFit"):
            wrap(core, nm)
        for nm in ("fit_beta", "fit_disp", "fit_disp_grid", "prefit", "nbinom_loglike", "two_sided_normal_p",
                   "take_rows"):
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
        return

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    dds = None
    for _ in range(args.steps):
        dds = None          # release the previous step's HBM tensors before allocating the next ones
        dds = step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        cdev = dev if comm_dev is not None else torch.device("cpu")
        tt = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        nn = torch.tensor([n], dtype=torch.int64, device=cdev)
        dist.all_reduce(nn, op=dist.ReduceOp.SUM)
        n_total = int(nn.item())
    else:
        n_total = n
    # per-kernel launch durations: one extra UNTIMED pass with HIP events around each fit kernel
    # (recorded inside the C library on the launch stream; reading them back synchronises, so this
    # pass is kept out of the throughput measurement)
    E.record = []
    for _ in range(2):
        step()
    rec, E.record = E.record, None

    if rank == 0:
        # ---- per-kernel launch durations (HIP events on the launch stream) -------------
        per = {}
        for name, ng, ms in rec:
            per.setdefault(name, []).append((ng, ms))
        def summary(sel):
            out = {}
            for k, v in per.items():
                v = [(g, t) for g, t in v if sel(g)]
                if v:
                    out[k] = {"launches": len(v), "avg_ms": float(np.mean([t for _, t in v])),
                              "genes_per_launch": float(np.mean([g for g, _ in v]))}
            return out
        big = n // (2 * max(1, args.chunks))
        kern = summary(lambda g: g >= big)                    # the full-size (chunk) launches of the chain
        kern_refit = summary(lambda g: g < big)               # refitWithoutOutliers: the replaced rows only
        # the two full-size kernels; dominant = larger share of the step
        share = {k: sum(t for _, t in per[k]) for k in per}
        dom = max(("fit_beta", "fit_disp"), key=lambda k: share.get(k, 0.0))
        nfull = max(g for g, _ in per[dom])
        full = [(g, t) for g, t in per[dom] if g == nfull]
        avg_ms = float(np.mean([t for _, t in full]))
        bytes_per_gene = algorithmic_bytes_per_gene(dom, m, nf_matrix=True, weights=False,
                                                    hat=(dom == "fit_beta"), mu=(dom == "fit_beta"))
        if dom == "fit_beta":
            # fitBeta#1 writes mu (no H), fitBeta#2 writes mu and H: average of the two launches
            bytes_per_gene = (algorithmic_bytes_per_gene("fit_beta", m, hat=False, mu=True) +
                              algorithmic_bytes_per_gene("fit_beta", m, hat=True, mu=True)) / 2.0
        achieved = bytes_per_gene * nfull / (avg_ms * 1e-3) / 1e9
        # HBM bytes per launch from the PMC counters: collected in separate rocprofv3 --pmc passes of
        # this same command (FETCH_SIZE doubled per the gfx950 note, + WRITE_SIZE) and committed
        # under profiles/ -- counters cannot be read from inside the process.
        traffic = None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc.json")))
            if n_req == 50000 and m == 500:     # PMC passes ran this workload; scale to the genes of one launch
                traffic = pmc[dom]["hbm_bytes_per_launch"] * nfull / pmc.get("_genes_per_launch", 50000)
        except (OSError, KeyError, ValueError):
            pass
        roofline = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                    "algorithmic_bytes_per_launch": bytes_per_gene * nfull, "genes_per_launch": nfull,
                    "avg_launch_ms": avg_ms,
                    "note": "f64-VALU/transcendental bound, not HBM bound (DESIGN.md); see profiles/"}

        mc = parallel.concat_mcols(dds, ["betaIter", "dispIter", "dispGeneIter"])
        # secondary view: the path is bound by f64 VALU issue, so also report the dominant kernel against the
        # VALU issue peak: wave-instructions per launch (PMC SQ_INSTS_VALU of the committed passes, scaled to the
        # genes of one launch) / live launch time, vs CUs x 4 SIMDs x clock / 4 cycles per 64-lane f64 instruction
        valu = None
        try:
            insts = pmc[dom]["SQ_INSTS_VALU"] * nfull / pmc.get("_genes_per_launch", 50000)
            prop = torch.cuda.get_device_properties(dev)
            clock_hz = float(getattr(prop, "clock_rate", 2400000)) * 1e3
            peak = prop.multi_processor_count * 4 * clock_hz / 4.0
            ach = insts / (avg_ms * 1e-3)
            valu = {"kernel": dom, "achieved": ach / 1e9, "peak": peak / 1e9, "unit": "G wave-instr/s",
                    "frac": ach / peak, "valu_instructions_per_launch": insts,
                    "note": "SQ_INSTS_VALU from profiles/r01_pmc.json (same workload), launch time measured live"}
        except (NameError, KeyError, TypeError, ValueError):
            pass
        it_beta = float(np.mean(mc["betaIter"]))
        it_disp = float(np.mean(mc["dispIter"]))
        out = {
            "metric": "genes/sec for DESeq() disp+beta+Wald fit, 50k x 500 x p=4",
            "value": n_total * args.steps / dt,
            "unit": "genes/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]: %d genes x %d samples per GPU, ~batch+condition (p=%d), "
                                   "Wald test; inputs resident in HBM in R layout (int32 counts, f64 nf matrix)"
                                   % (n, m, p),
                       "genes_per_gpu": n, "samples": m, "p": p,
                       "parallelism": "gene-shard x%d, %d chunk stream(s) per GPU" % (world, max(1, args.chunks))},
            "roofline": roofline,
            "valu_roofline": valu,
            "kernels": kern,
            "kernels_outlier_refit": kern_refit,
            "mean_iterations": {"fitBeta_final": it_beta, "fitDisp_MAP": it_disp,
                                "fitDisp_geneEst": float(np.mean(mc["dispGeneIter"]))},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(counts, d["size_factors"], x, args.cpu_sample_genes)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


class _ReferenceFns:
    """fns module for HostEngine: the three native routines from the REFERENCE's own src/DESeq2.cpp
    (oracle/_ref/libdeseq2_ref_fast.so: compiled against the stand-in headers of oracle/shim/, special
    functions in plain double), everything the reference does in R around them from the C oracle."""

    def __init__(self, R, O):
        self.R, self.O = R, O
        for name in ("prefitMoments", "nbinomLogLike", "parametricDispersionFit", "cooksDistance", "replaceOutliers",
                     "design_qr"):
            setattr(self, name, getattr(O, name))
        self.fitDisp, self.fitDispGrid = R.fitDisp, R.fitDispGrid

    def fitBeta(self, y, x, nf, alpha_hat, contrast, beta_mat, lam, w, useWeights, tol, maxit, useQR, minmu,
                want_mu=False, mu_floor=0.0, want_hat=True):
        r = self.R.fitBeta(y, x, nf, alpha_hat, contrast, beta_mat, lam, w, useWeights, tol, maxit, useQR, minmu)
        if want_mu:
            r["mu"] = self.O.fittedMu(x, nf, r["beta_mat"], mu_floor)
        return r


def cpu_baseline(counts, sf, x, k):
    """CPU baseline, 1 thread like the reference, on the first k genes of the same workload through the
    same host code: kind "reference" = the reference's own C++ source for fitBeta/fitDisp/fitDispGrid
    (when the prebuilt oracle/_ref library is present), else kind "port" = the C oracle."""
    from deseq2_amd import core
    from deseq2_amd.engine import HostEngine
    from oracle import oracle as O
    O.set_threads(1)
    sub = counts[:k]
    sub = sub[sub.sum(axis=1) > 0]

    def run(fns):
        t0 = time.perf_counter()
        core.DESeq(core.DESeqDataSet(sub, x, sizeFactors=sf, engine=HostEngine(fns)))
        return time.perf_counter() - t0
    dt_port = run(O)
    out = {"value": sub.shape[0] / dt_port, "unit": "genes/s", "cores": 1, "kind": "port",
           "sample": "first %d genes of the same %d-sample matrix, full DESeq() chain over the C oracle, %.1f s"
                     % (sub.shape[0], counts.shape[1], dt_port)}
    try:
        from oracle import reference as R
        R.use_fast(True)
        dt_ref = run(_ReferenceFns(R, O))
        out = {"value": sub.shape[0] / dt_ref, "unit": "genes/s", "cores": 1, "kind": "reference",
               "sample": "first %d genes of the same %d-sample matrix, full DESeq() chain; fitBeta/fitDisp/fitDispGrid "
                         "= the reference's src/DESeq2.cpp compiled against stand-in Rcpp/Armadillo headers (libm "
                         "special functions), the R-side steps in C, %.1f s" % (sub.shape[0], counts.shape[1], dt_ref),
               "port_value": sub.shape[0] / dt_port}
    except (OSError, ImportError):
        pass
    return out


if __name__ == "__main__":
    main()
