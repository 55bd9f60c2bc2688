#!/usr/bin/env python
"""Summarise rocprofv3 --pmc passes (csv) for the dsq kernels into profiles/rNN_pmc.json.
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KB; on gfx950 FETCH_SIZE counts 64 B per
128-B request for wide coalesced reads (MI355X_MICROARCH.md, HBM section) and is doubled here."""
import json
import sys

import pandas as pd


def load(d):
    df = pd.read_csv(d + "/p_counter_collection.csv")
    df = df[df["Kernel_Name"].str.contains("dsq::")].copy()
    df["k"] = df["Kernel_Name"].str.extract(r"dsq::(\w+?)_kernel")
    # full-size launches only (the outlier refit and the straggler passes re-launch the same kernels on a handful of
    # rows, sometimes with the same grid): joined with the kernel trace of the SAME pass on the dispatch id, a launch
    # counts when it ran for at least half of the kernel's longest launch (round 2 averaged both kinds together)
    try:
        kt = pd.read_csv(d + "/p_kernel_trace.csv")
        kt["dur"] = kt["End_Timestamp"] - kt["Start_Timestamp"]
        df = df.merge(kt[["Dispatch_Id", "dur"]], on="Dispatch_Id", how="left")
        df = df[df["dur"] >= 0.5 * df.groupby("k")["dur"].transform("max")]
    except (OSError, KeyError):
        df = df[df["Grid_Size"] == df.groupby("k")["Grid_Size"].transform("max")]
    return df.groupby(["k", "Counter_Name"])["Counter_Value"].mean().unstack()


def main(fetch_dir, write_dir, sq_dir, out):
    f, w, s = load(fetch_dir), load(write_dir), load(sq_dir)
    kt = pd.read_csv(sq_dir + "/p_kernel_trace.csv")
    kt = kt[kt["Kernel_Name"].str.contains("dsq::")].copy()
    kt["k"] = kt["Kernel_Name"].str.extract(r"dsq::(\w+?)_kernel")
    kt["ms"] = (kt["End_Timestamp"] - kt["Start_Timestamp"]) / 1e6
    kt = kt[kt["ms"] >= 0.5 * kt.groupby("k")["ms"].transform("max")]
    # further SQ passes (the f64 operation mix: SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64; int / cvt / smem): directories in
    # DSQ_PMC_EXTRA, ':'-separated; their columns join the SQ table
    import os
    for d in [e for e in os.environ.get("DSQ_PMC_EXTRA", "").split(":") if e and e != "."]:
        try:
            x = load(d)
            s = s.join(x[[c for c in x.columns if c not in s.columns]], how="left")
        except (OSError, KeyError, ValueError) as e:
            print("pmc_summary: extra pass %s skipped: %r" % (d, e))
    res = {}
    for k in s.index:
        r = {"fetch_bytes_per_launch": float(f.loc[k, "FETCH_SIZE"]) * 1024 * 2 if k in f.index else None,
             "fetch_size_raw_kb": float(f.loc[k, "FETCH_SIZE"]) if k in f.index else None,
             "write_bytes_per_launch": float(w.loc[k, "WRITE_SIZE"]) * 1024 if k in w.index else None,
             "avg_ms_under_pmc": float(kt[kt["k"] == k]["ms"].mean())}
        for c in s.columns:
            r[c] = float(s.loc[k, c])
        if r["fetch_bytes_per_launch"] is not None and r["write_bytes_per_launch"] is not None:
            r["hbm_bytes_per_launch"] = r["fetch_bytes_per_launch"] + r["write_bytes_per_launch"]
        # SQ_* cycle counters are in quad-cycles summed over SIMDs
        if "SQ_ACTIVE_INST_VALU" in r and "SQ_WAVE_CYCLES" in r and r["SQ_WAVE_CYCLES"]:
            r["valu_active_frac_of_wave_cycles"] = r["SQ_ACTIVE_INST_VALU"] / r["SQ_WAVE_CYCLES"]
        # the dynamic f64 mix: wave-instructions by class; flops = (add + mul + 2 fma) x 64 lanes (an upper bound: masked
        # lanes count); "useful" share of the VALU issue slots
        if all(("SQ_INSTS_VALU_%s_F64" % c) in r for c in ("ADD", "MUL", "FMA", "TRANS")) and r.get("SQ_INSTS_VALU"):
            a, mu_, fm, tr = (r["SQ_INSTS_VALU_%s_F64" % c] for c in ("ADD", "MUL", "FMA", "TRANS"))
            r["f64_arith_insts_per_launch"] = a + mu_ + fm + tr
            r["f64_arith_frac_of_valu"] = (a + mu_ + fm + tr) / r["SQ_INSTS_VALU"]
            r["f64_flops_per_launch"] = 64.0 * (a + mu_ + 2.0 * fm + tr)
        res[k] = r
    if "fit_beta_cell" in res and "fit_beta" not in res:
        res["fit_beta"] = dict(res["fit_beta_cell"])       # factor designs run the cell-collapsed fitBeta kernel
    if len(sys.argv) > 5:
        res["_genes_per_launch"] = int(sys.argv[5])
        res["_note"] = sys.argv[6] if len(sys.argv) > 6 else ""
    # which build of the library the passes ran on: bench.py uses the counters only for a library with this digest
    import hashlib
    so = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "deseq2_amd", "libdeseq2_mi355x.so")
    h = hashlib.sha256()
    with open(so, "rb") as fh:
        for blk in iter(lambda: fh.read(1 << 22), b""):
            h.update(blk)
    res["_library_sha256"] = h.hexdigest()
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:5])
