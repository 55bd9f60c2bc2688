#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd database (the default output of `rocprofv3 --kernel-trace
--stats` on ROCm 7.2) into the per-kernel table kept under profiles/.
usage: python profiles/summarize_rocpd.py gpurun_out/prof/x_results.db > profiles/rNN_kernel_stats.md"""
import sqlite3

NOTE = ("(* `arch vgpr`: the vgpr_count rocprofv3 records for the dispatch (architectural registers). The code object's own "
        ".vgpr_count -- the unified allocation that bounds the occupancy on gfx950, accumulation registers included -- is "
        "listed in profiles/rNN_isa_mix.md (fit_disp<4>: 84 here, 166 there).)")
import sys


def main(path, top=18):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute(
        "select name, count(*), sum(end-start)/1e6, avg(end-start)/1e6, min(end-start)/1e6, max(end-start)/1e6, "
        "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), "
        "max(grid_x), max(workgroup_x) from kernels group by name, grid_x order by 3 desc").fetchall()
    # one row per (kernel, grid): DESeq()'s outlier refit re-launches the fit kernels on a handful of rows
    tot = sum(r[2] for r in rows)
    print(NOTE + "\n")
    print("| kernel | calls | total ms | avg ms | min ms | max ms | % GPU time | arch vgpr* | agpr | sgpr | LDS B | scratch B | grid | wg |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    for r in rows[:top]:
        print("| `%s` | %d | %.3f | %.3f | %.3f | %.3f | %.1f | %s | %s | %s | %s | %s | %s | %s |" %
              ((r[0][:90], r[1], r[2], r[3], r[4], r[5], 100 * r[2] / tot) + tuple(r[6:13])))
    rest = rows[top:]
    print("| (%d other kernels: torch glue, copies) | %d | %.3f | | | | %.1f | | | | | | | |" %
          (len(rest), sum(r[1] for r in rest), sum(r[2] for r in rest), 100 * sum(r[2] for r in rest) / tot))
    print("\ntotal GPU kernel time: %.3f ms" % tot)


if __name__ == "__main__":
    main(sys.argv[1])
