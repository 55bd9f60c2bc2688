#!/usr/bin/env python
"""Timeline of one bench step (default: the last but one TIMED step, so that what follows it is another timed step) in a rocprofv3 rocpd database: every kernel in start order with its duration and
the idle gap before it.  usage: python tools/timeline.py x_results.db [n_steps_in_trace]"""
import sqlite3
import sys


def main(path):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select name, start, end, grid_x from kernels order by start").fetchall()
    # a step starts with the int32 R-layout -> gene-major conversion of the counts
    starts = [i for i, r in enumerate(rows) if "r_to_gm_kernel<int" in r[0]]
    which = int(sys.argv[2]) if len(sys.argv) > 2 else -4     # bench.py: warmup + K timed steps, then 2 passes with events
    lo = starts[which]
    hi = starts[which + 1] if which + 1 < 0 else len(rows)
    seg = rows[lo:hi]
    t0 = seg[0][1]
    prev_end = t0
    busy = 0.0
    big_gaps = []
    for name, s, e, g in seg:
        gap = (s - prev_end) / 1e3
        dur = (e - s) / 1e3
        busy += dur
        short = name.replace("void ", "").replace("dsq::", "")[:60]
        print("%9.1f us  gap %7.1f  dur %8.1f  grid %8d  %s" % ((s - t0) / 1e3, gap, dur, g, short))
        if gap > 15:
            big_gaps.append((gap, short))
        prev_end = max(prev_end, e)
    total = (prev_end - t0) / 1e3
    print("step span %.1f us, kernel busy %.1f us, idle %.1f us, launches %d" % (total, busy, total - busy, len(seg)))
    print("gaps > 15 us:", sorted(big_gaps, reverse=True)[:15])


if __name__ == "__main__":
    main(sys.argv[1])
