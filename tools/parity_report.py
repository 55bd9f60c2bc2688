#!/usr/bin/env python
"""Measured agreement rates against the LAPACK restatement (oracle/lapack_oracle.py): `python tools/parity_report.py hip`
on the GPU box (HIP path through the host-pointer C ABI), `python tools/parity_report.py oracle` anywhere (the C oracle)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tests.test_oracle_vs_lapack as T   # noqa: E402
from oracle import lapack_oracle          # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "oracle"
if which == "hip":
    from deseq2_amd import native as F
else:
    from oracle import oracle as F
rows = []
cases = [(n, T.shape_case(n)) for n in T.SHAPE_NAMES] + list(T.live_cases().items())
for name, d in cases:
    st = T.compare(T.run_all(F, d), T.run_all(lapack_oracle, d), d, name)
    n, m = d["counts"].shape
    fb = st["fitBeta"]
    rows.append("| %s | %d x %d, p=%d | %d / %d | %s | %.3f / %.3f | %d / %d | %.1e / %.1e | %.3f |" % (
        name, n, m, d["x"].shape[1], fb["iter_mismatch"], fb["n"],
        ("%d / %d" % (st["fitBetaPrior"]["iter_mismatch"], st["fitBetaPrior"]["n"])) if "fitBetaPrior" in st else "-",
        st["fitDispMLE"]["well"], st["fitDispMAP"]["well"], st["fitDispMLE"]["ties"], st["fitDispMAP"]["ties"],
        st["fitDispMLE"]["max_rel_log_alpha"], st["fitDispMAP"]["max_rel_log_alpha"], st["fitDispGrid"]["same"]))
print("# Agreement of the %s path with the LAPACK restatement (oracle/lapack_oracle.py)\n" % which.upper())
print("Budgets asserted by the tests: fitBeta$iter equal on every gene; fitDisp iter / iter_accept equal outside "
      "ulp-level ties, ties <= 1 %; well-conditioned share >= 0.95 (m > 12); grid agreement >= 0.95; values within "
      "1e-7 (beta, log_alpha) / 1e-8 (lp, H, deviance).\n")
print("| case | shape | fitBeta$iter mismatches | betaPrior pass | well (MLE / MAP) | ties (MLE / MAP) | max rel log alpha | grid same |")
print("|---|---|---|---|---|---|---|---|")
print("\n".join(rows))
