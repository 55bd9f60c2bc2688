#!/usr/bin/env python
"""profiles/r02_parity.md: measured parity rates against the compiled reference's stored outputs
(tests/golden/reference_golden.npz, reference_shapes.npz).  `python tools/parity_report.py hip` on the GPU box
(HIP path through the host-pointer C ABI), `python tools/parity_report.py oracle` anywhere (the C oracle)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tests.test_oracle_vs_reference as T   # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "oracle"
if which == "hip":
    from deseq2_amd import native as F
else:
    from oracle import oracle as F
rows = []
cases = [(n, T.shape_case(n), T.SHAPES) for n in T.SHAPE_NAMES] + [(n, d, T.GOLDEN) for n, d in T.golden_cases().items()]
for name, d, path in cases:
    st = T.compare(T.run_all(F, d), T.load_golden(path, name), d, name)
    n, m = d["counts"].shape
    fb = st["fitBeta"]
    rows.append("| %s | %d x %d, p=%d | %d / %d | %s | %.3f / %.3f | %d / %d | %.1e / %.1e | %.3f |" % (
        name, n, m, d["x"].shape[1], fb["iter_mismatch"], fb["n"],
        ("%d / %d" % (st["fitBetaPrior"]["iter_mismatch"], st["fitBetaPrior"]["n"])) if "fitBetaPrior" in st else "-",
        st["fitDispMLE"]["well"], st["fitDispMAP"]["well"], st["fitDispMLE"]["ties"], st["fitDispMAP"]["ties"],
        st["fitDispMLE"]["max_rel_log_alpha"], st["fitDispMAP"]["max_rel_log_alpha"], st["fitDispGrid"]["same"]))
print("# Parity of the %s path against the compiled reference (src/DESeq2.cpp over oracle/shim)\n" % which.upper())
print("Budgets asserted by the tests: fitBeta$iter equal on every gene; fitDisp iter / iter_accept equal outside "
      "ulp-level ties, ties <= 1 %; well-conditioned share >= 0.95 (m > 12); grid agreement >= 0.95; values within "
      "1e-7 (beta, log_alpha) / 1e-8 (lp, H, deviance).\n")
print("| case | shape | fitBeta iter mismatches | prior-pass iter mismatches | well share (MLE / MAP) | ties (MLE / MAP) | "
      "max rel. err log_alpha (MLE / MAP) | grid identical |")
print("|---|---|---|---|---|---|---|---|")
print("\n".join(rows))
