#!/usr/bin/env python
"""profiles/r03_parity.md (r02_parity.md in round 2): measured parity rates against the compiled reference's stored outputs
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

# ---- the dispersion-floor regime (tests/floor_regime.py): what R's callers see, next to the reference's own
# libm-double build against its binary128 build
from tests import floor_regime as FR             # noqa: E402
from tests.test_floor_regime import GOLDEN as FLOOR   # noqa: E402
print("\n## Dispersion-floor regime (500 x 60, p = 4, NB / Poisson mixture; tests/floor_regime.py)\n")
print("`floor` = genes whose start or final dispersion is below 1e-5.  Left: the %s path against the compiled reference "
      "(binary128 special functions); right: the reference's OWN libm-double build against the same.\n" % which.upper())
cols = ("floor_start", "floor", "beta_iter_mismatch", "iter_equal_floor", "iter_equal_rest", "conv_differs", "refit_differs",
        "dge_rel_gt_1e6", "dge_abs_max", "map_conv_differs", "map_iter_equal", "map_rel_max")
print("| seed | who | genes at alpha_0 = 1e-8 | floor genes | fitBeta$iter mismatches | fitDisp$iter equal (floor) | "
      "fitDisp$iter equal (rest) | dispGeneEstConv differs | refitDisp differs | dispGeneEst > 1e-6 rel. | max abs diff "
      "dispGeneEst | MAP dispConv differs | MAP iter equal | max rel. dispMAP |")
print("|" + "---|" * 14)
for seed in FR.SEEDS:
    ref = FR.load_floor_golden(FLOOR, seed)
    for who, got in ((which, FR.visible_chain(F, FR.floor_case(seed))), ("ref_fast", FR.load_floor_golden(FLOOR, seed, "ref_fast"))):
        s = FR.assert_visible_parity(got, ref, who)
        print("| %d | %s | " % (seed, who) + " | ".join(("%.3g" % s[c]) if isinstance(s[c], float) else str(s[c]) for c in cols) + " |")
