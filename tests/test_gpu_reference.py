"""-m gpu: the HIP path against the REFERENCE's own outputs (tests/golden/reference_golden.npz, produced
by /root/reference/src/DESeq2.cpp compiled against oracle/shim/) -- the same comparison and tolerances
the oracle is held to in tests/test_oracle_vs_reference.py, through the host-pointer C ABI; and, when the
prebuilt oracle/_ref/libdeseq2_ref.so travelled with the snapshot, live on larger cases."""
import numpy as np
import pytest

from deseq2_amd import native
from tests.test_oracle_vs_reference import GOLDEN, _case, _have_ref, compare, golden_cases, run_all

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", sorted(golden_cases()))
def test_hip_reproduces_reference_golden(name):
    z = np.load(GOLDEN)
    d = golden_cases()[name]
    ref = {}
    for key in z.files:
        c, fn, k = key.split("/")
        if c == name:
            ref.setdefault(fn, {})[k] = z[key]
    compare(run_all(native, d), ref, d, name)


@pytest.mark.skipif(not _have_ref(), reason="oracle/_ref/libdeseq2_ref.so not present")
@pytest.mark.parametrize("n,m,design,kw", [
    (1500, 50, "batch_condition", {}),
    (400, 120, ("factor", 8), {}),
    (600, 30, "two_group", {"weights": True}),
    (300, 64, "batch_condition", {"weights": True, "zero_w": True, "useQR": False}),
])
def test_hip_vs_compiled_reference_live(n, m, design, kw):
    from oracle import reference
    d = _case(n, m, design, seed=n + m, **kw)
    compare(run_all(native, d), run_all(reference, d), d, "live %dx%d" % (n, m))
