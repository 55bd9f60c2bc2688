"""-m gpu: the HIP path against the REFERENCE's own outputs (tests/golden/reference_golden.npz, produced
by /root/reference/src/DESeq2.cpp compiled against oracle/shim/) -- the same comparison and tolerances
the oracle is held to in tests/test_oracle_vs_reference.py, through the host-pointer C ABI; and, when the
prebuilt oracle/_ref/libdeseq2_ref.so travelled with the snapshot, live on larger cases."""
import numpy as np
import pytest

from deseq2_amd import native
from tests.test_oracle_vs_reference import (GOLDEN, HEAD, SHAPES, SHAPE_NAMES, _case, _close, _have_ref, compare,
                                             golden_cases, load_golden, run_all, shape_case)

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


@pytest.mark.skipif(not _have_ref(), reason="oracle/_ref/libdeseq2_ref.so not present")
def test_hip_vs_compiled_reference_beyond_the_resident_waves():
    """9 000 genes x 100 samples, ~ batch + condition: three times the 3 072 wave slots the persistent fit kernels keep
    resident, so every wave fits several genes in a row under the reference's eye.  Against the
    libm-double build of the reference's src/DESeq2.cpp (10 s of CPU; the binary128 build would take minutes): fitBeta$iter
    equal on every gene, fitDisp iterations equal outside <= 1 % ulp-level ties, values within 1e-6 / 1e-7."""
    from oracle import reference
    reference.use_fast(True)
    try:
        d = _case(9000, 100, "batch_condition", seed=9100)
        st = compare(run_all(native, d), run_all(reference, d), d, "live 9000x100 (libm-double reference build)", scale=10)
    finally:
        reference.use_fast(False)
    assert st["fitBeta"]["iter_mismatch"] == 0 and st["fitDispGrid"]["same"] >= 0.99


@pytest.mark.parametrize("name", SHAPE_NAMES)
def test_hip_reproduces_reference_at_baseline_shapes(name):
    """BASELINE.json configs C2..C5 at full shape (C3: 1000 x 500 p=4; C4: 200 x 2000, 10-level factor, QR; C5:
    1000 x 200, weights with zeros + the betaPrior pass on the expanded p=3 design; C2: 1000 x 100): the HIP path
    through the host-pointer C ABI against the outputs of the reference's own src/DESeq2.cpp
    (tests/golden/reference_shapes.npz, generator tests/golden/make_reference_shapes.py) -- iteration counts equal
    (ties <= 1 %), well-conditioned share >= 0.95, grid agreement >= 0.95, values within 1e-7 / 1e-8."""
    d = shape_case(name)
    got = run_all(native, d)
    ref = load_golden(SHAPES, name)
    _close(got["aux"]["mu"][:HEAD], ref["aux"]["mu"], name + " mu", rtol=1e-9)
    compare(got, ref, d, name, min_well=0.95)


@pytest.mark.parametrize("seed", (5, 6, 7))
def test_hip_floor_regime_vs_reference(seed):
    """the dispersion-floor regime (Poisson / NB mixture, > 25 % of the genes start at alpha_0 = 1e-8): everything R's
    callers see -- fitBeta$iter, clamped dispGeneEst, dispGeneEstConv / refitDisp, MAP dispConv and dispMAP -- against
    the compiled reference's stored outputs, with the budgets the reference's own libm build meets
    (tests/floor_regime.py, tests/golden/reference_floor.npz)"""
    from tests.floor_regime import assert_visible_parity, floor_case, load_floor_golden, visible_chain
    from tests.test_floor_regime import GOLDEN as FLOOR
    assert_visible_parity(visible_chain(native, floor_case(seed)), load_floor_golden(FLOOR, seed), "hip seed %d" % seed)
