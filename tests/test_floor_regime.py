"""CPU: the oracle in the dispersion-floor regime against the compiled reference's stored outputs
(tests/golden/reference_floor.npz; see tests/floor_regime.py for what is compared and why)."""
import os

import pytest

from tests.floor_regime import SEEDS, assert_visible_parity, floor_case, load_floor_golden, rates, visible_chain

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "reference_floor.npz")


@pytest.mark.parametrize("seed", SEEDS)
def test_oracle_floor_regime_vs_reference(oracle, seed):
    ref = load_floor_golden(GOLDEN, seed)
    assert_visible_parity(visible_chain(oracle, floor_case(seed)), ref, "oracle seed %d" % seed)


@pytest.mark.parametrize("seed", SEEDS)
def test_reference_disagrees_with_itself_at_the_floor(seed):
    """the premise: the reference's own libm-double build matches its binary128 build on hardly any floor-start gene's
    iteration count, while everything R's callers see still agrees -- the same budgets the engine is held to"""
    ref, fast = load_floor_golden(GOLDEN, seed), load_floor_golden(GOLDEN, seed, "ref_fast")
    s = assert_visible_parity(fast, ref, "ref_fast seed %d" % seed)
    assert s["iter_equal_floor"] < 0.2


def test_floor_golden_is_current():
    from oracle import reference
    if not reference.available():
        pytest.skip("oracle/_ref/libdeseq2_ref.so not built (needs /root/reference)")
    ref = load_floor_golden(GOLDEN, SEEDS[0])
    live = visible_chain(reference, floor_case(SEEDS[0]))
    s = rates(live, ref)
    assert s["beta_iter_mismatch"] == 0 and s["iter_equal_floor"] == 1.0 and s["dge_abs_max"] == 0.0
