"""CPU: the C oracle in the dispersion-floor regime against the LAPACK restatement (see tests/floor_regime.py for what
is compared and why)."""
import pytest

from tests.floor_regime import SEEDS, assert_visible_parity, floor_case, rates, visible_chain


@pytest.mark.parametrize("seed", SEEDS)
def test_oracle_floor_regime_vs_lapack(oracle, seed):
    from oracle import lapack_oracle
    d = floor_case(seed)
    ref = visible_chain(lapack_oracle, d)
    got = visible_chain(oracle, d)
    s = assert_visible_parity(got, ref, "oracle seed %d" % seed)
    # the premise of the module: on the floor genes the step COUNT is rounding noise of the special functions underneath
    # (two correct implementations agree on few of them) while everything R's callers see agrees
    assert s["iter_equal_floor"] < 0.5, rates(got, ref)
