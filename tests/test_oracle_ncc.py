"""CPU: the C restatement (oracle/ncc_oracle.c) against the golden vectors produced by the compiled reference."""
import numpy as np
import pytest

from oracle import ncc_oracle as N
from tests.golden_util import case_inputs


def _names(g):
    return [str(n) for n in g["names"]]


def test_golden_has_cases(ncc_golden):
    assert len(_names(ncc_golden)) >= 10


@pytest.mark.parametrize("idx", range(13))
def test_oracle_matches_reference_golden(ncc_golden, idx):
    g = ncc_golden
    name = _names(g)[idx]
    A, B, overlap, side, dmax = case_inputs(g, name)
    r = N.pdalgo_execute(A, B, dmax[0], dmax[1], dmax[2], side, overlap, kind="oracle", debug=True)
    assert r["rc"] == 0
    assert r["coord"] == list(g[f"{name}/coord"])
    assert r["NCC_widths"] == list(g[f"{name}/NCC_widths"])
    assert r["wRangeThr"] == list(g[f"{name}/wRangeThr"])
    assert r["delays"] == list(g[f"{name}/delays_ijk"])
    assert r["INF_W"] == int(g[f"{name}/INF_W"])
    # bit-for-bit, NaNs included
    assert np.array_equal(r["NCC_maxs"].view(np.uint32), g[f"{name}/NCC_maxs"].view(np.uint32))
    for m, nm in enumerate(["xy1", "xz1", "yz1", "xy2", "xz2", "yz2"]):
        assert np.array_equal(r["mips"][m], g[f"{name}/mip_{nm}"])
    for m, nm in enumerate(["xy", "xz", "yz"]):
        assert np.array_equal(r["maps"][m].view(np.uint32), g[f"{name}/map_{nm}"].view(np.uint32)), nm


@pytest.mark.skipif(not N.have_ref(), reason="oracle/_ref not built (reference sources absent)")
def test_oracle_matches_live_reference_random():
    rng = np.random.default_rng(99)
    for trial in range(4):
        side = trial % 2
        shift = tuple(int(v) for v in rng.integers(-4, 5, size=3))
        A, B = N.tile_pair((28, 112, 104), 36, side, shift, seed=100 + trial)
        r1 = N.pdalgo_execute(A, B, 8, 8, 3, side, 36, kind="ref")
        r2 = N.pdalgo_execute(A, B, 8, 8, 3, side, 36, kind="oracle")
        assert r1["coord"] == r2["coord"] and r1["NCC_widths"] == r2["NCC_widths"] and r1["wRangeThr"] == r2["wRangeThr"]
        assert np.array_equal(r1["NCC_maxs"].view(np.uint32), r2["NCC_maxs"].view(np.uint32))


def test_wrange_precondition_is_an_error():
    A, B = N.tile_pair((26, 64, 64), 30, 1, (0, 0, 0), seed=1)
    p = N.pdalgo_params(10, 10, 10)
    r = N.norm_cross_corr_mips(A, B, 0, 34, 2, 10, 10, 1, p, kind="oracle")  # wRangeThr_k = 10 > delayk = 2
    assert r["rc"] == -1  # libcrossmips.cpp:212-219 throws


def test_argmax_first_strict_and_leading_nan():
    import ctypes as C
    lib = N._lib("oracle")
    v = np.array([1.0, 3.0, 3.0, 2.0], np.float32)
    assert lib.orc_argmax(v.ctypes.data_as(C.POINTER(C.c_float)), 4) == 1
    v = np.array([np.nan, 3.0, 5.0], np.float32)
    assert lib.orc_argmax(v.ctypes.data_as(C.POINTER(C.c_float)), 3) == 0  # compute_funcs.cu:1294-1305
