"""Host logic of the XCD-aware workgroup order (vasp_amd/csrc/fsi_kernels.hpp: xcd_unit / xcd_span), no device.

k_residual and every tiled / row-block sweep of the preconditioner map their launch index to the unit they work on (a cell pair, a
tile of nodes, a block of rows) through this function: workgroups are dealt round-robin to the 8 XCDs of the chip, XCD k takes the
k-th eighth of the units.  A unit that no workgroup takes is a tile of the vector that a sweep never updates - on one mesh size in
eight, whichever the remainder hits - so the map is held here, for every n, to what the kernels assume: every unit exactly once,
the XCD of a logical workgroup is its index mod 8 (persistent kernels stride by a multiple of 8), consecutive units on one XCD."""
import numpy as np
import pytest

from vasp_amd import capi


def order(n):
    lib = capi.load_library()
    span = lib.fsi_xcd_order(n, None)
    out = np.full(max(span, 1), -7, dtype=np.int64)
    assert lib.fsi_xcd_order(n, capi._ptr(out)) == span
    return span, out[:span]


@pytest.mark.parametrize("n", list(range(0, 70)) + [255, 256, 257, 1000, 4093, 4096, 5917, 65535, 65537, 561601])
def test_every_unit_is_taken_exactly_once(n):
    span, unit = order(n)
    assert span % 8 == 0 and n <= span < n + 8
    taken = unit[unit >= 0]
    assert np.all(unit[unit < 0] == -1)
    assert len(taken) == n and np.array_equal(np.sort(taken), np.arange(n))


@pytest.mark.parametrize("n", [9, 64, 1000, 5917])
def test_an_xcd_works_through_one_contiguous_eighth(n):
    span, unit = order(n)
    chunk = (n + 7) // 8
    for k in range(8):
        mine = unit[k::8]                       # the logical workgroups the dispatcher hands to XCD k, in launch order
        mine = mine[mine >= 0]
        lo = k * chunk
        assert np.array_equal(mine, np.arange(lo, min(lo + chunk, n)))


def test_a_negative_count_is_refused():
    assert capi.load_library().fsi_xcd_order(-1, None) < 0
