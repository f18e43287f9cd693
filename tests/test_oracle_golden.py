"""The CPU restatement (oracle/tsdf_oracle.c) against the golden vectors generated from the
reference's own kernel body: bit-exact TSDF and weight on every fixture."""
import numpy as np
import pytest

from golden_util import NAMES, Golden


def test_fixture_set_is_complete():
    assert len(NAMES) >= 7, NAMES


@pytest.mark.parametrize("name", NAMES)
def test_oracle_reproduces_golden(oracle, name):
    g = Golden(name)
    t, w = oracle.init_grid(g.dims)
    n_upd = 0
    for c2b, depth in g.frames:
        n_upd += oracle.integrate(g.K, c2b, depth, g.dims, g.origin, g.vs, g.trunc, t, w, threads=2)
    assert np.array_equal(w, g.weight)
    assert np.array_equal(t.view(np.uint32), g.tsdf.view(np.uint32))
    assert n_upd == int(g.weight.astype(np.float64).sum())  # each update adds exactly 1


@pytest.mark.parametrize("name", NAMES)
def test_oracle_slab_form_reproduces_golden(oracle, name):
    """Integrating three z-slabs separately equals the whole-grid golden (global z is used)."""
    g = Golden(name)
    dx, dy, dz = g.dims
    cuts = [0, dz // 3, dz // 3 + 1, dz]
    parts_t, parts_w = [], []
    for a, b in zip(cuts[:-1], cuts[1:]):
        t, w = oracle.init_grid(g.dims, a, b)
        for c2b, depth in g.frames:
            oracle.integrate(g.K, c2b, depth, g.dims, g.origin, g.vs, g.trunc, t, w, z_begin=a, z_end=b)
        parts_t.append(t)
        parts_w.append(w)
    assert np.array_equal(np.concatenate(parts_w), g.weight)
    assert np.array_equal(np.concatenate(parts_t).view(np.uint32), g.tsdf.view(np.uint32))
