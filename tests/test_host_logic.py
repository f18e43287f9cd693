"""Host-side logic of the library that needs no device (the -m "not gpu" share of the product): the wavefront brick the
library chooses per grid (csrc/tsdf_capi.hip, choose_brick_for)."""
import numpy as np
import pytest

from semantic_slam_amd import capi


def shape_of(dims, z_range=None):
    cfg = capi.make_config(dims, 0.004, [0, 0, 1])
    if z_range is not None:
        cfg.z_begin, cfg.z_end = z_range
    return capi.default_brick_shape(cfg)


def test_default_brick_shape_is_valid_for_every_row_length():
    for dim_x in range(1, 1100):
        q, r, s = shape_of((dim_x, 37, 29))
        if dim_x % 4:
            assert (q, r, s) == (0, 0, 0), dim_x            # scalar kernel: no brick view
            continue
        quads = dim_x // 4
        assert q >= 1 and quads % q == 0 and 1 <= q * r * s <= 64 and r <= 37 and s <= 29, (dim_x, q, r, s)
        assert q * r * s >= 48, (dim_x, q, r, s)            # at most a quarter of the lanes idle


@pytest.mark.parametrize("dims,want", [((512, 512, 512), (2, 4, 8)), ((1024, 1024, 1024), (2, 4, 8)), ((200, 200, 200), (2, 4, 8)),
                                       ((2048, 2048, 256), (2, 4, 8)), ((36, 20, 12), (3, 3, 7))])
def test_default_brick_shape_of_the_configured_grids(dims, want):
    """Even quad counts: 8 x 4 x 8 voxels; 36-voxel rows (9 quads): 12 x 3 x 7."""
    assert shape_of(dims) == want


def test_default_brick_shape_respects_thin_slabs_and_short_columns():
    assert shape_of((512, 512, 512), (100, 101))[2] == 1                      # a one-slice slab: planar bricks
    q, r, s = shape_of((512, 512, 512), (0, 2))
    assert s <= 2 and q * r * s <= 64
    q, r, s = shape_of((512, 2, 512))
    assert r <= 2 and q * r * s <= 64 and 128 % q == 0
    q, r, s = shape_of((4, 1, 1))
    assert (q, r, s) == (1, 1, 1)
    with pytest.raises(capi.TsdfError, match="tsdf_default_brick_shape"):
        shape_of((0, 4, 4))


def test_default_brick_shape_keeps_a_bricks_span_below_4_GiB():
    """The brick kernels address a lane's quad as (start of the brick) + (32-bit byte offset): a brick of s slices spans
    s * dim_x * dim_y voxels, which must stay below 2^30 (csrc/tsdf_capi.hip, brick_shape_ok).  Slices of 2^28 voxels allow at
    most three slices per brick; slices of 2^30 voxels allow none, and the grid gets no brick view (the per-voxel launch)."""
    q, r, s = shape_of((16384, 16384, 8))
    assert 1 <= s <= 3 and q * r * s <= 64 and s * 16384 * 16384 < 2 ** 30
    assert shape_of((32768, 32768, 1)) == (0, 0, 1) or shape_of((32768, 32768, 1))[0] == 0
    q, r, s = shape_of((8192, 8192, 64))                     # 2^26 per slice: the usual eight slices still fit
    assert (q, r, s) == (2, 4, 8)
