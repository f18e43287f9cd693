"""The kernel's shared-reciprocal division (csrc/tsdf_kernels.hip.h, fast_div2) must be the IEEE
quotient bit for bit wherever the kernel uses it.  Checked on the device against the compiler's
own division on ~4 billion pseudo-random operand pairs (structured mantissas, zeros, tiny and
huge numerators, denominators across [2^-60, 2^60])."""
import numpy as np
import pytest

from semantic_slam_amd import capi

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [1, 0x9E3779B97F4A7C15, 20261004])
def test_fast_division_is_ieee_exact(cuda, seed):
    bad, first = capi.selftest_fastdiv(1 << 30, seed=seed)
    assert bad == 0, f"{bad} mismatches, first: n={first[0]!r} d={first[1]!r} got={first[2]!r} want={first[3]!r}"


@pytest.mark.parametrize("fx,cx", [(16383.0, 0.0), (16383.0, 0.5), (-16383.0, -0.49999997), (1e-3, 1e-30), (16383.0, 1e-20)])
def test_tiny_quotients_cannot_move_a_pixel_for_any_admitted_intrinsics(cuda, fx, cx):
    """Below 2^-42 the unscaled sequence may lose the last bit of a denormal residual; the pixel coordinate
    fl(fx*q + cx) must not notice, for the largest focal length the fast path admits and awkward centres."""
    bad, first = capi.selftest_fastdiv(1 << 28, seed=77, fx=fx, cx=cx)
    assert bad == 0, f"{bad} mismatches, first: n={first[0]!r} d={first[1]!r} got={first[2]!r} want={first[3]!r}"


def test_one_instruction_rounding_equals_roundf_everywhere(cuda):
    """v_cvt_rpi_i32_f32 == (int)roundf for EVERY fp32 value in (-0.5, 2^24] (exhaustive, ~2.3e9 values)."""
    bad, first = capi.selftest_round()
    assert bad == 0, f"{bad} mismatches, first: u={first[0]!r} got={first[1]!r} want={first[2]!r}"


@pytest.mark.parametrize("seed", [3, 0xC2B2AE3D27D4EB4F, 20261005])
def test_truncated_distance_division_is_ieee_exact(cuda, seed):
    """diff / trunc through the launch-wide refined reciprocal (fast_div_r, fused kernels): bit-identical to `/` over the
    whole admitted domain -- divisor in [2^-20, 2^20], numerator 0, NaN or of magnitude in [2^-81, 2^60] -- on 2^30
    operand pairs per seed (quotients around the clamp at 1, structured mantissas, both signs)."""
    bad, first = capi.selftest_fastdiv_band(1 << 30, seed=seed)
    assert bad == 0, f"{bad} mismatches, first: n={first[0]!r} d={first[1]!r} got={first[2]!r} want={first[3]!r}"


@pytest.mark.parametrize("h,w", [(480, 640), (481, 643), (97, 130), (768, 1024), (16, 16), (5, 7)])
def test_tile_table_kernels_agree(cuda, h, w):
    """The depth tile tables of the classified launches: whole-row strip summary + levels by doubling (what is launched)
    against one wavefront per tile + levels by scanning, bit for bit -- random depths with invalid, out-of-range, NaN and
    infinite pixels, with and without a mask, aligned and unaligned buffers."""
    rng = np.random.default_rng(h * 1000 + w)
    depth = rng.uniform(0.3, 5.5, (h, w)).astype(np.float32)
    depth[rng.uniform(0, 1, (h, w)) < 0.02] = 0.0
    depth[rng.uniform(0, 1, (h, w)) < 0.01] = 7.0
    depth[rng.uniform(0, 1, (h, w)) < 0.002] = np.nan
    depth[rng.uniform(0, 1, (h, w)) < 0.002] = np.inf
    depth[rng.uniform(0, 1, (h, w)) < 0.002] = -1.0
    depth[: h // 3, : w // 2] = rng.uniform(1.0, 1.2, (h // 3, w // 2)).astype(np.float32)   # a region of all-valid tiles
    mask = (rng.uniform(0, 1, (h, w)) < 0.8).astype(np.uint8) * 255
    mask[: h // 4] = 255
    for off in (0, 1):
        d_buf = cuda.zeros(h * w + 4, dtype=cuda.float32, device="cuda")
        m_buf = cuda.zeros(h * w + 4, dtype=cuda.uint8, device="cuda")
        d_buf[off:off + h * w].copy_(cuda.from_numpy(depth.ravel()))
        m_buf[off:off + h * w].copy_(cuda.from_numpy(mask.ravel()))
        cuda.cuda.synchronize()
        assert capi.selftest_tile_tables(d_buf.data_ptr() + 4 * off, None, h, w) == 0
        assert capi.selftest_tile_tables(d_buf.data_ptr() + 4 * off, m_buf.data_ptr() + off, h, w) == 0
