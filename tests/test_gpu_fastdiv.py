"""The kernel's shared-reciprocal division (csrc/tsdf_kernels.hip.h, fast_div2) must be the IEEE
quotient bit for bit wherever the kernel uses it.  Checked on the device against the compiler's
own division on ~4 billion pseudo-random operand pairs (structured mantissas, zeros, tiny and
huge numerators, denominators across [2^-60, 2^60])."""
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
