"""HIP kernel (through the C ABI) against the golden vectors made from the reference's own
kernel body: bit-exact on every fixture, for the default and the first-version kernel."""
import numpy as np
import pytest

from golden_util import NAMES, Golden
from semantic_slam_amd import capi

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("variant", capi.variants(0, 3, 1, 2))   # 0: collected frames, 3: one default kernel per frame, 1 / 2: first versions
@pytest.mark.parametrize("name", NAMES)
def test_hip_reproduces_golden(cuda, name, variant):
    g = Golden(name)
    h, w = g.depth.shape[1:]
    cfg = capi.make_config(g.dims, g.vs, g.origin, trunc=g.trunc, K=g.K, im_height=h, im_width=w)
    with capi.Volume(cfg) as vol:
        vol.set_kernel_variant(variant)
        for c2b, depth in g.frames:
            d = cuda.from_numpy(np.ascontiguousarray(depth)).cuda()
            vol.integrate_cam2base(d.data_ptr(), c2b)
            vol.sync()
        t, wgt = vol.download()
    assert np.array_equal(wgt, g.weight)
    assert np.array_equal(t.view(np.uint32), g.tsdf.view(np.uint32))
