"""Synthetic depth + pose streams for the TSDF Integrate path (numpy only, deterministic).

The reference ships no depth images (result/rgbd/associations.txt only names the TUM
files), so every workload is rendered analytically here, in the shapes SURVEY.md section 8(d)
fixes:

* S-full -- the roofline workload.  The volume sits wholly inside the frustum of the TUM fr3
  camera (include/tsdf.hpp:96 in the reference), the depth is a constant 5.9 m (inside the
  6 m cut-off of src/tsdf.cu:46 and behind the far face), so every voxel passes every test
  and is updated every frame: algorithmic bytes = 16 B x voxels.
* S-band -- S-full's geometry with a truncation margin of 4 m and a camera that also wobbles along its
  axis: every voxel is updated every frame AND lies inside the truncation band, so dist < 1, both
  divisions run and every TSDF value changes every frame -- nothing about the update can be elided,
  all 16 B per voxel have to move.  The workload the HBM-roofline fraction is quoted on.
* S-surf -- a sphere in front of a back wall seen from an orbit, optional uint16
  quantisation at the TUM depth factor 5000 (config/TUM3.yaml:34): a realistic mix of
  updated, truncated and out-of-frustum voxels.
"""
import math

import numpy as np

# TUM fr3 intrinsics, the reference's compile-time default (include/tsdf.hpp:96)
TUM_K = np.array([535.4, 0.0, 320.1, 0.0, 539.2, 247.6, 0.0, 0.0, 1.0], dtype=np.float32)
IM_H, IM_W = 480, 640
MAX_DEPTH = 6.0  # src/tsdf.cu:46
SFULL_Z0 = 3.2
SFULL_DEPTH = 5.9


def identity_pose():
    return np.eye(4, dtype=np.float32).ravel()


def rot_z(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])


def rot_y(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[c, 0.0, s], [0.0, 1.0, 0.0], [-s, 0.0, c]])


def rot_x(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[1.0, 0.0, 0.0], [0.0, c, -s], [0.0, s, c]])


def make_pose(R, t):
    """Row-major 4x4 cam2world from a rotation and a translation, as 16 float32."""
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = t
    return T.astype(np.float32).ravel()


# --------------------------------------------------------------------------------------
# S-full
# --------------------------------------------------------------------------------------
def sfull_volume(dim, voxel_size):
    """origin (base-camera frame) of a volume centred on the optical axis at z0 = 3.2 m that stays wholly
    inside the frustum for every pose of sfull_pose.  dim: edge in voxels, or (dim_x, dim_y, dim_z)."""
    dx, dy, dz = (dim, dim, dim) if np.isscalar(dim) else dim
    hx, hy = dx * voxel_size / 2.0, dy * voxel_size / 2.0
    wobble = math.cos(0.04) + math.sin(0.04)
    # in-image bounds from tsdf.cu:41-43: |x/z| < 0.5965, y/z < (479.5-247.6)/539.2 = 0.4301 (binds first)
    assert SFULL_Z0 * 0.4301 > max(hx, hy) * wobble + 1e-3, \
        "volume does not fit the frustum at z0: not a full-coverage workload"
    assert SFULL_Z0 + dz * voxel_size < SFULL_DEPTH, "far face beyond the constant depth"
    return np.array([-hx, -hy, SFULL_Z0], dtype=np.float32)


def sfull_depth(h=IM_H, w=IM_W):
    return np.full((h, w), SFULL_DEPTH, dtype=np.float32)


def sfull_pose(k):
    """Frame k: roll about the optical axis by 0.04 sin(0.1 k) rad, shift (sin k, cos k, 0) mm."""
    return make_pose(rot_z(0.04 * math.sin(0.1 * k)),
                     [1e-3 * math.sin(k), 1e-3 * math.cos(k), 0.0])


# --------------------------------------------------------------------------------------
# S-band
# --------------------------------------------------------------------------------------
SBAND_TRUNC = 4.0     # metres: wider than the deepest diff (5.9 - 3.16 = 2.74 m), so fmin(1, diff/trunc) < 1 everywhere
SBAND_WOBBLE_Z = 0.04  # metres along the optical axis


def sband_volume(dim, voxel_size):
    """S-full's volume; checks that it also stays inside the frustum and in front of the constant depth when
    the camera moves SBAND_WOBBLE_Z along its axis."""
    origin = sfull_volume(dim, voxel_size)
    dx, dy, dz = (dim, dim, dim) if np.isscalar(dim) else dim
    half = max(dx, dy) * voxel_size / 2.0 * (math.cos(0.04) + math.sin(0.04)) + 1e-3
    assert (SFULL_Z0 - SBAND_WOBBLE_Z) * 0.4301 > half, "volume leaves the frustum when the camera moves forward"
    assert SFULL_Z0 + dz * voxel_size + SBAND_WOBBLE_Z < SFULL_DEPTH, "far face reaches the constant depth"
    assert SFULL_DEPTH - (SFULL_Z0 - SBAND_WOBBLE_Z) < SBAND_TRUNC, "nearest voxel outside the truncation band"
    return origin


def sband_pose(k):
    """Frame k: S-full's roll and in-plane shift plus 4 cm sin(0.37 k) along the optical axis, so the distance
    of every voxel to the constant-depth surface -- and with it its TSDF value -- changes from frame to frame."""
    return make_pose(rot_z(0.04 * math.sin(0.1 * k)),
                     [1e-3 * math.sin(k), 1e-3 * math.cos(k), SBAND_WOBBLE_Z * math.sin(0.37 * k)])


# --------------------------------------------------------------------------------------
# S-surf
# --------------------------------------------------------------------------------------
class SurfScene:
    """Sphere of radius 0.35*extent centred in the volume, back wall at the far face."""

    def __init__(self, dims, voxel_size, origin, K=TUM_K, h=IM_H, w=IM_W):
        self.dims = tuple(int(d) for d in dims)
        self.vs = float(voxel_size)
        self.origin = np.asarray(origin, dtype=np.float64)
        ext = np.array(self.dims, dtype=np.float64) * self.vs
        self.center = self.origin + ext / 2.0
        self.radius = 0.35 * float(ext.min())
        self.wall_z = float(self.origin[2] + ext[2])
        self.K = np.asarray(K, dtype=np.float64)
        self.h, self.w = h, w
        u = np.arange(w, dtype=np.float64)
        v = np.arange(h, dtype=np.float64)
        self.dir_cam = np.stack(np.broadcast_arrays((u[None, :] - self.K[2]) / self.K[0],
                                                    (v[:, None] - self.K[5]) / self.K[4],
                                                    np.ones((h, w))), axis=-1)

    def pose(self, k, n=64, max_yaw_deg=15.0):
        """Orbit about the sphere centre: yaw sweeps +-max_yaw over n frames, camera looks at it."""
        yaw = math.radians(max_yaw_deg) * math.sin(2.0 * math.pi * k / n)
        R = rot_y(yaw)
        # camera position such that the sphere centre stays on the optical axis at its range
        rng = float(self.center[2])
        t = self.center - R @ np.array([0.0, 0.0, rng])
        return make_pose(R, t)

    def depth(self, cam2base, quantize=False, noise_sigma=0.0, seed=1234):
        """z-depth image (metres, fp32) of the scene from a row-major 4x4 camera-to-base pose."""
        T = np.asarray(cam2base, dtype=np.float64).reshape(4, 4)
        R, o = T[:3, :3], T[:3, 3]
        d = self.dir_cam @ R.T  # ray direction per unit camera z, in the base frame
        oc = o - self.center
        a = np.einsum("hwc,hwc->hw", d, d)
        b = 2.0 * np.einsum("hwc,c->hw", d, oc)
        c = float(oc @ oc) - self.radius ** 2
        disc = b * b - 4.0 * a * c
        with np.errstate(invalid="ignore", divide="ignore"):
            z_s = np.where(disc >= 0.0, (-b - np.sqrt(np.maximum(disc, 0.0))) / (2.0 * a), np.inf)
            z_s = np.where(z_s > 0.0, z_s, np.inf)
            z_w = (self.wall_z - o[2]) / d[..., 2]
        z_w = np.where(z_w > 0.0, z_w, np.inf)
        # the wall is finite: 1.5x the volume's xy extent around the centre
        hit = o[None, None, :] + z_w[..., None] * d
        ext = np.array(self.dims[:2]) * self.vs
        inside = np.all(np.abs(hit[..., :2] - self.center[:2]) <= 0.75 * ext, axis=-1)
        z_w = np.where(inside & np.isfinite(z_w), z_w, np.inf)
        z = np.minimum(z_s, z_w)
        z = np.where(np.isfinite(z), z, 0.0)
        if noise_sigma > 0.0:
            rng = np.random.default_rng(seed)
            z = np.where(z > 0.0, z + rng.normal(0.0, noise_sigma, z.shape), 0.0)
        if quantize:
            z = np.round(np.clip(z, 0.0, 13.0) * 5000.0) / 5000.0
        return z.astype(np.float32)


def surf_volume(dim, voxel_size, z0=1.0):
    """A dim^3 volume centred on the optical axis starting z0 metres in front of the base camera."""
    half = dim * voxel_size / 2.0
    return np.array([-half, -half, z0], dtype=np.float32)


# --------------------------------------------------------------------------------------
# a sensor's imperfections (Gaussian noise, sigma 2 mm; dropouts as zero blocks)
# --------------------------------------------------------------------------------------
def sensor_imperfections(depths, noise_mm=0.0, holes=0.0, seed=1234):
    """The frames through zero-mean Gaussian depth noise (sigma in mm, valid pixels only, re-quantised at the TUM factor
    1/5000 m) and dropouts (the given fraction of the image lost to invalid, i.e. zero, 8 x 8 pixel blocks), one
    generator for the whole sequence.  Returns new contiguous float32 frames."""
    rng = np.random.default_rng(seed)
    out = []
    for d in depths:
        d = d.copy()
        if noise_mm > 0:
            d = np.where(d > 0, d + rng.normal(0.0, noise_mm * 1e-3, d.shape).astype(np.float32), d)
            d = (np.round(d * 5000.0) / 5000.0).astype(np.float32)
        if holes > 0:
            drop = rng.uniform(0, 1, ((d.shape[0] + 7) // 8, (d.shape[1] + 7) // 8)) < holes      # (partial blocks at odd image sizes)
            d[np.kron(drop, np.ones((8, 8), bool))[:d.shape[0], :d.shape[1]]] = 0.0
        out.append(np.ascontiguousarray(d, np.float32))
    return out


# --------------------------------------------------------------------------------------
# random rigid poses for parity tests
# --------------------------------------------------------------------------------------
def random_pose(rng, max_angle=0.3, max_shift=0.3):
    ax, ay, az = rng.uniform(-max_angle, max_angle, 3)
    return make_pose(rot_z(az) @ rot_y(ay) @ rot_x(ax), rng.uniform(-max_shift, max_shift, 3))


def look_at_pose(rng, target, distance, jitter=0.15):
    """Camera at a random direction `distance` away from `target`, looking at it (optical axis = +z of the
    camera, as in the pinhole model of ref: src/tsdf.cu:41-42), random roll, small angular jitter."""
    d = rng.normal(size=3)
    d /= np.linalg.norm(d) + 1e-12
    pos = np.asarray(target, np.float64) - d * distance
    z = d
    up = rng.normal(size=3)
    x = np.cross(up, z)
    x /= np.linalg.norm(x) + 1e-12
    y = np.cross(z, x)
    R = np.stack([x, y, z], axis=1)                     # columns = camera axes in the base frame
    a = rng.uniform(-jitter, jitter, 3)
    R = R @ rot_z(a[2]) @ rot_y(a[1]) @ rot_x(a[0])
    return make_pose(R, pos)
