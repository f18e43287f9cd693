"""The C-ABI boundary without a GPU: the library loads, exports exactly the functions
include/tsdf_hip.h declares, its struct layout matches the ctypes mirror, and compute entry
points fail loudly (no CPU fallback) when no device is present."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from semantic_slam_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "tsdf_hip.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(tsdf_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree():
    assert declared_functions() == sorted(capi.ABI_SYMBOLS)


def test_library_exports_every_declared_symbol():
    lib = capi.load()
    for name in declared_functions():
        assert hasattr(lib, name), f"libtsdf_hip.so does not export {name}"
    assert b"gfx950" in lib.tsdf_version()


def test_no_oracle_in_product():
    """The shipped library must not link or reference the CPU checker."""
    out = subprocess.check_output(["ldd", capi.LIB_PATH]).decode()
    assert "oracle" not in out
    for fn in os.listdir(os.path.join(ROOT, "semantic_slam_amd")):
        if fn.endswith(".py"):
            assert "oracle" not in open(os.path.join(ROOT, "semantic_slam_amd", fn)).read().replace(
                "oracle/", ""), fn  # mentions of the directory in docstrings are fine, imports are not
    for fn in os.listdir(os.path.join(ROOT, "semantic_slam_amd", "csrc")):
        path = os.path.join(ROOT, "semantic_slam_amd", "csrc", fn)
        if os.path.isfile(path):
            assert "oracle" not in open(path).read(), fn


def test_struct_layout_matches_c(tmp_path):
    prog = tmp_path / "layout.c"
    fields = [f for f, _ in capi.TsdfConfig._fields_]
    body = "\n".join(f'printf("{f} %zu\\n", offsetof(tsdf_config, {f}));' for f in fields)
    prog.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "tsdf_hip.h"\n'
                    'int main(void){printf("size %zu\\n", sizeof(tsdf_config));\n' + body + "\nreturn 0;}\n")
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(prog), "-o", str(exe)])
    got = dict(line.split() for line in subprocess.check_output([str(exe)]).decode().splitlines())
    assert int(got["size"]) == C.sizeof(capi.TsdfConfig)
    for f in fields:
        assert int(got[f]) == getattr(capi.TsdfConfig, f).offset, f


def test_defaults_are_the_reference_constants():
    cfg = capi.default_config(480, 640)
    assert (cfg.dim_x, cfg.dim_y, cfg.dim_z) == (200, 200, 200)            # ref: include/tsdf.hpp:65-67
    assert np.float32(cfg.voxel_size) == np.float32(0.004)                 # ref: include/tsdf.hpp:63
    assert np.float32(cfg.trunc_margin) == np.float32(0.004) * np.float32(5)  # ref: include/tsdf.hpp:64
    assert cfg.max_depth == 6.0                                            # ref: src/tsdf.cu:46
    assert list(cfg.cam_K) == [np.float32(x) for x in (535.4, 0, 320.1, 0, 539.2, 247.6, 0, 0, 1)]
    assert list(cfg.base2world) == list(np.eye(4, dtype=np.float32).ravel())
    assert (cfg.z_begin, cfg.z_end, cfg.im_height, cfg.im_width) == (0, 200, 480, 640)


def test_invalid_arguments_are_reported():
    lib = capi.load()
    h = C.c_void_p()
    cfg = capi.default_config()
    cfg.dim_x = 0
    assert lib.tsdf_create(C.byref(cfg), C.byref(h)) == -1 and b"dims" in lib.tsdf_last_error()
    cfg = capi.default_config()
    cfg.z_end = 201
    assert lib.tsdf_create(C.byref(cfg), C.byref(h)) == -1 and b"slab" in lib.tsdf_last_error()
    cfg = capi.default_config()
    cfg.voxel_size = 0.0
    assert lib.tsdf_create(C.byref(cfg), C.byref(h)) == -1
    cfg = capi.default_config()
    cfg.dim_x, cfg.dim_y = 1 << 20, 1 << 12
    assert lib.tsdf_create(C.byref(cfg), C.byref(h)) == -1 and b"slice" in lib.tsdf_last_error()
    assert lib.tsdf_create(None, C.byref(h)) == -1
    assert lib.tsdf_sync(None) == -1 and lib.tsdf_integrate(None, None, None) == -1
    assert lib.tsdf_destroy(None) == 0
    with pytest.raises(capi.TsdfError):
        capi.check(-1, "x")


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(capi.TsdfError, match="no HIP device|CPU path"):
        capi.Volume(capi.default_config())


def test_measurement_build_compiles_and_knows_the_experiment_variants():
    """`make experiments` (-DTSDF_EXPERIMENTS): the product plus the kernel variants of csrc/tsdf_experiments.hip.h.  It is not
    what ships and nothing loads it by default; this keeps it compiling, and checks that the shipped library refuses the
    experiment numbers while the measurement build reports itself (host-only calls: no GPU needed)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "semantic_slam_amd", "csrc"), "experiments"])
    exp = os.path.join(root, "semantic_slam_amd", "libtsdf_hip_exp.so")
    assert os.path.isfile(exp)
    code = ("from semantic_slam_amd import capi; print(capi.experiments_build(), capi.variants(0, 1, 2, 3, 7, 8, 11, 13, 23, 119))")
    out = subprocess.check_output([sys.executable, "-c", code], cwd=root, env=dict(os.environ, TSDF_HIP_LIB=exp), text=True)
    assert out.strip() == "True [0, 1, 2, 3, 7, 8, 11, 13, 23, 119]"
    env = {k: v for k, v in os.environ.items() if k != "TSDF_HIP_LIB"}
    out = subprocess.check_output([sys.executable, "-c", code], cwd=root, env=env, text=True)
    assert out.strip() == "False [0, 1, 3, 7, 8]"
