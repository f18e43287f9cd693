"""CPU sanitizer run ("-fsanitize=address,undefined on the CPU restatement + host wrapper"; no GPU
sanitizer exists on this pool).  `make -C oracle asan` builds oracle/tsdf_oracle.c together with the product's host-side
arithmetic -- csrc/pose_math.h and csrc/host_derive.h behind oracle/asan_host.cpp -- with AddressSanitizer and
UndefinedBehaviorSanitizer, every finding fatal; tests/sanitized_checks.py then runs the seven golden vectors (whole grid and
slabs), NaN / inf / denormal depth frames, the pose known-answer tests on 206 matrices (restatement == product header, bit for
bit), the brick choice and the guards for degenerate configurations and poses, the streaming copy into the pinned ring for sizes and
alignments around its thresholds, the .ply / .bin writers, the extraction rules and
the label / colour rules in a child interpreter with the sanitizer runtimes preloaded."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from semantic_slam_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _runtime(name):
    p = subprocess.check_output(["gcc", f"-print-file-name={name}"]).decode().strip()
    return p if os.path.isabs(p) and os.path.isfile(p) else None


def test_oracle_and_host_arithmetic_are_clean_under_asan_and_ubsan(tmp_path):
    asan, ubsan = _runtime("libasan.so"), _runtime("libubsan.so")
    if not asan or not ubsan:
        pytest.skip("gcc's sanitizer runtimes are not installed")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"])
    lib = os.path.join(ROOT, "oracle", "_asan", "liboracle_asan.so")
    # what the product's library says outside the sanitizer (tsdf_default_brick_shape needs no device): the sanitized copy
    # of the same header must agree
    cases = []
    for dims, z, im in (((512, 512, 512), (0, 512), (480, 640)), ((200, 200, 200), (0, 200), (480, 640)), ((1024, 1024, 1024), (384, 512), (480, 640)),
                        ((2048, 2048, 2048), (512, 768), (480, 640)), ((36, 7, 5), (0, 5), (250, 402)), ((6, 6, 6), (0, 6), (48, 64)),
                        ((4, 1, 1), (0, 1), (48, 64)), ((33000, 33000, 4), (0, 4), (480, 640))):
        cfg = capi.make_config(dims, 2.56 / dims[0], np.array([-1.28, -1.28, 1.0], np.float32), z_begin=z[0], z_end=z[1],
                               im_height=im[0], im_width=im[1])
        cases.append({"cfg": {"im": list(im), "dims": list(dims), "z": list(z), "voxel_size": float(cfg.voxel_size),
                              "trunc": float(cfg.trunc_margin), "max_depth": float(cfg.max_depth), "origin": [float(x) for x in cfg.origin],
                              "K": [float(x) for x in cfg.cam_K]},
                      "shape": list(capi.default_brick_shape(cfg))})
    expect = tmp_path / "expect.json"
    expect.write_text(json.dumps({"bricks": cases}))
    env = dict(os.environ, LD_PRELOAD=f"{asan}:{ubsan}", TSDF_ORACLE_LIB=lib, OMP_NUM_THREADS="2",
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "sanitized_checks.py"), str(expect), str(tmp_path)],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    out, err = p.stdout.decode(), p.stderr.decode()
    assert p.returncode == 0, f"rc {p.returncode}\n{out[-2000:]}\n{err[-4000:]}"
    assert "SANITIZED_OK" in out and "runtime error" not in err and "AddressSanitizer" not in err, err[-4000:]
    assert int(out.split("SANITIZED_OK")[1].split()[0]) > 650
