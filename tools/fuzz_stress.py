#!/usr/bin/env python3
"""Longer run of the randomised parity test (tests/test_gpu_fuzz.py): seeds 40..399 under the four classification variants
(per launch / always with bricks / always with workgroup patches / never), 1440 configurations, each bit-exact against the
oracle; the forced-bricks runs draw a random valid brick shape (TSDF_BRICK3D) per seed.  Development probe; the suite itself
runs seeds 0..39."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from oracle.oracle import Oracle
import test_gpu_fuzz as F
orc = Oracle()
bad = 0
for seed in range(40, 400):
    for variant in (0, 8, 11, 7):
        os.environ.pop("TSDF_BRICK3D", None)
        dims = F.random_case(seed)[1]
        if variant == 8 and dims[0] % 4 == 0:
            rs = np.random.default_rng(seed)
            quads = dims[0] // 4
            q = int(rs.choice([d for d in range(1, min(quads, 64) + 1) if quads % d == 0]))
            r = int(rs.integers(1, 64 // q + 1))
            sl = int(rs.integers(1, 64 // (q * r) + 1))
            os.environ["TSDF_BRICK3D"] = f"{q},{r},{sl}"
        try:
            F.test_random_configuration.__wrapped__(torch, orc, seed, variant) if hasattr(F.test_random_configuration, '__wrapped__') else F.test_random_configuration(torch, orc, seed, variant)
        except AssertionError as e:
            bad += 1
            print("FAIL seed", seed, "variant", variant, str(e)[:200], flush=True)
print("done, failures:", bad)
