#!/usr/bin/env python3
"""Longer run of the randomised parity test (tests/test_gpu_fuzz.py): seeds 52..N under the shipped classification variants
(per launch / always over the brick list / never), the forced runs over a random valid brick shape, every third seed with a
sequence longer than one pass -- each bit-exact against the oracle.  Development probe; the suite itself runs seeds 0..51.
    python tools/fuzz_stress.py [last_seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402
import test_gpu_fuzz as F  # noqa: E402

orc = Oracle()
last = int(sys.argv[1]) if len(sys.argv) > 1 else 400
bad = n = 0
for seed in range(52, last):
    dims = F.random_case(seed)[1]
    for variant in (0, 8, 7):
        shape = F.random_brick_shape(seed, dims) if variant == 8 else None
        try:
            F.run_case(torch, orc, seed, variant, shape=shape, long=seed % 3 == 0)
            n += 1
        except AssertionError as e:
            bad += 1
            print("FAIL seed", seed, "variant", variant, "shape", shape, str(e)[:200], flush=True)
    if seed % 50 == 0:
        print("seed", seed, "cases", n, "failures", bad, flush=True)
print("done: cases", n, "failures", bad)
