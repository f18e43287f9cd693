#!/usr/bin/env python3
"""Extract the 194 keyframe poses of the reference's saved fr3_office run
(/root/reference/result/rgbd/bundle.txt, a data file) into tests/golden/fr3_office_keyframes.npz,
with the names of the depth images of each keyframe (result/rgbd/associations.txt).  Data only.

    python tests/golden/make_keyframes.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from semantic_slam_amd import ingest  # noqa: E402

REF = "/root/reference/result/rgbd"
Tcw = ingest.load_bundle_poses(os.path.join(REF, "bundle.txt"))
assoc = ingest.load_associations(os.path.join(REF, "associations.txt"))
assert len(Tcw) == len(assoc) == 194
np.savez_compressed(os.path.join(HERE, "fr3_office_keyframes.npz"), Tcw=Tcw,
                    depth_names=np.array([a[2] for a in assoc]), timestamps=np.array([a[0] for a in assoc]))
print(Tcw.shape, Tcw[1], assoc[0])
