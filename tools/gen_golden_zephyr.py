"""Writes tests/golden/zephyr_small.npz: a small synthetic scoring case (inputs) with the outputs of the CPU
oracle (expected values). The Zephyr reference itself is not in /root/reference, so these vectors
pin the BUILD's specification (SPEC.md), not the reference: parity unpinned. Run from the repo root:
    python tools/gen_golden_zephyr.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import zephyr_oracle as ozr  # noqa: E402
from ossid_code_amd import synth  # noqa: E402
from ossid_code_amd.zephyr.pointnet2 import PointNet2SSG, fold_pn2  # noqa: E402
from test_oracle import small_inputs, _oracle_features  # noqa: E402

WEIGHT_SEED = 7


def main():
    d = small_inputs(N=5, M=640, H=96, W=128, seed=11)
    d["pose_hypos"][3, 0, 3] += 0.1
    _, _, _, px, uv, cnt = _oracle_features(ozr, d)
    m = synth.random_pn2_state(PointNet2SSG(8).eval(), WEIGHT_SEED)
    scores, dbg = ozr.pn2_score(px, fold_pn2(m), debug=True)
    out = os.path.join(ROOT, "tests", "golden", "zephyr_small.npz")
    np.savez_compressed(out, img=d["img"], depth=d["depth"], cam_K=d["cam_K"], pose_hypos=d["pose_hypos"],
                        model_points=d["model_points"].astype(np.float32),
                        model_normals=d["model_normals"].astype(np.float32),
                        model_colors=d["model_colors"].astype(np.float32), point_x=px, uv_original=uv, inconst=cnt,
                        weight_seed=WEIGHT_SEED, fps1=dbg["fps1"], fps2=dbg["fps2"], scores=scores,
                        top1=int(np.argmax(scores)))
    print(out, os.path.getsize(out), "bytes; scores", scores)


if __name__ == "__main__":
    main()
