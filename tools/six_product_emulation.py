"""ONE experiment, decided on the CPU: would the hypothesis rank order survive a six-product split-bf16
form of the scorer's SA1 / SA2 layers (x = p0 + p1 + p2 in bf16 pieces, the six products with i + j <= 2: 192 instead of 512
matrix-pipe cycles per 16-deep slice)?  The oracle's product mode 2 (oracle/zephyr_oracle.c, experiment only) emulates that
arithmetic -- every v_mfma_f32_32x32x16_bf16 as the exact dot product of its slice added to the f32 accumulator with one
rounding -- and this tool scores the same seeded frames in both arithmetics:
    python tools/six_product_emulation.py --frames 50 --hyp 100 --points 2048 [--out profiles/r04_six_product_emulation.json]
Per frame: is the top-1 the same, is the WHOLE rank order the same, max |score difference|.  The rule (north star):
the headline stays on the exact-f32 instruction unless the rank order is identical on every frame."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import zephyr_oracle as ozr  # noqa: E402
from ossid_code_amd import synth, zephyr  # noqa: E402
from ossid_code_amd.zephyr.pointnet2 import fold_pn2  # noqa: E402


class Args:
    dataset, no_valid_proj, no_valid_depth, inconst_ratio_th, extra_bottleneck_dim = "HSVD_diff_uv_norm", True, True, 100, 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=50)
    ap.add_argument("--hyp", type=int, default=100)
    ap.add_argument("--points", type=int, default=2048)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    ozr.build()
    ozr.set_threads(max(1, (os.cpu_count() or 2) - 2))
    model = synth.random_pn2_state(zephyr.PointNet2SSG(8, Args(), num_class=1), 0).eval()
    w = fold_pn2(model)
    rows = []
    t0 = time.time()
    for f in range(a.frames):
        d = synth.make_scoring_inputs(N=a.hyp, M=a.points, seed=1000 + f)
        rgbd = ozr.pack_rgbd(ozr.u8_to_unit(ozr.blur5_u8(d["img"])), d["depth"])
        tab = ozr.prep_model(d["model_points"], d["model_normals"], d["model_colors"])
        px, _ = ozr.featurize(rgbd, d["pose_hypos"].astype(np.float32), tab, d["cam_K"])
        ozr.set_product_mode(0)
        exact = ozr.pn2_score(px, w)
        ozr.set_product_mode(2)
        six = ozr.pn2_score(px, w)
        ozr.set_product_mode(0)
        oe, os_ = np.argsort(-exact, kind="stable"), np.argsort(-six, kind="stable")
        gaps = np.abs(np.diff(exact[oe]))
        rows.append({"frame": f, "top1_same": bool(oe[0] == os_[0]), "order_same": bool(np.array_equal(oe, os_)),
                     "swapped_positions": int((oe != os_).sum()),
                     "max_abs_dscore": float(np.abs(six - exact).max()),
                     "max_rel_dscore": float((np.abs(six - exact) / np.maximum(np.abs(exact), 1e-30)).max()),
                     "bits_equal": int((six == exact).sum()), "min_adjacent_gap": float(gaps.min()),
                     "score_range": [float(exact.min()), float(exact.max())]})
        print(json.dumps(rows[-1]), "%.0f s" % (time.time() - t0), flush=True)
    out = {"what": "oracle product mode 2 (six-product split-bf16 emulation in SA1 / SA2) vs the exact fmaf chains, same frames",
           "frames": a.frames, "hypotheses_per_frame": a.hyp, "points": a.points,
           "frames_top1_same": sum(r["top1_same"] for r in rows), "frames_order_same": sum(r["order_same"] for r in rows),
           "max_abs_dscore": max(r["max_abs_dscore"] for r in rows), "max_rel_dscore": max(r["max_rel_dscore"] for r in rows),
           "pipe_cycles_per_16_slice": {"exact_f32": 512, "six_product": 192},
           "per_frame": rows}
    print(json.dumps({k: v for k, v in out.items() if k != "per_frame"}))
    if a.out:
        json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
