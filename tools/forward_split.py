"""Where a test-time frame's wall time goes (no profiler, hipGraph path as the product runs it):
  dense_graph_ms   the captured dense part (backbone + head over all templates) replayed back to back, per replay
  post_ms          Network.postprocess alone on the graph's static outputs (device launches + the one host sync it needs)
  frame_ms         forward_all_templates end to end
  python tools/forward_split.py [--nt 21] [--reps 50]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nt", type=int, default=21)
    ap.add_argument("--reps", type=int, default=50)
    a = ap.parse_args()
    from ossid_code_amd import dtoid
    from test_dtoid_gpu import _batch
    torch.manual_seed(0)
    cfg = dtoid.DtoidConfig()
    m = dtoid.DtoidNet(cfg).cuda().eval()
    b = _batch(cfg, 1, "cuda")
    test = {"img": b["img"], "obj_id": torch.tensor([1]), "limg": torch.rand(1, a.nt, 3, 124, 124).cuda(),
            "lmask": (torch.rand(1, a.nt, 1, 124, 124) > 0.5).float().cuda()}
    for _ in range(3):
        m.forwardTestTime(test)
    torch.cuda.synchronize()
    net = m.model
    local, glob = m._template_features(test, 1, b["img"].device)
    img = b["img"]

    def wall(fn, reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    out = {}
    out["frame_ms"] = wall(lambda: m.forwardTestTime(test), a.reps)
    out["forward_all_templates_ms"] = wall(lambda: net.forward_all_templates(img, local, glob, topk=500, seg_sigmoid=True,
                                                                             raw_image=True), a.reps)
    dense = net._graphed_dense(img, local, glob[0], raw_image=True)
    out["dense_graph_ms"] = wall(lambda: net._graphed_dense(img, local, glob[0], raw_image=True), a.reps)
    entry = [e for e in net.__dict__["_graph_cache"].values()][-1]
    out["graph_replay_only_ms"] = wall(entry[0].replay, a.reps)
    hw = (img.shape[2], img.shape[3])
    out["post_ms"] = wall(lambda: net.postprocess(*dense, hw, 500, True), a.reps)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
