"""Per-op time of forward_all_templates' post-processing (decode, top-k, NMS, gathers) at n_t templates."""
import os
import sys
import time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ossid_code_amd.dtoid import ops
from ossid_code_amd.dtoid.anchors import Anchors
nt = int(sys.argv[1]) if len(sys.argv) > 1 else 21
A = 27144
g = torch.Generator().manual_seed(0)
cls_all = torch.rand(nt, A, 2, generator=g).cuda()
reg_all = (torch.randn(nt, A, 4, generator=g) * 0.1).cuda()
seg_all = torch.randn(nt, 1, 480, 640, generator=g).cuda()
heat_all = torch.rand(nt, 1, 29, 39, generator=g).cuda()
anchors = Anchors(pyramid_levels=[4], ratios=[0.5, 1, 2], sizes=[30], scales=[1, 2, 3, 4, 5, 6, 7, 8])([[29, 39]], device="cuda")
def t(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        r = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3, r
ms, boxes = t(lambda: ops.decode_clip_boxes(anchors, reg_all, 640, 480).view(-1, 4)); print("decode_clip   %.3f" % ms)
ms, (max_score, max_id) = t(lambda: torch.topk(cls_all.reshape(-1, 2)[:, 1], 1000)); print("topk 1000     %.3f" % ms)
ms, anchors_pred = t(lambda: boxes[max_id]); print("boxes[max_id] %.3f" % ms)
ms, keep = t(lambda: ops.nms(anchors_pred, max_score, 0.5, sorted_desc=True)[:500]); print("nms           %.3f  (kept %d)" % (ms, len(keep)))
ms, _ = t(lambda: (max_score[keep], anchors_pred[keep], (max_id // A).float()[:, None][keep])); print("3 index ops   %.3f" % ms)
tid = (max_id // A)[keep].long()
ms, _ = t(lambda: (seg_all[:, 0][tid], heat_all[:, 0][tid])); print("seg/heat gather %.3f (k=%d)" % (ms, len(tid)))
ms, _ = t(lambda: torch.sigmoid(seg_all[:, 0][tid])); print("gather+sigmoid %.3f" % ms)
