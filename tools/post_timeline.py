"""Cumulative host+device time through forward_all_templates' post-processing, real tensors from the graph."""
import os
import sys
import time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ossid_code_amd import dtoid
from ossid_code_amd.dtoid import ops
from ossid_code_amd.dtoid.model import normalizeImageRange
torch.manual_seed(0)
m = dtoid.DtoidNet(dtoid.DtoidConfig()).cuda().eval()
g = torch.Generator().manual_seed(1)
nt = 21
test = {"img": torch.rand(1, 3, 480, 640, generator=g).cuda(), "obj_id": torch.tensor([1]),
        "limg": torch.rand(1, nt, 3, 124, 124, generator=g).cuda(),
        "lmask": (torch.rand(1, nt, 1, 124, 124, generator=g) > 0.5).float().cuda()}
for _ in range(3):
    m.forwardTestTime(test)
net = m.model
local, glob = m._template_features(test, 1, torch.device("cuda", 0))
img = normalizeImageRange(test["img"])
def run(stop):
    cls_all, reg_all, seg_all, heat_all, fmap = net._graphed_dense(img, local, glob[0])
    if stop == 0: return
    n_t, A = reg_all.shape[0], reg_all.shape[1]
    anchors = net.anchors([list(fmap)], device=reg_all.device)
    boxes = ops.decode_clip_boxes(anchors, reg_all, 640, 480).view(-1, 4)
    if stop == 1: return
    max_score, max_id = torch.topk(cls_all.reshape(-1, 2)[:, 1], 1000)
    if stop == 2: return
    anchors_pred = boxes[max_id]
    obj_indices = (max_id // A).to(torch.float32)[:, None]
    if stop == 3: return
    keep = ops.nms(anchors_pred, max_score, 0.5, sorted_desc=True)[:500]
    if stop == 4: return
    max_score, anchors_pred, obj_indices = max_score[keep], anchors_pred[keep], obj_indices[keep]
    tid = obj_indices.reshape(-1).long()
    if stop == 5: return
    seg = ops.gather_rows(seg_all[:, 0], tid, sigmoid=True)
    h = heat_all[:, 0][tid]
names = ["dense", "+decode", "+topk", "+index", "+nms(sync)", "+index3", "+gathers"]
for stop, name in enumerate(names):
    for _ in range(3):
        run(stop)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        run(stop)
    torch.cuda.synchronize()
    print("%-12s %.3f ms" % (name, (time.perf_counter() - t0) / 20 * 1e3))
# nms alone on the real tensors
cls_all, reg_all, seg_all, heat_all, fmap = net._graphed_dense(img, local, glob[0])
A = reg_all.shape[1]
anchors = net.anchors([list(fmap)], device=reg_all.device)
boxes = ops.decode_clip_boxes(anchors, reg_all, 640, 480).view(-1, 4)
max_score, max_id = torch.topk(cls_all.reshape(-1, 2)[:, 1], 1000)
anchors_pred = boxes[max_id]
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter(); keep = ops.nms(anchors_pred, max_score, 0.5, sorted_desc=True); t1 = time.perf_counter()
    print("nms on real boxes: %.3f ms, kept %d; unique scores %d" % ((t1 - t0) * 1e3, len(keep), len(torch.unique(max_score))))
from ossid_code_amd import _lib
n = 1000
ws = torch.empty(_lib.fn("ossid_nms_workspace_bytes")(n), dtype=torch.uint8, device="cuda")
keepb = torch.empty(n, dtype=torch.int32, device="cuda"); nk = torch.empty(1, dtype=torch.int32, device="cuda")
sb = anchors_pred.float().contiguous()
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter()
    _lib.fn("ossid_nms")(sb.data_ptr(), n, 0.5, ws.data_ptr(), ws.numel(), keepb.data_ptr(), nk.data_ptr(), _lib.stream())
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("raw ossid_nms: host %.3f ms, +sync %.3f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
