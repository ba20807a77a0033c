"""Where a DtoidNet.forwardTestTime frame goes: graph replay (dense part) vs decode/top-k/NMS/gather vs host time."""
import os
import sys
import time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ossid_code_amd import dtoid
from ossid_code_amd.dtoid.model import normalizeImageRange
torch.manual_seed(0)
m = dtoid.DtoidNet(dtoid.DtoidConfig()).cuda().eval()
g = torch.Generator().manual_seed(1)
nt = int(sys.argv[1]) if len(sys.argv) > 1 else 21
test = {"img": torch.rand(1, 3, 480, 640, generator=g).cuda(), "obj_id": torch.tensor([1]),
        "limg": torch.rand(1, nt, 3, 124, 124, generator=g).cuda(),
        "lmask": (torch.rand(1, nt, 1, 124, 124, generator=g) > 0.5).float().cuda()}
for _ in range(3):
    m.forwardTestTime(test)
torch.cuda.synchronize()
net = m.model
local, glob = m._template_features(test, 1, torch.device("cuda", 0))
img = normalizeImageRange(test["img"])
def t(fn, n=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("whole forwardTestTime      %.3f ms" % t(lambda: m.forwardTestTime(test)))
print("forward_all_templates      %.3f ms" % t(lambda: net.forward_all_templates(img, local, glob, topk=500)))
print("dense graph replay only    %.3f ms" % t(lambda: net._graphed_dense(img, local, glob[0])))
ent = list(net.__dict__["_graph_cache"].values())[0]
print("bare graph.replay()        %.3f ms" % t(lambda: ent[0].replay()))
def t_sync(fn, n=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
print("bare replay + sync each    %.3f ms" % t_sync(lambda: ent[0].replay()))
t0 = time.perf_counter()
for _ in range(20):
    ent[0].replay()
host = (time.perf_counter() - t0) / 20 * 1e3
torch.cuda.synchronize()
print("host time of replay() call %.3f ms" % host)
