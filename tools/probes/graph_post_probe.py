"""Does ossid_detect_post survive a hipGraph capture + replay? (standalone probe; prints the stage it reached)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ossid_code_amd.dtoid import ops  # noqa: E402

n_t, A = int(os.environ.get("NT", "3")), 27144
torch.manual_seed(0)
cls = torch.rand(n_t, A, 2, device="cuda")
reg = torch.randn(n_t, A, 4, device="cuda") * 0.5
anchors = torch.rand(1, A, 4, device="cuda") * 100
anchors[..., 2:] += anchors[..., :2] + 10
st = None
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        st = ops.detect_post(cls, reg, anchors, 640, 480, 1000, 0.5, st)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
print("eager ok, count", int(st.count.item()), flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    st = ops.detect_post(cls, reg, anchors, 640, 480, 1000, 0.5, st)
print("captured", flush=True)
g.replay()
torch.cuda.synchronize()
print("replayed, count", int(st.count.item()), flush=True)
