"""forwardTestTimeBatch (BASELINE configs[2]: 32 images x 21 templates) a few times, for a rocprofv3 kernel trace:
  rocprofv3 --kernel-trace -d out --output-format csv -- python3 tools/trace_forward_batch.py
  python tools/trace_step.py out/*/*kernel_trace.csv --marker im2col_stem --top 40      (one batch = between two stem launches)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ossid_code_amd import dtoid

torch.manual_seed(0)
m = dtoid.DtoidNet(dtoid.DtoidConfig()).cuda().eval()
g = torch.Generator().manual_seed(1)
nt, B = 21, 32
test = {"img": torch.rand(B, 3, 480, 640, generator=g).cuda(), "obj_id": torch.tensor([1]),
        "limg": torch.rand(1, nt, 3, 124, 124, generator=g).cuda(),
        "lmask": (torch.rand(1, nt, 1, 124, 124, generator=g) > 0.5).float().cuda()}
for _ in range(4):
    m.forwardTestTimeBatch(test)
torch.cuda.synchronize()
