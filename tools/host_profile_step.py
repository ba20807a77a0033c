"""cProfile of the host side of one eager finetune step (where the ~38 ms of enqueue time go)."""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ossid_code_amd import dtoid  # noqa: E402
from ossid_code_amd.dtoid import finetune  # noqa: E402

cfg = dtoid.DtoidConfig()
torch.manual_seed(0)
m = dtoid.DtoidNet(cfg).cuda().train()
flat = finetune.FlatParams(m)
opt = finetune.FusedAMSGrad(flat, lr=1e-4, weight_decay=1e-6)
g = torch.Generator().manual_seed(1)
B = 8
mask = torch.zeros(B, 1, 480, 640)
mask[:, :, 120:240, 160:320] = 1
batch = {"img": torch.rand(B, 3, 480, 640, generator=g), "limg": torch.rand(B, 3, 124, 124, generator=g),
         "lmask": (torch.rand(B, 1, 124, 124, generator=g) > 0.5).float(), "gimg": torch.rand(B, 3, 124, 124, generator=g),
         "gmask": (torch.rand(B, 1, 124, 124, generator=g) > 0.5).float(),
         "bbox_gt": torch.tensor([[[160.0, 120.0, 320.0, 240.0, 1.0]]]).repeat(B, 1, 1),
         "heatmap": torch.rand(B, 1, 29, 39, generator=g).double(), "mask": mask}
batch = {k: v.cuda() for k, v in batch.items()}
for _ in range(3):
    finetune.finetune_step(m, batch, opt)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    finetune.finetune_step(m, batch, opt)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
