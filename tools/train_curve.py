"""End-to-end sanity of the training arithmetic: the same 12 finetune steps (fixed batch, AMSGrad lr 1e-4) on the hand-written
kernels (split-bf16 / three-way-split convolutions) and on the nn.Module path (MIOpen, f32): the loss curves have to track
each other -- not bit for bit (two float32 paths through a 120-layer network drift apart chaotically after a few optimizer
steps), but step by step within a few per cent while both go down.  python tools/train_curve.py [--steps 12]"""
import argparse
import copy
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ossid_code_amd import dtoid  # noqa: E402
from ossid_code_amd.dtoid import finetune  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--batch", type=int, default=4)
    a = ap.parse_args()
    torch.manual_seed(0)
    base = dtoid.DtoidNet(dtoid.DtoidConfig()).cuda().train()
    with torch.no_grad():
        for conv in (base.model.classification.output, base.model.regression.output, base.model.correlation_model.seg_final,
                     base.model.correlation_model.corr_conv_heatmap):
            conv.weight.normal_(0, 0.02)
    g = torch.Generator().manual_seed(1)
    B = a.batch
    mask = torch.zeros(B, 1, 480, 640)
    mask[:, :, 120:240, 160:320] = 1
    batch = {"img": torch.rand(B, 3, 480, 640, generator=g), "limg": torch.rand(B, 3, 124, 124, generator=g),
             "lmask": (torch.rand(B, 1, 124, 124, generator=g) > 0.5).float(), "gimg": torch.rand(B, 3, 124, 124, generator=g),
             "gmask": (torch.rand(B, 1, 124, 124, generator=g) > 0.5).float(),
             "bbox_gt": torch.tensor([[[160.0, 120.0, 320.0, 240.0, 1.0]]]).repeat(B, 1, 1),
             "heatmap": torch.rand(B, 1, 29, 39, generator=g).double(), "mask": mask}
    batch = {k: v.cuda() for k, v in batch.items()}
    curves = {}
    for impl in ("hip", "miopen"):
        m = copy.deepcopy(base)
        m.model.use_hip_training = impl == "hip"
        flat = finetune.FlatParams(m)
        opt = finetune.FusedAMSGrad(flat, lr=1e-4, weight_decay=1e-6)
        curves[impl] = [float(finetune.finetune_step(m, batch, opt)) for _ in range(a.steps)]
    rel = [abs(h - r) / abs(r) for h, r in zip(curves["hip"], curves["miopen"])]
    print(json.dumps({"steps": a.steps, "batch": B, "loss_hip": [round(v, 5) for v in curves["hip"]],
                      "loss_module_path": [round(v, 5) for v in curves["miopen"]], "rel_diff": [round(v, 6) for v in rel]}))


if __name__ == "__main__":
    main()
