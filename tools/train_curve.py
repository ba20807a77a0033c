"""End-to-end sanity of the training arithmetic: the same 12 finetune steps (fixed batch, AMSGrad lr 1e-4) on the hand-written
kernels (split-bf16 / three-way-split convolutions) and on the nn.Module path (MIOpen, f32): the loss curves have to track
each other -- not bit for bit (two float32 paths through a 120-layer network drift apart chaotically after a few optimizer
steps), but step by step within a few per cent while both go down.  python tools/train_curve.py [--steps 12]

Attribution (round 4): the module path is run TWICE (its own run-to-run spread: MIOpen's atomics) and, with
`--perturb 1e-6`, once more from weights moved by one part in a million -- how far ANY f32-level difference drifts over the
same steps. Run the same command on an all-exact build (OSSID_HIPCC_EXTRA="-DOSSID_CONV_F32 -DOSSID_WINO_F32 -DOSSID_WGRAD_F32
-DOSSID_SEGTAIL_F32") for the kernel path without split-bf16 products; `--tag` names the run in the output."""
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
    ap.add_argument("--perturb", type=float, default=1e-6)
    ap.add_argument("--tag", default="default build")
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
    for name, impl, eps in (("hip", "hip", 0.0), ("hip_again", "hip", 0.0), ("miopen", "miopen", 0.0), ("miopen_again", "miopen", 0.0),
                            ("miopen_perturbed", "miopen", a.perturb), ("hip_perturbed", "hip", a.perturb)):
        m = copy.deepcopy(base)
        if eps:
            gp = torch.Generator(device="cuda").manual_seed(7)
            with torch.no_grad():
                for p in m.parameters():
                    p.mul_(1.0 + eps * (2.0 * torch.rand(p.shape, generator=gp, device=p.device) - 1.0))
        m.model.use_hip_training = impl == "hip"
        flat = finetune.FlatParams(m)
        opt = finetune.FusedAMSGrad(flat, lr=1e-4, weight_decay=1e-6)
        curves[name] = [float(finetune.finetune_step(m, batch, opt)) for _ in range(a.steps)]

    def rel(x, y):
        return [round(abs(h - r) / abs(r), 6) for h, r in zip(curves[x], curves[y])]
    from ossid_code_amd import _lib
    print(json.dumps({"tag": a.tag, "steps": a.steps, "batch": B, "perturb": a.perturb,
                      "split_bf16": {"conv": _lib.lib().ossid_conv_split_bf16(), "wino": _lib.lib().ossid_conv_wino_split_bf16(),
                                     "wgrad": _lib.lib().ossid_conv_wgrad_split_bf16(), "segtail": _lib.lib().ossid_seg_tail_split_bf16()},
                      "loss_hip": [round(v, 5) for v in curves["hip"]],
                      "loss_module_path": [round(v, 5) for v in curves["miopen"]],
                      "rel_diff_hip_vs_module": rel("hip", "miopen"),
                      "rel_diff_hip_vs_hip_again": rel("hip_again", "hip"),
                      "rel_diff_module_vs_module_again": rel("miopen_again", "miopen"),
                      "rel_diff_module_perturbed_vs_module": rel("miopen_perturbed", "miopen"),
                      "rel_diff_hip_perturbed_vs_hip": rel("hip_perturbed", "hip")}))


if __name__ == "__main__":
    main()
