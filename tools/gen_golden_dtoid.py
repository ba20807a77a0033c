"""Writes tests/golden/dtoid_head.npz: outputs of the REFERENCE DTOID head classes (imported from /root/reference in
the build container, SURVEY.md 8c recipe) on seeded inputs and seeded weights.

Only data is stored: inputs are re-derived from seeds at test time, the expected outputs/gradients are saved. The
reference source is never copied; the fixture pins this build's CorrelationModel / ClassificationModel /
RegressionModel / BBoxTransform / anchors / DetectionLoss / normalizeImageRange / heatmapGaussain against it.
Run from the repo root:   python tools/gen_golden_dtoid.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ref_import  # noqa: E402

IMG = (32, 40)      # "image" size of the reduced case: feature grid 4x5, three x2 upsamples -> 32x40
GRID = (4, 5)
B = 2
SEED = 1234


def seeded_state(module, seed):
    """Deterministic weights for any module: every tensor of the state_dict from one seeded generator, in key order.
    Convolution outputs are kept O(1); BatchNorm statistics are non-trivial."""
    g = torch.Generator().manual_seed(seed)
    sd = module.state_dict()
    out = {}
    for k, v in sd.items():
        if k.endswith("num_batches_tracked"):
            out[k] = v.clone()
        elif k.endswith("running_var"):
            out[k] = 0.5 + torch.rand(v.shape, generator=g)
        elif k.endswith("running_mean"):
            out[k] = 0.1 * torch.randn(v.shape, generator=g)
        elif v.dim() == 1 and ("norm" in k or ".n" in k or k.split(".")[-2].startswith("n")) and k.endswith("weight"):
            out[k] = 1.0 + 0.2 * torch.randn(v.shape, generator=g)
        elif v.dim() == 1:
            out[k] = 0.1 * torch.randn(v.shape, generator=g)
        else:
            fan_in = v[0].numel()
            out[k] = torch.randn(v.shape, generator=g) * (1.0 / fan_in) ** 0.5
    return out


def seeded_inputs(seed):
    g = torch.Generator().manual_seed(seed)
    feat = torch.randn(B, 640, *GRID, generator=g)
    tmpl = torch.randn(B, 640, 7, 7, generator=g)
    ann = torch.tensor([[[6.0, 4.0, 30.0, 28.0, 1.0]], [[10.0, 2.0, 36.0, 20.0, 1.0]]])
    heat_t = torch.rand(B, 1, *GRID, generator=g).double()
    mask_t = (torch.rand(B, 1, *IMG, generator=g) > 0.5).float()
    return feat, tmpl, ann, heat_t, mask_t


def main():
    network, loss_mod, anchors_mod, utils = ref_import.load()
    torch.manual_seed(0)
    corr = network.CorrelationModel(IMG, 640)
    cls = network.ClassificationModel(512, num_anchors=24)
    reg = network.RegressionModel(512, num_anchors=24)
    for i, m in enumerate((corr, cls, reg)):
        m.load_state_dict(seeded_state(m, SEED + i))
        m.train()                     # BatchNorm in training mode, as in finetuneDtoid (online_learning.py:656)
    feat, tmpl, ann, heat_t, mask_t = seeded_inputs(SEED + 10)
    feat.requires_grad_(True)
    tmpl.requires_grad_(True)
    x2, heat, seg = corr(feat, tmpl)
    c, _ = cls(x2)
    r = reg(x2)
    base = anchors_mod.generate_anchors(base_size=30, ratios=np.array([0.5, 1, 2]), scales=np.array([1, 2, 3, 4, 5, 6, 7, 8]))
    anc = torch.from_numpy(anchors_mod.shift(GRID, 16, base).astype(np.float32))[None]
    boxes = network.BBoxTransform(mean=torch.zeros(4), std=torch.tensor([0.1, 0.1, 0.2, 0.2]))(anc, r)
    lc, lr = loss_mod.DetectionLoss()(c, r, anc, ann)
    l_center = torch.nn.L1Loss()(heat_t, heat)
    l_seg = torch.nn.BCELoss()(torch.sigmoid(seg), mask_t)
    total = 20 * l_seg + 20 * l_center + lc + lr
    total.sum().backward()
    # eval-mode forward too (test-time path)
    for m in (corr, cls, reg):
        m.eval()
    with torch.no_grad():
        x2e, heate, sege = corr(feat, tmpl)
        ce, _ = cls(x2e)
        re_ = reg(x2e)
    # a case with no annotation and one with two boxes, for the loss alone
    ann2 = torch.tensor([[[-1.0, -1, -1, -1, -1], [-1.0, -1, -1, -1, -1]],
                         [[6.0, 4.0, 30.0, 28.0, 1.0], [20.0, 10.0, 38.0, 30.0, 1.0]]])
    lc2, lr2 = loss_mod.DetectionLoss()(c.detach(), r.detach(), anc, ann2)
    img = torch.rand(2, 3, 8, 8, generator=torch.Generator().manual_seed(5))
    out = dict(
        x2=x2.detach().numpy(), heat=heat.detach().numpy(), seg=seg.detach().numpy(), cls=c.detach().numpy(),
        reg=r.detach().numpy(), anchors=anc.numpy(), boxes=boxes.detach().numpy(),
        loss_cls=lc.detach().numpy(), loss_reg=lr.detach().numpy(), loss_center=l_center.detach().numpy(),
        loss_seg=l_seg.detach().numpy(), grad_feat=feat.grad.numpy(), grad_tmpl=tmpl.grad.numpy(),
        grad_c1=corr.c1.weight.grad.numpy()[:8], grad_cf_bias=corr.cf.bias.grad.numpy(),
        grad_cls_conv1_bias=cls.conv1.bias.grad.numpy(), grad_reg_out=reg.output.weight.grad.numpy()[:4],
        x2_eval=x2e.numpy(), heat_eval=heate.numpy(), seg_eval=sege.numpy(), cls_eval=ce.numpy(), reg_eval=re_.numpy(),
        loss_cls2=lc2.numpy(), loss_reg2=lr2.numpy(),
        norm_img=utils.normalizeImageRange(img).numpy(),
        gauss=utils.heatmapGaussain(29, 39, 12.3, 7.9, np.sqrt(1.5)),
        anchors_29x39=anchors_mod.shift((29, 39), 16, base).astype(np.float32)[::997],
        seed=SEED)
    path = os.path.join(ROOT, "tests", "golden", "dtoid_head.npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in out.items()})
    print(path, os.path.getsize(path), "bytes", float(lc), float(lr), float(l_center), float(l_seg))


if __name__ == "__main__":
    main()
