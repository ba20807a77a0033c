"""CPU tests of the DTOID restatement against golden vectors captured from the REFERENCE head classes
(tools/gen_golden_dtoid.py) and of the host logic (state_dict layout, anchors, loss, caching).
The three HIP-backed ops are swapped for oracle/dtoid_oracle.py here; the GPU tests check the HIP ops themselves."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from gen_golden_dtoid import B, GRID, IMG, SEED, seeded_inputs, seeded_state  # noqa: E402

from oracle import dtoid_oracle  # noqa: E402
from ossid_code_amd import dtoid  # noqa: E402

G = np.load(os.path.join(ROOT, "tests", "golden", "dtoid_head.npz"))


def close(a, b, rtol=1e-4, atol=1e-5):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    return np.allclose(a, b, rtol=rtol, atol=atol)


def build_head(device="cpu", train=True):
    corr = dtoid.CorrelationModel(IMG, 640)
    cls = dtoid.ClassificationModel(512, num_anchors=24)
    reg = dtoid.RegressionModel(512, num_anchors=24)
    for i, m in enumerate((corr, cls, reg)):
        m.load_state_dict(seeded_state(m, SEED + i))       # same keys as the reference classes
        m.train(train)
        m.to(device)
    return corr, cls, reg


def run_head(device="cpu"):
    corr, cls, reg = build_head(device, train=True)
    feat, tmpl, ann, heat_t, mask_t = (t.to(device) for t in seeded_inputs(SEED + 10))
    feat.requires_grad_(True)
    tmpl.requires_grad_(True)
    x2, heat, seg = corr(feat, tmpl)
    c, _ = cls(x2)
    r = reg(x2)
    anc = dtoid.Anchors(pyramid_levels=[4], ratios=[0.5, 1, 2], sizes=[30], scales=[1, 2, 3, 4, 5, 6, 7, 8])(
        [GRID], device=device)
    boxes = dtoid.BBoxTransform()(anc, r)
    lc, lr = dtoid.DetectionLoss()(c, r, anc, ann)
    l_center = torch.nn.L1Loss()(heat_t, heat)
    l_seg = torch.nn.BCELoss()(torch.sigmoid(seg), mask_t)
    (20 * l_seg + 20 * l_center + lc + lr).sum().backward()
    return dict(x2=x2, heat=heat, seg=seg, cls=c, reg=r, anchors=anc, boxes=boxes, loss_cls=lc, loss_reg=lr,
                loss_center=l_center, loss_seg=l_seg, grad_feat=feat.grad, grad_tmpl=tmpl.grad,
                grad_c1=corr.c1.weight.grad[:8], grad_cf_bias=corr.cf.bias.grad,
                grad_cls_conv1_bias=cls.conv1.bias.grad, grad_reg_out=reg.output.weight.grad[:4]), (corr, cls, reg)


def test_head_forward_backward_matches_reference_golden():
    with dtoid_oracle.cpu_ops():
        out, _ = run_head("cpu")
    for k, v in out.items():
        assert close(v, G[k], rtol=2e-4, atol=2e-5), k


def test_head_eval_matches_reference_golden():
    """The fixture's eval pass ran after its train pass, i.e. with BatchNorm running statistics updated once."""
    with dtoid_oracle.cpu_ops():
        _, (corr, cls, reg) = run_head("cpu")
        for m in (corr, cls, reg):
            m.eval()
        feat, tmpl, _, _, _ = seeded_inputs(SEED + 10)
        with torch.no_grad():
            x2, heat, seg = corr(feat, tmpl)
            c, _ = cls(x2)
            r = reg(x2)
    for k, v in dict(x2_eval=x2, heat_eval=heat, seg_eval=seg, cls_eval=c, reg_eval=r).items():
        assert close(v, G[k]), k


def test_loss_edge_cases_match_reference_golden():
    anc = torch.from_numpy(G["anchors"])
    c, r = torch.from_numpy(G["cls"]), torch.from_numpy(G["reg"])
    ann2 = torch.tensor([[[-1.0, -1, -1, -1, -1], [-1.0, -1, -1, -1, -1]],
                         [[6.0, 4.0, 30.0, 28.0, 1.0], [20.0, 10.0, 38.0, 30.0, 1.0]]])
    lc, lr = dtoid.DetectionLoss()(c, r, anc, ann2)       # sample 0 has no box, sample 1 has two
    assert close(lc, G["loss_cls2"]) and close(lr, G["loss_reg2"])


def test_small_helpers_match_reference_golden():
    img = torch.rand(2, 3, 8, 8, generator=torch.Generator().manual_seed(5))
    assert close(dtoid.normalizeImageRange(img), G["norm_img"], rtol=1e-6, atol=1e-6)
    a = dtoid.Anchors(pyramid_levels=[4], ratios=[0.5, 1, 2], sizes=[30], scales=[1, 2, 3, 4, 5, 6, 7, 8])
    full = a([[29, 39]], device="cpu")
    assert full.shape == (1, 27144, 4) and full.dtype == torch.float32
    assert close(full[0, ::997], G["anchors_29x39"], rtol=0, atol=0)
    assert a([[29, 39]], device="cpu") is full            # cached, not rebuilt per call
    assert a([[29, 29]], device="cpu").shape == (1, 20184, 4)


def test_state_dict_layout():
    """Key names and shapes a reference checkpoint carries (SURVEY.md 8b 'state_dict compatibility')."""
    m = dtoid.DtoidNet(dtoid.DtoidConfig())
    sd = m.state_dict()
    want = {
        "model.image_feature_extractor.backdense_0.0.weight": (64, 3, 7, 7),
        "model.image_feature_extractor.backdense_1.0.running_mean": (64,),
        "model.image_feature_extractor.backdense_1.3.denselayer6.conv2.weight": (32, 128, 3, 3),
        "model.image_feature_extractor.backdense_2.0.conv.weight": (128, 256, 1, 1),
        "model.image_feature_extractor.backdense_2.5.denselayer16.conv1.weight": (128, 992, 1, 1),
        "model.image_feature_extractor.backdense_2.6.bias": (1024,),
        "model.image_feature_extractor.c1.weight": (640, 1024, 1, 1),
        "model.template_feature_extractor.backbone_0.0.weight": (64, 4, 3, 3),
        "model.template_feature_extractor.backbone.features.0.weight": (64, 3, 3, 3),
        "model.template_feature_extractor.backbone_2.7.expand3x3.weight": (256, 64, 3, 3),
        "model.template_feature_extractor.backbone.classifier.1.weight": (1000, 512, 1, 1),
        "model.template_feature_extractor_global.final_conv_1.weight": (128, 640, 3, 3),
        "model.template_feature_extractor_global.final_norm_2.running_var": (64,),
        "model.correlation_model.cf.weight": (512, 768, 3, 3),
        "model.correlation_model.corr_conv_dot3x3.weight": (256, 640, 3, 3),
        "model.correlation_model.seg_final.bias": (1,),
        "model.classification.output.weight": (48, 256, 3, 3),
        "model.regression.output.weight": (96, 256, 3, 3),
    }
    for k, shape in want.items():
        assert k in sd and tuple(sd[k].shape) == shape, k
    # the Fire modules are shared between backbone.features and the backbone_1/2 slices, as in the reference
    tf = m.model.template_feature_extractor
    assert tf.backbone.features[3].squeeze.weight is tf.backbone_1[2].squeeze.weight
    nparam = sum(p.numel() for p in m.parameters())
    assert 33.5e6 < nparam < 34.5e6                         # ~34 M
    # zero-init output layers with the prior bias (network.py:408-419)
    assert float(m.model.classification.output.weight.abs().sum()) == 0
    assert abs(float(m.model.classification.output.bias[0]) + np.log(99.0)) < 1e-6
    m2 = dtoid.DtoidNet(dtoid.DtoidConfig())
    m2.load_state_dict(sd)


def test_backbone_shapes_small():
    """Shape truth of the explicit backbones at a reduced image size (the full 480x640 pass runs on the GPU)."""
    net = dtoid.Network(img_size=(96, 128), heatmap_size=(5, 7)).eval()
    with dtoid_oracle.cpu_ops(), torch.no_grad():
        tmpl = torch.rand(2, 4, 124, 124)
        g = net.compute_template_global(tmpl[:1])
        loc = net.compute_template_local(tmpl)
        assert g.shape == (1, 64, 3, 3) and loc.shape == (2, 640, 7, 7)
        f = net.image_feature_extractor(torch.rand(1, 3, 96, 128), g)
        assert f.shape == (1, 640, 5, 7)
        out = net.forward_all_templates(torch.rand(1, 3, 96, 128), [loc], [g], topk=5)
        assert out[1].shape[1] == 4 and out[3].shape[1:] == (96, 128) and out[4].shape[1:] == (5, 7)
        assert out[0].shape[0] == out[1].shape[0] == out[2].shape[0] <= 5


def test_dtoidnet_forward_dict_and_loss_backward_small():
    cfg = dtoid.DtoidConfig(img_h=96, img_w=128, heatmap_h=5, heatmap_w=7)
    m = dtoid.DtoidNet(cfg).train()
    g = torch.Generator().manual_seed(0)
    Bn = 2
    batch = {"img": torch.rand(Bn, 3, 96, 128, generator=g), "limg": torch.rand(Bn, 3, 124, 124, generator=g),
             "lmask": (torch.rand(Bn, 1, 124, 124, generator=g) > 0.5).float(),
             "gimg": torch.rand(Bn, 3, 124, 124, generator=g),
             "gmask": (torch.rand(Bn, 1, 124, 124, generator=g) > 0.5).float(),
             "bbox_gt": torch.tensor([[[20.0, 10.0, 90.0, 80.0, 1.0]], [[5.0, 5.0, 60.0, 50.0, 1.0]]]),
             "heatmap": torch.rand(Bn, 1, 5, 7, generator=g).double(),
             "mask": (torch.rand(Bn, 1, 96, 128, generator=g) > 0.5).float()}
    with dtoid_oracle.cpu_ops():
        out = m(batch)
        for k in ("classifications", "regressions", "anchors", "heat_map", "segmentation", "transformed_anchors",
                  "loss", "loss_seg", "loss_center", "loss_cls", "loss_reg", "seg_IoU", "seg_IoU_50"):
            assert k in out, k
        assert out["classifications"].shape == (Bn, 5 * 7 * 24, 2)
        out["loss"].backward()
    grads = [p.grad for p in m.parameters()]
    used = [g_ is not None for g_ in grads]
    assert sum(used) > 500 and not all(used)          # the SqueezeNet classifiers / 3-ch stems never get gradients
    assert all(torch.isfinite(g_).all() for g_ in grads if g_ is not None)


def test_nms_oracle_is_greedy():
    boxes = torch.tensor([[0, 0, 10, 10], [1, 1, 11, 11], [20, 20, 30, 30], [0, 0, 10, 10.0]])
    scores = torch.tensor([0.9, 0.8, 0.7, 0.95])
    assert dtoid_oracle.nms(boxes, scores, 0.5).tolist() == [3, 2]


def test_dense_block_inplace_gradient_function_equals_cat_path():
    """Training-mode dense block: the one-buffer autograd function (no torch.cat, L in-place gradient adds) against the
    plain concatenating forward under ordinary autograd: outputs, input gradient, every parameter gradient and the
    BatchNorm running statistics."""
    import copy
    from ossid_code_amd.dtoid import backbones as bb
    torch.manual_seed(0)
    blk = bb.DenseBlock(5, 16, growth=8, bn_size=2).train()
    blk2 = copy.deepcopy(blk)
    x = torch.randn(3, 16, 6, 7, requires_grad=True)
    x2 = x.detach().clone().requires_grad_()
    old = bb.DENSE_BLOCK_INPLACE_GRAD
    bb.DENSE_BLOCK_INPLACE_GRAD = False
    try:
        y = blk(x)
    finally:
        bb.DENSE_BLOCK_INPLACE_GRAD = old
    y2 = bb._DenseBlockFn.apply(x2, blk2)
    w = torch.arange(y.numel()).reshape(y.shape).float().cos()
    (y * w).sum().backward()
    (y2 * w).sum().backward()
    assert torch.allclose(y, y2, rtol=1e-5, atol=1e-5)
    assert torch.allclose(x.grad, x2.grad, rtol=1e-4, atol=1e-4)
    for (n, p), (_, q) in zip(blk.named_parameters(), blk2.named_parameters()):
        assert torch.allclose(p.grad, q.grad, rtol=1e-4, atol=1e-4), n
    for (n, p), (_, q) in zip(blk.named_buffers(), blk2.named_buffers()):
        assert torch.allclose(p.float(), q.float(), rtol=1e-6, atol=1e-6), n


def test_winograd_dispatch_policy_is_pure_host_logic():
    """train_ops.wino_fits: plain 3x3 layers with the reduction channels in 16s, >= 64 output channels and enough workgroups
    (32 tiles of 2x2 outputs x 64 channels each) -- nothing else."""
    from ossid_code_amd.dtoid import train_ops as T
    if not T.USE_WINO:
        return
    assert T.wino_fits(8, 29, 39, 768, 512, 9)                  # the fusion layer at batch 8: 75 x 8 workgroups
    assert T.wino_fits(8, 120, 160, 32, 128, 9)                 # a dense block's 3x3 data gradient
    assert not T.wino_fits(8, 29, 39, 768, 512, 1)              # 1x1
    assert not T.wino_fits(8, 29, 39, 768, 48, 9)               # under 64 output channels
    assert not T.wino_fits(8, 29, 39, 24, 64, 9)                # reduction channels not in 16s
    assert not T.wino_fits(1, 8, 8, 64, 64, 9)                  # one workgroup
    assert not T.wino_fits(8, 58, 78, 256, 128, 9, plain=False)  # behind a fused up-sampling
