"""GPU tests (pytest -m gpu) of the DTOID path: the hand-written HIP ops through the C ABI against the CPU
restatements (oracle/dtoid_oracle.py), the detector head against golden vectors from the REFERENCE classes, and the
finetune step against torch.optim.Adam. Floating point: fp32 everywhere; tolerances are written at each check
(the convolutions run in MIOpen on the GPU and in torch-CPU for the fixture, so sums are reordered)."""
import numpy as np
import pytest
import torch

from oracle import dtoid_oracle
from ossid_code_amd import dtoid
from ossid_code_amd.dtoid import finetune, ops
from test_dtoid_cpu import G, close, run_head

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,C,H,W", [(2, 640, 29, 39), (1, 64, 240, 320), (3, 5, 7, 9), (2, 64, 1, 1), (1, 3, 4, 70)])
def test_dw_xcorr_forward_and_both_gradients(hiplib, B, C, H, W):
    g = torch.Generator().manual_seed(B * 1000 + C)
    x = torch.randn(B, C, H, W, generator=g)
    k = torch.randn(B, C, 3, 3, generator=g)
    go = torch.randn(B, C, H, W, generator=g)
    xr, kr = x.clone().requires_grad_(True), k.clone().requires_grad_(True)
    want = dtoid_oracle.dw_xcorr(xr, kr)
    want.backward(go)
    xd, kd = x.cuda().requires_grad_(True), k.cuda().requires_grad_(True)
    got = ops.dw_xcorr(xd, kd)
    got.backward(go.cuda())
    assert close(got, want.detach().numpy(), rtol=1e-5, atol=1e-5)                 # 9-term sums
    assert close(xd.grad, xr.grad.numpy(), rtol=1e-5, atol=1e-5)
    assert close(kd.grad, kr.grad.numpy(), rtol=1e-4, atol=1e-4 * max(1.0, (H * W) ** 0.5))   # H*W-term sums


def test_dw_xcorr_broadcasts_one_image_over_templates(hiplib):
    x = torch.randn(1, 16, 9, 11).cuda()
    k = torch.randn(4, 16, 3, 3).cuda()
    got = ops.dw_xcorr(x, k)
    assert close(got, dtoid_oracle.dw_xcorr(x.cpu(), k.cpu()).numpy(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("n", [1, 63, 64, 65, 1000, 3000])
def test_nms_matches_greedy_oracle(hiplib, n):
    g = torch.Generator().manual_seed(n)
    ctr = torch.rand(n, 2, generator=g) * 200
    wh = torch.rand(n, 2, generator=g) * 60 + 2
    boxes = torch.cat([ctr - wh / 2, ctr + wh / 2], 1)
    scores = torch.rand(n, generator=g)
    if n > 10:
        boxes[5] = boxes[2]                       # exact duplicates
        scores[7] = scores[3]                     # tied scores: stable order decides
    want = dtoid_oracle.nms(boxes, scores, 0.5)
    got = ops.nms(boxes.cuda(), scores.cuda(), 0.5)
    assert got.dtype == torch.long and got.cpu().tolist() == want.tolist()
    assert ops.nms(boxes[:0].cuda(), scores[:0].cuda(), 0.5).numel() == 0


def test_decode_clip_matches_reference_golden(hiplib):
    anc, reg = torch.from_numpy(G["anchors"]).cuda(), torch.from_numpy(G["reg"]).cuda()
    got = ops.decode_clip_boxes(anc, reg, 40, 32)
    want = dtoid_oracle.decode_clip_boxes(anc.cpu(), reg.cpu(), 40, 32)
    assert close(got, want.numpy(), rtol=1e-5, atol=1e-4)
    unclipped = torch.from_numpy(G["boxes"])            # the reference's BBoxTransform output
    ref = unclipped.clone()
    ref[..., 0].clamp_(min=0), ref[..., 1].clamp_(min=0), ref[..., 2].clamp_(max=40), ref[..., 3].clamp_(max=32)
    assert close(got, ref.numpy(), rtol=1e-5, atol=1e-4)


def test_head_forward_backward_on_gpu_matches_reference_golden(hiplib):
    out, _ = run_head("cuda")
    for k, v in out.items():
        assert close(v, G[k], rtol=2e-3, atol=2e-4), k        # MIOpen vs torch-CPU summation order, 5760-term convs


@pytest.mark.parametrize("steps", [1, 3])
def test_fused_amsgrad_matches_torch_adam(hiplib, steps):
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(37, 19), torch.nn.Linear(19, 5)).cuda()
    ref = torch.nn.Sequential(torch.nn.Linear(37, 19), torch.nn.Linear(19, 5)).cuda()
    ref.load_state_dict(net.state_dict())
    flat = finetune.FlatParams(net, unused_filter=lambda n: n.startswith("1.bias"))   # pretend one tensor is unused
    opt = finetune.FusedAMSGrad(flat, lr=1e-2, weight_decay=1e-3)
    used = [p for n, p in ref.named_parameters() if not n.startswith("1.bias")]
    ropt = torch.optim.Adam(used, lr=1e-2, weight_decay=1e-3, amsgrad=True)
    frozen = net[1].bias.detach().clone()
    for s in range(steps):
        x = torch.randn(11, 37, device="cuda")
        for m, o in ((net, opt), (ref, ropt)):
            o.zero_grad()
            m(x).square().mean().backward()
            o.step()
    for (n, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
        if n.startswith("1.bias"):
            assert torch.equal(p, frozen)                     # never touched, like a grad-less tensor under Adam
        else:
            assert torch.allclose(p, q, rtol=1e-5, atol=1e-6), n


def _batch(cfg, B, dev, seed=0):
    g = torch.Generator().manual_seed(seed)
    H, W, hh, hw = cfg.model.img_h, cfg.model.img_w, cfg.model.heatmap_h, cfg.model.heatmap_w
    mask = torch.zeros(B, 1, H, W)
    mask[:, :, H // 4: H // 2, W // 4: W // 2] = 1
    b = {"img": torch.rand(B, 3, H, W, generator=g), "limg": torch.rand(B, 3, 124, 124, generator=g),
         "lmask": (torch.rand(B, 1, 124, 124, generator=g) > 0.5).float(),
         "gimg": torch.rand(B, 3, 124, 124, generator=g), "gmask": (torch.rand(B, 1, 124, 124, generator=g) > 0.5).float(),
         "bbox_gt": torch.tensor([[[W / 4.0, H / 4.0, W / 2.0, H / 2.0, 1.0]]]).repeat(B, 1, 1),
         "heatmap": torch.rand(B, 1, hh, hw, generator=g).double(), "mask": mask}
    return {k: v.to(dev) for k, v in b.items()}


def test_full_network_shapes_and_finetune_step_480x640(hiplib):
    """D11/D13/D14/D16 at the real size: shapes of every output, loss decreases under the fused step,
    template cache stays on the device, and test-time inference returns the reference's dict."""
    cfg = dtoid.DtoidConfig()
    torch.manual_seed(0)
    m = dtoid.DtoidNet(cfg).cuda()
    flat = finetune.FlatParams(m)
    assert flat.n_used < flat.total and m.model.classification.conv1.weight.data_ptr() >= flat.param.data_ptr()
    opt = finetune.FusedAMSGrad(flat, lr=1e-4, weight_decay=1e-6)
    m.train()
    batch = _batch(cfg, 2, "cuda")
    out = m(batch)
    assert out["classifications"].shape == (2, 27144, 2) and out["regressions"].shape == (2, 27144, 4)
    assert out["anchors"].shape == (1, 27144, 4) and out["heat_map"].shape == (2, 1, 29, 39)
    assert out["segmentation"].shape == (2, 1, 480, 640) and out["transformed_anchors"].shape == (2, 27144, 4)
    losses = [float(finetune.finetune_step(m, batch, opt)) for _ in range(4)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    m.eval()
    nt = 3
    test = {"img": batch["img"][:1], "obj_id": torch.tensor([5]), "limg": torch.rand(1, nt, 3, 124, 124).cuda(),
            "lmask": (torch.rand(1, nt, 1, 124, 124) > 0.5).float().cuda(), "mask": batch["mask"][:1],
            "heatmap": batch["heatmap"][:1]}
    res = m.forwardTestTime(test)
    k = res["pred_scores"].shape[0]
    assert 1 <= k <= 500 and res["pred_bbox"].shape == (k, 4) and res["segmentation"].shape == (k, 1, 480, 640)
    assert res["heat_map"].shape == (k, 1, 29, 39) and res["final_bbox"][0] is res["pred_bbox"]
    assert (res["pred_scores"][:-1] >= res["pred_scores"][1:]).all() and "seg_IoU" in res
    assert (res["pred_template_ids"] >= 0).all() and (res["pred_template_ids"] < nt).all()
    local, glob = m.template_feature_cache[5]
    assert local[0].is_cuda and local[0].shape == (nt, 640, 7, 7) and glob[0].shape == (1, 64, 3, 3)
    res2 = m.forwardTestTime(test)                           # second frame: served from the device-resident cache
    assert torch.equal(res2["pred_bbox"], res["pred_bbox"])
