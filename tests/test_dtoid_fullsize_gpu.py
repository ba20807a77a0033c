"""BASELINE configs[2] at its stated size (32 images x 21 templates, 480x640) through the batched test-time API, checked by
properties that need no stored expectation, and D13's z-filter / IoU metric against a float64 numpy restatement of
models/dtoid/__init__.py:125-146, 163-169."""
import numpy as np
import pytest
import torch

from ossid_code_amd import dtoid

pytestmark = pytest.mark.gpu


def _net(seed):
    cfg = dtoid.DtoidConfig()
    torch.manual_seed(seed)
    m = dtoid.DtoidNet(cfg).cuda().eval()
    with torch.no_grad():   # non-degenerate outputs (the reference zero-initialises the four output layers)
        for conv in (m.model.classification.output, m.model.regression.output, m.model.correlation_model.seg_final,
                     m.model.correlation_model.corr_conv_heatmap):
            conv.weight.normal_(0, 0.05)
    return m


KEYS = ("pred_scores", "pred_bbox", "pred_template_ids", "segmentation", "heat_map")


def test_configs2_full_size_batch_32_images_21_templates(hiplib):
    """forwardTestTimeBatch at 32 x 21: (a) per-image results agree with forwardTestTime on images 0, 15 and 31 (the
    batch-1 backbone picks other tilings than the batched one, so to rounding: top score, detection count, and the dense
    maps of the best detection); (b) duplicate images give bit-identical rows; (c) permuting the batch permutes the
    results bit for bit -- an image's result depends on nothing but that image."""
    m = _net(21)
    g = torch.Generator().manual_seed(9)
    B, nt = 32, 21
    imgs = torch.rand(B, 3, 480, 640, generator=g)
    imgs[5] = imgs[3]
    imgs[30] = imgs[3]
    imgs = imgs.cuda()
    test = {"obj_id": torch.tensor([1]), "limg": torch.rand(1, nt, 3, 124, 124, generator=g).cuda(),
            "lmask": (torch.rand(1, nt, 1, 124, 124, generator=g) > 0.5).float().cuda()}
    outs = m.forwardTestTimeBatch(dict(test, img=imgs))
    assert len(outs) == B
    for o in outs:
        k = o["pred_scores"].shape[0]
        assert 1 <= k <= 500 and o["pred_bbox"].shape == (k, 4) and o["segmentation"].shape == (k, 1, 480, 640)
        assert o["heat_map"].shape == (k, 1, 29, 39) and o["pred_template_ids"].shape == (k,)
        assert float(o["pred_template_ids"].max()) < nt and bool(torch.isfinite(o["pred_scores"]).all())
        assert bool((o["pred_scores"][:-1] >= o["pred_scores"][1:]).all())          # sorted by score
    # (b) duplicates
    for dup in (5, 30):
        for key in KEYS:
            assert torch.equal(outs[3][key], outs[dup][key]), key
    assert not torch.equal(outs[3]["pred_scores"], outs[4]["pred_scores"])
    # (a) against the per-image API
    for i in (0, 15, 31):
        one = m.forwardTestTime(dict(test, img=imgs[i:i + 1]))
        assert abs(float(one["pred_scores"][0]) - float(outs[i]["pred_scores"][0])) <= 1e-4
        assert abs(one["pred_scores"].shape[0] - outs[i]["pred_scores"].shape[0]) <= 3      # NMS survivors near a tie
        if int(one["pred_template_ids"][0]) == int(outs[i]["pred_template_ids"][0]):
            assert float((one["pred_bbox"][0] - outs[i]["pred_bbox"][0]).abs().max()) <= 0.05      # pixels
            assert float((one["segmentation"][0] - outs[i]["segmentation"][0]).abs().max()) <= 2e-3
            assert float((one["heat_map"][0] - outs[i]["heat_map"][0]).abs().max()) <= 2e-3
    # (c) permutation
    perm = torch.randperm(B, generator=g)
    outs_p = m.forwardTestTimeBatch(dict(test, img=imgs[perm.cuda()]))
    for j in range(B):
        for key in KEYS:
            assert torch.equal(outs_p[j][key], outs[int(perm[j])][key]), (j, key)


def _z_filter_numpy(boxes, tids, z_values):
    """float64 restatement of the reference's filter (models/dtoid/__init__.py:125-146): predicted distance from the box's
    larger side, 124 px being the template size; keep 0.4 < z < 2; nothing kept -> keep detection 0."""
    boxes = boxes.astype(np.float32)
    side = np.maximum(boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1])      # float32, as the reference's numpy arrays
    pred_z = (124 / side) * -z_values[0, tids]                                      # f32 quotient, promoted by the f64 z
    ids = np.where((pred_z > 0.4) & (pred_z < 2))[0]
    return ids if len(ids) else np.array([0])


@pytest.mark.parametrize("case", ["some", "none", "all"])
def test_forward_test_time_z_filter_and_seg_iou(hiplib, case):
    """D13 with `filter_z` and a ground-truth mask: the kept detections are exactly those of the numpy restatement applied
    to the UNFILTERED call's outputs, in the same order; seg_IoU is the foreground IoU of the first kept detection's
    mask (pl.metrics iou(..., ignore_index=0), :163-169) computed in float64 numpy."""
    m = _net(5)
    g = torch.Generator().manual_seed(12)
    nt = 6
    mask = torch.zeros(1, 1, 480, 640)
    mask[:, :, 100:300, 200:420] = 1
    test = {"img": torch.rand(1, 3, 480, 640, generator=g).cuda(), "obj_id": torch.tensor([1]),
            "limg": torch.rand(1, nt, 3, 124, 124, generator=g).cuda(),
            "lmask": (torch.rand(1, nt, 1, 124, 124, generator=g) > 0.5).float().cuda(),
            "mask": mask.cuda(), "heatmap": torch.rand(1, 1, 29, 39, generator=g).double().cuda()}
    m.cfg.filter_z = False
    ref = m.forwardTestTime(test)
    boxes = ref["pred_bbox"].cpu().numpy()
    tids = ref["pred_template_ids"].long().cpu().numpy()
    side = np.maximum(boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1]).astype(np.float64)
    # template "z values" (negative: the reference multiplies by -z) chosen so that the band 0.4 .. 2 cuts the list
    if case == "some":
        zt = -(np.median(side) / 124.0) * np.linspace(0.3, 2.4, nt)
    elif case == "none":
        zt = -np.full(nt, 50.0)
    else:
        zt = -(side.min() / 124.0 + side.max() / 124.0) / 2 * np.ones(nt)
        if not ((124 / side.max()) * -zt[0] > 0.4 and (124 / side.min()) * -zt[0] < 2):
            pytest.skip("box sizes spread over more than the 0.4 .. 2 band")
    z_values = zt[None].astype(np.float64)
    want = _z_filter_numpy(boxes, tids, z_values)
    if case == "some":
        assert 0 < len(want) < len(boxes)
    if case == "none":
        assert list(want) == [0]
    m.cfg.filter_z = True
    got = m.forwardTestTime(dict(test, template_z_values=torch.from_numpy(z_values)))
    m.cfg.filter_z = False
    for key in KEYS:
        sel = ref[key][torch.from_numpy(want).cuda()]
        # (every key bit for bit: since round 4 the 512 -> 1 heat-map convolution is this repo's deterministic kernel too)
        assert torch.equal(got[key], sel), key
    assert got["final_bbox"][0] is got["pred_bbox"] and got["final_score"][0] is got["pred_scores"]
    # IoU of the first kept detection's mask with the ground truth, float64 on the host
    seg = got["segmentation"][0, 0].cpu().numpy() > 0.5
    gt = mask[0, 0].numpy() > 0
    union = np.logical_or(seg, gt).sum()
    iou = float(np.logical_and(seg, gt).sum()) / float(union) if union else 0.0
    assert abs(float(got["seg_IoU"]) - iou) <= 1e-6
    assert float(got["seg_IoU_50"]) == float(iou > 0.5)
