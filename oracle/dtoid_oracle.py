"""CPU restatements of the DTOID device ops -- TEST INFRASTRUCTURE ONLY (tests/ and smoke() may import this; the
product package never does).

Each function cites the reference line it follows (/root/reference/python/ossid/models/dtoid/...). They are pinned
against the reference itself through tests/golden/dtoid_head.npz, which tools/gen_golden_dtoid.py produced by running
the reference's own head classes in the build container.
"""
import contextlib

import numpy as np
import torch
import torch.nn.functional as F


def dw_xcorr(x, kernel):
    """network.py:186-192 / :365-371: grouped convolution with groups = B*C and the template feature as weight."""
    if x.shape[0] != kernel.shape[0]:
        x = x.expand(kernel.shape[0], -1, -1, -1)
    B, C = kernel.shape[:2]
    y = F.conv2d(x.contiguous().view(1, B * C, x.size(2), x.size(3)), kernel.reshape(B * C, 1, 3, 3), groups=B * C,
                 padding=1)
    return y.view(B, C, y.size(2), y.size(3))


def nms(boxes, scores, thr, sorted_desc=False):
    """torchvision.ops.nms (network.py:563): greedy, by decreasing score, suppress IoU > thr."""
    b = boxes.detach().cpu().double().numpy()
    order = torch.sort(scores.detach().cpu(), descending=True, stable=True).indices.numpy()
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    keep, dead = [], np.zeros(len(b), bool)
    for oi, i in enumerate(order):
        if dead[i]:
            continue
        keep.append(i)
        rest = order[oi + 1:]
        iw = np.clip(np.minimum(b[i, 2], b[rest, 2]) - np.maximum(b[i, 0], b[rest, 0]), 0, None)
        ih = np.clip(np.minimum(b[i, 3], b[rest, 3]) - np.maximum(b[i, 1], b[rest, 1]), 0, None)
        inter = iw * ih
        with np.errstate(divide="ignore", invalid="ignore"):
            iou = inter / (area[i] + area[rest] - inter)
        dead[rest[iou > thr]] = True
    return torch.as_tensor(np.asarray(keep, dtype=np.int64), device=boxes.device)


def decode_clip_boxes(anchors, deltas, img_w, img_h):
    """network.py:42-70 (BBoxTransform) + :78-88 (ClipBoxes)."""
    a = anchors.reshape(1, -1, 4).float()
    d = deltas.detach().float()
    w, h = a[..., 2] - a[..., 0], a[..., 3] - a[..., 1]
    cx, cy = a[..., 0] + 0.5 * w, a[..., 1] + 0.5 * h
    dx, dy, dw, dh = d[..., 0] * 0.1, d[..., 1] * 0.1, d[..., 2] * 0.2, d[..., 3] * 0.2
    pcx, pcy, pw, ph = cx + dx * w, cy + dy * h, torch.exp(dw) * w, torch.exp(dh) * h
    out = torch.stack([pcx - 0.5 * pw, pcy - 0.5 * ph, pcx + 0.5 * pw, pcy + 0.5 * ph], 2)
    out[..., 0].clamp_(min=0)
    out[..., 1].clamp_(min=0)
    out[..., 2].clamp_(max=img_w)
    out[..., 3].clamp_(max=img_h)
    return out


def amsgrad_reference(param, grads, lr, wd, steps=1):
    """torch.optim.Adam(amsgrad=True) as online_learning.py:258-263 builds it; returns the updated parameter."""
    p = torch.nn.Parameter(param.detach().clone().cpu())
    opt = torch.optim.Adam([p], lr=lr, weight_decay=wd, amsgrad=True)
    for g in grads[:steps]:
        p.grad = g.detach().clone().cpu()
        opt.step()
    return p.detach()


@contextlib.contextmanager
def cpu_ops():
    """Run the product's DTOID modules on the CPU for oracle checks by swapping the three HIP-backed ops for the
    restatements above (tests only)."""
    from ossid_code_amd.dtoid import ops
    saved = (ops.dw_xcorr, ops.nms, ops.decode_clip_boxes)
    ops.dw_xcorr, ops.nms, ops.decode_clip_boxes = dw_xcorr, nms, decode_clip_boxes
    try:
        yield
    finally:
        ops.dw_xcorr, ops.nms, ops.decode_clip_boxes = saved
