"""Detection losses of the DTOID finetune step (reference: models/dtoid/loss.py:10-37 calc_iou, :46-175
DetectionLoss.forward -- focal classification loss (alpha .25, gamma 2) with IoU anchor assignment (< 0.4 negative,
>= 0.5 positive, in between ignored) and smooth-L1 box regression (beta 1/9) on the positives).

The reference walks the batch in a Python loop with data-dependent branches (host syncs per sample); this is the same
arithmetic as whole-batch tensor expressions, so the loss stays on the device and is graph-capturable."""
import torch
import torch.nn as nn

from .. import _lib


class _FusedDetectionLoss(torch.autograd.Function):
    """ossid_focal_smoothl1_loss_{fwd,bwd}: the whole of DetectionLoss.forward and its gradient as three launches."""

    @staticmethod
    def forward(ctx, cls, reg, anchors, ann, alpha, gamma):
        B, A, C = cls.shape
        G = int(ann.shape[1])
        cls, reg = cls.float().contiguous(), reg.float().contiguous()
        anc, ann = anchors.reshape(-1, 4).float().contiguous(), ann.float().contiguous()
        dev = cls.device
        dcls, dreg = torch.empty_like(cls), torch.empty_like(reg)
        ws = torch.empty(_lib.fn("ossid_focal_smoothl1_loss_workspace_floats")(B, A), dtype=torch.float32, device=dev)
        losses, scales = torch.empty(2, dtype=torch.float32, device=dev), torch.empty(2 * B, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = _lib.fn("ossid_focal_smoothl1_loss_fwd")(cls.data_ptr(), reg.data_ptr(), anc.data_ptr(), ann.data_ptr(), B, A, C, G,
                                                          float(alpha), float(gamma), dcls.data_ptr(), dreg.data_ptr(),
                                                          ws.data_ptr(), losses.data_ptr(), scales.data_ptr(), _lib.stream())
        _lib.check(rc, "ossid_focal_smoothl1_loss_fwd")
        ctx.save_for_backward(dcls, dreg, scales)
        ctx.set_materialize_grads(False)
        # two outputs (not two slices of one: their gradients came back through two slice-backward fills, two copies and an add)
        return losses[0:1], losses[1:2]

    @staticmethod
    def backward(ctx, g_cls, g_reg):
        dcls_raw, dreg_raw, scales = ctx.saved_tensors
        B, A, C = dcls_raw.shape
        zero = None
        if g_cls is None or g_reg is None:
            zero = torch.zeros(1, dtype=torch.float32, device=dcls_raw.device)
        g = torch.cat([(zero if g_cls is None else g_cls).float().reshape(1), (zero if g_reg is None else g_reg).float().reshape(1)])
        dcls, dreg = torch.empty_like(dcls_raw), torch.empty_like(dreg_raw)
        with torch.cuda.device(g.device):
            rc = _lib.fn("ossid_focal_smoothl1_loss_bwd")(dcls_raw.data_ptr(), dreg_raw.data_ptr(), scales.data_ptr(), g.data_ptr(),
                                                          B, A, C, dcls.data_ptr(), dreg.data_ptr(), _lib.stream())
        _lib.check(rc, "ossid_focal_smoothl1_loss_bwd")
        return dcls, dreg, None, None, None, None


def calc_iou(a, b):
    """pair-wise IoU of boxes a [n1,4] and b [n2,4] -> [n1,n2]"""
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    iw = (torch.min(a[:, None, 2], b[None, :, 2]) - torch.max(a[:, None, 0], b[None, :, 0])).clamp(min=0)
    ih = (torch.min(a[:, None, 3], b[None, :, 3]) - torch.max(a[:, None, 1], b[None, :, 1])).clamp(min=0)
    inter = iw * ih
    union = ((a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1]))[:, None] + area_b[None] - inter
    return inter / union.clamp(min=1e-8)


class DetectionLoss(nn.Module):
    use_fused = True     # on the GPU: csrc/train.hip's fused kernels; False = the tensor expressions below

    def __init__(self, alpha=0.25, gamma=2.0):
        super().__init__()
        self.alpha, self.gamma = alpha, gamma

    def forward(self, classifications, regressions, anchors, annotations):
        """classifications [B,A,C] probabilities, regressions [B,A,4], anchors [1,A,4], annotations [B,G,5]
        (x1,y1,x2,y2,label; label -1 marks padding). Returns (cls_loss [1], reg_loss [1])."""
        alpha, gamma = self.alpha, self.gamma
        B, A, C = classifications.shape
        dev = classifications.device
        if classifications.is_cuda and self.use_fused and annotations.shape[1] <= 16 and C <= 64:
            return _FusedDetectionLoss.apply(classifications, regressions, anchors, annotations.to(dev), alpha, gamma)
        annotations = annotations.to(dev)
        anchor = anchors[0]
        aw, ah = anchor[:, 2] - anchor[:, 0], anchor[:, 3] - anchor[:, 1]
        acx, acy = anchor[:, 0] + 0.5 * aw, anchor[:, 1] + 0.5 * ah

        p = classifications.clamp(1e-4, 1.0 - 1e-4)
        valid = annotations[:, :, 4] != -1                                  # [B,G]
        has_gt = valid.any(1)                                               # [B]
        # IoU of every anchor with every (valid) annotation: padded rows can never win the max
        g = annotations[:, None, :, :4]                                     # [B,1,G,4]
        a4 = anchor[None, :, None, :]                                       # [1,A,1,4]
        iw = (torch.min(a4[..., 2], g[..., 2]) - torch.max(a4[..., 0], g[..., 0])).clamp(min=0)
        ih = (torch.min(a4[..., 3], g[..., 3]) - torch.max(a4[..., 1], g[..., 1])).clamp(min=0)
        union = (aw * ah)[None, :, None] + ((g[..., 2] - g[..., 0]) * (g[..., 3] - g[..., 1])) - iw * ih
        iou = (iw * ih) / union.clamp(min=1e-8)                             # [B,A,G]
        iou = torch.where(valid[:, None, :], iou, torch.full_like(iou, -1.0))
        iou_max, iou_arg = iou.max(2)                                       # [B,A]
        assigned = torch.gather(annotations, 1, iou_arg[:, :, None].expand(-1, -1, 5))       # [B,A,5]
        positive = (iou_max >= 0.5) & has_gt[:, None]
        negative = (iou_max < 0.4) | ~has_gt[:, None]                       # no annotation: everything is background
        npos = positive.sum(1)

        onehot = torch.zeros_like(p).scatter_(2, assigned[:, :, 4:5].long().clamp(0, C - 1), 1.0)
        targets = torch.where(positive[:, :, None], onehot, torch.zeros_like(p))
        counted = (positive | negative)[:, :, None]
        is_pos = targets == 1.0
        focal = torch.where(is_pos, alpha * (1.0 - p) ** gamma, (1.0 - alpha) * p ** gamma)
        bce = -torch.where(is_pos, torch.log(p), torch.log(1.0 - p))
        cls = torch.where(counted, focal * bce, torch.zeros_like(p)).sum((1, 2))
        cls = cls / torch.where(has_gt, npos.float().clamp(min=1.0), torch.ones_like(cls))

        gw = (assigned[:, :, 2] - assigned[:, :, 0])
        gh = (assigned[:, :, 3] - assigned[:, :, 1])
        gcx, gcy = assigned[:, :, 0] + 0.5 * gw, assigned[:, :, 1] + 0.5 * gh
        gw, gh = gw.clamp(min=1), gh.clamp(min=1)
        t = torch.stack([(gcx - acx) / aw, (gcy - acy) / ah, torch.log(gw / aw), torch.log(gh / ah)], 2)
        t = torch.stack([t[..., 0] / 0.1, t[..., 1] / 0.1, t[..., 2] / 0.2, t[..., 3] / 0.2], 2)   # no host tensor: graph-safe
        diff = (t - regressions).abs()
        sl1 = torch.where(diff <= 1.0 / 9.0, 0.5 * 9.0 * diff * diff, diff - 0.5 / 9.0)
        reg = torch.where(positive[:, :, None], sl1, torch.zeros_like(sl1)).sum((1, 2))
        reg = reg / (4.0 * npos.float()).clamp(min=1.0)                      # mean over the positives' 4 coordinates
        return cls.mean(0, keepdim=True), reg.mean(0, keepdim=True)


class SegBceIou(torch.autograd.Function):
    """(sigmoid(logit), BCELoss(sigmoid(logit), mask), per-image foreground IoU) in one pass on the GPU
    (ossid_seg_bce_iou_fwd; reference: models/dtoid/__init__.py:210 `torch.sigmoid`, :216 `seg_loss_func`, :228-232 the
    IoU metric) instead of ~20 elementwise / reduction launches. Only the loss is differentiable; the probability map is
    returned detached, as the reference only feeds it to the loss and to metrics."""

    @staticmethod
    def forward(ctx, logit, mask):
        from .. import _lib
        _lib.require_cuda(logit, mask)
        x = logit.float().contiguous()
        y = mask.to(x.device).float().contiguous()
        B = x.shape[0]
        hw = x.numel() // B
        prob, dsum = torch.empty_like(x), torch.empty_like(x)
        out = torch.empty(1 + B, dtype=torch.float32, device=x.device)
        nbytes = _lib.fn("ossid_seg_bce_iou_workspace_bytes")(B)
        ws = torch.empty(nbytes // 8, dtype=torch.float64, device=x.device)
        with _lib.on_device(x.device):
            _lib.check(_lib.fn("ossid_seg_bce_iou_fwd")(x.data_ptr(), y.data_ptr(), B, hw, prob.data_ptr(), dsum.data_ptr(),
                                                        out.data_ptr(), ws.data_ptr(), nbytes, _lib.stream()),
                       "ossid_seg_bce_iou_fwd")
        ctx.save_for_backward(dsum)
        ctx.n = x.numel()
        iou = out[1:]
        ctx.mark_non_differentiable(prob, iou)
        ctx.set_materialize_grads(False)         # (no zero tensors for the two outputs nothing differentiates)
        return prob, out[0], iou

    @staticmethod
    def backward(ctx, _gprob, gloss, _giou):
        (dsum,) = ctx.saved_tensors
        if gloss is None:
            return None, None
        return dsum * (gloss / ctx.n), None
