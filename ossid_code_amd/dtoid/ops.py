"""Torch-facing wrappers of the DTOID device ops in libossid_hip.so (include/ossid_hip.h, "DTOID ops").
Tensors only cross as raw pointers; autograd sees DwXcorr as one differentiable node."""
import torch

from .. import _lib


class _DwXcorr(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, k):
        _lib.require_cuda(x, k)
        B, C, H, W = x.shape
        x = x.contiguous().float()
        k = k.contiguous().float()
        out = torch.empty_like(x)
        with torch.cuda.device(x.device):
            _lib.check(_lib.fn("ossid_dw_xcorr_fwd")(x.data_ptr(), k.data_ptr(), B * C, H, W, out.data_ptr(),
                                                     _lib.stream()), "ossid_dw_xcorr_fwd")
        ctx.save_for_backward(x, k)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, k = ctx.saved_tensors
        B, C, H, W = x.shape
        dout = dout.contiguous()
        dx = dk = None
        with torch.cuda.device(x.device):
            if ctx.needs_input_grad[0]:
                dx = torch.empty_like(x)
                _lib.check(_lib.fn("ossid_dw_xcorr_bwd_x")(dout.data_ptr(), k.data_ptr(), B * C, H, W, dx.data_ptr(),
                                                           _lib.stream()), "ossid_dw_xcorr_bwd_x")
            if ctx.needs_input_grad[1]:
                dk = torch.empty_like(k)
                _lib.check(_lib.fn("ossid_dw_xcorr_bwd_k")(x.data_ptr(), dout.data_ptr(), B * C, H, W, dk.data_ptr(),
                                                           _lib.stream()), "ossid_dw_xcorr_bwd_k")
        return dx, dk


def dw_xcorr(x, kernel):
    """x [B,C,H,W], kernel [B,C,3,3] -> [B,C,H,W]: out[b,c] = x[b,c] cross-correlated with kernel[b,c], padding 1
    (the reference's conv2d_dw_group, network.py:186-192 / :365-371). A batch-1 x is broadcast over kernel's batch."""
    if kernel.shape[-2:] != (3, 3):
        raise ValueError("dw_xcorr is built for 3x3 kernels")
    if x.shape[0] != kernel.shape[0]:
        x = x.expand(kernel.shape[0], -1, -1, -1)
    return _DwXcorr.apply(x, kernel)


def nms(boxes, scores, iou_threshold):
    """torchvision.ops.nms semantics: indices of the kept boxes, by decreasing score."""
    _lib.require_cuda(boxes, scores)
    n = int(boxes.shape[0])
    if n == 0:
        return torch.zeros(0, dtype=torch.long, device=boxes.device)
    order = torch.sort(scores, descending=True, stable=True).indices
    sb = boxes.float()[order].contiguous()
    keep = torch.empty(n, dtype=torch.int32, device=boxes.device)
    nkeep = torch.empty(1, dtype=torch.int32, device=boxes.device)
    nbytes = _lib.fn("ossid_nms_workspace_bytes")(n)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=boxes.device)
    with torch.cuda.device(boxes.device):
        _lib.check(_lib.fn("ossid_nms")(sb.data_ptr(), n, float(iou_threshold), ws.data_ptr(), nbytes, keep.data_ptr(),
                                        nkeep.data_ptr(), _lib.stream()), "ossid_nms")
    return order[keep[: int(nkeep.item())].long()]


def decode_clip_boxes(anchors, deltas, img_w, img_h):
    """anchors [1,A,4] or [A,4], deltas [R,A,4] -> clipped boxes [R,A,4] (BBoxTransform + ClipBoxes, no autograd)."""
    _lib.require_cuda(anchors, deltas)
    a = anchors.reshape(-1, 4).float().contiguous()
    d = deltas.detach().float().contiguous()
    R, A = d.shape[0], d.shape[1]
    out = torch.empty_like(d)
    with torch.cuda.device(d.device):
        _lib.check(_lib.fn("ossid_decode_clip_boxes")(a.data_ptr(), d.data_ptr(), R, A, float(img_w), float(img_h),
                                                      out.data_ptr(), _lib.stream()), "ossid_decode_clip_boxes")
    return out
