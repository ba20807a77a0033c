"""Torch-facing wrappers of the DTOID device ops in libossid_hip.so (include/ossid_hip.h, "DTOID ops").
Tensors only cross as raw pointers; autograd sees DwXcorr as one differentiable node."""
import torch

from .. import _lib


class _DwXcorr(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, k):
        _lib.require_cuda(x, k)
        B, C, H, W = k.shape[0], x.shape[1], x.shape[2], x.shape[3]
        x = x.contiguous().float()          # [B,C,H,W], or [1,C,H,W] shared by all B kernels
        k = k.contiguous().float()
        out = torch.empty((B, C, H, W), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(_lib.fn("ossid_dw_xcorr_fwd")(x.data_ptr(), x.shape[0] * C, k.data_ptr(), B * C, H, W,
                                                     out.data_ptr(), _lib.stream()), "ossid_dw_xcorr_fwd")
        if x.shape[0] != B:
            x = x.expand(B, -1, -1, -1)     # only materialised if a backward pass asks for it
        ctx.save_for_backward(x, k)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, k = ctx.saved_tensors
        x = x.contiguous()
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
        if x.shape[0] != 1:
            raise ValueError("dw_xcorr: batch of x must be 1 or equal the kernel's")
        if x.requires_grad:
            x = x.expand(kernel.shape[0], -1, -1, -1)
        else:
            x = x[:1]
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


class PackedConv3x3:
    """Weights of one nn.Conv2d(k=3, stride 1, padding 1) in the MFMA operand layout of csrc/conv.hip, plus the fused
    epilogue vectors (bias, and the eval-mode BatchNorm affine that follows the ELU in the reference head)."""

    def __init__(self, conv, bn=None, act=False):
        w = conv.weight.detach().float().contiguous()
        _lib.require_cuda(w)
        self.cout, self.cin = int(w.shape[0]), int(w.shape[1])
        if tuple(w.shape[2:]) != (3, 3) or conv.stride != (1, 1) or conv.padding != (1, 1) or self.cin % 16 or self.cout % 4:
            raise ValueError("PackedConv3x3 handles 3x3 / stride 1 / padding 1 with Cin % 16 == 0 and Cout % 4 == 0")
        n = _lib.fn("ossid_conv3x3_packed_floats")(self.cout, self.cin)
        self.wpk = torch.empty(n, dtype=torch.float32, device=w.device)
        with torch.cuda.device(w.device):
            _lib.check(_lib.fn("ossid_conv3x3_pack_weights")(w.data_ptr(), self.cout, self.cin, self.wpk.data_ptr(),
                                                             _lib.stream()), "ossid_conv3x3_pack_weights")
        self.bias = None if conv.bias is None else conv.bias.detach().float().contiguous()
        self.act = 1 if act else 0
        self.scale = self.shift = None
        if bn is not None:
            inv = torch.rsqrt(bn.running_var.detach().float() + bn.eps)
            self.scale = (bn.weight.detach().float() * inv).contiguous()
            self.shift = (bn.bias.detach().float() - bn.running_mean.detach().float() * self.scale).contiguous()

    def __call__(self, x, size=None):
        """x: logical [B,Cin,Hs,Ws] tensor (any memory format; channels_last is consumed in place) -> logical
        [B,Cout,H,W] tensor in channels_last memory format. size=(H, W) >= (Hs, Ws): the input is nearest-neighbour
        up-sampled to that size on the fly (F.interpolate(mode="nearest") fused into the patch staging)."""
        _lib.require_cuda(x)
        B, C, Hs, Ws = x.shape
        H, W = (Hs, Ws) if size is None else (int(size[0]), int(size[1]))
        if C != self.cin:
            raise ValueError("expected %d input channels, got %d" % (self.cin, C))
        x = x.float().contiguous(memory_format=torch.channels_last)
        out = torch.empty((B, self.cout, H, W), dtype=torch.float32, device=x.device,
                          memory_format=torch.channels_last)
        p = lambda t: None if t is None else t.data_ptr()  # noqa: E731
        with torch.cuda.device(x.device):
            rc = _lib.fn("ossid_conv3x3_nhwc_fwd")(x.data_ptr(), self.wpk.data_ptr(), p(self.bias), p(self.scale),
                                                   p(self.shift), out.data_ptr(), B, H, W, self.cin, self.cout, self.act,
                                                   Hs, Ws, _lib.stream())
        _lib.check(rc, "ossid_conv3x3_nhwc_fwd")
        return out
