"""Torch-facing wrappers of the DTOID device ops in libossid_hip.so (include/ossid_hip.h, "DTOID ops").
Tensors only cross as raw pointers; autograd sees DwXcorr as one differentiable node."""
import ctypes
import os

import torch

from .. import _lib

C_byref = ctypes.byref


class _DwXcorr(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, k):
        _lib.require_cuda(x, k)
        B, C, H, W = k.shape[0], x.shape[1], x.shape[2], x.shape[3]
        x = x.contiguous().float()          # [B,C,H,W], or [1,C,H,W] shared by all B kernels
        k = k.contiguous().float()
        out = torch.empty((B, C, H, W), dtype=torch.float32, device=x.device)
        with _lib.on_device(x.device):
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
        with _lib.on_device(x.device):
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


def dw_xcorr_nhwc_bcast(x, kernel):
    """x [1,C,H,W] (channels_last memory), kernel [B,C,3,3] -> [B,C,H,W] in channels_last memory: dw_xcorr with the one
    image broadcast over the B kernel sets, produced directly in the layout the next convolution reads. No autograd."""
    _lib.require_cuda(x, kernel)
    B, C = int(kernel.shape[0]), int(kernel.shape[1])
    H, W = int(x.shape[2]), int(x.shape[3])
    x = x.float().contiguous(memory_format=torch.channels_last)
    k = kernel.detach().float().contiguous()
    out = torch.empty((B, C, H, W), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    with _lib.on_device(x.device):
        _lib.check(_lib.fn("ossid_dw_xcorr_nhwc_bcast")(x.data_ptr(), k.data_ptr(), B, C, H, W, out.data_ptr(),
                                                        _lib.stream()), "ossid_dw_xcorr_nhwc_bcast")
    return out


def nms(boxes, scores, iou_threshold, sorted_desc=False):
    """torchvision.ops.nms semantics: indices of the kept boxes, by decreasing score. sorted_desc=True: the caller
    guarantees scores are already in decreasing order (the output of torch.topk) and the sort + gather are skipped."""
    _lib.require_cuda(boxes, scores)
    n = int(boxes.shape[0])
    if n == 0:
        return torch.zeros(0, dtype=torch.long, device=boxes.device)
    if sorted_desc:
        order, sb = None, boxes.float().contiguous()
    else:
        order = torch.sort(scores, descending=True, stable=True).indices
        sb = boxes.float()[order].contiguous()
    keep = torch.empty(n, dtype=torch.int32, device=boxes.device)
    nkeep = torch.empty(1, dtype=torch.int32, device=boxes.device)
    nbytes = _lib.fn("ossid_nms_workspace_bytes")(n)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=boxes.device)
    with _lib.on_device(boxes.device):
        _lib.check(_lib.fn("ossid_nms")(sb.data_ptr(), n, float(iou_threshold), ws.data_ptr(), nbytes, keep.data_ptr(),
                                        nkeep.data_ptr(), _lib.stream()), "ossid_nms")
    kept = keep[: int(nkeep.item())].long()
    return kept if order is None else order[kept]


def topk_scores(scores, k):
    """(values, indices) of the k largest entries of a 1-D score vector, in decreasing order (network.py:555,
    `torch.topk(classifications, 1000, dim=1)` on the object column). Equal scores are ordered, and at the cut chosen,
    by increasing index. On the GPU: ossid_topk (radix select + sort of the k survivors); k > 2048 or CPU: torch."""
    n = int(scores.shape[0])
    if not scores.is_cuda or k > 2048 or k <= 0:
        return torch.topk(scores, k)
    s = scores.float().contiguous()
    vals = torch.empty(k, dtype=torch.float32, device=s.device)
    idx = torch.empty(k, dtype=torch.int64, device=s.device)
    nbytes = _lib.fn("ossid_topk_workspace_bytes")(n, k)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=s.device)
    with _lib.on_device(s.device):
        _lib.check(_lib.fn("ossid_topk")(s.data_ptr(), n, k, ws.data_ptr(), nbytes, vals.data_ptr(), idx.data_ptr(),
                                         _lib.stream()), "ossid_topk")
    return vals, idx


class DetectPost:
    """Device-side state of one frame's post-processing (ossid_detect_post): the k best object scores in decreasing order, their
    flat indices (template * A + anchor), their decoded + clipped boxes, the NMS keep list over them and its count. Everything is
    written by launches only -- no read-back -- so the call can sit inside the frame's captured graph; `count_host` (pinned) is
    where the caller copies the count to afterwards (outside any capture: a device-to-host copy node aborted at replay)."""

    def __init__(self, n, k, A, device):
        self.n, self.k, self.A = n, k, A
        self.scores = torch.empty(k, dtype=torch.float32, device=device)
        self.indices = torch.empty(k, dtype=torch.int64, device=device)
        self.boxes = torch.empty((k, 4), dtype=torch.float32, device=device)
        self.keep = torch.empty(k, dtype=torch.int32, device=device)
        self.count = torch.zeros(1, dtype=torch.int32, device=device)
        self.count_host = torch.zeros(1, dtype=torch.int32).pin_memory()
        self.ws = torch.empty(_lib.fn("ossid_detect_post_workspace_bytes")(n, k), dtype=torch.uint8, device=device)


def detect_post(cls_all, reg_all, anchors, img_w, img_h, k, iou_threshold=0.5, state=None):
    """network.py:543-566 on the device: cls_all [n_t, A, 2] (class 1 = object), reg_all [n_t, A, 4], anchors [1, A, 4] ->
    DetectPost. Launches only (capturable)."""
    _lib.require_cuda(cls_all, reg_all, anchors)
    n_t, A = int(reg_all.shape[0]), int(reg_all.shape[1])
    n = n_t * A
    cls_all, reg_all = cls_all.detach(), reg_all.detach()
    if cls_all.dtype != torch.float32 or not cls_all.is_contiguous() or cls_all.shape[-1] != 2:
        cls_all = cls_all.float().contiguous()
    if reg_all.dtype != torch.float32 or not reg_all.is_contiguous():
        reg_all = reg_all.float().contiguous()
    a = anchors.reshape(-1, 4)
    if a.dtype != torch.float32 or not a.is_contiguous():
        a = a.float().contiguous()
    if state is None or state.n != n or state.k != k or state.A != A or state.scores.device != reg_all.device:
        state = DetectPost(n, k, A, reg_all.device)
    with _lib.on_device(reg_all.device):
        _lib.check(_lib.fn("ossid_detect_post")(cls_all.data_ptr() + 4, n, 2, k, a.data_ptr(), reg_all.data_ptr(), A, float(img_w),
                                                float(img_h), float(iou_threshold), state.ws.data_ptr(), state.ws.numel(),
                                                state.scores.data_ptr(), state.indices.data_ptr(), state.boxes.data_ptr(),
                                                state.keep.data_ptr(), state.count.data_ptr(), _lib.stream()), "ossid_detect_post")
    state._inputs = (cls_all, reg_all, a)          # (alive until the launches have run)
    return state


def detect_emit(state, count, seg_all, heat_all, seg_sigmoid=False):
    """The detection list of network.py:566-581 for the first `count` kept candidates of a DetectPost, ONE launch:
    [scores [c], boxes [c,4], template index [c,1] (float), seg [c,H,W], heat [c,hh,hw]]. seg_all [n_t,H,W], heat_all [n_t,hh,hw]."""
    dev = state.scores.device
    seg_all = seg_all if (seg_all.dtype == torch.float32 and seg_all.is_contiguous()) else seg_all.float().contiguous()
    heat_all = heat_all if (heat_all.dtype == torch.float32 and heat_all.is_contiguous()) else heat_all.float().contiguous()
    o_s = torch.empty(count, dtype=torch.float32, device=dev)
    o_b = torch.empty((count, 4), dtype=torch.float32, device=dev)
    o_o = torch.empty((count, 1), dtype=torch.float32, device=dev)
    o_seg = torch.empty((count,) + tuple(seg_all.shape[1:]), dtype=torch.float32, device=dev)
    o_heat = torch.empty((count,) + tuple(heat_all.shape[1:]), dtype=torch.float32, device=dev)
    if count:
        seg_row = int(seg_all[0].numel())
        heat_row = int(heat_all[0].numel())
        with _lib.on_device(dev):
            _lib.check(_lib.fn("ossid_detect_emit")(state.scores.data_ptr(), state.indices.data_ptr(), state.boxes.data_ptr(),
                                                    state.keep.data_ptr(), count, state.A, seg_all.data_ptr(), seg_row,
                                                    heat_all.data_ptr(), heat_row, 1 if seg_sigmoid else 0, o_s.data_ptr(),
                                                    o_b.data_ptr(), o_o.data_ptr(), o_seg.data_ptr(), o_heat.data_ptr(),
                                                    _lib.stream()), "ossid_detect_emit")
    return [o_s, o_b, o_o, o_seg, o_heat]


_IMNORM = {}


def im2col_stem(img, k, stride, pad, kpad, normalize=False, out=None):
    """ossid_im2col_stem: img [B,Cin,H,W] (NCHW) -> [B, kpad, Ho, Wo] logical tensor in channels_last memory (rows of
    receptive fields, column (ky*k + kx)*Cin + ci), optionally with normalizeImageRange applied on the way. out: a
    tensor of that shape and layout to write into (a persistent buffer of a recorded launch sequence)."""
    _lib.require_cuda(img)
    img = img.float().contiguous()
    B, Cin, H, W = img.shape
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    if out is None:
        out = torch.empty((B, Ho, Wo, kpad), dtype=torch.float32, device=img.device).permute(0, 3, 1, 2)
    elif tuple(out.shape) != (B, kpad, Ho, Wo) or not out.is_contiguous(memory_format=torch.channels_last):
        raise ValueError("im2col_stem: out must be a channels-last [B, kpad, Ho, Wo] tensor")
    mean = inv = None
    if normalize:
        key = str(img.device)
        if key not in _IMNORM:
            _IMNORM[key] = (torch.tensor([0.485, 0.456, 0.406], device=img.device),
                            1.0 / torch.tensor([0.229, 0.224, 0.225], device=img.device))
        mean, inv = _IMNORM[key]
    with _lib.on_device(img.device):
        rc = _lib.fn("ossid_im2col_stem")(img.data_ptr(), B, Cin, H, W, k, stride, pad, kpad, None if mean is None else
                                          mean.data_ptr(), None if inv is None else inv.data_ptr(), out.data_ptr(), _lib.stream())
    _lib.check(rc, "ossid_im2col_stem")
    return out


def conv1x1_c1(x, conv, sigmoid=False):
    """nn.Conv2d(C, 1, 1) (`corr_conv_heatmap`, network.py:334) on a channels-last x [B,C,H,W] -> [B,1,H,W], no autograd,
    optionally through the sigmoid that follows it (network.py:349): one deterministic pass (ossid_conv1x1_c1_fwd)."""
    _lib.require_cuda(x)
    B, C, H, W = x.shape
    x = x.float().contiguous(memory_format=torch.channels_last)
    w = conv.weight.detach().float().reshape(-1).contiguous()
    out = torch.empty((B, 1, H, W), dtype=torch.float32, device=x.device)
    with _lib.on_device(x.device):
        _lib.check(_lib.fn("ossid_conv1x1_c1_fwd")(x.data_ptr(), B * H * W, C, w.data_ptr(),
                                                   None if conv.bias is None else conv.bias.detach().data_ptr(), 1 if sigmoid else 0,
                                                   out.data_ptr(), _lib.stream()), "ossid_conv1x1_c1_fwd")
    return out


class SpatialMean(torch.autograd.Function):
    """F.avg_pool2d(x, (H, W)) of a [B,C,H,W] tensor in either memory format -> [B,C,1,1] (network.py:343: the average of the
    7x7 template features), ossid_spatial_mean forward and backward."""

    @staticmethod
    def forward(ctx, x):
        B, C, H, W = x.shape
        cl = x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous()
        if not cl:
            x = x.contiguous()
        x = x.float()
        out = torch.empty((B, C, 1, 1), dtype=torch.float32, device=x.device)
        with _lib.on_device(x.device):
            _lib.check(_lib.fn("ossid_spatial_mean")(x.data_ptr(), B, H * W, C, 1 if cl else 0, 0, out.data_ptr(), _lib.stream()),
                       "ossid_spatial_mean")
        ctx.cfg = (B, C, H, W, cl)
        return out

    @staticmethod
    def backward(ctx, g):
        B, C, H, W, cl = ctx.cfg
        g = g.float().contiguous()
        dx = torch.empty((B, C, H, W), dtype=torch.float32, device=g.device,
                         memory_format=torch.channels_last if cl else torch.contiguous_format)
        with _lib.on_device(g.device):
            _lib.check(_lib.fn("ossid_spatial_mean")(g.data_ptr(), B, H * W, C, 1 if cl else 0, 1, dx.data_ptr(), _lib.stream()),
                       "ossid_spatial_mean")
        return dx


def spatial_mean(x):
    """F.avg_pool2d(x, full window) on the GPU through SpatialMean; torch on the CPU."""
    if not x.is_cuda:
        return torch.nn.functional.avg_pool2d(x, (x.shape[2], x.shape[3]))
    return SpatialMean.apply(x)


def small_matmul(a, b):
    """a [M,K] @ b [K,N] for a handful of rows, no autograd (ossid_small_matmul: no library GEMM on the per-object path)."""
    _lib.require_cuda(a, b)
    a, b = a.float().contiguous(), b.float().contiguous()
    M, K = a.shape
    N = b.shape[1]
    out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    with _lib.on_device(a.device):
        _lib.check(_lib.fn("ossid_small_matmul")(a.data_ptr(), b.data_ptr(), M, K, N, out.data_ptr(), _lib.stream()), "ossid_small_matmul")
    return out


def avgpool2_nhwc(x, stride):
    """nn.AvgPool2d(2, stride) on a channels-last tensor, no autograd (DenseNet transitions at test time)."""
    B, C, H, W = x.shape
    Ho, Wo = (H - 2) // stride + 1, (W - 2) // stride + 1
    out = torch.empty((B, C, Ho, Wo), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    with _lib.on_device(x.device):
        _lib.check(_lib.fn("ossid_avgpool2_nhwc")(x.data_ptr(), B, H, W, C, stride, out.data_ptr(), 0, _lib.stream()), "ossid_avgpool2_nhwc")
    return out


def _imnorm(device):
    key = str(device)
    if key not in _IMNORM:
        _IMNORM[key] = (torch.tensor([0.485, 0.456, 0.406], device=device), 1.0 / torch.tensor([0.229, 0.224, 0.225], device=device))
    return _IMNORM[key]


def stem_conv(img, conv, normalize=False):
    """DenseNet conv0 (nn.Conv2d(3, 64, 7, stride 2, padding 3)) on the NCHW image -> channels-last [B,64,Ho,Wo], no autograd:
    ossid_stem_conv_fwd reads the parameter's own layout (nothing to pack or refresh; a captured graph sees weight updates)
    and applies normalizeImageRange while staging when `normalize`."""
    _lib.require_cuda(img, conv.weight)
    img = img.float().contiguous()
    B, Cin, H, W = img.shape
    w = conv.weight.detach()
    k = int(w.shape[2])
    Ho, Wo = (H + 6 - k) // 2 + 1, (W + 6 - k) // 2 + 1
    out = torch.empty((B, int(w.shape[0]), Ho, Wo), dtype=torch.float32, device=img.device, memory_format=torch.channels_last)
    mean, inv = _imnorm(img.device) if normalize else (None, None)
    with _lib.on_device(img.device):
        rc = _lib.fn("ossid_stem_conv_fwd")(img.data_ptr(), B, Cin, H, W, w.data_ptr(), int(w.shape[0]), k, conv.stride[0],
                                            conv.padding[0], None if conv.bias is None else conv.bias.detach().data_ptr(),
                                            None if mean is None else mean.data_ptr(), None if inv is None else inv.data_ptr(),
                                            out.data_ptr(), _lib.stream())
    _lib.check(rc, "ossid_stem_conv_fwd")
    return out


def stem_tail(x0, kernels, scale, shift):
    """relu(scale * (x0 + dw_xcorr(x0, kernels)) + shift) on a channels-last x0 [B,C,H,W]; kernels [B or 1, C, 3, 3]."""
    B, C, H, W = x0.shape
    k = kernels.detach().float().contiguous()
    out = torch.empty_like(x0)
    with _lib.on_device(x0.device):
        rc = _lib.fn("ossid_stem_tail_nhwc")(x0.data_ptr(), k.data_ptr(), 0 if k.shape[0] == 1 else C * 9, scale.data_ptr(),
                                             shift.data_ptr(), B, H, W, C, out.data_ptr(), _lib.stream())
    _lib.check(rc, "ossid_stem_tail_nhwc")
    return out


def stem_tail_pool(x0, kernels, scale, shift, out=None):
    """maxpool(3, 2, 1) of relu(scale * (x0 + dw_xcorr(x0, kernels)) + shift) in one pass (test time: stem_tail + pool0 of
    DenseNet). out: a channels-last [B, >= C, Ho, Wo] tensor whose first C channels receive the result (a dense block's
    resident buffer); None = a fresh [B, C, Ho, Wo]."""
    B, C, H, W = x0.shape
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    k = kernels.detach().float().contiguous()
    if out is None:
        out = torch.empty((B, C, Ho, Wo), dtype=torch.float32, device=x0.device, memory_format=torch.channels_last)
    elif tuple(out.shape[2:]) != (Ho, Wo) or out.shape[0] != B or out.shape[1] < C or not out.is_contiguous(memory_format=torch.channels_last):
        raise ValueError("stem_tail_pool: out must be a channels-last [B, >= C, Ho, Wo] tensor")
    with _lib.on_device(x0.device):
        rc = _lib.fn("ossid_stem_tail_pool_nhwc")(x0.data_ptr(), k.data_ptr(), 0 if k.shape[0] == 1 else C * 9, scale.data_ptr(),
                                                  shift.data_ptr(), B, H, W, C, out.data_ptr(), int(out.shape[1]), _lib.stream())
    _lib.check(rc, "ossid_stem_tail_pool_nhwc")
    return out


def bn_relu_avgpool2(x, C, scale, shift, stride):
    """avgpool2x2(relu(scale * x[:, :C] + shift), stride) on a channels-last x [B, >= C, H, W] -> [B, C, Ho, Wo] (the front of
    a DenseNet transition with the pool moved in front of its bias-free 1x1 convolution)."""
    B, ctot, H, W = x.shape
    Ho, Wo = (H - 2) // stride + 1, (W - 2) // stride + 1
    out = torch.empty((B, C, Ho, Wo), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    with _lib.on_device(x.device):
        _lib.check(_lib.fn("ossid_bn_relu_avgpool2_nhwc")(x.data_ptr(), B, H, W, C, int(ctot), scale.data_ptr(), shift.data_ptr(),
                                                          stride, out.data_ptr(), _lib.stream()), "ossid_bn_relu_avgpool2_nhwc")
    return out


def maxpool_nhwc(x, k, stride, pad=0, ceil_mode=False):
    """nn.MaxPool2d(k, stride, pad, ceil_mode=ceil_mode) on a channels-last tensor."""
    B, C, H, W = x.shape

    def osz(n):
        o = -(-(n + 2 * pad - k) // stride) + 1 if ceil_mode else (n + 2 * pad - k) // stride + 1
        return o - 1 if ceil_mode and (o - 1) * stride >= n + pad else o
    out = torch.empty((B, C, osz(H), osz(W)), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    with _lib.on_device(x.device):
        rc = _lib.fn("ossid_maxpool_nhwc")(x.data_ptr(), B, H, W, C, k, stride, pad, 1 if ceil_mode else 0, out.data_ptr(),
                                           _lib.stream())
    _lib.check(rc, "ossid_maxpool_nhwc")
    return out


class _StemAsMatrix:
    """A k x k strided stem convolution presented to PackedConv as the 1x1 convolution that follows ossid_im2col_stem:
    weight [Cout, Cin, k, k] re-laid to [Cout, kpad, 1, 1] in the im2col column order (ky*k + kx)*Cin + ci."""
    padding, stride, groups = (0, 0), (1, 1), 1

    def __init__(self, conv, kpad):
        self.conv, self.kpad = conv, kpad

    @property
    def weight(self):
        w = self.conv.weight.detach().float()
        cout = w.shape[0]
        flat = w.permute(0, 2, 3, 1).reshape(cout, -1)
        out = torch.zeros((cout, self.kpad), dtype=torch.float32, device=w.device)
        out[:, : flat.shape[1]] = flat
        return out.view(cout, self.kpad, 1, 1)

    @property
    def bias(self):
        return self.conv.bias


def gather_rows(src, idx, sigmoid=False):
    """src [R, ...] float32, idx [k] int64 -> src[idx] (optionally through a sigmoid), one pass over the data."""
    _lib.require_cuda(src, idx)
    src = src.float().contiguous()
    idx = idx.long().contiguous()
    k, row = int(idx.shape[0]), int(src[0].numel()) if src.shape[0] else 0
    out = torch.empty((k,) + tuple(src.shape[1:]), dtype=torch.float32, device=src.device)
    if k == 0 or row == 0:
        return out
    if row % 4:
        g = src[idx]
        return torch.sigmoid(g) if sigmoid else g
    with _lib.on_device(src.device):
        _lib.check(_lib.fn("ossid_gather_rows")(src.data_ptr(), int(src.shape[0]), row, idx.data_ptr(), k,
                                                1 if sigmoid else 0, out.data_ptr(), _lib.stream()), "ossid_gather_rows")
    return out


def decode_clip_boxes(anchors, deltas, img_w, img_h):
    """anchors [1,A,4] or [A,4], deltas [R,A,4] -> clipped boxes [R,A,4] (BBoxTransform + ClipBoxes, no autograd)."""
    _lib.require_cuda(anchors, deltas)
    a = anchors.reshape(-1, 4).float().contiguous()
    d = deltas.detach().float().contiguous()
    R, A = d.shape[0], d.shape[1]
    out = torch.empty_like(d)
    with _lib.on_device(d.device):
        _lib.check(_lib.fn("ossid_decode_clip_boxes")(a.data_ptr(), d.data_ptr(), R, A, float(img_w), float(img_h),
                                                      out.data_ptr(), _lib.stream()), "ossid_decode_clip_boxes")
    return out


def _bn_affine(bn):
    """eval-mode BatchNorm2d as a per-channel (scale, shift)"""
    inv = torch.rsqrt(bn.running_var.detach().float() + bn.eps)
    scale = (bn.weight.detach().float() * inv).contiguous()
    return scale, (bn.bias.detach().float() - bn.running_mean.detach().float() * scale).contiguous()


USE_PHASE_CONV = os.environ.get("OSSID_PHASE_CONV", "1") != "0"


USE_WINO = os.environ.get("OSSID_WINO", "1") != "0"
WINO_MIN_WGS = int(os.environ.get("OSSID_WINO_MIN_WGS", "256"))


_WINO_WS = {}


def wino_workspace(descs, device):
    """Scratch for the Winograd launch's tail split (include/ossid_hip.h ossid_conv3x3_wino_workspace_bytes), one grow-only
    buffer per (device, stream); sets it in the (first) descriptor. descs: one ConvDesc, or two for the pair entry."""
    if len(descs) == 1:
        n = _lib.fn("ossid_conv3x3_wino_workspace_bytes")(C_byref(descs[0]))
    else:
        n = _lib.fn("ossid_conv3x3_wino_pair_workspace_bytes")(C_byref(descs[0]), C_byref(descs[1]))
    if not n:
        return
    rec = _lib.recording()
    if rec is not None:                        # a recorded sequence owns its scratch (train_ops._scratch has the reason)
        buf = rec.scratch("wino", n, device)
    else:
        key = (str(device), _lib.stream())
        buf = _WINO_WS.get(key)
        if buf is None or buf.numel() < n:
            buf = _WINO_WS[key] = torch.empty(max(int(n), 16 << 20), dtype=torch.uint8, device=device)
    descs[0].scratch, descs[0].scratch_bytes = buf.data_ptr(), buf.numel()


class PackedConv:
    """One nn.Conv2d (3x3 / stride 1 / padding 1, or 1x1) in the MFMA operand layout of csrc/conv.hip with its fused
    neighbours: `pre_bn` (+ReLU) = eval-mode BatchNorm in FRONT of the conv (DenseNet's BN-ReLU-Conv), `act` = ELU and
    `bn` = eval-mode BatchNorm BEHIND it (the head's norm(F.elu(conv(x))))."""

    def __init__(self, conv, bn=None, act=False, pre_bn=None, pre_relu=False, phases=False, wino=True):
        w = conv.weight.detach().float().contiguous()
        _lib.require_cuda(w)
        self.cout, self.cin = int(w.shape[0]), int(w.shape[1])
        k = tuple(w.shape[2:])
        ok = (k == (3, 3) and conv.padding == (1, 1)) or (k == (1, 1) and conv.padding == (0, 0))
        if not ok or conv.stride != (1, 1) or conv.groups != 1 or self.cin % 16 or self.cout % 4:
            raise ValueError("PackedConv handles 3x3/pad 1 and 1x1, stride 1, Cin % 16 == 0, Cout % 4 == 0")
        self.taps = k[0] * k[1]
        n = _lib.fn("ossid_conv_packed_floats")(self.cout, self.cin, self.taps)
        self.wpk = torch.empty(n, dtype=torch.float32, device=w.device)
        self.bias = None if conv.bias is None else torch.empty_like(conv.bias, dtype=torch.float32)
        self.act = int(act) if not isinstance(act, bool) else (1 if act else 0)       # 0 none, 1 ELU, 2 ReLU
        mk = lambda m: (None, None) if m is None else tuple(torch.empty_like(t) for t in _bn_affine(m))  # noqa: E731
        self.scale, self.shift = mk(bn)
        self.pre_scale, self.pre_shift = mk(pre_bn)
        self.pre_relu = 1 if pre_relu else 0
        self._src = (conv, bn, pre_bn)
        # 3x3 layers that may be called behind an exact 2x nearest up-sampling also keep the four PHASE weight sets
        # (2x2 kernels with merged rows / columns, csrc/conv.hip TAPS = 4): 4/9 of the multiply-adds
        self.wpk4 = None
        if self.taps == 9 and pre_bn is None and phases:      # only the decoder layers behind a 2x up-sampling ask for them
            n4 = _lib.fn("ossid_conv_packed_floats")(self.cout, self.cin, 4)
            self.wpk4 = torch.empty(4 * n4, dtype=torch.float32, device=w.device)
        # 3x3 layers with at least one full pair of channel tiles also keep U = G g G^T for the Winograd F(2x2,3x3) kernel
        # (csrc/wino.hip: 16 instead of 36 multiplies per 2x2 outputs); run() takes it when the launch fills the chip
        self.wpk_wino = None
        if self.taps == 9 and self.cout >= 64 and USE_WINO and wino:
            self.wpk_wino = torch.empty(_lib.fn("ossid_conv_wino_packed_floats")(self.cout, self.cin), dtype=torch.float32,
                                        device=w.device)
        self.refresh()

    def refresh(self):
        """(Re)read the source modules INTO THE SAME device buffers: a captured hipGraph that launches this layer stays
        valid across parameter updates (the online loop finetunes the detector every few frames)."""
        conv, bn, pre_bn = self._src
        w = conv.weight.detach().float().contiguous()
        with _lib.on_device(w.device):
            _lib.check(_lib.fn("ossid_conv_pack_weights")(w.data_ptr(), self.cout, self.cin, self.taps,
                                                          self.wpk.data_ptr(), _lib.stream()), "ossid_conv_pack_weights")
        if self.wpk_wino is not None:
            with _lib.on_device(w.device):
                _lib.check(_lib.fn("ossid_conv_pack_weights_wino")(w.data_ptr(), self.cout, self.cin, 0, self.wpk_wino.data_ptr(),
                                                                   _lib.stream()), "ossid_conv_pack_weights_wino")
        if self.wpk4 is not None:
            n4 = self.wpk4.numel() // 4
            rows = ((w[:, :, 0], w[:, :, 1] + w[:, :, 2]), (w[:, :, 0] + w[:, :, 1], w[:, :, 2]))       # phase a: [Cout,Cin,3] x2
            for a in range(2):
                for b in range(2):
                    r0, r1 = rows[a]
                    cols = (lambda r: (r[:, :, 0], r[:, :, 1] + r[:, :, 2])) if b == 0 else \
                        (lambda r: (r[:, :, 0] + r[:, :, 1], r[:, :, 2]))
                    w4 = torch.stack([torch.stack(cols(r0), -1), torch.stack(cols(r1), -1)], -2).contiguous()   # [Cout,Cin,2,2]
                    with _lib.on_device(w.device):
                        _lib.check(_lib.fn("ossid_conv_pack_weights")(
                            w4.data_ptr(), self.cout, self.cin, 4, self.wpk4[(a * 2 + b) * n4:].data_ptr(), _lib.stream()),
                            "ossid_conv_pack_weights")
        if self.bias is not None:
            self.bias.copy_(conv.bias.detach())
        for mod, sc, sh in ((bn, self.scale, self.shift), (pre_bn, self.pre_scale, self.pre_shift)):
            if mod is not None:
                a, b = _bn_affine(mod)
                sc.copy_(a)
                sh.copy_(b)

    def _desc(self, x_nhwc, B, H, W, out_nhwc, in_cs=0, out_cs=0, out_coff=0, src_hw=(0, 0), in_bs=-1, pre=None, skip_pre=False):
        """(ossid_conv_desc, entry name) of one launch; the Winograd form when the layer has it and the launch fills the chip."""
        d = _lib.ConvDesc()
        p = lambda t: None if t is None else t.data_ptr()  # noqa: E731
        d.x, d.wpk, d.bias, d.out = x_nhwc.data_ptr(), self.wpk.data_ptr(), p(self.bias), out_nhwc.data_ptr()
        d.pre_scale, d.pre_shift, d.post_scale, d.post_shift = p(self.pre_scale), p(self.pre_shift), p(self.scale), p(self.shift)
        d.in_batch_stride, d.pre_batch_stride = in_bs, 0
        if skip_pre:                 # the caller has applied the input affine (+ReLU) itself (pooled transitions)
            d.pre_scale = d.pre_shift = None
        if pre is not None:
            d.pre_scale, d.pre_shift, d.pre_batch_stride = pre[0].data_ptr(), pre[1].data_ptr(), self.cin
        d.batch, d.height, d.width, d.cin, d.cout, d.taps = B, H, W, self.cin, self.cout, self.taps
        d.act, d.pre_relu = self.act, (0 if skip_pre else self.pre_relu)
        d.src_height, d.src_width = int(src_hw[0]), int(src_hw[1])
        d.in_channel_stride, d.out_channel_stride, d.out_channel_offset = in_cs, out_cs, out_coff
        name = "ossid_conv_nhwc_fwd"
        if d.src_height in (0, H) and d.src_width in (0, W) and self.use_wino(B, H, W):
            name, d.wpk = "ossid_conv3x3_wino_fwd", self.wpk_wino.data_ptr()
        return d, name

    def use_wino(self, B, H, W):
        """Does a launch on [B][H][W] (no fused up-sampling) take the Winograd kernel? The layer must have the layout, and
        the launch enough workgroups (32 tiles of 2x2 outputs x 64 channels each): under one per CU the direct kernel's
        split-reduction variants are the better fit."""
        if self.wpk_wino is None:
            return False
        return ((B * ((H + 1) // 2) * ((W + 1) // 2) + 31) // 32) * ((self.cout + 63) // 64) >= WINO_MIN_WGS

    def run(self, x_nhwc, B, H, W, out_nhwc, in_cs=0, out_cs=0, out_coff=0, src_hw=(0, 0), in_bs=-1, pre=None, skip_pre=False):
        """Raw call on physical [B][H][W][C] buffers (tensors only provide pointers). in_bs = 0: x is ONE image shared by
        the whole batch; pre = (scale [B,Cin], shift [B,Cin]): a per-image input affine instead of the stored one."""
        d, name = self._desc(x_nhwc, B, H, W, out_nhwc, in_cs, out_cs, out_coff, src_hw, in_bs, pre, skip_pre)
        with _lib.on_device(out_nhwc.device):
            if name == "ossid_conv3x3_wino_fwd":
                wino_workspace((d,), out_nhwc.device)
            _lib.check(_lib.fn(name)(C_byref(d), _lib.stream()), name)
        return out_nhwc

    @staticmethod
    def run_pair(pk0, args0, pk1, args1):
        """Two independent layers (args = the positional / keyword arguments of run() as (tuple, dict)): ONE grid when
        both take the Winograd kernel (ossid_conv3x3_wino_fwd_pair), two launches otherwise."""
        d0, n0 = pk0._desc(*args0[0], **args0[1])
        d1, n1 = pk1._desc(*args1[0], **args1[1])
        dev = args0[0][4].device
        with _lib.on_device(dev):
            if n0 == n1 == "ossid_conv3x3_wino_fwd":
                wino_workspace((d0, d1), dev)
                _lib.check(_lib.fn("ossid_conv3x3_wino_fwd_pair")(C_byref(d0), C_byref(d1), _lib.stream()),
                           "ossid_conv3x3_wino_fwd_pair")
            else:
                for d, n in ((d0, n0), (d1, n1)):
                    if n == "ossid_conv3x3_wino_fwd":
                        wino_workspace((d,), dev)
                    _lib.check(_lib.fn(n)(C_byref(d), _lib.stream()), n)

    def __call__(self, x, size=None):
        """x: logical [B,Cin,Hs,Ws] tensor (any memory format; channels_last is consumed in place) -> logical
        [B,Cout,H,W] tensor in channels_last memory format. size=(H, W) >= (Hs, Ws) (3x3 only): the input is
        nearest-neighbour up-sampled to that size on the fly."""
        _lib.require_cuda(x)
        B, C, Hs, Ws = x.shape
        H, W = (Hs, Ws) if size is None else (int(size[0]), int(size[1]))
        if C != self.cin:
            raise ValueError("expected %d input channels, got %d" % (self.cin, C))
        x = x.float().contiguous(memory_format=torch.channels_last)
        out = torch.empty((B, self.cout, H, W), dtype=torch.float32, device=x.device,
                          memory_format=torch.channels_last)
        if self.wpk4 is not None and USE_PHASE_CONV and (H, W) == (2 * Hs, 2 * Ws) and self.pre_scale is None:
            if self.run_phases(x, B, Hs, Ws, out):
                return out
        return self.run(x, B, H, W, out, src_hw=(Hs, Ws) if size is not None else (0, 0))

    def run_phases(self, x_nhwc, B, Hs, Ws, out_nhwc):
        """conv3x3(nearest_upsample_2x(x)) as four 2x2 phase convolutions of the source (one launch). False: not taken."""
        d = _lib.ConvDesc()
        p = lambda t: None if t is None else t.data_ptr()  # noqa: E731
        d.x, d.wpk, d.bias, d.out = x_nhwc.data_ptr(), self.wpk4.data_ptr(), p(self.bias), out_nhwc.data_ptr()
        d.post_scale, d.post_shift = p(self.scale), p(self.shift)
        d.batch, d.height, d.width, d.cin, d.cout, d.taps, d.act = B, Hs, Ws, self.cin, self.cout, 4, self.act
        d.in_batch_stride = -1
        with _lib.on_device(out_nhwc.device):
            rc = _lib.fn("ossid_conv_nhwc_fwd")(C_byref(d), _lib.stream())
        if rc == -22:
            return False
        _lib.check(rc, "ossid_conv_nhwc_fwd")
        return True


def dense_block_table(layers, growth):
    """The device table ossid_dense_entry / ossid_dense_layer read for one DenseNet block: per layer conv1's packed weights,
    norm1's scale / shift and the layer's input width in 16-channel units. layers: [(PackedConv conv1, PackedConv conv2)].
    None when the block is not densenet121-shaped (growth 32, bottleneck 128) or the library has no split form: the caller
    then keeps the two-launch path. The table holds ADDRESSES of the PackedConv buffers, which refresh() rewrites in place."""
    if growth != 32 or not layers or not _lib.fn("ossid_dense_fused_available")():
        return None
    rows = []
    for li, (c1, c2) in enumerate(layers):
        if c1.cout != 128 or c1.taps != 1 or c2.cin != 128 or c2.cout != 32 or c2.taps != 9 or c1.cin != layers[0][0].cin + 32 * li:
            return None
        rows.append([c1.wpk.data_ptr(), c1.pre_scale.data_ptr(), c1.pre_shift.data_ptr(), c1.cin // 16])
    return torch.tensor(rows, dtype=torch.int64).to(layers[0][0].wpk.device)


def dense_block_fused(buf, B, H, W, C0, layers, table):
    """A DenseNet block at test time on csrc/dense.hip: buf [B][H][W][ctot] (channels-last) holds the block's input in its
    first C0 channels; every layer appends its 32 channels in place. One launch per layer plus the block entry."""
    L = len(layers)
    ctot = int(buf.shape[1])
    P = B * H * W
    y = torch.empty((L, P, 128), dtype=torch.float32, device=buf.device)
    with _lib.on_device(buf.device):
        s = _lib.stream()
        _lib.check(_lib.fn("ossid_dense_entry")(buf.data_ptr(), ctot, C0, P, L, table.data_ptr(), y.data_ptr(), s), "ossid_dense_entry")
        for li, (c1, c2) in enumerate(layers):
            _lib.check(_lib.fn("ossid_dense_layer")(y.data_ptr(), buf.data_ptr(), B, H, W, ctot, C0, li, L, c2.wpk.data_ptr(),
                                                    c2.pre_scale.data_ptr(), c2.pre_shift.data_ptr(), table.data_ptr(), s),
                       "ossid_dense_layer")
    return buf


class SegTail:
    """The decoder's last two layers as ONE launch (csrc/segtail.hip): nearest up-sample to `size` -> conv3x3 32->16 ->
    ELU -> eval BatchNorm -> conv3x3 16->1 (network.py:357-362: s5/ns5 after F.interpolate, then seg_final).
    `__call__` returns None when the kernel does not take the shape (up-sampling ratio under ~1.5): the caller then
    runs the two layers separately."""

    def __init__(self, conv1, bn1, conv2):
        w1 = conv1.weight.detach().float().contiguous()
        w2 = conv2.weight.detach().float().contiguous()
        _lib.require_cuda(w1)
        if tuple(w1.shape) != (16, 32, 3, 3) or tuple(w2.shape) != (1, 16, 3, 3) or conv1.padding != (1, 1) or \
                conv2.padding != (1, 1) or conv1.stride != (1, 1) or conv2.stride != (1, 1):
            raise ValueError("SegTail is the 32->16->1 tail of the DTOID decoder")
        self.w1p = torch.empty(_lib.fn("ossid_seg_tail_packed_floats")(), dtype=torch.float32, device=w1.device)
        z = lambda n: torch.zeros(n, dtype=torch.float32, device=w1.device)  # noqa: E731
        self.b1, self.scale, self.shift, self.w2, self.b2 = z(16), z(16), z(16), torch.empty(16, 9, device=w1.device), z(1)
        self._src = (conv1, bn1, conv2)
        self.refresh()

    def refresh(self):
        """Re-read the three source modules into the same device buffers (see PackedConv.refresh)."""
        conv1, bn1, conv2 = self._src
        w1 = conv1.weight.detach().float().contiguous()
        with _lib.on_device(w1.device):
            _lib.check(_lib.fn("ossid_seg_tail_pack_weights")(w1.data_ptr(), self.w1p.data_ptr(), _lib.stream()),
                       "ossid_seg_tail_pack_weights")
        if conv1.bias is not None:
            self.b1.copy_(conv1.bias.detach())
        a, b = _bn_affine(bn1)
        self.scale.copy_(a)
        self.shift.copy_(b)
        self.w2.copy_(conv2.weight.detach().float().reshape(16, 9))
        if conv2.bias is not None:
            self.b2.copy_(conv2.bias.detach().reshape(1))

    def __call__(self, x, size):
        _lib.require_cuda(x)
        B, C, Hs, Ws = x.shape
        H, W = int(size[0]), int(size[1])
        if C != 32:
            raise ValueError("expected 32 input channels, got %d" % C)
        if -(-17 * Hs // H) + 2 > 12 or -(-33 * Ws // W) + 2 > 20 or Hs > H or Ws > W:   # the kernel's patch capacity
            return None
        x = x.float().contiguous(memory_format=torch.channels_last)
        out = torch.empty((B, 1, H, W), dtype=torch.float32, device=x.device)
        with _lib.on_device(x.device):
            rc = _lib.fn("ossid_seg_tail_fwd")(x.data_ptr(), B, Hs, Ws, 32, H, W, self.w1p.data_ptr(),
                                               self.b1.data_ptr(), self.scale.data_ptr(), self.shift.data_ptr(),
                                               self.w2.data_ptr(), self.b2.data_ptr(), out.data_ptr(), _lib.stream())
        _lib.check(rc, "ossid_seg_tail_fwd")
        return out


PackedConv3x3 = PackedConv   # the head's name for it


def conv3x3(x, weight, bias=None):
    """Differentiable 3x3 / stride 1 / padding 1 convolution on the hand-written kernels (forward, data gradient on the
    rotated weights, LDS-staged MFMA weight gradient): train_ops.FusedConv without prologue or activation."""
    from . import train_ops
    _lib.require_cuda(x, weight)
    return train_ops.FusedConv.apply(x, weight, bias, None, None, False, False, None, False)


def conv_module(mod, x):
    """Apply an nn.Conv2d of the nn.Module path (eval on the CPU, the MIOpen reference path of the tests). The training
    step does not come through here: Network._forward_train_hip drives csrc/conv.hip + csrc/train.hip directly."""
    return mod(x)
