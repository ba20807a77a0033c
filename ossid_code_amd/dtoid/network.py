"""DTOID network with the reference's module tree, call signatures and outputs
(/root/reference/python/ossid/models/dtoid/network.py: ImageFeatExtract :160-192, TemplateFeatExtractGlobal :195-239,
TemplateFeatExtract :242-279, CorrelationModel :282-371, ClassificationModel :96-128, RegressionModel :131-157,
Network :373-581). Attribute names are the reference's, so its state_dict keys load unchanged.

What is different is how it runs on MI355X:
  - the per-sample depthwise correlation (conv2d_dw_group) is a hand-written HIP stencil with both gradients
    (csrc/dtoid.hip) instead of a groups=B*C grouped convolution;
  - backbones are written out (backbones.py), no torchvision, no weight download;
  - forward_all_templates keeps everything on the device: anchors cached, box decode+clip and NMS in HIP, the
    template-index bookkeeping is an arange instead of an O(n_t^2) torch.cat loop, and the per-template
    segmentation maps are only gathered for the boxes that survive NMS.
"""
import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .anchors import Anchors
from .backbones import SqueezeNet11, densenet121_features

PRIOR = 0.01


class BBoxTransform(nn.Module):
    """anchors + regression deltas -> boxes (network.py:30-70); differentiable torch form used by the train path."""

    def __init__(self, mean=None, std=None):
        super().__init__()
        self.mean = torch.zeros(4) if mean is None else mean
        self.std = torch.tensor([0.1, 0.1, 0.2, 0.2]) if std is None else std

    def forward(self, boxes, deltas):
        key = (deltas.device, deltas.dtype)
        if getattr(self, "_dev_key", None) != key:      # move the constants once, not per call (graph-capturable)
            self._dev_key, self._dev_ms = key, (self.mean.to(deltas), self.std.to(deltas))
        mean, std = self._dev_ms
        w, h = boxes[..., 2] - boxes[..., 0], boxes[..., 3] - boxes[..., 1]
        cx, cy = boxes[..., 0] + 0.5 * w, boxes[..., 1] + 0.5 * h
        d = deltas * std + mean
        pcx, pcy = cx + d[..., 0] * w, cy + d[..., 1] * h
        pw, ph = torch.exp(d[..., 2]) * w, torch.exp(d[..., 3]) * h
        return torch.stack([pcx - 0.5 * pw, pcy - 0.5 * ph, pcx + 0.5 * pw, pcy + 0.5 * ph], dim=2)


class ClipBoxes(nn.Module):
    """clamp boxes to the image, in place like the reference (network.py:74-88)."""

    def forward(self, boxes, img):
        height, width = img.shape[2], img.shape[3]
        boxes[:, :, 0].clamp_(min=0)
        boxes[:, :, 1].clamp_(min=0)
        boxes[:, :, 2].clamp_(max=width)
        boxes[:, :, 3].clamp_(max=height)
        return boxes


def _head_trunk(mod, cin, feature_size):
    mod.conv1 = nn.Conv2d(cin, feature_size, kernel_size=3, padding=1)
    mod.act1 = nn.ELU()
    mod.conv2 = nn.Conv2d(feature_size, feature_size, kernel_size=3, padding=1)
    mod.act2 = nn.ELU()
    mod.conv3 = nn.Conv2d(feature_size, feature_size, kernel_size=3, padding=1)
    mod.act3 = nn.ELU()
    mod.conv4 = nn.Conv2d(feature_size, feature_size, kernel_size=3, padding=1)
    mod.act4 = nn.ELU()


def _run_trunk(mod, x):
    for i in (1, 2, 3, 4):
        x = getattr(mod, "act%d" % i)(ops.conv_module(getattr(mod, "conv%d" % i), x))
    return ops.conv_module(mod.output, x)


class ClassificationModel(nn.Module):
    def __init__(self, num_features_in, num_anchors=3, num_classes=2, prior=0.01, feature_size=256):
        super().__init__()
        self.num_anchors, self.num_classes = num_anchors, num_classes
        _head_trunk(self, num_features_in, feature_size)
        self.output = nn.Conv2d(feature_size, num_anchors * num_classes, kernel_size=3, padding=1)
        self.output_act = nn.Sigmoid()

    def forward(self, x):
        out = self.output_act(_run_trunk(self, x))                       # [B, A*C, h, w]
        B = x.shape[0]
        flat = out.permute(0, 2, 3, 1).reshape(B, -1, self.num_classes)   # cell-major, anchor, class
        return flat, out


class RegressionModel(nn.Module):
    def __init__(self, num_features_in, num_anchors=3, feature_size=256):
        super().__init__()
        _head_trunk(self, num_features_in, feature_size)
        self.output = nn.Conv2d(feature_size, num_anchors * 4, kernel_size=3, padding=1)

    def forward(self, x):
        out = _run_trunk(self, x)
        return out.permute(0, 2, 3, 1).reshape(out.shape[0], -1, 4)


class ImageFeatExtract(nn.Module):
    """DenseNet-121 trunk whose stem output is modulated by the global template feature (network.py:160-192)."""

    def __init__(self):
        super().__init__()
        dense = densenet121_features()
        dense.transition3.pool = nn.AvgPool2d(kernel_size=2, stride=1, padding=0)
        children = list(dense.children())
        self.backdense_0 = nn.Sequential(*children[:1])
        self.backdense_1 = nn.Sequential(*children[1:5])
        self.backdense_2 = nn.Sequential(*children[5:])
        self.c1 = nn.Conv2d(1024, 640, 1)
        self.n1 = nn.BatchNorm2d(640, affine=True)

    def forward(self, image, template_feat):
        x0 = self.backdense_0(image)
        x0 = x0 + ops.dw_xcorr(x0, template_feat)
        x2 = self.backdense_2(self.backdense_1(x0))
        return self.n1(F.elu(self.c1(x2)))


def _squeezenet_trunk(mod):
    """4-channel stem + the two slices of SqueezeNet-1.1's feature stack, sharing the Fire modules with
    mod.backbone exactly as the reference does (so both key families exist in the state_dict)."""
    mod.backbone = SqueezeNet11()
    feats = list(mod.backbone.features)
    stem = nn.Conv2d(4, 64, kernel_size=3, stride=2)
    with torch.no_grad():
        stem.weight[:, :3] = feats[0].weight
        stem.bias.copy_(feats[0].bias)
    mod.backbone_0 = nn.Sequential(stem)
    mod.backbone_1 = nn.Sequential(*feats[1:5])
    mod.backbone_2 = nn.Sequential(*feats[5:])
    mod.norm_1 = nn.BatchNorm2d(128, affine=True)
    mod.norm_2 = nn.BatchNorm2d(512, affine=True)


_BILINEAR = {}


def _bilinear_matrix(n_in, n_out, device):
    """[n_out, n_in] matrix of F.interpolate(mode="bilinear", align_corners=False) along one axis (the torch formula:
    src = (dst + 0.5) * in/out - 0.5 clamped at 0, neighbours floor(src) and min(floor(src)+1, in-1))."""
    key = (n_in, n_out, str(device))
    if key not in _BILINEAR:
        src = ((torch.arange(n_out, dtype=torch.float32) + 0.5) * (float(n_in) / float(n_out)) - 0.5).clamp(min=0)
        i0 = src.floor().long().clamp(max=n_in - 1)
        i1 = (i0 + 1).clamp(max=n_in - 1)
        lam = src - i0.float()
        A = torch.zeros(n_out, n_in)
        A[torch.arange(n_out), i0] += 1.0 - lam
        A[torch.arange(n_out), i1] += lam
        _BILINEAR[key] = A.to(device)
    return _BILINEAR[key]


def _bilinear_resize(x, size):
    """F.interpolate(x, size=size, mode="bilinear", align_corners=False). On the GPU as two small matrix products (the
    30x30 -> 7x7 resize of the template encoders, network.py:236/:276: torch's kernel takes 0.3 ms per call there --
    one thread per output pixel looping over batch x channels -- and its backward as long again)."""
    if not x.is_cuda:
        return F.interpolate(x, size=size, mode="bilinear", align_corners=False)
    A = _bilinear_matrix(x.shape[2], int(size), x.device)
    Bm = _bilinear_matrix(x.shape[3], int(size), x.device)
    return torch.matmul(torch.matmul(A, x), Bm.t())


def _squeezenet_features(mod, img):
    x1 = mod.backbone_1(mod.backbone_0(img))
    x2 = mod.backbone_2(x1)
    x1n, x2n = mod.norm_1(x1), mod.norm_2(x2)
    x1d = _bilinear_resize(x1n, x2.size(3))
    return torch.cat([x2n, x1d], dim=1)


class TemplateFeatExtractGlobal(nn.Module):
    """object-attention branch: template -> [B,64,3,3] depthwise kernels (network.py:195-239)"""

    def __init__(self, output_dim=1024):
        super().__init__()
        _squeezenet_trunk(self)
        self.final_conv_1 = nn.Conv2d(640, 128, 3)
        self.final_conv_2 = nn.Conv2d(128, 64, 3)
        self.final_norm_1 = nn.BatchNorm2d(128, affine=True)
        self.final_norm_2 = nn.BatchNorm2d(64, affine=True)

    def forward(self, img):
        xf = _squeezenet_features(self, img)
        xf = self.final_norm_1(F.elu(self.final_conv_1(xf)))
        return self.final_norm_2(F.elu(self.final_conv_2(xf)))


class TemplateFeatExtract(nn.Module):
    """pose-specific branch: template -> [B,640,7,7] (network.py:242-279)"""

    def __init__(self, output_dim=1024):
        super().__init__()
        _squeezenet_trunk(self)

    def forward(self, img):
        return _squeezenet_features(self, img)


class FusedTemplateEncoder:
    """Test-time execution plan of a SqueezeNet-1.1 template encoder (TemplateFeatExtract / TemplateFeatExtractGlobal,
    network.py:195-279) on this repo's kernels: the 4-channel stride-2 stem as im2col + 1x1 MFMA convolution, the
    max-pools channels-last, every Fire module as three convolutions with the ReLU in their epilogues writing into
    channel slices of one buffer (no torch.cat); the global branch's two valid 3x3 convolutions as padded convolutions
    whose interior is kept (an interior output of a pad-1 conv never touches the padding: identical values). BatchNorm
    (eval) on the two taps, the 30 -> 7 bilinear resize and the final concatenation are small torch ops."""

    def __init__(self, mod):
        P = ops.PackedConv
        self.mod = mod
        self.stem = P(ops._StemAsMatrix(mod.backbone_0[0], 48), act=2)
        self.stages = []                                    # ("pool", module) | ("fire", (squeeze, e1, e3), e1_channels)
        for part in (list(mod.backbone_1), list(mod.backbone_2)):
            plan = []
            for m in part:
                if isinstance(m, nn.MaxPool2d):
                    plan.append(("pool", m))
                elif isinstance(m, nn.ReLU):
                    continue                                # the stem's ReLU rides in its epilogue
                else:
                    plan.append(("fire", (P(m.squeeze, act=2), P(m.expand1x1, act=2), P(m.expand3x3, act=2))))
            self.stages.append(plan)
        self.final = None
        if hasattr(mod, "final_conv_1"):
            self.final = [P(_PadOne(mod.final_conv_1), mod.final_norm_1, act=1), P(_PadOne(mod.final_conv_2), mod.final_norm_2, act=1)]

    def refresh(self):
        self.stem.refresh()
        for plan in self.stages:
            for kind, item in plan:
                if kind == "fire":
                    for pk in item:
                        pk.refresh()
        for pk in self.final or []:
            pk.refresh()

    def _run(self, plan, x):
        for kind, item in plan:
            if kind == "pool":
                k = item.kernel_size if isinstance(item.kernel_size, int) else item.kernel_size[0]
                st = item.stride if isinstance(item.stride, int) else item.stride[0]
                pd = item.padding if isinstance(item.padding, int) else item.padding[0]
                x = ops.maxpool_nhwc(x, k, st, pd, item.ceil_mode)
            else:
                sq, e1, e3 = item
                B, _, H, W = x.shape
                s = sq(x)
                out = torch.empty((B, e1.cout + e3.cout, H, W), dtype=torch.float32, device=x.device,
                                  memory_format=torch.channels_last)
                e1.run(s, B, H, W, out, out_cs=e1.cout + e3.cout, out_coff=0)
                e3.run(s, B, H, W, out, out_cs=e1.cout + e3.cout, out_coff=e1.cout)
                x = out
        return x

    def __call__(self, img):
        mod = self.mod
        x = self.stem(ops.im2col_stem(img, 3, 2, 0, 48))
        x1 = self._run(self.stages[0], x)
        x2 = self._run(self.stages[1], x1)
        x1n, x2n = mod.norm_1(x1), mod.norm_2(x2)
        xf = torch.cat([x2n, _bilinear_resize(x1n, x2.size(3))], dim=1)
        if self.final is not None:
            for pk in self.final:                          # valid 3x3: pad-1 convolution, keep the interior
                xf = pk(xf)[:, :, 1:-1, 1:-1]
        return xf.contiguous()


class _PadOne:
    """An nn.Conv2d(3x3, padding 0) presented to PackedConv as the pad-1 convolution whose interior it equals."""
    padding, stride, groups = (1, 1), (1, 1), 1

    def __init__(self, conv):
        assert conv.kernel_size == (3, 3) and conv.padding == (0, 0) and conv.stride == (1, 1)
        self.conv = conv

    @property
    def weight(self):
        return self.conv.weight

    @property
    def bias(self):
        return self.conv.bias


# Side-stream slots of the training step (Network._fork). HIP maps streams onto a few hardware queues, and two streams on one
# queue serialise: with the encoders on slots of their own (six streams in all with the weight-gradient one) the step ran
# 42.3 ms in a fresh process but 48.4 ms once a test-time graph had been captured in it (other streams created first, another
# mapping); with the encoders REUSING the two branch slots -- they are never busy at the same time as the branches -- it is
# 41.9 ms either way. (GPU_MAX_HW_QUEUES=8 made it 79 ms; left alone.)
ENC_G_STREAM = int(os.environ.get("OSSID_ENC_G_STREAM", "0"))
ENC_L_STREAM = int(os.environ.get("OSSID_ENC_L_STREAM", "1"))
# The global template encoder's node created BEFORE the stem convolution's (1) or behind it (0). Autograd runs ready nodes in
# reverse order of creation: created first, the encoder's backward -- a chain of ~110 small launches on its side stream that
# needs the stem's kernel gradient and ends the step -- is enqueued AFTER the stem convolution's weight gradient instead of in
# front of it (where that launch and its host work became a tail of their own behind the chain).
ENC_G_FIRST = os.environ.get("OSSID_ENC_G_FIRST", "1") != "0"


class _CatConv:
    """Several nn.Conv2d layers that read the SAME input (same kernel, padding, stride), presented to PackedConv as one
    layer with their output channels concatenated (weights are re-read at every refresh)."""
    stride, groups = (1, 1), 1

    def __init__(self, convs):
        self.convs = list(convs)
        self.padding = self.convs[0].padding
        assert all(c.padding == self.padding and c.stride == (1, 1) and c.groups == 1 and
                   c.weight.shape[1:] == self.convs[0].weight.shape[1:] and (c.bias is None) == (self.convs[0].bias is None)
                   for c in self.convs)

    @property
    def weight(self):
        return torch.cat([c.weight.detach() for c in self.convs], 0)

    @property
    def bias(self):
        return None if self.convs[0].bias is None else torch.cat([c.bias.detach() for c in self.convs], 0)


class CorrelationModel(nn.Module):
    def __init__(self, img_size=(480, 480), input_dim=1024):
        super().__init__()
        self.img_size = img_size
        self.c1 = nn.Conv2d(input_dim, input_dim, 3, padding=0)
        self.n1 = nn.BatchNorm2d(input_dim, affine=True)
        self.c2 = nn.Conv2d(input_dim, input_dim, 3, padding=0)
        self.n2 = nn.BatchNorm2d(input_dim, affine=True)
        for name in ("dot", "dot3x3", "sub"):
            setattr(self, "corr_conv_" + name, nn.Conv2d(input_dim, 256, 3, padding=1))
            setattr(self, "norm_corr_" + name, nn.BatchNorm2d(256, affine=True))
        self.cf = nn.Conv2d(768, 512, 3, padding=1)
        self.nf = nn.BatchNorm2d(512, affine=True)
        cin = 512
        for i, cout in enumerate((256, 128, 64, 32, 16), 1):
            setattr(self, "s%d" % i, nn.Conv2d(cin, cout, 3, padding=1))
            setattr(self, "ns%d" % i, nn.BatchNorm2d(cout, affine=True))
            cin = cout
        self.seg_final = nn.Conv2d(16, 1, 3, padding=1)
        self.corr_conv_heatmap = nn.Conv2d(512, 1, 1)

    def _cab(self, conv, norm, x):
        return norm(F.elu(ops.conv_module(conv, x)))

    def forward(self, image_feat, template_feat, test=False):
        t2 = self._cab(self.c2, self.n2, self._cab(self.c1, self.n1, template_feat))       # 7x7 -> 5x5 -> 3x3
        dot3x3 = ops.dw_xcorr(image_feat, t2)
        avg = F.avg_pool2d(template_feat, 7)
        parts = [self._cab(self.corr_conv_dot, self.norm_corr_dot, image_feat * avg),
                 self._cab(self.corr_conv_sub, self.norm_corr_sub, image_feat - avg),
                 self._cab(self.corr_conv_dot3x3, self.norm_corr_dot3x3, dot3x3)]
        x2 = self._cab(self.cf, self.nf, torch.cat(parts, dim=1))
        heat_map = torch.sigmoid(self.corr_conv_heatmap(x2))
        s = x2
        for i in (1, 2, 3):
            s = F.interpolate(self._cab(getattr(self, "s%d" % i), getattr(self, "ns%d" % i), s), scale_factor=2,
                              mode="nearest")
        s = F.interpolate(self._cab(self.s4, self.ns4, s), size=self.img_size, mode="nearest")
        segmentation = self.seg_final(self._cab(self.s5, self.ns5, s))
        return x2, heat_map, segmentation


# bumped by every writer that changes parameters behind the modules' back (finetune.FusedAMSGrad.step): see
# Network.optimistic_checks
PARAM_EPOCH = [0]


class FusedHead:
    """Test-time execution plan of the correlation / detection head on the hand-written MFMA convolution
    (csrc/conv.hip): every 3x3 stride-1 conv runs channels-last with its ELU and eval-mode BatchNorm folded into the
    epilogue. Built from (and re-built whenever they change) the parameters of the nn.Modules, which stay the single
    source of truth -- training and state_dict handling never see this object."""

    def __init__(self, corr, cls, reg):
        P = ops.PackedConv3x3
        self.corr = corr
        self.tc1 = P(_PadOne(corr.c1), corr.n1, act=True)      # the two VALID 3x3 convs on the 7x7 template features:
        self.tc2 = P(_PadOne(corr.c2), corr.n2, act=True)      # pad-1 convolutions whose interior is kept
        self.dot = P(corr.corr_conv_dot, corr.norm_corr_dot, act=True)
        self.sub = P(corr.corr_conv_sub, corr.norm_corr_sub, act=True)
        self.sub_raw = P(corr.corr_conv_sub)                  # conv + bias only: the template-independent half of `sub`
        self._sub_wsum = self._dot_wcto = None
        self._fill_derived()
        self.dot3 = P(corr.corr_conv_dot3x3, corr.norm_corr_dot3x3, act=True)
        self.cf = P(corr.cf, corr.nf, act=True)
        self.seg = [P(getattr(corr, "s%d" % i), getattr(corr, "ns%d" % i), act=True, phases=i in (2, 3, 4)) for i in (1, 2, 3, 4, 5)]
        self.tail = ops.SegTail(corr.s5, corr.ns5, corr.seg_final)
        self.cls = [P(getattr(cls, "conv%d" % i), act=True) for i in (1, 2, 3, 4)] + [P(cls.output)]
        self.reg = [P(getattr(reg, "conv%d" % i), act=True) for i in (1, 2, 3, 4)] + [P(reg.output)]
        # both trunks' first convolutions read x2: one 512 -> 512 layer (3.08 rounds of workgroups instead of 2 x 2)
        self.cr1 = P(_CatConv([cls.conv1, reg.conv1]), act=True)
        self.num_classes = cls.num_classes

    def refresh(self):
        for pk in [self.tc1, self.tc2, self.dot, self.sub, self.sub_raw, self.dot3, self.cf, self.tail, self.cr1] + self.seg + self.cls + self.reg:
            pk.refresh()
        self._fill_derived()

    # from this many templates on the channel-contraction-last form of `dot` wins: G costs 0.30 ms per frame (0.74 GB) and
    # the GEMM ~0.010 ms per template, against 0.0174 ms per template for the Winograd convolution it replaces since that
    # kernel's tail split (round 3: 0.37 ms at 21 templates) -- the crossover moved from 16 to ~40 templates
    DOT_GEMM_MIN_TEMPLATES = int(os.environ.get("OSSID_DOT_GEMM_MIN", "40"))

    def _fill_derived(self):
        """Weight re-layouts the linearity tricks need, (re)written IN PLACE (a captured graph reads `_dot_wcto`):
        _dot_wcto [c][tap][o]: corr_conv_dot for ossid_dot_expand;
        _sub_wsum [640, 9*256]: for each border pattern p = 3*rowclass + colclass the weights of corr_conv_sub summed over the
        taps that fall inside the frame (class 0: first row/column -> tap 0 is outside; 2: last -> tap 2 is outside)."""
        with torch.no_grad():
            w = self.corr.corr_conv_dot.weight.detach().float()                  # [256, 640, 3, 3]
            wcto = w.permute(1, 2, 3, 0).reshape(w.shape[1], 9, w.shape[0])
            w = self.corr.corr_conv_sub.weight.detach().float()
            sel = ([1, 2], [0, 1, 2], [0, 1])
            parts = [w[:, :, sel[rc]][:, :, :, sel[cc]].sum((2, 3)) for rc in range(3) for cc in range(3)]   # 9 x [256, 640]
            wsum = torch.stack(parts, 0).permute(2, 0, 1).reshape(w.shape[1], -1)
            if self._dot_wcto is None:
                self._dot_wcto, self._sub_wsum = wcto.contiguous(), wsum.contiguous()
            else:
                self._dot_wcto.copy_(wcto)
                self._sub_wsum.copy_(wsum)

    def dot_wcto(self):
        return self._dot_wcto

    def sub_wsum(self):
        return self._sub_wsum

    @staticmethod
    def version_key(*mods):
        """(storage identity, value versions): a change of the first means the parameters were re-homed (rebuild, and
        drop captured graphs: they read the old addresses), of the second only that values changed (refresh in place).
        Runs once per frame, so the walk over the module tree (~1 ms for DenseNet-121: it was 8 % of a frame) is not
        repeated: the tensor list is kept ON the first module (it dies with it -- no global table holding parameter
        storage alive) and only pointers and version counters are read per call. The list is re-derived when a
        Parameter OBJECT was replaced (load_state_dict(assign=True), module surgery): the cheap check is the count and
        identity of the modules' direct `_parameters` / `_buffers` values."""
        owner = mods[0]
        ent = owner.__dict__.get("_ossid_version_tensors")
        sig = tuple(id(m) for m in mods)
        if ent is not None and ent[0] == sig:
            # a replaced Parameter shows up as a different object in its module's dict
            if any(d[k] is not t for d, k, t in ent[3]):
                ent = None
        else:
            ent = None
        if ent is None:
            params = [t for m in mods for t in m.parameters()]
            homes = []
            for m in mods:
                for sub in m.modules():
                    homes += [(sub._parameters, k, t) for k, t in sub._parameters.items() if t is not None]
                    homes += [(sub._buffers, k, t) for k, t in sub._buffers.items() if t is not None]
            ent = (sig, params, params + [t for m in mods for t in m.buffers()], homes)
            owner.__dict__["_ossid_version_tensors"] = ent
        return (tuple(t.data_ptr() for t in ent[1]), tuple(t._version for t in ent[2]))

    def template_side(self, template_feat):
        """Everything the correlation needs from the templates alone (network.py:333-343: the two valid 3x3 convs on the
        7x7 template features and the global average): constant across frames for an object, so the graphed path computes
        it once per (templates, weights) instead of once per frame."""
        corr = self.corr
        if template_feat.is_cuda:
            t2 = self.tc2(self.tc1(template_feat)[:, :, 1:-1, 1:-1])[:, :, 1:-1, 1:-1].contiguous()      # 7 -> 5 -> 3
        else:
            t2 = corr._cab(corr.c2, corr.n2, corr._cab(corr.c1, corr.n1, template_feat)).contiguous()
        avg = ops.spatial_mean(template_feat) if tuple(template_feat.shape[2:]) == (7, 7) else F.avg_pool2d(template_feat, 7)
        a2 = avg.reshape(avg.shape[0], avg.shape[1]).float().contiguous()
        # [n_t, 9*256]: conv_sub's response to the constant image a_t
        csub = ops.small_matmul(a2, self.sub_wsum()) if a2.is_cuda else (a2 @ self.sub_wsum()).contiguous()
        return [t2, avg, a2, csub]

    def correlation(self, image_feat, template_feat, side=None, frame=None, decoder=True):
        """frame: a dict shared by the template chunks of ONE image (the template-independent tensors -- channels-last image,
        G of the dot reassociation, S of the sub one -- are then computed once per frame, not once per chunk).
        decoder=False: seg is None, the caller runs self.decoder(x2) itself (beside the two detection trunks)."""
        corr = self.corr
        frame = {} if frame is None else frame
        t2, avg, a2, csub = self.template_side(template_feat) if side is None else side
        bcast = image_feat.shape[0] == 1 and image_feat.is_cuda and image_feat.shape[1] % 4 == 0
        dot3x3 = None if bcast else ops.dw_xcorr(image_feat, t2)
        if bcast:
            # ONE image against B templates: `image_feat * avg_t` and `image_feat - avg_t` (network.py:344-347) are per-template
            # affine views of the same feature map, so the two convolutions read the single image (batch stride 0, it stays
            # in L2) and apply avg_t as a per-(template, channel) input scale / shift while staging; the three results land
            # in channel slices of one buffer (no torch.cat). Saves two 61 MB elementwise passes and a 73 MB copy per frame.
            B = int(template_feat.shape[0])
            H, W = int(image_feat.shape[2]), int(image_feat.shape[3])
            if "xin" not in frame:
                frame["xin"] = image_feat.float().contiguous(memory_format=torch.channels_last)
            xin = frame["xin"]
            ones, zeros = self._const(a2)
            ctot = self.dot.cout + self.sub.cout + self.dot3.cout
            x = torch.empty((B, ctot, H, W), dtype=torch.float32, device=xin.device, memory_format=torch.channels_last)
            if B >= self.DOT_GEMM_MIN_TEMPLATES and self.dot.cout % 4 == 0 and 256 % (self.dot.cout // 4) == 0:
                # conv(image * avg_t) = sum_c avg_t[c] * G[c]: G once per frame, then ONE [B x C] x [C x HW*Cout] GEMM
                C = int(xin.shape[1])
                if "G" not in frame:
                    G = torch.empty((C, H * W * self.dot.cout), dtype=torch.float32, device=xin.device)
                    with torch.cuda.device(xin.device):
                        ops._lib.check(ops._lib.fn("ossid_dot_expand")(xin.data_ptr(), self.dot_wcto().data_ptr(), C,
                                                                       self.dot.cout, H, W, G.data_ptr(), ops._lib.stream()),
                                       "ossid_dot_expand")
                    frame["G"] = G
                G = frame["G"]
                with torch.cuda.device(xin.device):
                    z = a2 @ G                                                   # [B, HW*Cout] = [t][px][o]
                    ops._lib.count_mfma.add(2.0 * B * C * H * W * self.dot.cout)
                    ops._lib.check(ops._lib.fn("ossid_bias_elu_affine_slice")(
                        z.data_ptr(), B * H * W, self.dot.cout, None if self.dot.bias is None else self.dot.bias.data_ptr(),
                        self.dot.scale.data_ptr(), self.dot.shift.data_ptr(), x.data_ptr(), ctot, 0, ops._lib.stream()),
                        "ossid_bias_elu_affine_slice")
            else:
                self.dot.run(xin, B, H, W, x, out_cs=ctot, out_coff=0, in_bs=0, pre=(a2, zeros))
            if H >= 2 and W >= 2:       # conv(image - avg_t) = conv(image) - conv(avg_t): ONE convolution per frame
                if "S" not in frame:
                    frame["S"] = torch.empty((1, self.sub.cout, H, W), dtype=torch.float32, device=xin.device,
                                             memory_format=torch.channels_last)
                    self.sub_raw.run(xin, 1, H, W, frame["S"])
                S = frame["S"]
                with torch.cuda.device(xin.device):
                    rc = ops._lib.fn("ossid_bcast_sub_epilogue")(
                        S.data_ptr(), csub.data_ptr(), B, H, W, self.sub.cout, self.sub.scale.data_ptr(),
                        self.sub.shift.data_ptr(), x.data_ptr(), ctot, self.dot.cout, ops._lib.stream())
                ops._lib.check(rc, "ossid_bcast_sub_epilogue")
            else:
                self.sub.run(xin, B, H, W, x, out_cs=ctot, out_coff=self.dot.cout, in_bs=0, pre=(ones, -a2))
            d3 = ops.dw_xcorr_nhwc_bcast(xin, t2)             # born channels-last: no transposing copy
            self.dot3.run(d3, B, H, W, x, out_cs=ctot, out_coff=self.dot.cout + self.sub.cout)
        else:
            x = torch.cat([self.dot(image_feat * avg), self.sub(image_feat - avg), self.dot3(dot3x3)], dim=1)
        x2 = self.cf(x)
        heat_map = ops.conv1x1_c1(x2, corr.corr_conv_heatmap, sigmoid=True) if x2.is_cuda else \
            torch.sigmoid(corr.corr_conv_heatmap(x2))
        return x2, heat_map, (self.decoder(x2) if decoder else None)

    def decoder(self, x2):
        """Segmentation decoder (network.py:350-362) on the fused feature map."""
        corr = self.corr
        # each F.interpolate(mode="nearest") is folded into the NEXT conv's patch staging, so the up-sampled
        # tensors (up to 21 x 32 x 480 x 640 floats) are never written or re-read
        s = self.seg[0](x2)
        for i in (1, 2, 3):
            s = self.seg[i](s, size=(2 * s.shape[2], 2 * s.shape[3]))
        seg = self.tail(s, size=corr.img_size)        # up-sample + s5/ELU/ns5 + seg_final in one launch
        if seg is None:
            seg = corr.seg_final(self.seg[4](s, size=corr.img_size))
        return seg

    def _const(self, like):
        key = (tuple(like.shape), str(like.device))
        c = self.__dict__.setdefault("_consts", {})
        if key not in c:
            c[key] = (torch.ones_like(like), torch.zeros_like(like))
        return c[key]

    @staticmethod
    def _trunk(convs, x):
        for cv in convs:
            x = cv(x)
        return x

    def detection(self, x2):
        """Both trunks (network.py:113-121, :146-154) in lockstep: the first convolutions as one layer, the i-th
        convolutions of the two trunks as ONE grid each. Returns (classifications [B,A,classes], regression [B,A,4])."""
        if not x2.is_cuda:
            return self.classification(x2), self.regression(x2)
        B, _, H, W = x2.shape
        x2 = x2.float().contiguous(memory_format=torch.channels_last)
        new = lambda c: torch.empty((B, c, H, W), dtype=torch.float32, device=x2.device, memory_format=torch.channels_last)  # noqa: E731
        c1 = self.cls[0].cout
        h1 = new(self.cr1.cout)
        self.cr1.run(x2, B, H, W, h1)
        hc, hr, cs = h1, h1[:, c1:], self.cr1.cout             # channel slices of one buffer: pointer + channel stride
        for i in (1, 2, 3):
            oc, orr = new(self.cls[i].cout), new(self.reg[i].cout)
            ops.PackedConv.run_pair(self.cls[i], ((hc, B, H, W, oc), {"in_cs": cs}), self.reg[i], ((hr, B, H, W, orr), {"in_cs": cs}))
            hc, hr, cs = oc, orr, 0
        c = torch.sigmoid(self.cls[4](hc))
        r = self.reg[4](hr)
        return (c.permute(0, 2, 3, 1).reshape(B, -1, self.num_classes), r.permute(0, 2, 3, 1).reshape(B, -1, 4))

    def classification(self, x2):
        out = torch.sigmoid(self._trunk(self.cls, x2))
        return out.permute(0, 2, 3, 1).reshape(x2.shape[0], -1, self.num_classes)

    def regression(self, x2):
        out = self._trunk(self.reg, x2)
        return out.permute(0, 2, 3, 1).reshape(x2.shape[0], -1, 4)


class FusedBackbone:
    """Test-time execution plan of the DenseNet-121 part of ImageFeatExtract on csrc/conv.hip. Each dense layer is two
    launches -- 1x1 conv with norm1+ReLU folded into its input staging, 3x3 conv with norm2+ReLU folded likewise -- and
    writes its 32 channels straight into the block's resident channels-last buffer; transitions are one fused 1x1 conv
    plus the average pool; norm5 -> c1 (1x1) -> ELU -> n1 is ONE launch. The stem: csrc/stem.hip's 7x7 stride-2 kernel,
    then template modulation + norm0 + ReLU in one pass and the max-pool. Rebuilt whenever the parameters change."""

    def __init__(self, ife):
        P = ops.PackedConv
        self.ife = ife
        seq = list(ife.backdense_1) + list(ife.backdense_2)      # norm0 relu0 pool0 block1 | trans1 block2 ... norm5
        self.stem = seq[:3]
        # stem on this repo's kernels: implicit-im2col 7x7 / 2 MFMA convolution (+ normalizeImageRange; reads the parameter
        # itself: nothing to pack) -> template modulation + norm0 + ReLU in one pass -> max-pool, channels-last throughout
        self.norm0 = self.stem[0]
        self.norm0_affine = [t.clone() for t in ops._bn_affine(self.norm0)]
        self.stages = []
        for m in seq[3:]:
            if hasattr(m, "nlayers"):                            # DenseBlock
                layers = [(P(l.conv1, pre_bn=l.norm1, pre_relu=True), P(l.conv2, pre_bn=l.norm2, pre_relu=True))
                          for l in m.values()]
                self.stages.append(("block", m, layers))
                m.__dict__["_ossid_dense_table"] = ops.dense_block_table(layers, m.growth)
            elif isinstance(m, nn.BatchNorm2d):                  # norm5, folded into c1's input staging
                self.final = P(ife.c1, bn=ife.n1, act=True, pre_bn=m, pre_relu=False)
            else:                                                # Transition: norm relu conv pool
                self.stages.append(("trans", m, P(m.conv, pre_bn=m.norm, pre_relu=True)))

    def refresh(self):
        for kind, mod, packed in self.stages:
            for pk in ([p for pair in packed for p in pair] if kind == "block" else [packed]):
                pk.refresh()
        self.final.refresh()
        for dst, src in zip(self.norm0_affine, ops._bn_affine(self.norm0)):
            dst.copy_(src)

    use_fused_stem = os.environ.get("OSSID_FUSED_STEM", "1") != "0"
    # stem_tail + pool0 in one pass, transitions with the pool in front of the 1x1 convolution, both written straight into the
    # next dense block's buffer
    use_pooled_transitions = os.environ.get("OSSID_POOLED_TRANSITIONS", "1") != "0"
    # a dense block of at most this many pixels (batch x height x width) takes the one-launch-per-layer form
    DENSE_FUSED_MAX_PIXELS = int(os.environ.get("OSSID_DENSE_FUSED_MAX_PIXELS", "20000"))

    def __call__(self, image, template_feat, raw_image=False):
        """raw_image: `image` is in [0, 1] and normalizeImageRange is applied inside the stem's gather (D1)."""
        ife = self.ife
        def block_buffer(si, B, C, H, W):
            """The resident buffer of the dense block at stage si (None if that stage is no block): its producer -- the stem's
            pooling pass, a transition's convolution -- writes the block's input straight into the first C channels."""
            if si >= len(self.stages) or self.stages[si][0] != "block":
                return None
            mod = self.stages[si][1]
            return torch.empty((B, C + mod.nlayers * mod.growth, H, W), dtype=torch.float32, device=image.device,
                               memory_format=torch.channels_last)
        pending = None                      # the next block's buffer, already holding its input
        if self.use_fused_stem:
            x0 = ops.stem_conv(image, ife.backdense_0[0], normalize=raw_image)
            B, C, H0, W0 = x0.shape
            pending = block_buffer(0, B, C, (H0 - 1) // 2 + 1, (W0 - 1) // 2 + 1) if self.use_pooled_transitions else None
            x = ops.stem_tail_pool(x0, template_feat, *self.norm0_affine, out=pending) if self.use_pooled_transitions else \
                ops.maxpool_nhwc(ops.stem_tail(x0, template_feat, *self.norm0_affine), 3, 2, 1)
            if pending is not None:
                x = pending[:, :C]
        else:
            if raw_image:
                from .model import normalizeImageRange
                image = normalizeImageRange(image)
            x0 = ife.backdense_0(image)
            if template_feat.shape[0] == 1 and x0.shape[0] > 1:    # a batch of images of ONE object (batched test time)
                template_feat = template_feat.expand(x0.shape[0], -1, -1, -1)
            x = x0 + ops.dw_xcorr(x0, template_feat)
            for m in self.stem:
                x = m(x)
            x = x.contiguous(memory_format=torch.channels_last)
        for si, (kind, mod, packed) in enumerate(self.stages):
            B, C, H, W = x.shape
            if kind == "block":
                ctot = C + mod.nlayers * mod.growth
                if pending is not None:
                    buf, pending = pending, None
                else:
                    buf = torch.empty((B, ctot, H, W), dtype=torch.float32, device=x.device,
                                      memory_format=torch.channels_last)
                    buf[:, :C] = x
                table = mod.__dict__.get("_ossid_dense_table")
                if table is not None and B * H * W <= self.DENSE_FUSED_MAX_PIXELS and C in (64, 128, 256, 512):
                    # few pixels (a single frame): one launch per layer, the later layers' bottleneck sums kept up to date
                    # incrementally (csrc/dense.hip) instead of two launches with the 1x1 over the whole prefix
                    ops.dense_block_fused(buf, B, H, W, C, packed, table)
                else:
                    tmp = torch.empty((B, 128, H, W), dtype=torch.float32, device=x.device,
                                      memory_format=torch.channels_last)
                    c = C
                    for conv1, conv2 in packed:
                        conv1.run(buf, B, H, W, tmp, in_cs=ctot)
                        conv2.run(tmp, B, H, W, buf, out_cs=ctot, out_coff=c)
                        c += mod.growth
                x = buf
            else:
                st = mod.pool.stride if isinstance(mod.pool.stride, int) else mod.pool.stride[0]
                if self.use_pooled_transitions and mod.conv.bias is None and st in (1, 2) and H >= 2 and W >= 2:
                    # norm -> relu -> conv 1x1 -> avg-pool with the pool moved in front of the (bias-free, linear, pixelwise)
                    # convolution: one pass that normalises, rectifies and averages, then the convolution on the pooled
                    # pixels (a quarter of them at stride 2), written straight into the next block's buffer
                    pooled = ops.bn_relu_avgpool2(x, C, packed.pre_scale, packed.pre_shift, st)
                    Ho, Wo = int(pooled.shape[2]), int(pooled.shape[3])
                    pending = block_buffer(si + 1, B, packed.cout, Ho, Wo)
                    out = pending if pending is not None else torch.empty((B, packed.cout, Ho, Wo), dtype=torch.float32,
                                                                          device=x.device, memory_format=torch.channels_last)
                    packed.run(pooled, B, Ho, Wo, out, out_cs=int(out.shape[1]), skip_pre=True)
                    x = out[:, :packed.cout]
                else:
                    x = ops.avgpool2_nhwc(packed(x), st)
        return self.final(x)


class Network(nn.Module):
    use_fused_backbone = True  # test-time DenseNet blocks on csrc/conv.hip; False = the nn.Module path (MIOpen)
    use_fused_head = True     # test-time head on csrc/conv.hip; False = the nn.Module path (MIOpen convolutions)
    # Replay the dense part of forward_all_templates from a captured hipGraph. OSSID_NO_GRAPH=1 forces eager; eager is
    # also chosen automatically when the process runs under a rocprofiler-sdk tool (rocprofv3 preloads
    # librocprofiler-sdk-tool.so and sets ROCP_TOOL_LIBRARIES): with its launch interception active a kernel launch
    # into a CAPTURING stream segfaulted on the host in round 1 ("capture under the profiler"), and
    # kernel-trace rows of graph-replayed kernels would carry no per-launch correlation anyway.
    under_profiler = ("rocprofiler" in os.environ.get("LD_PRELOAD", "") or
                      bool(os.environ.get("ROCP_TOOL_LIBRARIES")) or bool(os.environ.get("ROCPROFILER_LIBRARY_CTOR")))
    use_graph = os.environ.get("OSSID_NO_GRAPH", "0") != "1" and (
        not under_profiler or os.environ.get("OSSID_GRAPH_UNDER_PROFILER", "0") == "1")

    def __init__(self, img_size=(480, 480), heatmap_size=(29, 29), template_size=124):
        super().__init__()
        self.img_size, self.heatmap_size = img_size, heatmap_size
        self.template_feature_extractor_global = TemplateFeatExtractGlobal()
        self.image_feature_extractor = ImageFeatExtract()
        self.template_feature_extractor = TemplateFeatExtract()
        self.correlation_model = CorrelationModel(self.img_size, 640)
        self.anchors = Anchors(pyramid_levels=[4], ratios=[0.5, 1, 2], sizes=[30], scales=[1, 2, 3, 4, 5, 6, 7, 8])
        self.classification = ClassificationModel(512, num_anchors=24)
        self.regression = RegressionModel(512, num_anchors=24)
        bias = -math.log((1.0 - PRIOR) / PRIOR)
        for conv, b in ((self.classification.output, bias), (self.regression.output, 0.0),
                        (self.correlation_model.corr_conv_heatmap, bias), (self.correlation_model.seg_final, bias)):
            conv.weight.data.fill_(0)
            conv.bias.data.fill_(b)
        self.regressBoxes = BBoxTransform()
        self.clipBoxes = ClipBoxes()

    def _fused_backbone(self):
        key = FusedHead.version_key(self.image_feature_extractor)
        cached = self.__dict__.get("_fused_bb_cache")
        if cached is None or cached[0][0] != key[0]:
            self.__dict__.pop("_graph_cache", None)
            cached = (key, FusedBackbone(self.image_feature_extractor))
        elif cached[0] != key:           # parameters changed (finetune): re-pack in place, graphs stay valid
            cached[1].refresh()
            cached = (key, cached[1])
        self.__dict__["_fused_bb_cache"] = cached
        return cached[1]

    def _fused_head(self):
        mods = (self.correlation_model, self.classification, self.regression)
        key = FusedHead.version_key(*mods)
        cached = self.__dict__.get("_fused_cache")
        if cached is None or cached[0][0] != key[0]:
            self.__dict__.pop("_graph_cache", None)
            cached = (key, FusedHead(*mods))
            self.__dict__["_plan_epoch"] = self.__dict__.get("_plan_epoch", 0) + 1
        elif cached[0] != key:
            cached[1].refresh()
            cached = (key, cached[1])
            self.__dict__["_plan_epoch"] = self.__dict__.get("_plan_epoch", 0) + 1
        self.__dict__["_fused_cache"] = cached
        return cached[1]

    def load(self, path):
        ckpt = torch.load(path, map_location="cpu")
        self.load_state_dict(ckpt["state_dict"])

    def freeze_bn(self):
        for layer in self.modules():
            if isinstance(layer, nn.BatchNorm2d):
                layer.eval()

    use_fused_templates = os.environ.get("OSSID_FUSED_TEMPLATES", "1") != "0"

    def _fused_template_encoder(self, mod, slot):
        key = FusedHead.version_key(mod)
        cached = self.__dict__.get(slot)
        if cached is None or cached[0][0] != key[0]:
            cached = (key, FusedTemplateEncoder(mod))
        elif cached[0] != key:
            cached[1].refresh()
            cached = (key, cached[1])
        self.__dict__[slot] = cached
        return cached[1]

    def _encode_templates(self, mod, slot, img):
        if self.use_fused_templates and img.is_cuda and not self.training and not torch.is_grad_enabled():
            return self._fused_template_encoder(mod, slot)(img)
        return mod(img)

    def compute_template_local(self, img):
        return self._encode_templates(self.template_feature_extractor, "_fused_tfe_local", img)

    def compute_template_global(self, img):
        return self._encode_templates(self.template_feature_extractor_global, "_fused_tfe_global", img)

    # finetune forward/backward on the hand-written kernels (train_ops.py); False = the nn.Module path (MIOpen)
    use_hip_training = os.environ.get("OSSID_TRAIN_IMPL", "hip") != "miopen"
    # The two SqueezeNet template encoders train on this repo's kernels as one autograd node each, replaying recorded launch
    # sequences (dtoid/train_encoders.py; round 3 -- round 2's form, ~25 autograd nodes per encoder, was 0.3-1.1 ms slower
    # than torch / MIOpen because of its host cost). OSSID_TRAIN_TEMPLATES=0: the nn.Module path (MIOpen).
    use_hip_template_training = os.environ.get("OSSID_TRAIN_TEMPLATES", "1") != "0"
    # The 7x7 stem + template modulation + norm0 + pool0 on this repo's kernels (csrc/stem.hip: implicit-im2col MFMA
    # convolution and weight gradient, fused statistics / pooling passes; train_ops.StemConv / StemTail). False = the
    # nn.Module path (MIOpen) for these layers: what the tests compare against.
    use_hip_stem_training = True

    # Independent branches of the training step on side HIP streams (eager execution only -- under a graph capture the
    # branches run in line): the local template encoder beside the image backbone, the three correlation convolutions
    # side by side, the two detection trunks beside the segmentation decoder. At batch 8 each of these launches fills
    # 60-75 % of the chip's workgroup slots; two or three of them in flight fill the rest. autograd runs every node's
    # backward on the stream of its forward, so the backward pass is concurrent in the same way. Tensors that cross
    # streams are record_stream()ed for the caching allocator.
    use_train_streams = os.environ.get("OSSID_TRAIN_STREAMS", "1") != "0"

    def _branches_on(self, device):
        return (self.use_train_streams and device.type == "cuda" and not torch.cuda.is_current_stream_capturing())

    def _fork(self, k, inputs, fn):
        """Run fn() on side stream k behind everything queued on the current stream; returns (outputs, stream)."""
        dev = inputs[0].device
        from . import train_ops
        side = train_ops.side_streams(dev)["b%d" % (k & 1)]     # two branch slots (train_ops.side_streams: probed once)
        side.wait_stream(torch.cuda.current_stream(dev))
        ev = self.__dict__.get("_pack_event")
        if ev is not None:                                       # this step's packed weights (packed beside the stem, below)
            side.wait_event(ev)
        for t in inputs:
            t.record_stream(side)
        with torch.cuda.stream(side):
            out = fn()
        return out, side

    @staticmethod
    def _join(side, outputs):
        main = torch.cuda.current_stream(outputs[0].device)
        main.wait_stream(side)
        for t in outputs:
            t.record_stream(main)

    def _forward_train_hip(self, image, g, local, lazy_g=None, lazy_local=None):
        """Training-mode forward of everything behind the two template encoders on csrc/conv.hip + csrc/train.hip,
        channels-last end to end (same arithmetic as the module path below; BatchNorm batch statistics folded into the
        next convolution's input staging). Returns (classifications, regression, anchors, heat_map, segmentation)."""
        from . import train_ops as T
        from .backbones import DenseBlock, Transition
        ife = self.image_feature_extractor
        # every conv weight -> MFMA layouts, one launch (0.64 ms: 136 MB read, 272 MB written). Nothing needs it before the
        # first dense block and the template encoders, so with side streams on it runs on the weight-gradient stream (idle
        # at the start of a step: the previous step's weight gradients were joined before its optimizer ran) beside the stem
        # In two launches: the backbone's and the encoders' weights (7 of the 34 M) first, with an event of their own -- the stem
        # on this repo's kernels takes 0.5 ms, not the 1.5 ms that used to cover the whole packing -- then the head's.
        # (round 4: three launches -- the GLOBAL template encoder's weights first, in a launch of a few microseconds: that encoder
        # is the head of the step's critical path (stem modulation needs its output) and used to wait ~1 ms for the whole
        # first part; tools/step_gaps.py)
        pack_event = pack_event_all = pack_event_g = None
        if self._branches_on(image.device):
            main = torch.cuda.current_stream(image.device)
            ps = T.side_streams(image.device)["wgrad"]
            ps.wait_stream(main)                                 # behind the optimizer step that wrote the weights
            with torch.cuda.stream(ps):
                evs = self._train_pack_plan().run(want_events=True)
            pack_event_all = evs[-1]
            pack_event = evs[-2] if len(evs) >= 2 else pack_event_all
            pack_event_g = evs[0] if len(evs) >= 3 else pack_event
        else:
            self._train_pack_plan().run()
        self.__dict__["_pack_event"] = pack_event_g              # what a fork waits for: the global encoder only its own part
        seq = list(ife.backdense_1) + list(ife.backdense_2)      # norm0 relu0 pool0 block1 | trans1 block2 ... norm5
        if self.use_hip_stem_training:
            # stem on this repo's kernels, channels-last from the first one
            if lazy_g is not None and ENC_G_FIRST:
                g, s_g = lazy_g()
            x0 = T.stem_conv(image, ife.backdense_0[0])           # implicit-im2col 7x7 / 2 kernel, exact f32 (csrc/stem.hip)
            if lazy_g is not None:
                if not ENC_G_FIRST:
                    g, s_g = lazy_g()
                self._join(s_g, [g])
            self.__dict__["_pack_event"] = pack_event            # (later forks -- the local encoder -- wait for the first big part)
            x = T.stem_tail(x0, g, seq[0])                        # modulation + norm0 + ReLU + pool0: three passes
        else:
            x0 = ife.backdense_0(image)
            if lazy_g is not None:
                g, s_g = lazy_g()
                self._join(s_g, [g])
            self.__dict__["_pack_event"] = pack_event
            x0 = x0 + ops.dw_xcorr(x0, g)
            x = x0
            for m in seq[:3]:                                    # stem: 64 channels at 240x320, on torch
                x = m(x)
            x = T.nhwc(x)
        if pack_event is not None:
            torch.cuda.current_stream(image.device).wait_event(pack_event)
        norm5 = None
        join_local, n_blocks = None, 0
        for m in seq[3:]:
            if isinstance(m, DenseBlock):
                x = T.dense_block_train(x, m)
                n_blocks += 1
                if n_blocks == 2 and lazy_local is not None:          # (see forward(): the middle of the backbone)
                    local, join_local = lazy_local()
            elif isinstance(m, Transition):
                stride = m.pool.stride if isinstance(m.pool.stride, int) else m.pool.stride[0]
                x = T.AvgPool2.apply(T.bn_relu_conv(x, m.norm, m.conv, deciding=True), stride)
            else:
                norm5 = m
        B = x.shape[0]
        n_px = B * x.shape[2] * x.shape[3]
        u, sums = T.bn_relu_conv(x, norm5, ife.c1, relu=False, act_elu=True, want_stats=True)
        s, t = T.bn_fold(sums, n_px, ife.n1)
        feat = T.AffineAct.apply(u, s, t, False)                             # n1(elu(c1(norm5(.)))), [B,640,29,39]: one pass
        if join_local is not None:
            self._join(join_local, [local])
        if pack_event_all is not None:                           # the head's weights: packed long since
            torch.cuda.current_stream(image.device).wait_event(pack_event_all)
            self.__dict__["_pack_event"] = pack_event_all
        out = self._head_train_hip(feat, local)
        # the folded BatchNorms update running_mean / running_var in their kernel; the counters in one launch
        ts = self.__dict__.get("_folded_bn_counters")
        if ts is None:
            folded = seq[3:] + [ife.n1] + ([seq[0]] if self.use_hip_stem_training else [])
            ts = [b.num_batches_tracked for m in folded for b in m.modules() if isinstance(b, nn.BatchNorm2d)]
            self.__dict__["_folded_bn_counters"] = ts
        torch._foreach_add_(ts, 1)
        return out

    def _train_pack_convs(self):
        """(convolutions in packing order, end of the first launch's part, end of the second's): the module structure is fixed,
        so _train_pack_plan walks it once, not every step (the step's first launch waits for this host work)."""
        from .backbones import DenseBlock, Transition
        ife, corr = self.image_feature_extractor, self.correlation_model
        convs = []
        if self.use_hip_template_training:
            from .train_encoders import encoder_convs
            convs += encoder_convs(self.template_feature_extractor_global)
        n_global = len(convs)                                    # first launch: the global template encoder alone (tiny)
        for m in list(ife.backdense_1) + list(ife.backdense_2):
            if isinstance(m, DenseBlock):
                for layer in m.values():
                    convs += [layer.conv1, layer.conv2]
            elif isinstance(m, Transition):
                convs.append(m.conv)
        convs.append(ife.c1)
        if self.use_hip_template_training:
            convs += encoder_convs(self.template_feature_extractor)
        n_first = len(convs)                                     # second launch: everything else in front of the head
        convs += [corr.c1, corr.c2, corr.corr_conv_dot, corr.corr_conv_sub, corr.corr_conv_dot3x3, corr.cf] + \
            [getattr(corr, "s%d" % i) for i in (1, 2, 3, 4, 5)]
        for mod in (self.classification, self.regression):
            convs += [getattr(mod, "conv%d" % i) for i in (1, 2, 3, 4)] + [mod.output]
        return convs, n_global, n_first

    def _train_pack_plan(self):
        """The PackPlan over every convolution the hip training path runs (DenseNet blocks and transitions, c1, the head's
        3x3 convolutions), rebuilt when a weight tensor was re-homed."""
        from . import train_ops as T
        from .backbones import DenseBlock, Transition
        ife, corr = self.image_feature_extractor, self.correlation_model
        cached = self.__dict__.get("_pack_convs")
        if cached is not None and cached[0] == (self.use_hip_template_training,):
            convs, n_global, n_first = cached[1]
        else:
            convs, n_global, n_first = self._train_pack_convs()
            self.__dict__["_pack_convs"] = ((self.use_hip_template_training,), (convs, n_global, n_first))
        plan = self.__dict__.get("_pack_plan")
        if plan is not None and plan.misses:
            # layouts a layer had to pack by itself last step (a launch on the critical chain): the decoder layers' data gradient
            # runs on the UP-SAMPLED grid, where the Winograd form applies although their forward (fused up-sampling) is direct.
            # The plan learns them: rebuilt once with those layouts in, the direct data-gradient layout they replace out
            self.__dict__.setdefault("_pack_learned", set()).update(plan.misses)
            plan = None
        if plan is None or not plan.valid_for(convs):
            # which layouts the step asks for at finetune batch sizes (train_ops.wino_fits): the head's plain 3x3 layers run
            # forward and data gradient on the Winograd kernel, the dense blocks' 3x3 only the data gradient; everything
            # else (1x1, the decoder layers behind an up-sampling, layers with fewer than 64 output channels) the direct one
            kinds = {}
            plain = [corr.corr_conv_dot, corr.corr_conv_sub, corr.corr_conv_dot3x3, corr.cf, corr.s1] + \
                [getattr(mod, "conv%d" % i) for mod in (self.classification, self.regression) for i in (1, 2, 3, 4)] + \
                [self.regression.output]
            for cv in plain:
                kinds[cv] = ("wino_fwd", "wino_dgrad") if T.USE_WINO else ("fwd", "dgrad")
            for m in list(ife.backdense_1) + list(ife.backdense_2):
                if isinstance(m, DenseBlock):
                    for layer in m.values():
                        kinds[layer.conv1] = (T.FWD_DECIDING, "dgrad")
                        kinds[layer.conv2] = (T.FWD_DECIDING, "wino_dgrad") if (T.USE_WINO and not T.dense_bwd3_fused()) else \
                            (T.FWD_DECIDING, "dgrad")
                elif isinstance(m, Transition):
                    kinds[m.conv] = (T.FWD_DECIDING, "dgrad")
            if self.use_hip_template_training:
                from .train_encoders import encoder_convs
                for enc in (self.template_feature_extractor_global, self.template_feature_extractor):
                    for cv in encoder_convs(enc):
                        final = any(cv is getattr(enc, n, None) for n in ("final_conv_1", "final_conv_2"))
                        kinds[cv] = ("fwd_exact", "dgrad_exact") if final else (T.FWD_ENCODER, "dgrad")
            learned = self.__dict__.get("_pack_learned")
            if learned:
                by_ptr = {cv.weight.data_ptr(): cv for cv in convs}
                for ptr, kind in learned:
                    cv = by_ptr.get(ptr)
                    if cv is None:
                        continue
                    have = tuple(kinds.get(cv, ("fwd", "dgrad")))
                    if kind == "wino_dgrad":
                        have = tuple(k for k in have if k != "dgrad" or (ptr, "dgrad") in learned)
                    if kind not in have:
                        kinds[cv] = have + (kind,)
            plan = self.__dict__["_pack_plan"] = T.PackPlan(convs, kinds, split_at=([n_global] if n_global else []) + [n_first])
        return plan

    def _head_train_hip(self, feat, local):
        """CorrelationModel + ClassificationModel + RegressionModel in training mode (network.py:328-363, :113-157) on the
        hand-written kernels: feat [B,640,h,w] image features, local [B,640,7,7] template features."""
        from . import train_ops as T
        corr = self.correlation_model
        B = feat.shape[0]

        def cab(inp, conv, bn, pre=None, size=None):
            """conv -> ELU, and the folded BatchNorm that follows it as (scale, shift) for whoever reads u next"""
            u, sums = T.fused_conv(inp, conv, pre=pre, act_elu=True, size=size, want_stats=True)
            sc, sh = T.bn_fold(sums, u.shape[0] * u.shape[2] * u.shape[3], bn)
            return u, sc, sh

        # 7x7 -> 5x5 -> 3x3: the two VALID 3x3 convolutions on the template features as padded convolutions on this repo's
        # kernel whose interior is kept (an interior output never touches the padding), ELU in the epilogue, training
        # BatchNorm by the generic passes (round 3; MIOpen's igemm kernels + layout transposes for them were ~0.8 ms of GPU
        # time per step in front of the head)
        t2 = local
        for conv, bn in ((corr.c1, corr.n1), (corr.c2, corr.n2)):
            t2 = T.bn_act_train(T.fused_conv(t2, conv, act_elu=True)[:, :, 1:-1, 1:-1], bn)
        dot3x3 = ops.dw_xcorr(feat, t2)
        avg = ops.spatial_mean(local)                        # F.avg_pool2d(local, 7) on the 7x7 template features
        par = self._branches_on(feat.device)
        if par:
            (p1, s_a) = self._fork(0, [feat, avg], lambda: cab(feat - avg, corr.corr_conv_sub, corr.norm_corr_sub))
            (p2, s_b) = self._fork(1, [dot3x3], lambda: cab(dot3x3, corr.corr_conv_dot3x3, corr.norm_corr_dot3x3))
            p0 = cab(feat * avg, corr.corr_conv_dot, corr.norm_corr_dot)
            self._join(s_a, list(p1))
            self._join(s_b, list(p2))
            parts = [p0, p1, p2]
        else:
            parts = [cab(feat * avg, corr.corr_conv_dot, corr.norm_corr_dot),
                     cab(feat - avg, corr.corr_conv_sub, corr.norm_corr_sub),
                     cab(dot3x3, corr.corr_conv_dot3x3, corr.norm_corr_dot3x3)]
        ucat = torch.cat([p[0] for p in parts], 1)
        pre = (torch.cat([p[1] for p in parts]), torch.cat([p[2] for p in parts]))
        u2, s2, t2_ = cab(ucat, corr.cf, corr.nf, pre=pre)

        def trunk(mod):
            h = T.fused_conv(u2, mod.conv1, pre=(s2, t2_), act_elu=True)
            for i in (2, 3, 4):
                h = T.fused_conv(h, getattr(mod, "conv%d" % i), act_elu=True)
            return T.fused_conv(h, mod.output)
        if par:      # the two detection trunks beside the heat map + segmentation decoder
            (cls_raw, s_a) = self._fork(0, [u2, s2, t2_], lambda: trunk(self.classification))
            (reg_raw, s_b) = self._fork(1, [u2, s2, t2_], lambda: trunk(self.regression))
        x2 = T.AffineAct.apply(u2, s2, t2_, False)                          # materialised once, for the 1-channel heat conv
        heat_map = torch.sigmoid(T.conv1x1_c1(x2, corr.corr_conv_heatmap))
        u, sc, sh = cab(u2, corr.s1, corr.ns1, pre=(s2, t2_))
        for i in (2, 3, 4):
            u, sc, sh = cab(u, getattr(corr, "s%d" % i), getattr(corr, "ns%d" % i), pre=(sc, sh),
                            size=(2 * u.shape[2], 2 * u.shape[3]))
        u, sc, sh = cab(u, corr.s5, corr.ns5, pre=(sc, sh), size=corr.img_size)
        segmentation = T.conv3x3_c1(T.AffineAct.apply(u, sc, sh, False), corr.seg_final)      # 16 -> 1: vector-ALU kernels
        if par:
            self._join(s_a, [cls_raw])
            self._join(s_b, [reg_raw])
        else:
            cls_raw, reg_raw = trunk(self.classification), trunk(self.regression)
        cls = torch.sigmoid(cls_raw)
        classifications = cls.permute(0, 2, 3, 1).reshape(B, -1, self.classification.num_classes)
        regression = reg_raw.permute(0, 2, 3, 1).reshape(B, -1, 4)
        anchors = self.anchors([[u2.size(2), u2.size(3)]], device=u2.device)
        ts = self.__dict__.get("_folded_bn_counters_head")
        if ts is None:
            mods = [corr.norm_corr_dot, corr.norm_corr_sub, corr.norm_corr_dot3x3, corr.nf, corr.n1, corr.n2] + \
                [getattr(corr, "ns%d" % i) for i in (1, 2, 3, 4, 5)]
            ts = [m.num_batches_tracked for m in mods]
            self.__dict__["_folded_bn_counters_head"] = ts
        torch._foreach_add_(ts, 1)
        return classifications, regression, anchors, heat_map, segmentation

    def forward(self, image, template, template_mask, global_template, global_template_mask):
        """(B,3,H,W), (B,3,h,w), (B,1,h,w), (B,3,h,w), (B,1,h,w) ->
        classifications [B,A,2], regression [B,A,4], anchors [1,A,4], heat_map [B,1,hh,hw], segmentation [B,1,H,W]"""
        hip_train = image.is_cuda and self.training and torch.is_grad_enabled() and self.use_hip_training
        if hip_train and self.use_hip_template_training:
            # each encoder = one autograd node replaying recorded launch sequences (dtoid/train_encoders.py)
            from .train_encoders import template_encoder_train
            enc_g = lambda: template_encoder_train(self.template_feature_extractor_global,                  # noqa: E731
                                                   torch.cat([global_template, global_template_mask], dim=1))
            enc_l = lambda: template_encoder_train(self.template_feature_extractor,                         # noqa: E731
                                                   torch.cat([template, template_mask], dim=1))
        else:
            enc_g = lambda: self.template_feature_extractor_global(                                         # noqa: E731
                torch.cat([global_template, global_template_mask], dim=1))
            enc_l = lambda: self.template_feature_extractor(torch.cat([template, template_mask], dim=1))    # noqa: E731
        if hip_train and self._branches_on(image.device):
            # both encoders run on side streams, and WHEN the host enqueues them matters as much as where they run
            # (autograd replays backward in reverse order of creation): the global one right behind the stem convolution
            # that does not need it; the local one in the middle of the backbone, so that its backward is enqueued in the
            # middle of the backbone's backward, while the host is ahead of the device, instead of as a tail behind
            # everything else
            lazy_g = lambda: self._fork(ENC_G_STREAM, [global_template, global_template_mask], enc_g)       # noqa: E731
            lazy_l = lambda: self._fork(ENC_L_STREAM, [template, template_mask], enc_l)                     # noqa: E731
            return self._forward_train_hip(image, None, None, lazy_g=lazy_g, lazy_local=lazy_l)
        g, local = enc_g(), enc_l()
        if hip_train:
            return self._forward_train_hip(image, g, local)
        if image.is_cuda and not self.training and not torch.is_grad_enabled() and self.use_fused_head:
            # inference on (image, template) PAIRS (BASELINE configs[2] (ii)): backbone and head on csrc/conv.hip
            features = self._features(image, g)
            fused = self._fused_head()
            xcors, heat_map, segmentation = fused.correlation(features, local)
            anchors = self.anchors([[xcors.size(2), xcors.size(3)]], device=xcors.device)
            return fused.detection(xcors) + (anchors, heat_map, segmentation)
        features = self.image_feature_extractor(image, g)
        xcors, heat_map, segmentation = self.correlation_model(features, local)
        anchors = self.anchors([[xcors.size(2), xcors.size(3)]], device=xcors.device)
        classifications, _ = self.classification(xcors)
        return classifications, self.regression(xcors), anchors, heat_map, segmentation

    def _features(self, image, template_global, raw_image=False):
        """D4: image feature map [B,640,h,w] (test time: stem + DenseNet blocks on this repo's kernels). raw_image: the
        image is in [0, 1] and still needs normalizeImageRange (fused into the stem's gather on the GPU path)."""
        if self.use_fused_backbone and image.is_cuda and not self.training:
            return self._fused_backbone()(image, template_global, raw_image=raw_image)
        if raw_image:
            from .model import normalizeImageRange
            image = normalizeImageRange(image)
        if template_global.shape[0] == 1 and image.shape[0] > 1:
            template_global = template_global.expand(image.shape[0], -1, -1, -1)
        return self.image_feature_extractor(image, template_global)

    def _dense_head(self, features, template_features, sides=None):
        """Head of ONE image (features [1,640,h,w]) per template chunk -> dense (cls [n_t,A,2], reg [n_t,A,4],
        seg [n_t,1,H,W], heat [n_t,1,hh,hw], feature-map shape)."""
        fused = self._fused_head() if (self.use_fused_head and features.is_cuda and not self.training) else None
        cls_out, reg_out, seg_out, heat_out = [], [], [], []
        frame = {}
        for ci, chunk in enumerate(template_features):
            if fused is not None:
                xc, heat, seg = fused.correlation(features, chunk, None if sides is None else sides[ci], frame)
                c, r = fused.detection(xc)
                cls_out.append(c)
                reg_out.append(r)
            else:
                xc, heat, seg = self.correlation_model(features.expand(chunk.size(0), -1, -1, -1), chunk, True)
                cls_out.append(self.classification(xc)[0])
                reg_out.append(self.regression(xc))
            seg_out.append(seg)
            heat_out.append(heat)
        cat = lambda parts: parts[0] if len(parts) == 1 else torch.cat(parts, 0)   # noqa: E731  (one chunk: no copy)
        return (cat(cls_out), cat(reg_out), cat(seg_out), cat(heat_out), (xc.size(2), xc.size(3)))

    def _dense_all_templates(self, image, template_features, template_global, sides=None, raw_image=False):
        """Backbone once + head per template chunk. No host syncs, no data-dependent shapes: capturable in a hipGraph."""
        return self._dense_head(self._features(image, template_global, raw_image), template_features, sides)

    def _graphed_dense(self, *args, **kwargs):
        """_graphed_dense_impl under torch.no_grad(): test-time inference builds no autograd graph. (Called with gradients
        enabled, the layers still on torch modules would record a graph that the cached static outputs keep alive for
        good -- and with it AccumulateGrad nodes carrying the CAPTURE stream into every later backward pass: the source of
        torch's "AccumulateGrad node's stream does not match" warning in round 2's GPU test log)"""
        with torch.no_grad():
            return self._graphed_dense_impl(*args, **kwargs)

    # The parameter checks of a frame (FusedHead.version_key over ~1 000 tensors: 0.15-0.2 ms of host time) AFTER its launch
    # instead of in front of it: the frame is launched on the plans of the previous call, the checks run while the GPU works,
    # and only if they find a change are the plans refreshed and the frame launched again (its first results are never
    # read). Known writers (FusedAMSGrad.step) bump PARAM_EPOCH, and the first call behind a bump checks first.
    optimistic_checks = os.environ.get("OSSID_OPTIMISTIC_CHECKS", "1") != "0"

    def _graphed_dense_impl(self, image, template_features, template_global, head_only=False, raw_image=False, post_hw=None,
                            _checked=False):
        """The dense part replayed from a captured hipGraph (the B=1 backbone alone is ~500 launches and otherwise
        host-bound). One graph per (input shape, chunk sizes, packed-head identity); inputs are copied into the
        graph's static buffers, outputs are read from them. head_only: `image` is already the feature map [1,640,h,w]
        (the batched test-time path runs the backbone once for the whole batch and replays this graph per image)."""
        want_bb = self.use_fused_backbone and not head_only
        fc, bc = self.__dict__.get("_fused_cache"), self.__dict__.get("_fused_bb_cache")
        optimistic = (self.optimistic_checks and not _checked and self.use_fused_head and fc is not None and
                      (not want_bb or bc is not None) and self.__dict__.get("_checked_epoch") == PARAM_EPOCH[0])
        if optimistic:
            fused, fused_bb = fc[1], (bc[1] if want_bb else None)
        else:
            fused = self._fused_head() if self.use_fused_head else None
            fused_bb = self._fused_backbone() if want_bb else None
            self.__dict__["_checked_epoch"] = PARAM_EPOCH[0]
        # post_hw = (H, W) of the frame: the device-only part of the post-processing (candidates, NMS: post_dense) is captured
        # behind the dense part; the ops.DetectPost it fills is left in self._graph_post for post_emit
        if post_hw is not None and not (self.use_fused_post and self.use_fused_head):
            post_hw = None
        key = (tuple(image.shape), tuple(int(c.shape[0]) for c in template_features), id(fused), id(fused_bb),
               str(image.device), bool(head_only), bool(raw_image), None if post_hw is None else tuple(post_hw))
        cache = self.__dict__.setdefault("_graph_cache", {})
        entry = cache.get(key)
        if entry is None and optimistic:          # a capture is due: with checked plans
            return self._graphed_dense_impl(image, template_features, template_global, head_only, raw_image, post_hw, True)
        if entry is None:
            if len(cache) >= 4:
                cache.clear()
            s_img, s_tf = image.clone(), [c.clone() for c in template_features]
            s_g = None if head_only else template_global.clone()
            side = torch.cuda.Stream(device=image.device)
            side.wait_stream(torch.cuda.current_stream())
            # the template-only part of the correlation lives OUTSIDE the graph, in static buffers refreshed only when the
            # templates or the weights change (below)
            s_sides = [fused.template_side(c) for c in s_tf] if fused is not None else None
            dense = (lambda: self._dense_head(s_img, s_tf, s_sides)) if head_only else \
                (lambda: self._dense_all_templates(s_img, s_tf, s_g, s_sides, raw_image))
            post_box = [None]

            def run():
                outs = dense()
                if post_hw is not None and outs[1].shape[0] * outs[1].shape[1] >= 1:
                    post_box[0] = self.post_dense(outs[0], outs[1], outs[4], post_hw, post_box[0])
                return outs
            with torch.cuda.stream(side):           # warm-up off the capture: MIOpen picks its kernels here
                for _ in range(2):
                    run()
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                outs = run()
            entry = [graph, s_img, s_tf, s_g, outs, (fused, fused_bb), s_sides, None, post_box[0]]
            cache[key] = entry
        graph, s_img, s_tf, s_g, outs, _, s_sides, last, post = entry
        self.__dict__["_graph_post"] = post
        s_img.copy_(image)
        if s_g is not None:
            s_g.copy_(template_global)
        # identity of the template tensors (held by reference below, so their addresses cannot be recycled for other data
        # while an entry is cached) + their version counters + the weight epoch
        srcs = list(template_features)
        src_key = (tuple((id(c), c._version) for c in srcs), self.__dict__.get("_plan_epoch", 0))
        if last is None or last[0] != src_key:
            for dst, src in zip(s_tf, template_features):
                dst.copy_(src)
            if s_sides is not None:
                # per object (= per set of template tensors) the template-only tensors are computed once and kept: a loop
                # over the objects of a scene only pays the small copies into the graph's static buffers
                sc = self.__dict__.setdefault("_side_cache", {})
                hit = sc.get(src_key)
                if hit is None:
                    if len(sc) >= 64:
                        sc.clear()
                    hit = sc[src_key] = (srcs, [[t.clone() for t in fused.template_side(c)] for c in srcs])
                for bufs, vals in zip(s_sides, hit[1]):
                    for dst, src in zip(bufs, vals):
                        dst.copy_(src)
            entry[7] = (src_key, srcs)
        graph.replay()
        if optimistic:
            # the checks, while the GPU runs the frame; a change (the cached (key, plan) pair is replaced whenever values or
            # storage changed) -> plans refreshed / rebuilt by the calls below, frame launched again behind that
            self._fused_head()
            if want_bb:
                self._fused_backbone()
            if self.__dict__.get("_fused_cache") is not fc or (want_bb and self.__dict__.get("_fused_bb_cache") is not bc):
                return self._graphed_dense_impl(image, template_features, template_global, head_only, raw_image, post_hw, True)
        return outs

    def forward_all_templates_batch(self, images, template_features, template_features_global, topk=1, seg_sigmoid=False,
                                    raw_image=False):
        """ADDITIVE API (the reference handles one image per call, models/dtoid/__init__.py:64): `forward_all_templates`
        semantics for a batch of images [B,3,H,W] of ONE object (BASELINE configs[2]: 32 images x 21 templates). The
        image backbone runs once on the whole batch (at batch 1 its 120 dependent launches are latency-bound; batched
        they are not), the head graph is replayed per image on that image's feature map, post-processing per image.
        Returns a list of B result lists, each exactly what forward_all_templates(images[i:i+1], ...) returns."""
        with torch.no_grad():
            feats = self._features(images, template_features_global[0], raw_image)
            out = []
            hw = (images.shape[2], images.shape[3])
            for i in range(images.shape[0]):
                f = feats[i:i + 1]
                if self.use_graph and f.is_cuda and not self.training:
                    dense = self._graphed_dense(f, template_features, None, head_only=True, post_hw=hw)
                    post = self.__dict__.get("_graph_post")
                    if post is not None:
                        out.append(self.post_emit(post, dense[2], dense[3], topk, seg_sigmoid))
                        continue
                else:
                    dense = self._dense_head(f, template_features)
                out.append(self.postprocess(*dense, hw, topk, seg_sigmoid))
            return out

    def forward_all_templates(self, image, template_features, template_features_global, topk=1, seg_sigmoid=False,
                              raw_image=False):
        """image [1,3,H,W]; template_features: list of [n_i,640,7,7] chunks; template_features_global: [[1,64,3,3]]
        -> [max_score [k], anchors_pred [k,4], obj_indices [k,1], seg_pred [k,H,W], heatmap_pred [k,hh,hw]]
        raw_image (additive): the image is in [0, 1] and normalizeImageRange happens inside the stem (D1 fused)."""
        with torch.no_grad():
            if self.use_graph and image.is_cuda and not self.training:
                cls_all, reg_all, seg_all, heat_all, fmap = self._graphed_dense(image, template_features,
                                                                                template_features_global[0],
                                                                                raw_image=raw_image,
                                                                                post_hw=(image.shape[2], image.shape[3]))
                post = self.__dict__.get("_graph_post")
                if post is not None:       # candidates + NMS ran inside the graph: only the count and one gather launch are left
                    return self.post_emit(post, seg_all, heat_all, topk, seg_sigmoid)
            else:
                cls_all, reg_all, seg_all, heat_all, fmap = self._dense_all_templates(image, template_features,
                                                                                      template_features_global[0],
                                                                                      raw_image=raw_image)
            return self.postprocess(cls_all, reg_all, seg_all, heat_all, fmap, (image.shape[2], image.shape[3]), topk,
                                    seg_sigmoid)

    use_fused_post = os.environ.get("OSSID_FUSED_POST", "1") != "0"

    def post_dense(self, cls_all, reg_all, fmap, img_hw, state=None):
        """The part of postprocess that needs no host decision (decode + clip of the candidates, top-1000 object scores over
        all templates, NMS 0.5: network.py:543-566) as launches only (ops.detect_post): it can be captured with the dense
        part of the frame. Returns the ops.DetectPost holding the candidates, the keep list and its count."""
        anchors = self.anchors([list(fmap)], device=reg_all.device)
        n_t, A = int(reg_all.shape[0]), int(reg_all.shape[1])
        return ops.detect_post(cls_all, reg_all, anchors, img_hw[1], img_hw[0], min(1000, n_t * A), 0.5, state)

    def post_emit(self, post, seg_all, heat_all, topk, seg_sigmoid):
        """The host's one decision -- how many candidates survived -- and the detection list for the first `topk` of them
        (network.py:566-581) in one launch."""
        post.count_host.copy_(post.count, non_blocking=True)
        torch.cuda.current_stream(post.scores.device).synchronize()
        count = min(int(post.count_host[0]), int(topk))
        return ops.detect_emit(post, count, seg_all[:, 0], heat_all[:, 0], seg_sigmoid)

    def postprocess(self, cls_all, reg_all, seg_all, heat_all, fmap, img_hw, topk=1, seg_sigmoid=False):
        """Dense head outputs of ONE image (cls [n_t,A,2], reg [n_t,A,4], seg [n_t,1,H,W], heat [n_t,1,hh,hw]) -> the
        reference's detection list (network.py:543-581: decode + clip, top-1000 object scores over all templates, NMS 0.5,
        first `topk`, per-detection segmentation / heat map of the template that fired). Everything stays on the device."""
        with torch.no_grad():
            n_t, A = reg_all.shape[0], reg_all.shape[1]
            if reg_all.is_cuda and self.use_fused_post and min(1000, n_t * A) <= 2048:
                post = self.post_dense(cls_all, reg_all, fmap, img_hw)
                return self.post_emit(post, seg_all, heat_all, topk, seg_sigmoid)
            anchors = self.anchors([list(fmap)], device=reg_all.device)
            boxes = ops.decode_clip_boxes(anchors, reg_all, img_hw[1], img_hw[0]).view(-1, 4)
            k = min(1000, n_t * A)
            max_score, max_id = ops.topk_scores(cls_all.reshape(-1, 2)[:, 1], k)  # class 1 = object
            anchors_pred = boxes[max_id]
            obj_indices = (max_id // A).to(torch.float32)[:, None]                # which local template fired
            keep = ops.nms(anchors_pred, max_score, 0.5, sorted_desc=True)[:topk]   # topk returns them sorted
            max_score, anchors_pred, obj_indices = max_score[keep], anchors_pred[keep], obj_indices[keep]
            tid = obj_indices.reshape(-1).long()
            if seg_all.is_cuda:  # seg_sigmoid (additive argument): the sigmoid DtoidNet applies afterwards, folded into the gather
                seg = ops.gather_rows(seg_all[:, 0], tid, sigmoid=seg_sigmoid)
            else:
                seg = torch.sigmoid(seg_all[:, 0][tid]) if seg_sigmoid else seg_all[:, 0][tid]
            return [max_score, anchors_pred, obj_indices, seg, heat_all[:, 0][tid]]
