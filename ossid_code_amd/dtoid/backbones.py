"""DenseNet-121 and SqueezeNet-1.1 feature extractors written out explicitly (torchvision is not a dependency).

The reference builds them with torchvision.models.densenet121 / squeezenet1_1(pretrained=True)
(/root/reference/python/ossid/models/dtoid/network.py:164,199,246) and then slices their children by position
(:166-169, :203-212). What must match is therefore the MODULE TREE -- attribute names and child order -- so that
every state_dict key of a reference checkpoint ("model.image_feature_extractor.backdense_1.3.denselayer1.norm1.weight",
"...backbone.features.3.squeeze.weight", ...) lands on a parameter of the same shape here. The bodies follow the
published DenseNet-BC (growth 32, blocks 6/12/24/16, bn_size 4) and SqueezeNet v1.1 definitions that torchvision 0.9.1
implements. No pretrained weights are fetched (there is no network); weights come from a checkpoint or random init.
"""
from collections import OrderedDict

import torch
import torch.nn as nn


class DenseLayer(nn.Module):
    """BN-ReLU-Conv1x1(4k) - BN-ReLU-Conv3x3(k) on the concatenation of all earlier feature maps."""

    def __init__(self, cin, growth=32, bn_size=4):
        super().__init__()
        self.norm1 = nn.BatchNorm2d(cin)
        self.relu1 = nn.ReLU(inplace=True)
        self.conv1 = nn.Conv2d(cin, bn_size * growth, kernel_size=1, stride=1, bias=False)
        self.norm2 = nn.BatchNorm2d(bn_size * growth)
        self.relu2 = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(bn_size * growth, growth, kernel_size=3, stride=1, padding=1, bias=False)

    def forward(self, x):
        y = self.conv1(self.relu1(self.norm1(x)))
        return self.conv2(self.relu2(self.norm2(y)))


DENSE_BLOCK_INPLACE_GRAD = True   # training: one resident feature buffer and one gradient buffer per dense block


def _alias(buf, channels):
    """The first `channels` channels of `buf` as a tensor with its OWN autograd version counter (a plain view would share
    buf's, and the later in-place appends to other channels would invalidate what the layer saved for backward)."""
    t = torch.empty(0, dtype=buf.dtype, device=buf.device)
    t.set_(buf.untyped_storage(), buf.storage_offset(), (buf.shape[0], channels, buf.shape[2], buf.shape[3]), buf.stride())
    return t


class _DenseBlockFn(torch.autograd.Function):
    """A dense block in training without torch.cat and without autograd's O(L^2) gradient accumulation.
    Forward: every layer reads a channel-prefix alias of ONE [B, C_total, H, W] buffer and appends its 32 channels to it;
    each layer's own little graph (BN-ReLU-conv-BN-ReLU-conv, ordinary autograd, same kernels as before) is kept.
    Backward: one gradient buffer of the same shape; layers are replayed last to first, each adds its input gradient
    onto the channel prefix in place - L strided adds instead of ~L^2/2 small ones (DenseNet-121 at batch 8: 1 364
    elementwise launches, 6 ms of a 57 ms step). Parameter gradients accumulate into .grad as usual."""

    @staticmethod
    def forward(ctx, x, block):
        B, C, H, W = x.shape
        buf = x.new_empty(B, C + block.nlayers * block.growth, H, W)
        buf[:, :C] = x
        graphs, c = [], C
        for layer in block.values():
            with torch.enable_grad():
                inp = _alias(buf, c).requires_grad_()
                out = layer(inp)
            buf[:, c:c + block.growth] = out.detach()
            graphs.append((inp, out))
            c += block.growth
        ctx.graphs, ctx.cin, ctx.growth = graphs, C, block.growth
        return buf

    @staticmethod
    def backward(ctx, gbuf):
        g = gbuf.clone(memory_format=torch.contiguous_format)
        c = ctx.cin + len(ctx.graphs) * ctx.growth
        for inp, out in reversed(ctx.graphs):
            c -= ctx.growth
            torch.autograd.backward([out], [g[:, c:c + ctx.growth]])
            g[:, :c] += inp.grad
            inp.grad = None
        ctx.graphs = None
        return g[:, :ctx.cin], None


class DenseBlock(nn.ModuleDict):
    def __init__(self, nlayers, cin, growth=32, bn_size=4):
        super().__init__()
        for i in range(nlayers):
            self["denselayer%d" % (i + 1)] = DenseLayer(cin + i * growth, growth, bn_size)
        self.cin, self.growth, self.nlayers = cin, growth, nlayers

    def forward(self, x):
        if torch.is_grad_enabled() and x.requires_grad or self.training:
            if DENSE_BLOCK_INPLACE_GRAD and torch.is_grad_enabled() and x.requires_grad and x.is_cuda:
                return _DenseBlockFn.apply(x, self)
            feats = [x]
            for layer in self.values():
                feats.append(layer(torch.cat(feats, 1)))
            return torch.cat(feats, 1)
        # inference: one resident buffer for the whole block; each layer appends its 32 channels in place
        # instead of re-concatenating (and re-reading) everything before it.
        B, C, H, W = x.shape
        buf = x.new_empty(B, C + self.nlayers * self.growth, H, W)
        buf[:, :C] = x
        c = C
        for layer in self.values():
            buf[:, c:c + self.growth] = layer(buf[:, :c])
            c += self.growth
        return buf


class Transition(nn.Sequential):
    def __init__(self, cin, cout):
        super().__init__()
        self.norm = nn.BatchNorm2d(cin)
        self.relu = nn.ReLU(inplace=True)
        self.conv = nn.Conv2d(cin, cout, kernel_size=1, stride=1, bias=False)
        self.pool = nn.AvgPool2d(kernel_size=2, stride=2)


def densenet121_features():
    """nn.Sequential with torchvision's child names/order: conv0 norm0 relu0 pool0 denseblock1 transition1 ...
    denseblock4 norm5 (12 children)."""
    layers = OrderedDict()
    layers["conv0"] = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
    layers["norm0"] = nn.BatchNorm2d(64)
    layers["relu0"] = nn.ReLU(inplace=True)
    layers["pool0"] = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
    c = 64
    for i, n in enumerate((6, 12, 24, 16)):
        layers["denseblock%d" % (i + 1)] = DenseBlock(n, c)
        c += n * 32
        if i != 3:
            layers["transition%d" % (i + 1)] = Transition(c, c // 2)
            c //= 2
    layers["norm5"] = nn.BatchNorm2d(c)
    return nn.Sequential(layers)


class Fire(nn.Module):
    def __init__(self, cin, squeeze, e1, e3):
        super().__init__()
        self.squeeze = nn.Conv2d(cin, squeeze, kernel_size=1)
        self.squeeze_activation = nn.ReLU(inplace=True)
        self.expand1x1 = nn.Conv2d(squeeze, e1, kernel_size=1)
        self.expand1x1_activation = nn.ReLU(inplace=True)
        self.expand3x3 = nn.Conv2d(squeeze, e3, kernel_size=3, padding=1)
        self.expand3x3_activation = nn.ReLU(inplace=True)

    def forward(self, x):
        x = self.squeeze_activation(self.squeeze(x))
        return torch.cat([self.expand1x1_activation(self.expand1x1(x)), self.expand3x3_activation(self.expand3x3(x))], 1)


class SqueezeNet11(nn.Module):
    """features (13 children) + classifier, as torchvision's squeezenet1_1; the reference keeps the WHOLE model as an
    attribute (network.py:199,246), so the never-used classifier and 3-channel stem stay in the state_dict."""

    def __init__(self, num_classes=1000):
        super().__init__()
        self.features = nn.Sequential(
            nn.Conv2d(3, 64, kernel_size=3, stride=2), nn.ReLU(inplace=True),
            nn.MaxPool2d(kernel_size=3, stride=2, ceil_mode=True),
            Fire(64, 16, 64, 64), Fire(128, 16, 64, 64),
            nn.MaxPool2d(kernel_size=3, stride=2, ceil_mode=True),
            Fire(128, 32, 128, 128), Fire(256, 32, 128, 128),
            nn.MaxPool2d(kernel_size=3, stride=2, ceil_mode=True),
            Fire(256, 48, 192, 192), Fire(384, 48, 192, 192), Fire(384, 64, 256, 256), Fire(512, 64, 256, 256))
        self.classifier = nn.Sequential(nn.Dropout(p=0.5), nn.Conv2d(512, num_classes, kernel_size=1),
                                        nn.ReLU(inplace=True), nn.AdaptiveAvgPool2d((1, 1)))

    def forward(self, x):
        return torch.flatten(self.classifier(self.features(x)), 1)
