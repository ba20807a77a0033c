"""Anchor boxes of the DTOID detection head (reference: models/dtoid/anchors.py:30-42, :45-76, :111-132; configured
at network.py:404 with pyramid level 4 / stride 16, base size 30, ratios {0.5,1,2}, scales 1..8 -> 24 per cell).

The reference regenerates the anchors in numpy and uploads them on every forward call; here they are computed once per
(feature-map shape, device) and kept on the device."""
import numpy as np
import torch
import torch.nn as nn


def base_anchors(base_size, ratios, scales):
    """[len(ratios)*len(scales), 4] boxes (x1,y1,x2,y2) centred on the origin; ratio-major, scale-minor order."""
    ratios, scales = np.asarray(ratios, dtype=np.float64), np.asarray(scales, dtype=np.float64)
    r = np.repeat(ratios, len(scales))
    s = np.tile(scales, len(ratios)) * base_size
    area = s * s
    w = np.sqrt(area / r)
    h = w * r
    return np.stack([-0.5 * w, -0.5 * h, 0.5 * w, 0.5 * h], 1)


def grid_anchors(shape, stride, base):
    """Shift the base anchors to every cell centre ((i+0.5)*stride), row-major cells, anchors innermost."""
    hh, ww = int(shape[0]), int(shape[1])
    cx = (np.arange(ww) + 0.5) * stride
    cy = (np.arange(hh) + 0.5) * stride
    gx, gy = np.meshgrid(cx, cy)
    shifts = np.stack([gx.ravel(), gy.ravel(), gx.ravel(), gy.ravel()], 1)
    return (shifts[:, None, :] + base[None, :, :]).reshape(-1, 4)


class Anchors(nn.Module):
    def __init__(self, pyramid_levels=None, strides=None, sizes=None, ratios=None, scales=None):
        super().__init__()
        self.pyramid_levels = [3, 4, 5, 6, 7] if pyramid_levels is None else pyramid_levels
        self.strides = [2 ** x for x in self.pyramid_levels] if strides is None else strides
        self.sizes = [2 ** (x + 2) for x in self.pyramid_levels] if sizes is None else sizes
        self.ratios = np.array([0.5, 1, 2]) if ratios is None else ratios
        self.scales = np.array([2 ** 0, 2 ** (1.0 / 3.0), 2 ** (2.0 / 3.0)]) if scales is None else scales
        self._cache = {}

    def forward(self, image_shapes, device=None):
        """image_shapes: one (h, w) feature-map shape per pyramid level -> float32 [1, sum(h*w*A), 4]."""
        device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        key = (tuple(tuple(int(v) for v in s) for s in image_shapes), str(device))
        if key not in self._cache:
            parts = [grid_anchors(image_shapes[i], self.strides[i], base_anchors(self.sizes[i], self.ratios, self.scales))
                     for i in range(len(self.pyramid_levels))]
            a = np.concatenate(parts, 0).astype(np.float32)[None]
            self._cache[key] = torch.from_numpy(a).to(device)
        return self._cache[key]
