"""PointNet2SSG -- host-side mirror of zephyr.models.pointnet2.PointNet2SSG.

Reference interface (the class itself lives in the un-vendored `zephyr` package):
  ctor  PointNet2SSG(dim_point, args, num_class=1)   /root/reference/python/ossid/scripts/online_learning.py:212,218,224
  use   .load_state_dict(ckpt['state_dict']); .to(0).eval(); .device; model({"point_x": x})
                                                      scripts/online_learning.py:213-227, utils/zephyr_utils.py:34
The module tree reproduces pointnet2_ops v3.0.0's PointNet2ClassificationSSG (SA_modules.{0,1,2}.mlps.0 =
Sequential(Conv2d 1x1 no-bias, BatchNorm2d, ReLU) x3; fc_layer = Linear/BN1d/ReLU/Linear/BN1d/ReLU/Dropout/
Linear) so a state_dict saved from that architecture loads key for key. The forward pass does not run in
torch: the parameters are folded (BN into the convolution), permuted into SPEC.md's canonical channel order,
packed into the MFMA operand layout of csrc/pn2.hip and handed to libossid_hip.so.
"""
import os

import numpy as np
import torch
import torch.nn as nn

from .. import _lib

BN_EPS = 1e-5
# (kpad, cout) of the 12 dense layers after folding, SPEC.md 4.3
LAYER_K = (8, 64, 64, 136, 128, 128, 264, 256, 512, 1024, 512, 256)
LAYER_C = (64, 64, 128, 128, 128, 256, 256, 512, 1024, 512, 256, 1)


def _shared_mlp(spec):
    layers = []
    for i in range(1, len(spec)):
        layers += [nn.Conv2d(spec[i - 1], spec[i], kernel_size=1, bias=False), nn.BatchNorm2d(spec[i]), nn.ReLU(True)]
    return nn.Sequential(*layers)


class _SAModule(nn.Module):
    """Parameter container shaped like pointnet2_ops.PointnetSAModule (single-scale)."""

    def __init__(self, npoint, radius, nsample, mlp, use_xyz=True):
        super().__init__()
        self.npoint, self.radius, self.nsample = npoint, radius, nsample
        spec = list(mlp)
        if use_xyz:
            spec[0] += 3
        self.mlps = nn.ModuleList([_shared_mlp(spec)])


def _fold_bn(weight, bn):
    """(W[cout,cin], BN) -> (W * s, beta - mean * s) with s = gamma / sqrt(var + eps); float32 numpy ops."""
    W = weight.detach().cpu().numpy().astype(np.float32).reshape(weight.shape[0], -1)
    gamma = bn.weight.detach().cpu().numpy().astype(np.float32)
    beta = bn.bias.detach().cpu().numpy().astype(np.float32)
    mean = bn.running_mean.detach().cpu().numpy().astype(np.float32)
    var = bn.running_var.detach().cpu().numpy().astype(np.float32)
    s = gamma / np.sqrt(var + np.float32(bn.eps))
    return (W * s[:, None]).astype(np.float32), (beta - mean * s).astype(np.float32)


def fold_pn2(model):
    """12 folded layers [(W'[cout,kpad], b'[cout])] in canonical input-channel order (SPEC.md 4.3)."""
    out = []
    for si, sa in enumerate(model.SA_modules):
        seq = sa.mlps[0]
        for li in range(3):
            W, b = _fold_bn(seq[3 * li].weight, seq[3 * li + 1])
            if li == 0 and si > 0:
                # torch order is (grouped_xyz(3), features(C)); canonical is (features(C), xyz(3), 0-pad to x8)
                C = W.shape[1] - 3
                Wc = np.zeros((W.shape[0], C + 8), np.float32)
                Wc[:, :C] = W[:, 3:]
                Wc[:, C:C + 3] = W[:, :3]
                W = Wc
            out.append((np.ascontiguousarray(W), np.ascontiguousarray(b)))
    fc = model.fc_layer
    out.append(_fold_bn(fc[0].weight, fc[1]))
    out.append(_fold_bn(fc[3].weight, fc[4]))
    out.append((fc[7].weight.detach().cpu().numpy().astype(np.float32).reshape(1, -1),
                fc[7].bias.detach().cpu().numpy().astype(np.float32).reshape(1)))
    for i, (W, b) in enumerate(out):
        assert W.shape == (LAYER_C[i], LAYER_K[i]), (i, W.shape)
    return out


def _pack_mfma(W):
    """W'[cout,kpad] -> [cout/32][kpad/8][64 lanes][4]: lane (c, h) holds W'[32mt+c][8kb+4h+0..3], the A operands
    of the four chained 32x32x2 MFMAs of k-block kb (csrc/pn2.hip)."""
    cout, kpad = W.shape
    return np.ascontiguousarray(W.reshape(cout // 32, 32, kpad // 8, 2, 4).transpose(0, 2, 3, 1, 4)).reshape(-1)


def pack_pn2(folded):
    """Folded layers -> (blob float32[...], w_off[12], b_off[12], wxyz2_off), offsets in floats, 64-float aligned."""
    parts, w_off, b_off = [], [], []
    pos = 0

    def add(a):
        nonlocal pos
        a = np.ascontiguousarray(a, dtype=np.float32).reshape(-1)
        off = pos
        pad = (-a.size) % 64
        parts.append(a)
        if pad:
            parts.append(np.zeros(pad, np.float32))
        pos += a.size + pad
        return off

    wxyz2_off = 0
    for i, (W, b) in enumerate(folded):
        if i == 3:  # SA2 L1: feature columns on the matrix core, xyz columns applied per sample
            w_off.append(add(_pack_mfma(W[:, :128])))
            wxyz2_off = add(np.stack([W[:, 128], W[:, 129], W[:, 130]], 0))
        elif i <= 8:
            w_off.append(add(_pack_mfma(W)))
        elif i <= 10:  # FC1/FC2 transposed [k][cout]
            w_off.append(add(W.T))
        else:
            w_off.append(add(W))
        b_off.append(add(b))
    return np.concatenate(parts), w_off, b_off, wxyz2_off


class PointNet2SSG(nn.Module):
    MAX_CHUNK = 4096  # hypotheses per C-ABI call (bounds the workspace at ~0.72 MB each)
    # Measured and NOT adopted (round 3, profiles/r03_scorer_overlap.txt): overlapping the sampling kernels (fps / ball query:
    # vector-ALU and LDS work, 6.5 % of the step) with the matrix-core stages INSIDE one call. The sampling stages read
    # coordinates only, so they can run ahead on a side stream -- but the MLP kernels fill the register file and 51-128 KB of LDS
    # per workgroup, so nothing co-resides with them: all a second stream can do is fill the tail of each launch, and cutting
    # one call into pieces adds more tails (and partial rounds: sa3 runs one workgroup per hypothesis, 1000 = 3.9 rounds of
    # 256) than it fills. One call, one stream: 15.19 ms; 4 pieces alternating two streams 15.78; sampling ahead on a side
    # stream in pieces [128, 872] 16.00, [504, 496] 15.55. Two whole FRAMES in flight on two streams (scoring.networkInferenceMany,
    # bench.py's timed region) do gain: 14.67 ms per frame.

    def __init__(self, dim_in, args=None, num_class=1):
        super().__init__()
        if dim_in != 8:
            raise NotImplementedError("the MI355X scorer is built for dim_point=8 (HSVD_diff_uv_norm), got %r" % dim_in)
        if num_class != 1:
            raise NotImplementedError("num_class must be 1 (online_learning.py:212)")
        self.args = args
        self.dim_in = dim_in
        extra = getattr(args, "extra_bottleneck_dim", 0) if args is not None else 0
        if extra:
            raise NotImplementedError("extra_bottleneck_dim must be 0 (online_learning.py:209)")
        self.SA_modules = nn.ModuleList([
            _SAModule(512, 0.2, 64, [dim_in - 3, 64, 64, 128]),
            _SAModule(128, 0.4, 64, [128, 128, 128, 256]),
            _SAModule(None, None, None, [256, 256, 512, 1024]),
        ])
        self.fc_layer = nn.Sequential(
            nn.Linear(1024, 512, bias=False), nn.BatchNorm1d(512), nn.ReLU(True),
            nn.Linear(512, 256, bias=False), nn.BatchNorm1d(256), nn.ReLU(True),
            nn.Dropout(0.5), nn.Linear(256, num_class))
        self._packed = None  # (version key, device blob, PN2Weights)
        self._ws = None

    @property
    def device(self):  # LightningModule attribute the caller reads (utils/zephyr_utils.py:34)
        return next(self.parameters()).device

    def _version_key(self, device):
        return (str(device),) + tuple(int(t._version) for t in list(self.parameters()) + list(self.buffers()))

    def packed_weights(self, device):
        key = self._version_key(device)
        if self._packed is None or self._packed[0] != key:
            blob, w_off, b_off, wxyz2_off = pack_pn2(fold_pn2(self))
            dblob = torch.from_numpy(blob).to(device)
            st = _lib.PN2Weights()
            st.blob = dblob.data_ptr()
            for i in range(12):
                st.w_off[i], st.b_off[i] = w_off[i], b_off[i]
            st.wxyz2_off = wxyz2_off
            st.npoint1, st.npoint2 = self.SA_modules[0].npoint, self.SA_modules[1].npoint
            st.radius1, st.radius2 = self.SA_modules[0].radius, self.SA_modules[1].radius
            self._packed = (key, dblob, st)
        return self._packed[2]

    def _workspace(self, nbytes, device):
        """Grow-only scratch buffer, one per (device, stream): frames in flight on different HIP streams must not
        share stage buffers."""
        if self._ws is None:
            self._ws = {}
        key = (str(device), torch.cuda.current_stream(device).cuda_stream)
        ws = self._ws.get(key)
        if ws is None or ws.numel() < nbytes:
            ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
            self._ws[key] = ws
        return ws

    def score(self, point_x, debug=False, stage_events=None):
        """point_x [B, M, 8] float32 on the GPU -> scores [B] (and the stage tensors when debug).
        stage_events: a _lib.StageEvents to time every stage of the call."""
        _lib.require_cuda(point_x)
        if self.training:
            raise NotImplementedError("PointNet2SSG is inference-only on this path: call .eval() "
                                      "(online_learning.py:215,221,227)")
        if point_x.dtype != torch.float32 or point_x.dim() != 3 or point_x.shape[2] != 8:
            raise ValueError("point_x must be float32 [B, M, 8]")
        point_x = point_x.contiguous()
        B, M, _ = point_x.shape
        dev = point_x.device
        w = self.packed_weights(dev)
        np1, np2 = w.npoint1, w.npoint2
        if B and M < np1:
            raise ValueError("need at least npoint=%d model points per hypothesis, got %d" % (np1, M))
        scores = torch.empty(B, dtype=torch.float32, device=dev)
        dbg = None
        if debug:
            i32, f32 = dict(dtype=torch.int32, device=dev), dict(dtype=torch.float32, device=dev)
            dbg = dict(fps1=torch.empty(B, np1, **i32), ball1=torch.empty(B, np1, 64, **i32),
                       feat1=torch.empty(B, np1, 128, **f32), fps2=torch.empty(B, np2, **i32),
                       ball2=torch.empty(B, np2, 64, **i32), feat2=torch.empty(B, np2, 256, **f32),
                       feat3=torch.empty(B, 1024, **f32))
        f = _lib.fn("ossid_pn2_score")
        wsb = _lib.fn("ossid_pn2_workspace_bytes")

        def launch(b0, nb, events):
            nbytes = wsb(nb, M, np1, np2)
            ws = self._workspace(nbytes, dev)
            dargs = [None] * 7
            if dbg is not None:
                dargs = [dbg[k][b0:b0 + nb].data_ptr() for k in ("fps1", "ball1", "feat1", "fps2", "ball2", "feat2", "feat3")]
            rc = f(point_x[b0:b0 + nb].data_ptr(), nb, M, w, ws.data_ptr(), nbytes, scores[b0:b0 + nb].data_ptr(),
                   *dargs, None if events is None else events.arr, _lib.stream())
            _lib.check(rc, "ossid_pn2_score")

        with torch.cuda.device(dev):
            for b0 in range(0, B, self.MAX_CHUNK):
                launch(b0, min(self.MAX_CHUNK, B - b0), stage_events)
        return (scores, dbg) if debug else scores

    def forward(self, data):
        x = data["point_x"] if isinstance(data, dict) else data
        return self.score(x).unsqueeze(1)
