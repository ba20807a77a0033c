"""Training-mode (finetune) execution of DTOID's two SqueezeNet-1.1 template encoders on this repo's kernels, each as ONE
autograd node whose forward and backward are recorded launch sequences (_lib.Seq) over persistent buffers.

Reference: models/dtoid/network.py:195-239 (TemplateFeatExtractGlobal) and :242-279 (TemplateFeatExtract), run in training
mode by scripts/online_learning.py:650-679. Same arithmetic as the nn.Module path (dtoid/network.py), channels-last:

  stem      4-channel 3x3 stride-2 convolution = ossid_im2col_stem + a 1x1 MFMA convolution (48 columns), ReLU in its epilogue
  Fire      squeeze 1x1 -> ReLU -> (expand 1x1 | expand 3x3) -> ReLU, the two expands writing channel slices of one buffer
  max-pool  with argmax positions (ossid_maxpool_idx_nhwc / _bwd_nhwc)
  taps      training BatchNorm on the 128-channel 30x30 tap and the 512-channel 7x7 output (column sums -> fold -> one
            generic pass), bilinear 30 -> 7 resize by tap table (ossid_resample_taps_nhwc), both written into slices of the
            640-channel result
  global    two VALID 3x3 convolutions = padded convolutions whose interior is kept (crop by tap table), ELU in the
            epilogue, training BatchNorm folded into the next convolution's input staging / one generic pass

Backward is the mirror image with the same few kernels (ReLU / ELU masks, bias sums and BatchNorm backward are instances of
ossid_chan_op; data gradients are the forward convolution on rotated weights; all ~25 weight gradients of an encoder are ONE
grouped launch on the weight-gradient stream). Why one node per encoder: each encoder is ~60 forward and ~200 backward
launches of a few microseconds each, they sit on the critical path twice per step (the image stem needs the global one's
output, the stem's backward feeds its backward), and the host thread that enqueues every stream of the step spent 1-2.5 ms
per encoder and direction issuing them through torch / autograd -- visible as holes on the main stream
(profiles/r02_finetune_step_timeline.txt at ms 2-3, 12-13, 46-47, 58-60). Replayed, an encoder direction is ~0.2 ms of host.
"""
import torch
import torch.nn as nn

from .. import _lib
from . import train_ops as T
from .train_ops import bn_fold_bwd, bn_fold_fwd, batch_stats, chan_op, conv_raw, flat, new_buf

_TAPS = {}


def _tables_from_matrix(A, device):
    """Dense [n_out, n_in] resampling matrix -> (idx [n_out, T] int32, w [n_out, T] float32, T) tap tables."""
    n_out = A.shape[0]
    nz = [torch.nonzero(A[o]).flatten().tolist() for o in range(n_out)]
    Tn = max(1, max(len(r) for r in nz))
    idx = torch.full((n_out, Tn), -1, dtype=torch.int32)
    w = torch.zeros((n_out, Tn), dtype=torch.float32)
    for o, r in enumerate(nz):
        for j, i in enumerate(r):
            idx[o, j], w[o, j] = i, A[o, i]
    return idx.to(device), w.to(device), Tn


def tap_tables(kind, n_in, n_out, device, adjoint=False):
    """Tap tables of ossid_resample_taps_nhwc along one axis. kind "bilinear": F.interpolate(mode="bilinear",
    align_corners=False) from n_in to n_out; "crop": drop one row at each border (n_out = n_in - 2). adjoint: the
    transposed operator (n_out -> n_in), i.e. the gradient."""
    key = (kind, n_in, n_out, str(device), adjoint)
    if key not in _TAPS:
        if kind == "bilinear":
            from .network import _bilinear_matrix
            A = _bilinear_matrix(n_in, n_out, torch.device("cpu")).clone()
        else:
            assert n_out == n_in - 2
            A = torch.zeros(n_out, n_in)
            A[torch.arange(n_out), torch.arange(n_out) + 1] = 1.0
        _TAPS[key] = _tables_from_matrix(A.t().contiguous() if adjoint else A, device)
    return _TAPS[key]


def _pad_taps(t, Tn):
    idx, w, n = t
    if n == Tn:
        return t
    idx2 = torch.full((idx.shape[0], Tn), -1, dtype=torch.int32, device=idx.device)
    w2 = torch.zeros((w.shape[0], Tn), dtype=torch.float32, device=w.device)
    idx2[:, :n], w2[:, :n] = idx, w
    return idx2, w2, Tn


def resample(x_flat, B, Hin, Win, C, x_cs, Hout, Wout, ty, tx, out_flat, out_cs=0, out_coff=0):
    """ossid_resample_taps_nhwc on raw channels-last buffers (flat views; a channel slice = pointer + channel stride)."""
    Tn = max(ty[2], tx[2])
    if ty[2] != tx[2]:                                    # non-square images: one table has fewer taps per output index
        key = (id(ty[0]), id(tx[0]), Tn)
        if key not in _TAPS:
            _TAPS[key] = (_pad_taps(ty, Tn), _pad_taps(tx, Tn), ty, tx)      # (the originals are kept alive with the key)
        ty, tx = _TAPS[key][0], _TAPS[key][1]
    with _lib.on_device(x_flat.device):
        rc = _lib.fn("ossid_resample_taps_nhwc")(x_flat.data_ptr(), B, Hin, Win, C, int(x_cs), Hout, Wout, ty[0].data_ptr(),
                                                 ty[1].data_ptr(), tx[0].data_ptr(), tx[1].data_ptr(), Tn,
                                                 out_flat.data_ptr(), int(out_cs), int(out_coff), _lib.stream())
    _lib.check(rc, "ossid_resample_taps_nhwc")


def _pool_cfg(m):
    g = lambda v: v if isinstance(v, int) else v[0]      # noqa: E731
    return g(m.kernel_size), g(m.stride), g(m.padding), bool(m.ceil_mode)


def _pool_out(n, k, stride, pad, ceil_mode):
    o = -(-(n + 2 * pad - k) // stride) + 1 if ceil_mode else (n + 2 * pad - k) // stride + 1
    return o - 1 if ceil_mode and (o - 1) * stride >= n + pad else o


def encoder_params(mod):
    """The parameters an encoder's node differentiates, in the order its backward returns their gradients."""
    ps = [mod.backbone_0[0].weight, mod.backbone_0[0].bias]
    for part in (mod.backbone_1, mod.backbone_2):
        for m in part:
            if hasattr(m, "squeeze"):
                for cv in (m.squeeze, m.expand1x1, m.expand3x3):
                    ps += [cv.weight, cv.bias]
    ps += [mod.norm_1.weight, mod.norm_1.bias, mod.norm_2.weight, mod.norm_2.bias]
    if hasattr(mod, "final_conv_1"):
        ps += [mod.final_conv_1.weight, mod.final_conv_1.bias, mod.final_norm_1.weight, mod.final_norm_1.bias,
               mod.final_conv_2.weight, mod.final_conv_2.bias, mod.final_norm_2.weight, mod.final_norm_2.bias]
    return ps


def encoder_convs(mod):
    """Every nn.Conv2d of the encoder that goes through the step's PackPlan (all but the stem, whose weight is re-laid)."""
    cs = []
    for part in (mod.backbone_1, mod.backbone_2):
        for m in part:
            if hasattr(m, "squeeze"):
                cs += [m.squeeze, m.expand1x1, m.expand3x3]
    if hasattr(mod, "final_conv_1"):
        cs += [mod.final_conv_1, mod.final_conv_2]
    return cs


STEM_KPAD = 48        # 3 x 3 x 4 = 36 im2col columns padded to the convolution kernel's 16-channel granularity


def stem_relayout(src, dst, cout, cin, k, kpad, inverse=False):
    """ossid_stem_weight_relayout: [cout, cin, k, k] -> [cout, kpad] in ossid_im2col_stem's column order, or back."""
    with _lib.on_device(src.device):
        _lib.check(_lib.fn("ossid_stem_weight_relayout")(src.data_ptr(), dst.data_ptr(), cout, cin, k, kpad, 1 if inverse else 0,
                                                         _lib.stream()), "ossid_stem_weight_relayout")


def _bn_train(x_flat, n, C, bn, cs=0):
    """Column sums + fold of a training BatchNorm on a channels-last tensor: [4, C] = scale, shift, mean, rstd."""
    st = batch_stats(x_flat, n, C, cs=cs)
    return bn_fold_fwd(st, C, n, bn.weight, bn.bias, bn.eps, T._mom(bn), bn.running_mean if bn.track_running_stats else None,
                       bn.running_var if bn.track_running_stats else None)


def _bn_back(g_flat, x_flat, n, C, f, bn, out_flat, g_cs=0, pre_scaled=False):
    """Backward of y = x * scale + shift with (scale, shift) a training BatchNorm's fold `f`, given g = dL/dy: writes
    dL/dx (through scale AND through the batch statistics) to out, returns (d gamma, d beta). Three launches + the fold."""
    dev = g_flat.device
    s = chan_op(g_flat, n, C, x=x_flat, out=out_flat, g_cs=g_cs, alpha=f[0], sum_mode=1, defer=True)     # out = g * scale
    r = new_buf((4, C), dev)
    bn_fold_bwd(None, None, bn.weight, f[2], f[3], C, n, r[0], r[1], r[2], r[3], partials=s)
    chan_op(out_flat, n, C, x=x_flat, out=out_flat, beta=r[2], kappa=r[3])                               # + coef_x * x + coef_1
    return r[0], r[1]


def _encoder_forward(mod, cols):
    """Forward launches of one encoder on the im2col'd templates `cols` [B,48,61,61] (raw ops only: recordable).
    Returns (output tensor, saved dict)."""
    dev = cols.device
    B, _, H, W = cols.shape
    stem = mod.backbone_0[0]
    sv = {"cols": cols, "stages": []}
    w_stem = new_buf((64, STEM_KPAD), dev)
    stem_relayout(stem.weight.detach(), w_stem, 64, 4, 3, STEM_KPAD)
    form = T._KIND_FORM[T.FWD_ENCODER]     # every convolution in front of a ReLU / max-pool runs at f32-level accuracy (T.FWD_ENCODER)
    wpk = new_buf((_lib.fn("ossid_conv_packed_floats_form")(64, STEM_KPAD, 1, form),), dev)
    with _lib.on_device(dev):
        _lib.check(_lib.fn("ossid_conv_pack_weights_form")(w_stem.data_ptr(), 64, STEM_KPAD, 1, 0, form, wpk.data_ptr(), _lib.stream()),
                   "ossid_conv_pack_weights_form")
    wpk._ossid_exact = form
    x = new_buf((B, 64, H, W), dev, channels_last=True)
    conv_raw(cols, wpk, B, H, W, STEM_KPAD, 64, 1, x, bias=stem.bias.detach(), act=2)
    sv["x0"] = x
    taps = []
    for part in (mod.backbone_1, mod.backbone_2):
        for m in part:
            C = x.shape[1]
            if isinstance(m, nn.MaxPool2d):
                k, st, pd, ceil = _pool_cfg(m)
                Ho, Wo = _pool_out(H, k, st, pd, ceil), _pool_out(W, k, st, pd, ceil)
                out = new_buf((B, C, Ho, Wo), dev, channels_last=True)
                idx = torch.empty(B * Ho * Wo * C, dtype=torch.uint8, device=dev)
                _lib.recording() and _lib.recording().keep(idx)
                with _lib.on_device(dev):
                    _lib.check(_lib.fn("ossid_maxpool_idx_nhwc")(x.data_ptr(), B, H, W, C, k, st, pd, 1 if ceil else 0, out.data_ptr(),
                                                                 idx.data_ptr(), _lib.stream()), "ossid_maxpool_idx_nhwc")
                sv["stages"].append(("pool", (C, H, W, k, st, pd, Ho, Wo), idx))
                x, H, W = out, Ho, Wo
            elif isinstance(m, nn.ReLU):
                continue
            else:
                sq, e1, e3 = m.squeeze, m.expand1x1, m.expand3x3
                nsq, n1, n3 = sq.out_channels, e1.out_channels, e3.out_channels
                s = new_buf((B, nsq, H, W), dev, channels_last=True)
                conv_raw(x, T._pack(sq.weight, T.FWD_ENCODER), B, H, W, C, nsq, 1, s, bias=sq.bias.detach(), act=2)
                out = new_buf((B, n1 + n3, H, W), dev, channels_last=True)
                conv_raw(s, T._pack(e1.weight, T.FWD_ENCODER), B, H, W, nsq, n1, 1, out, bias=e1.bias.detach(), act=2, out_cs=n1 + n3)
                conv_raw(s, T._pack(e3.weight, T.FWD_ENCODER), B, H, W, nsq, n3, 9, out, bias=e3.bias.detach(), act=2, out_cs=n1 + n3,
                         out_coff=n1)
                sv["stages"].append(("fire", m, x, s, out, (H, W)))
                x = out
        taps.append((x, H, W))
    (x1, H1, W1), (x2, H2, W2) = taps
    C1, C2 = x1.shape[1], x2.shape[1]
    n1, n2 = B * H1 * W1, B * H2 * W2
    f1, f2 = _bn_train(flat(x1), n1, C1, mod.norm_1), _bn_train(flat(x2), n2, C2, mod.norm_2)
    Cf = C1 + C2
    xf = new_buf((B, Cf, H2, W2), dev, channels_last=True)
    chan_op(flat(x2), n2, C2, out=flat(xf), out_cs=Cf, alpha=f2[0], kappa=f2[1])                     # norm_2(x2) -> xf[:, :512]
    x1n = new_buf((B, C1, H1, W1), dev, channels_last=True)
    chan_op(flat(x1), n1, C1, out=flat(x1n), alpha=f1[0], kappa=f1[1])                               # norm_1(x1)
    resample(flat(x1n), B, H1, W1, C1, 0, H2, W2, tap_tables("bilinear", H1, H2, dev), tap_tables("bilinear", W1, W2, dev),
             flat(xf), out_cs=Cf, out_coff=C2)                                                       # resized -> xf[:, 512:]
    sv.update(x1=x1, x2=x2, f1=f1, f2=f2, xf=xf, dims=(H1, W1, H2, W2))
    out = xf
    if hasattr(mod, "final_conv_1"):
        fin = []
        a, Ha, Wa, pre = xf, H2, W2, None
        for conv, bn in ((mod.final_conv_1, mod.final_norm_1), (mod.final_conv_2, mod.final_norm_2)):
            cin, cout = conv.in_channels, conv.out_channels
            u = new_buf((B, cout, Ha, Wa), dev, channels_last=True)                                  # padded conv + ELU
            # (exact-f32 launches, forward and data gradient: the BatchNorm behind each of the two normalises over B x 5 x 5 and
            # B x 3 x 3 values per channel and amplifies rounding differences ~100x; the layers are tiny)
            conv_raw(a, T._pack(conv.weight, "fwd_exact"), B, Ha, Wa, cin, cout, 9, u, bias=conv.bias.detach(), pre=pre, act=1)
            c = new_buf((B, cout, Ha - 2, Wa - 2), dev, channels_last=True)                          # its interior = the valid conv
            resample(flat(u), B, Ha, Wa, cout, 0, Ha - 2, Wa - 2, tap_tables("crop", Ha, Ha - 2, dev),
                     tap_tables("crop", Wa, Wa - 2, dev), flat(c))
            f = _bn_train(flat(c), B * (Ha - 2) * (Wa - 2), cout, bn)
            fin.append((conv, bn, a, pre, u, c, f, (Ha, Wa)))
            a, Ha, Wa, pre = c, Ha - 2, Wa - 2, (f[0], f[1])
        out = new_buf((B, a.shape[1], Ha, Wa), dev, channels_last=True)
        chan_op(flat(a), B * Ha * Wa, a.shape[1], out=flat(out), alpha=pre[0], kappa=pre[1])         # final_norm_2(...)
        sv["final"] = fin
    return out, sv


def _encoder_backward(mod, gout, sv, side, direct=False):
    """Backward launches of one encoder given gout = dL/d(output) in a persistent buffer (raw ops only).
    Returns the parameter gradients in encoder_params order."""
    dev = gout.device
    cols = sv["cols"]
    B = cols.shape[0]
    H1, W1, H2, W2 = sv["dims"]
    x1, x2, f1, f2, xf = sv["x1"], sv["x2"], sv["f1"], sv["f2"], sv["xf"]
    C1, C2 = x1.shape[1], x2.shape[1]
    Cf = C1 + C2
    deferred, grads = [], {}

    def wgrad_later(conv, x, dy, Bh, Hh, Wh, cin, cout, taps, pre=None, in_cs=0, dy_cs=0, key=None):
        dw = T.grad_home(conv.weight, direct) if conv is not None else new_buf((cout, cin, 1, 1), dev)
        deferred.append(dict(x=x, dy=dy, B=Bh, H=Hh, W=Wh, cin=cin, cout=cout, taps=taps, dw=dw, pre=pre, pre_relu=False,
                             in_cs=in_cs, dy_cs=dy_cs))
        grads[key if key is not None else conv.weight] = dw
        return dw

    # ---- the global branch's two valid convolutions + BatchNorms
    if "final" in sv:
        g = gout
        fin = sv["final"]
        for li in (1, 0):
            conv, bn, a, pre, u, c, f, (Ha, Wa) = fin[li]
            cin, cout = conv.in_channels, conv.out_channels
            n_c = B * (Ha - 2) * (Wa - 2)
            # g = d/d(c * scale + shift): the encoder's output (li = 1, materialised) or the prologue'd input of the second
            # convolution (li = 0: its data gradient already is the gradient w.r.t. the affine's output)
            dc = new_buf(c.shape, dev, channels_last=True)
            grads[bn.weight], grads[bn.bias] = _bn_back(flat(g), flat(c), n_c, cout, f, bn, flat(dc))
            du = new_buf(u.shape, dev, channels_last=True)                                             # zero-pad back
            resample(flat(dc), B, Ha - 2, Wa - 2, cout, 0, Ha, Wa, tap_tables("crop", Ha, Ha - 2, dev, adjoint=True),
                     tap_tables("crop", Wa, Wa - 2, dev, adjoint=True), flat(du))
            sums = chan_op(flat(du), B * Ha * Wa, cout, x=flat(u), out=flat(du), mask_mode=2, sum_mode=2)   # ELU', bias sums
            grads[conv.bias] = sums[0]
            wgrad_later(conv, a, du, B, Ha, Wa, cin, cout, 9, pre=pre)
            g = new_buf(a.shape, dev, channels_last=True)                                              # d/d(prologue'd input)
            conv_raw(du, T._pack(conv.weight, "dgrad_exact"), B, Ha, Wa, cout, cin, 9, g)
        dxf = g
    else:
        dxf = gout
    # ---- the two taps: xf[:, :C2] = norm_2(x2), xf[:, C2:] = resize(norm_1(x1))
    dx2 = new_buf(x2.shape, dev, channels_last=True)
    grads[mod.norm_2.weight], grads[mod.norm_2.bias] = _bn_back(flat(dxf), flat(x2), B * H2 * W2, C2, f2, mod.norm_2, flat(dx2),
                                                                g_cs=Cf)
    dx1n = new_buf(x1.shape, dev, channels_last=True)
    resample(flat(dxf, C2), B, H2, W2, C1, Cf, H1, W1, tap_tables("bilinear", H1, H2, dev, adjoint=True),
             tap_tables("bilinear", W1, W2, dev, adjoint=True), flat(dx1n))
    dx1 = new_buf(x1.shape, dev, channels_last=True)
    grads[mod.norm_1.weight], grads[mod.norm_1.bias] = _bn_back(flat(dx1n), flat(x1), B * H1 * W1, C1, f1, mod.norm_1, flat(dx1))
    # ---- the stages, last to first
    n_stage2 = len(list(mod.backbone_2)) - sum(isinstance(m, nn.ReLU) for m in mod.backbone_2)
    g = dx2
    stages = sv["stages"]
    for si in range(len(stages) - 1, -1, -1):
        if si == len(stages) - 1 - n_stage2:          # crossing the 30x30 tap: add the resize branch's gradient
            chan_op(flat(dx1), B * H1 * W1, C1, out=flat(g), accumulate=True)
        st = stages[si]
        if st[0] == "pool":
            _, (C, H, W, k, s_, pd, Ho, Wo), idx = st
            dx = new_buf((B, C, H, W), dev, channels_last=True)
            with _lib.on_device(dev):
                _lib.check(_lib.fn("ossid_maxpool_bwd_nhwc")(g.data_ptr(), idx.data_ptr(), B, H, W, C, k, s_, pd, Ho, Wo,
                                                             dx.data_ptr(), _lib.stream()), "ossid_maxpool_bwd_nhwc")
            g = dx
        else:
            _, m, x_in, s, out, (H, W) = st
            sq, e1, e3 = m.squeeze, m.expand1x1, m.expand3x3
            cin, nsq, n1, n3 = sq.in_channels, sq.out_channels, e1.out_channels, e3.out_channels
            n = B * H * W
            dv = new_buf(out.shape, dev, channels_last=True)
            sums = chan_op(flat(g), n, n1 + n3, x=flat(out), out=flat(dv), mask_mode=3, sum_mode=2)     # ReLU', both biases
            grads[e1.bias], grads[e3.bias] = sums[0, :n1], sums[0, n1:]
            wgrad_later(e1, s, flat(dv), B, H, W, nsq, n1, 1, dy_cs=n1 + n3)
            wgrad_later(e3, s, flat(dv, n1), B, H, W, nsq, n3, 9, dy_cs=n1 + n3)
            ds = new_buf(s.shape, dev, channels_last=True)
            ds3 = new_buf(s.shape, dev, channels_last=True)
            conv_raw(flat(dv), T._pack(e1.weight, "dgrad"), B, H, W, n1, nsq, 1, ds, in_cs=n1 + n3)
            conv_raw(flat(dv, n1), T._pack(e3.weight, "dgrad"), B, H, W, n3, nsq, 9, ds3, in_cs=n1 + n3)
            chan_op(flat(ds3), n, nsq, out=flat(ds), accumulate=True)                                   # ds += ds3
            sums = chan_op(flat(ds), n, nsq, x=flat(s), out=flat(ds), mask_mode=3, sum_mode=2)           # ReLU', squeeze bias
            grads[sq.bias] = sums[0]
            wgrad_later(sq, x_in, ds, B, H, W, cin, nsq, 1)
            dx = new_buf(x_in.shape, dev, channels_last=True)
            conv_raw(ds, T._pack(sq.weight, "dgrad"), B, H, W, nsq, cin, 1, dx)
            g = dx
    # ---- stem: ReLU', bias, weight gradient on the im2col columns (no data gradient: the templates are inputs)
    x0 = sv["x0"]
    _, _, H0, W0 = x0.shape
    stem = mod.backbone_0[0]
    sums = chan_op(flat(g), B * H0 * W0, 64, x=flat(x0), out=flat(g), mask_mode=3, sum_mode=2)
    grads[stem.bias] = sums[0]
    dw_cols = wgrad_later(None, cols, g, B, H0, W0, STEM_KPAD, 64, 1, key="stem")
    dw_stem = T.grad_home(stem.weight, direct)
    # everything the side stream reads or writes (the caching allocator must not hand an operand's memory to the next
    # allocation on THIS stream while the grouped launch is still reading it: outside a recorded sequence -- whose buffers
    # are persistent -- the intermediate gradients die when this function returns; recording only the outputs left
    # tests/test_dtoid_gpu.py::test_template_encoder_training_node_matches_module_path[False-*] with one layer's weight
    # gradient a few 1e-3 off once in a few full-suite runs)
    touched = [it["dw"] for it in deferred] + [dw_stem, dw_cols]
    for it in deferred:
        touched += [it["x"], it["dy"]] + (list(it["pre"]) if it.get("pre") is not None else [])

    def weight_gradients():            # one grouped launch, then the stem's from im2col column order back to [64, 4, 3, 3]
        T.wgrad_group(deferred)
        stem_relayout(dw_cols, dw_stem, 64, 4, 3, STEM_KPAD, inverse=True)
    T._wgrad_async(touched, weight_gradients, dev, side=side)
    sv_out = [dw_stem, grads[stem.bias]]
    for p in encoder_params(mod)[2:]:
        sv_out.append(grads[p])
    return sv_out


class TemplateEncoderTrain(torch.autograd.Function):
    """One SqueezeNet template encoder (TemplateFeatExtract / TemplateFeatExtractGlobal) in training mode: see the module
    docstring. forward(img [B,4,h,w], mod, *encoder_params(mod)) -> [B,640,7,7] / [B,64,3,3] channels-last."""

    @staticmethod
    def forward(ctx, img, mod, *params):
        from . import ops
        dev = img.device
        B = img.shape[0]
        plan = None
        if T.SEQ_REPLAY and not torch.cuda.is_current_stream_capturing():
            plan = T._plan_for(mod, (tuple(img.shape), str(dev), params[0].data_ptr(), params[-1].data_ptr(),
                                     mod.norm_1.running_mean.data_ptr(),
                                     T._Packed.get(params[2].detach(), T.FWD_ENCODER).data_ptr()))
        if plan is None:
            out, sv = _encoder_forward(mod, ops.im2col_stem(img, 3, 2, 0, STEM_KPAD))
        else:
            # eager, into a persistent buffer: the templates' address changes every step
            plan.t["cols"] = ops.im2col_stem(img, 3, 2, 0, STEM_KPAD, out=plan.t.get("cols"))
            if plan.fwd is None:
                seq = _lib.Seq()
                with _lib.record(seq):
                    plan.t["out"], plan.t["sv"] = _encoder_forward(mod, plan.t["cols"])
                plan.fwd = seq
            else:
                plan.fwd.run((T._cur_stream(dev),))
            out, sv = T._alias(plan.t["out"]), plan.t["sv"]
            plan.gen += 1
            ctx.gen = plan.gen
        ctx.mod, ctx.sv, ctx.plan, ctx.params = mod, sv, plan, params
        return out

    @staticmethod
    def backward(ctx, gout):
        mod, sv, plan, params = ctx.mod, ctx.sv, ctx.plan, ctx.params
        dev = gout.device
        weights = [p for p in params if p.dim() == 4]
        side = T._wgrad_side_ok(dev, weights)
        direct = all(T._grad_taken_unread(w) for w in weights)
        if plan is not None and ctx.gen != plan.gen:
            raise RuntimeError("TemplateEncoderTrain: this encoder ran another training forward since the one being "
                               "differentiated; its persistent buffers hold the later pass")
        if plan is not None and torch.cuda.is_current_stream_capturing():
            plan = None
        if plan is None:
            g = gout.float().clone(memory_format=torch.channels_last)
            grads = _encoder_backward(mod, g, sv, side, direct)
        else:
            if "gout" not in plan.t:
                plan.t["gout"] = torch.empty_like(gout, dtype=torch.float32, memory_format=torch.channels_last)
            plan.t["gout"].copy_(gout)
            if plan.bwd is None or plan.side != (side, direct):
                seq = _lib.Seq()
                with _lib.record(seq):
                    plan.t["grads"] = _encoder_backward(mod, plan.t["gout"], sv, side, direct)
                plan.bwd, plan.side = seq, (side, direct)
            else:
                T._run_seq(plan.bwd, dev)
            grads = plan.t["grads"]
        ctx.sv = None
        return (None, None) + tuple(T._alias(g) for g in grads)


def template_encoder_train(mod, img):
    """Apply a template encoder in training mode through TemplateEncoderTrain (and count the batch on its BatchNorms:
    the folded BatchNorms update running_mean / running_var in their kernel, the counters in one multi-tensor launch)."""
    out = TemplateEncoderTrain.apply(img, mod, *encoder_params(mod))
    ts = mod.__dict__.get("_folded_bn_counters")
    if ts is None:
        ts = mod.__dict__["_folded_bn_counters"] = [b.num_batches_tracked for b in mod.modules()
                                                    if isinstance(b, nn.BatchNorm2d) and b.num_batches_tracked is not None]
    if ts:
        torch._foreach_add_(ts, 1)
    return out
