"""GPU tests (pytest -m gpu) of the finetune-step kernels (csrc/train.hip) and of the autograd Functions built on them
(ossid_code_amd/dtoid/train_ops.py), each against a plain PyTorch restatement of the same op (float64 on the CPU where a
sum is long). fp32 throughout; summation orders differ (MFMA tiles / split-K slabs vs torch), tolerances are stated at
each check, relative to the largest magnitude of the expected tensor."""
import copy

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from ossid_code_amd.dtoid import backbones
from ossid_code_amd.dtoid import train_ops as T

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-12))


def cl(t):
    return t.cuda().contiguous(memory_format=torch.channels_last)


@pytest.mark.parametrize("N,C,Ct,off", [(1000, 32, 96, 32), (4097, 640, 640, 0), (77, 8, 8, 0), (50000, 128, 128, 0),
                                        (3, 256, 512, 128)])
@pytest.mark.parametrize("mask_mode,sum_mode,acc", [(0, 1, False), (1, 1, True), (2, 2, False), (0, 0, True), (1, 0, False)])
def test_chan_op_all_modes(hiplib, N, C, Ct, off, mask_mode, sum_mode, acc):
    g = torch.Generator().manual_seed(N + C + mask_mode)
    G, X, O = (torch.randn(N, Ct, generator=g) for _ in range(3))
    al, be, ka, ms, mt = (torch.randn(C, generator=g) for _ in range(5))
    gs, xs = G[:, off:off + C].double(), X[:, off:off + C].double()
    if mask_mode == 0:
        m = torch.ones_like(xs)
    elif mask_mode == 1:
        m = ((ms.float() * X[:, off:off + C] + mt.float()) > 0).double()
    else:
        m = torch.where(xs > 0, torch.ones_like(xs), xs + 1)
    r = (al.double() * gs + be.double() * xs + ka.double()) * m
    want_out = O.double().clone()
    want_out[:, off:off + C] = (want_out[:, off:off + C] + r) if acc else r
    Gd, Xd, Od = G.cuda(), X.cuda(), O.cuda()
    sums = T.chan_op(Gd.view(-1)[off:], N, C, x=Xd.view(-1)[off:], out=Od.view(-1)[off:], g_cs=Ct, x_cs=Ct, out_cs=Ct,
                     alpha=al.cuda(), beta=be.cuda(), kappa=ka.cuda(), mask_mode=mask_mode, mask_scale=ms.cuda(),
                     mask_shift=mt.cuda(), accumulate=acc, sum_mode=sum_mode)
    assert rel(Od, want_out) < 1e-5
    if sum_mode == 1:
        assert rel(sums[0], (gs * m).sum(0)) < 2e-5 and rel(sums[1], (gs * m * xs).sum(0)) < 2e-5
    elif sum_mode == 2:
        assert rel(sums[0], r.sum(0)) < 2e-5 and rel(sums[1], (r * xs).sum(0)) < 2e-5
    else:
        assert sums is None


def test_chan_op_sums_only_and_strided_sum_table(hiplib):
    x = torch.randn(5000, 64).cuda()
    table = torch.zeros(2, 200).cuda()
    T.chan_op(x, 5000, 64, x=x, sum_mode=1, sums=table.view(-1)[40:], sums_row_stride=200)
    assert rel(table[0, 40:104], x.double().sum(0)) < 1e-5 and rel(table[1, 40:104], (x.double() ** 2).sum(0)) < 1e-5
    assert float(table[:, :40].abs().sum()) == 0 and float(table[:, 104:].abs().sum()) == 0


@pytest.mark.parametrize("B,Cin,Cout,H,W,taps,in_extra,dy_extra,pre,up", [
    (8, 640, 256, 29, 39, 9, 0, 0, False, None),       # the head's correlation convs at batch 8
    (2, 128, 32, 30, 40, 9, 0, 96, True, None),        # a dense layer's 3x3 (dy = a 32-channel slice of the block buffer)
    (2, 96, 128, 15, 20, 1, 160, 0, True, None),       # a dense layer's 1x1 (x = a channel prefix of the block buffer)
    (1, 16, 16, 5, 7, 9, 0, 0, False, None),
    (2, 256, 48, 9, 13, 9, 0, 0, False, None),         # cls output layer
    (2, 256, 96, 9, 13, 9, 0, 0, True, None),          # reg output layer
    (2, 64, 32, 46, 62, 9, 0, 0, True, (23, 31)),      # decoder layer behind a 2x nearest up-sampling
    (1, 32, 16, 48, 64, 9, 0, 0, True, (23, 31)),      # ... behind a non-integer one
    (3, 512, 640, 7, 9, 1, 0, 0, True, None),
    (1, 768, 512, 29, 39, 9, 0, 0, False, None),
    (8, 128, 32, 15, 15, 1, 0, 0, False, None),        # SqueezeNet Fire squeeze convs (pixel count no multiple of 32)
    (8, 256, 48, 7, 7, 1, 0, 0, False, None),
    (8, 64, 16, 30, 30, 1, 0, 0, False, None),
    (8, 32, 128, 15, 15, 9, 0, 0, False, None),        # ... and expand convs (few input channels)
    (8, 16, 64, 30, 30, 9, 0, 0, False, None),
    (8, 48, 192, 7, 7, 1, 0, 0, False, None),
    (2, 160, 64, 24, 32, 1, 0, 0, False, None),        # the 7x7 stem as a 1x1 conv on im2col rows
    (2, 32, 16, 61, 77, 9, 0, 0, True, (29, 37)),      # csrc/wgrad_fc.hip: ragged 4 x 32 pixel tiles, non-integer up-sampling
    (3, 64, 32, 10, 70, 9, 0, 32, False, None),        # ... plain input, dy a channel slice of a wider buffer
    (1, 32, 16, 3, 5, 9, 32, 0, True, None),           # ... smaller than one tile, x a channel prefix
])
def test_wgrad_matches_torch(hiplib, B, Cin, Cout, H, W, taps, in_extra, dy_extra, pre, up):
    g = torch.Generator().manual_seed(Cin + Cout + H)
    Hs, Ws = (H, W) if up is None else up
    xs = torch.randn(B, Cin + in_extra, Hs, Ws, generator=g)
    dy = torch.randn(B, Cout + dy_extra, H, W, generator=g)
    ps, pt = torch.randn(Cin, generator=g), torch.randn(Cin, generator=g)
    xin = xs[:, :Cin].double()
    if pre:
        xin = F.relu(xin * ps.double().view(1, -1, 1, 1) + pt.double().view(1, -1, 1, 1))
    if up is not None:
        xin = F.interpolate(xin, size=(H, W), mode="nearest")
    k = 3 if taps == 9 else 1
    w = torch.zeros(Cout, Cin, k, k, dtype=torch.float64, requires_grad=True)
    F.conv2d(xin, w, padding=k // 2).backward(dy[:, dy_extra:dy_extra + Cout].double())
    xd, dyd = cl(xs), cl(dy)
    dw = torch.empty(Cout, Cin, k, k, device="cuda")
    T.wgrad_raw(xd, T.flat(dyd, dy_extra), B, H, W, Cin, Cout, taps, dw, pre=(ps.cuda(), pt.cuda()) if pre else None,
                pre_relu=pre, in_cs=Cin + in_extra, dy_cs=Cout + dy_extra, src_hw=(Hs, Ws) if up is not None else (0, 0))
    assert rel(dw, w.grad) < 5e-5
    dw2 = dw.clone()
    T.wgrad_raw(xd, T.flat(dyd, dy_extra), B, H, W, Cin, Cout, taps, dw2, pre=(ps.cuda(), pt.cuda()) if pre else None,
                pre_relu=pre, in_cs=Cin + in_extra, dy_cs=Cout + dy_extra, src_hw=(Hs, Ws) if up is not None else (0, 0),
                accumulate=True)
    assert torch.equal(dw2, dw + dw)                               # deterministic split-K: bit-reproducible, accumulates


@pytest.mark.parametrize("B,Cin,Cout,H,W,k", [(2, 64, 32, 9, 13, 3), (8, 256, 256, 29, 39, 3), (2, 224, 128, 15, 20, 1),
                                              (1, 32, 16, 40, 52, 3), (2, 256, 48, 9, 11, 3)])
def test_data_gradient_is_the_forward_kernel_on_dgrad_packed_weights(hiplib, B, Cin, Cout, H, W, k):
    g = torch.Generator().manual_seed(Cin * 3 + Cout)
    x = torch.randn(B, Cin, H, W, generator=g).double().requires_grad_(True)
    w = torch.randn(Cout, Cin, k, k, generator=g)
    dy = torch.randn(B, Cout, H, W, generator=g)
    F.conv2d(x, w.double(), padding=k // 2).backward(dy.double())
    wd = w.cuda()
    dx = T.empty_nhwc(B, Cin, H, W, "cuda")
    T.conv_raw(cl(dy), T._pack(wd, "dgrad"), B, H, W, Cout, Cin, k * k, dx)
    assert rel(dx, x.grad) < 2e-5


@pytest.mark.parametrize("B,Cin,Cout,H,W,k", [(1, 992, 128, 30, 40, 1),     # batch-1 dense layer: split-K over the waves, ragged chunk
                                              (8, 128, 32, 30, 40, 3),      # dense 3x3: one channel tile, 128-channel chunk
                                              (8, 768, 512, 29, 39, 3),     # head: four channel tiles x NT pixel tiles
                                              (2, 32, 128, 12, 16, 3),      # a dense layer's data gradient at a small size
                                              (2, 64, 48, 120, 160, 3)])    # wide image: 2-D pixel tiles
def test_convolution_forms_exact_f32_and_split_bf16_against_float64(hiplib, B, Cin, Cout, H, W, k):
    """ossid_conv_desc.exact: the same launch on v_mfma_f32_32x32x2_f32 (exact f32 products) and as three bf16 matrix-core
    products per f32 product, forward and data-gradient weight layouts, against float64. The stated bounds -- 5e-6 (f32
    accumulation over up to 6 912 terms: measured 1e-6 .. 3.3e-6) and 2e-5 of the largest output -- are what every other
    tolerance in this file builds on; the split form must also be the less exact of the two (otherwise the flag would
    select nothing)."""
    g = torch.Generator().manual_seed(Cin + 7 * Cout + k)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    want = F.conv2d(x.double(), w.double(), padding=k // 2)
    wd, xd = w.cuda(), cl(x)
    err = {}
    for kind in ("fwd_exact", "fwd_x6", "fwd"):
        out = T.empty_nhwc(B, Cout, H, W, "cuda")
        T.conv_raw(xd, T._pack(wd, kind), B, H, W, Cin, Cout, k * k, out)
        err[kind] = rel(out, want)
    # (fwd_x6: the three-way split, six bf16 products per f32 product -- held to the exact form's bound)
    assert err["fwd_exact"] < 5e-6 and err["fwd_x6"] < 5e-6 and err["fwd"] < 2e-5, err
    if hiplib.lib().ossid_conv_split_bf16():
        assert err["fwd"] > err["fwd_exact"], err
    if Cout % 16 == 0:
        dy = torch.randn(B, Cout, H, W, generator=g)
        xg = x.double().requires_grad_(True)
        F.conv2d(xg, w.double(), padding=k // 2).backward(dy.double())
        for kind, tol in (("dgrad_exact", 5e-6), ("dgrad", 2e-5)):
            dx = T.empty_nhwc(B, Cin, H, W, "cuda")
            T.conv_raw(cl(dy), T._pack(wd, kind), B, H, W, Cout, Cin, k * k, dx)
            assert rel(dx, xg.grad) < tol, (kind, rel(dx, xg.grad))


@pytest.mark.parametrize("stride,H,W", [(2, 120, 160), (1, 30, 40), (2, 7, 9)])
def test_avgpool2_forward_backward(hiplib, stride, H, W):
    x = torch.randn(2, 16, H, W).cuda().requires_grad_(True)
    xr = x.detach().clone().requires_grad_(True)
    y, yr = T.AvgPool2.apply(x, stride), F.avg_pool2d(xr, 2, stride)
    go = torch.randn_like(yr)
    y.backward(go)
    yr.backward(go)
    assert rel(y, yr) < 1e-6 and rel(x.grad, xr.grad) < 1e-6


@pytest.mark.parametrize("Hs,Ws,H,W", [(29, 39, 58, 78), (232, 312, 480, 640), (5, 7, 5, 7), (3, 4, 10, 9)])
def test_upsample_backward_is_the_window_sum(hiplib, Hs, Ws, H, W):
    """The exact adjoint of the forward's index map min(floor(dst * in/out), in-1) = what torch's CPU autograd computes
    (and what the reference-generated golden files pin). torch's GPU backward kernel derives its windows from a second,
    ceil-based formula that disagrees with its own forward where src * out/in is an integer in exact arithmetic but not
    in float32 (312 -> 640: 7 of 312 columns), so the comparison is against the CPU."""
    x = torch.randn(2, 8, Hs, Ws).requires_grad_(True)
    go = torch.randn(2, 8, H, W)
    F.interpolate(x, size=(H, W), mode="nearest").backward(go)
    got = T.upsample_bwd(cl(go), 2, Hs, Ws, H, W, 8)
    assert rel(got, x.grad) < 1e-5


def _copy_grads(mods):
    return [p.grad.detach().clone() for m in mods for p in m.parameters()]


def test_conv_elu_batchnorm_conv_chain_matches_torch_autograd(hiplib):
    """`conv -> ELU -> BatchNorm(train) -> [nearest up-sample] -> conv -> ELU` (the decoder pattern, network.py:350-357),
    BatchNorm folded into the second conv's input staging: outputs, every parameter gradient, the input gradient and the
    running statistics against nn modules."""
    torch.manual_seed(0)
    c1, bn, c2 = torch.nn.Conv2d(32, 64, 3, padding=1).cuda(), torch.nn.BatchNorm2d(64).cuda(), torch.nn.Conv2d(64, 16, 3, padding=1).cuda()
    with torch.no_grad():
        bn.weight.normal_(1, 0.2)
        bn.bias.normal_(0, 0.2)
    x = torch.randn(3, 32, 11, 14, device="cuda")
    go = torch.randn(3, 16, 22, 28, device="cuda")
    import copy
    r1, rbn, r2 = copy.deepcopy(c1), copy.deepcopy(bn), copy.deepcopy(c2)
    xr = x.clone().requires_grad_(True)
    yr = F.elu(r2(F.interpolate(rbn(F.elu(r1(xr))), scale_factor=2, mode="nearest")))
    yr.backward(go)
    xm = x.clone().requires_grad_(True)
    u, sums = T.fused_conv(xm, c1, act_elu=True, want_stats=True)
    scale, shift = T.bn_fold(sums, 3 * 11 * 14, bn)
    y = T.fused_conv(u, c2, pre=(scale, shift), act_elu=True, size=(22, 28))
    y.backward(go)
    assert rel(y, yr) < 2e-5
    assert rel(xm.grad, xr.grad) < 2e-4
    for a, b in zip(_copy_grads((c1, bn, c2)), _copy_grads((r1, rbn, r2))):
        assert rel(a, b) < 2e-4
    assert rel(bn.running_mean, rbn.running_mean) < 1e-5 and rel(bn.running_var, rbn.running_var) < 1e-5


def test_bn_relu_conv_matches_torch_autograd(hiplib):
    """DenseNet's BN-ReLU-Conv (transition / norm5 pattern) through ColStats + BNFold + the conv's fused prologue."""
    torch.manual_seed(1)
    bn, conv = torch.nn.BatchNorm2d(96).cuda(), torch.nn.Conv2d(96, 48, 1, bias=False).cuda()
    import copy
    rbn, rconv = copy.deepcopy(bn), copy.deepcopy(conv)
    x = (torch.randn(4, 96, 9, 12, device="cuda") * 2 + 0.5)
    go = torch.randn(4, 48, 4, 6, device="cuda")
    xr = x.clone().requires_grad_(True)
    F.avg_pool2d(rconv(F.relu(rbn(xr))), 2, 2).backward(go)
    xm = x.clone().requires_grad_(True)
    T.AvgPool2.apply(T.bn_relu_conv(xm, bn, conv), 2).backward(go)
    assert rel(xm.grad, xr.grad) < 2e-4
    for a, b in zip(_copy_grads((bn, conv)), _copy_grads((rbn, rconv))):
        assert rel(a, b) < 2e-4
    assert rel(bn.running_var, rbn.running_var) < 1e-5


@pytest.mark.parametrize("replay", [False, True])
@pytest.mark.parametrize("L,C0,B,H,W", [(3, 64, 2, 12, 16), (6, 64, 2, 30, 40), (4, 256, 1, 7, 9)])
def test_dense_block_training_path_matches_module_path(hiplib, L, C0, B, H, W, replay, monkeypatch):
    """One resident buffer + shared batch statistics + in-place gradient accumulation vs the nn.Module dense block
    (torch.cat, one BatchNorm per layer over the whole concatenation): output, input gradient, every parameter gradient,
    every running statistic -- three rounds with fresh data each. With `replay` round 0 records the block's launch
    sequences (train_ops.SEQ_REPLAY) and rounds 1, 2 replay them from the persistent buffers: a launch missing from the
    recording, or a torch kernel inside it, would leave round 1 with round 0's values."""
    monkeypatch.setattr(T, "SEQ_REPLAY", replay)
    torch.manual_seed(2)
    blk = backbones.DenseBlock(L, C0).cuda().train()
    with torch.no_grad():
        for m in blk.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.normal_(1, 0.2)
                m.bias.normal_(0, 0.2)
    import copy
    ref = copy.deepcopy(blk)
    ref64 = copy.deepcopy(blk).double().cpu()            # the anchor: the same module in float64 on the CPU

    def module_path(mod, xx, g):
        xr = xx.clone().requires_grad_(True)
        feats = [xr]
        for layer in mod.values():
            feats.append(layer(torch.cat(feats, 1)))
        yr = torch.cat(feats, 1)
        yr.backward(g)
        return yr, xr.grad

    def l2(a, b):
        a, b = a.detach().double().cpu(), b.detach().double().cpu()
        return float((a - b).norm() / b.norm().clamp(min=1e-30))
    for rnd in range(3):
        x = torch.randn(B, C0, H, W, device="cuda") * (1 + rnd)
        go = torch.randn(B, C0 + 32 * L, H, W, device="cuda")
        for mod in (blk, ref, ref64):
            for p in mod.parameters():
                p.grad = None
        yr, xr_grad = module_path(ref, x, go)
        y64, x64_grad = module_path(ref64, x.double().cpu(), go.double().cpu())
        xm = x.clone().requires_grad_(True)
        y = T.dense_block_train(xm, blk)
        y.backward(go)
        assert rel(y, yr) < 5e-5, rnd
        # against float64: this path (three-way-split forward: f32-level) must be as close as torch's own f32 path is, within 3x
        # (max-norm for the output; the gradients in L2 -- a single ReLU decision flipped between two float32 paths moves single
        # elements by O(1) in either path -- with the floor at what such a flip costs; what tools/debug_dense_sb.py used to print)
        assert rel(y, y64) < max(2e-6, 3 * rel(yr, y64)), (rnd, rel(y, y64), rel(yr, y64))
        # (floors = the split-bf16 allowance of the data / weight gradients, 5-7e-6 per product through L layers and sums over
        # thousands of pixels with cancellation: measured 1e-3 on the input gradient and 2.1e-3 on a BatchNorm weight where
        # torch's f32 path sits at 6e-7 of float64; the all-exact build holds 5e-4 / 1e-3, profiles/r03_split_vs_exact.txt)
        assert l2(xm.grad, x64_grad) < max(2e-3, 3 * l2(xr_grad, x64_grad)), (rnd, l2(xm.grad, x64_grad), l2(xr_grad, x64_grad))
        assert l2(xm.grad, xr_grad) < 2e-3, rnd
        for (n, p), q, q64 in zip(blk.named_parameters(), ref.parameters(), ref64.parameters()):
            assert l2(p.grad, q.grad) < 5e-3, (rnd, n)
            assert l2(p.grad, q64.grad) < max(5e-3, 3 * l2(q.grad, q64.grad)), (rnd, n, l2(p.grad, q64.grad), l2(q.grad, q64.grad))
        for (n, b), q, q64 in zip(blk.named_buffers(), ref.buffers(), ref64.buffers()):
            if b.dtype.is_floating_point:
                assert rel(b, q) < 1e-4 and rel(b, q64) < 1e-4, (rnd, n)
    plans = blk.__dict__.get("_train_plans", {})
    if replay:
        (plan,) = plans.values()
        # recorded launches per layer -- forward: fold, 1x1 with its statistics, fold, 3x3, slab statistics; backward: own-slab
        # pass, 3x3 data gradient with its mask pass fused, fold, 1x1 data gradient with the accumulation fused, fold
        assert len(plan.fwd) >= 5 * L and len(plan.bwd) >= 5 * L and plan.gen == 3
        # the persistent buffers belong to the LAST forward: differentiating an older one is refused
        x1 = torch.randn(B, C0, H, W, device="cuda", requires_grad=True)
        y1 = T.dense_block_train(x1, blk)
        y2 = T.dense_block_train(torch.randn(B, C0, H, W, device="cuda", requires_grad=True), blk)
        with pytest.raises(RuntimeError, match="another training forward"):
            y1.sum().backward()
        y2.sum().backward()
    else:
        assert not plans


def test_stem_training_pieces_match_torch_autograd(hiplib):
    """x + conv2d_dw_group(x, k) (both gradients), channels-last max-pool with argmax indices (3/2/1 and 3/2/0 ceil) and
    the materialised training BatchNorm+ReLU against torch autograd."""
    from oracle import dtoid_oracle
    torch.manual_seed(5)
    x = torch.randn(2, 16, 21, 27, device="cuda")
    k = torch.randn(2, 16, 3, 3, device="cuda") * 0.3
    go = torch.randn(2, 16, 21, 27, device="cuda")
    xr, kr = x.clone().requires_grad_(True), k.clone().requires_grad_(True)
    (xr + dtoid_oracle.dw_xcorr(xr, kr)).backward(go)
    xm, km = x.clone().requires_grad_(True), k.clone().requires_grad_(True)
    y = T.DwXcorrAdd.apply(xm, km)
    y.backward(go)
    assert rel(y, x + dtoid_oracle.dw_xcorr(x, k)) < 1e-5 and rel(xm.grad, xr.grad) < 1e-5 and rel(km.grad, kr.grad) < 1e-4
    for kk, st, pd, ceil in ((3, 2, 1, False), (3, 2, 0, True)):
        xr = x.clone().requires_grad_(True)
        yr = F.max_pool2d(xr, kk, st, pd, ceil_mode=ceil)
        g2 = torch.randn_like(yr)
        yr.backward(g2)
        xm = x.clone().requires_grad_(True)
        ym = T.MaxPoolNHWC.apply(xm, kk, st, pd, ceil)
        ym.backward(g2)
        assert torch.equal(ym, yr) and rel(xm.grad, xr.grad) < 1e-6
    bn = torch.nn.BatchNorm2d(16).cuda().train()
    with torch.no_grad():
        bn.weight.normal_(1, 0.3)
        bn.bias.normal_(0, 0.3)
    import copy
    rbn = copy.deepcopy(bn)
    xr = x.clone().requires_grad_(True)
    F.relu(rbn(xr)).backward(go)
    xm = x.clone().requires_grad_(True)
    T.bn_act_train(xm, bn, relu=True).backward(go)
    assert rel(xm.grad, xr.grad) < 2e-4 and rel(bn.weight.grad, rbn.weight.grad) < 2e-4 and rel(bn.bias.grad, rbn.bias.grad) < 2e-4
    assert rel(bn.running_var, rbn.running_var) < 1e-5


@pytest.mark.parametrize("cin,sq,e,hw", [(64, 16, 64, 30), (128, 16, 64, 30), (128, 32, 128, 15), (256, 32, 128, 15), (256, 48, 192, 7),
                                         (384, 48, 192, 7), (384, 64, 256, 7), (512, 64, 256, 7)])
def test_fire_module_on_fused_convs_matches_torch_autograd(hiplib, cin, sq, e, hw):
    """One SqueezeNet Fire module (squeeze 1x1 -> ReLU -> expand 1x1 | 3x3 -> ReLU -> cat) at every configuration the
    template encoders use, through FusedConv with ReLU epilogues: output, input gradient, all six parameter gradients."""
    import copy
    torch.manual_seed(cin + sq)
    fire = backbones.Fire(cin, sq, e, e).cuda()
    ref = copy.deepcopy(fire)
    x = torch.randn(8, cin, hw, hw, device="cuda")
    go = torch.randn(8, 2 * e, hw, hw, device="cuda")
    xr = x.clone().requires_grad_(True)
    yr = ref(xr)
    yr.backward(go)
    xm = x.clone().requires_grad_(True)
    s = T.fused_conv(xm, fire.squeeze, act=2)
    y = torch.cat([T.fused_conv(s, fire.expand1x1, act=2), T.fused_conv(s, fire.expand3x3, act=2)], 1)
    y.backward(go)
    assert rel(y, yr) < 2e-5 and rel(xm.grad, xr.grad) < 2e-4
    for (n, p), q in zip(fire.named_parameters(), ref.parameters()):
        assert rel(p.grad, q.grad) < 5e-4, (n, rel(p.grad, q.grad))


def test_batch_statistics_survive_a_large_mean_with_a_small_spread(hiplib):
    """A channel at 100 +- 0.01 (or 0.5 +- 1e-3, a nearly dead ReLU channel): E[x^2] - E[x]^2 in float32 would lose the
    variance entirely; the pivoted sums of ossid_chan_op (sum_mode 3) + ossid_bn_fold_fwd keep it. Forward output and the
    input gradient of a training BatchNorm against float64."""
    import copy
    torch.manual_seed(3)
    x = torch.randn(4, 8, 30, 40, device="cuda")
    x[:, 0] = 100.0 + 0.01 * x[:, 0]
    x[:, 1] = 0.5 + 1e-3 * x[:, 1]
    x[:, 2] = -3000.0 + x[:, 2]
    bn = torch.nn.BatchNorm2d(8).cuda().train()
    r64 = copy.deepcopy(bn).double().cpu()
    a = x.clone().requires_grad_(True)
    y = T.bn_act_train(a, bn)
    c = x.double().cpu().requires_grad_(True)
    y64 = r64(c)
    go = torch.randn_like(y64)
    y.backward(go.float().cuda())
    y64.backward(go)
    assert rel(y, y64) < 2e-3 and rel(bn.running_var, r64.running_var) < 1e-3      # (the inputs themselves carry ~1e-7 * 100 / 0.01)
    assert rel(bn.weight.grad, r64.weight.grad) < 5e-3


def test_side_stream_probe_returns_streams_that_run_beside_the_main_one(hiplib):
    """train_ops.side_streams: three distinct streams per device, chosen once; at least the weight-gradient stream must
    really run beside the main stream (a spin kernel on the main stream does not delay a kernel on it)."""
    dev = torch.device("cuda", torch.cuda.current_device())
    T._side_pools.clear()
    pool = T.side_streams(dev)
    assert len({pool["wgrad"].cuda_stream, pool["b0"].cuda_stream, pool["b1"].cuda_stream}) == 3
    assert pool["concurrent"] >= 1 and T.side_streams(dev) is pool
    main = torch.cuda.current_stream()
    small = torch.zeros(8, device="cuda")
    torch.cuda.synchronize()
    ev_m, ev_s = torch.cuda.Event(), torch.cuda.Event()
    torch.cuda._sleep(4000000)
    ev_m.record(main)
    with torch.cuda.stream(pool["wgrad"]):
        small.add_(1)
        ev_s.record(pool["wgrad"])
    ev_s.synchronize()
    assert not ev_m.query()
    torch.cuda.synchronize()


@pytest.mark.parametrize("B,C,H,W", [(2, 16, 37, 45), (1, 16, 480, 640), (3, 8, 5, 7), (2, 32, 9, 11), (1, 4, 1, 3)])
def test_one_output_channel_conv3x3_matches_torch(hiplib, B, C, H, W):
    """nn.Conv2d(C, 1, 3, padding=1) in training (the decoder's seg_final 16 -> 1, network.py:362) on ossid_conv3x3_c1_*:
    output, input gradient, weight and bias gradient against torch autograd in float64 (the kernels sum in a fixed order:
    two runs are bit-identical)."""
    torch.manual_seed(B * 1000 + C)
    conv = torch.nn.Conv2d(C, 1, 3, padding=1).cuda()
    x = cl(torch.randn(B, C, H, W))
    go = torch.randn(B, 1, H, W, device="cuda")
    xr = x.double().detach().requires_grad_(True)
    ref = copy.deepcopy(conv).double()
    yr = ref(xr)
    yr.backward(go.double())
    outs = []
    for _ in range(2):
        conv.weight.grad = conv.bias.grad = None
        xm = x.clone().requires_grad_(True)
        y = T.conv3x3_c1(xm, conv)
        y.backward(go)
        torch.cuda.synchronize()
        outs.append((y.detach().clone(), xm.grad.clone(), conv.weight.grad.clone(), conv.bias.grad.clone()))
    y, dx, dw, db = outs[0]
    assert y.shape == (B, 1, H, W) and dw.shape == conv.weight.shape and db.shape == (1,)
    assert rel(y, yr) < 2e-6 and rel(dx, xr.grad) < 2e-6
    assert rel(dw, ref.weight.grad) < 1e-5 and rel(db, ref.bias.grad) < 1e-5
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)


def test_weight_gradient_of_a_non_leaf_weight_is_complete_when_backward_returns_it(hiplib):
    """FusedConv handed a NON-LEAF weight (a re-laid / scaled view of a parameter: `.grad` is always None there,
    which the side-stream guard used to read as "AccumulateGrad will take it unread"). The next autograd node (here the
    multiplication's backward) reads dw on the caller's stream straight away, so the launch must stay in line. With the
    weight-gradient stream kept busy by a long kernel, a side-stream launch would hand over an unwritten buffer."""
    torch.manual_seed(5)
    dev = torch.device("cuda")
    x = cl(torch.randn(2, 32, 20, 24))
    w = torch.nn.Parameter(torch.randn(64, 32, 3, 3, device=dev) * 0.1)
    go = cl(torch.randn(2, 64, 20, 24))
    scale = torch.full((64, 1, 1, 1), 1.5, device=dev)
    assert T.WGRAD_SIDE
    side = T.side_streams(dev)["wgrad"]
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        torch.cuda._sleep(int(4e8))                      # ~0.2 s: anything queued behind it is late
    wn = w * scale                                      # non-leaf
    assert not wn.is_leaf and not T._grad_taken_unread(wn) and T._grad_taken_unread(w)
    y = T.FusedConv.apply(x, wn, None, None, None, False, 0, None, False)
    y.backward(go)
    got = w.grad.clone()                                # read on the main stream, as an optimizer would
    torch.cuda.synchronize()
    w64 = w.detach().double().cpu().requires_grad_(True)
    y64 = F.conv2d(x.double().cpu(), w64 * scale.double().cpu(), padding=1)
    y64.backward(go.double().cpu())
    assert rel(got, w64.grad) < 5e-5


@pytest.mark.parametrize("B,H,W,normalize", [(2, 480, 640, False), (1, 480, 640, True), (3, 53, 77, False), (1, 16, 40, True),
                                             (2, 7, 9, False)])
def test_stem_conv7x7s2_implicit_im2col_forward_and_weight_gradient(hiplib, B, H, W, normalize):
    """csrc/stem.hip: DenseNet conv0 (3 -> 64, 7x7, stride 2, padding 3; network.py:164-170) as an implicit-im2col MFMA
    kernel on the NCHW image, and its weight gradient, against float64 conv2d / autograd on the CPU. Exact-f32 products
    (an fmaf chain per output): 2e-6 of the output scale; the weight gradient sums up to 600 k products per element in a
    fixed order: 2e-5. Ragged sizes exercise the tile masks; `normalize` = normalizeImageRange fused into the staging
    (test time)."""
    from ossid_code_amd.dtoid import ops
    g = torch.Generator().manual_seed(B * 1000 + H + W)
    conv = torch.nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(64, 3, 7, 7, generator=g) * 0.1)
    img = torch.rand(B, 3, H, W, generator=g)
    ref_in = img.double()
    if normalize:
        mean = torch.tensor([0.485, 0.456, 0.406], dtype=torch.float64)[None, :, None, None]
        std = torch.tensor([0.229, 0.224, 0.225], dtype=torch.float64)[None, :, None, None]
        ref_in = (ref_in - mean) / std
    w64 = conv.weight.detach().double().requires_grad_(True)
    want = F.conv2d(ref_in, w64, stride=2, padding=3)
    convg = copy.deepcopy(conv).cuda()
    got = ops.stem_conv(img.cuda(), convg, normalize=normalize)
    assert got.shape == want.shape and got.is_contiguous(memory_format=torch.channels_last)
    assert rel(got, want) < (2e-5 if normalize else 2e-6)          # (normalised: (x - m) * (1/s) in f32 vs (x - m) / s in f64)
    if not normalize:
        go = torch.randn(want.shape, generator=g)
        want.backward(go.double())
        y = T.stem_conv(img.cuda(), convg)
        assert torch.equal(y, got)
        y.backward(cl(go))
        torch.cuda.synchronize()
        assert rel(convg.weight.grad, w64.grad) < 2e-5
        # bit-reproducible: fixed-order sums
        g1 = convg.weight.grad.clone()
        convg.weight.grad = None
        T.stem_conv(img.cuda(), convg).backward(cl(go))
        torch.cuda.synchronize()
        assert torch.equal(convg.weight.grad, g1)


@pytest.mark.parametrize("B,C,H,W,kb", [(2, 64, 30, 44, 2), (3, 16, 21, 27, 1), (1, 64, 240, 320, 1), (2, 8, 5, 6, 2)])
def test_stem_tail_fused_modulation_batchnorm_relu_pool_matches_torch_autograd(hiplib, B, C, H, W, kb):
    """train_ops.StemTail (csrc/stem.hip: modulation with the batch statistics in the same pass, pool of relu(norm0) without
    the normalised tensor, the two-pass backward) against torch autograd of
    max_pool2d(relu(batch_norm(x + conv2d_dw_group(x, k))), 3, 2, 1) in float64 on the CPU: output, gradients with respect to
    x, k, gamma, beta, and the running statistics. A pre-activation within rounding of zero may take the other side of the
    ReLU in f32: compared in relative L2 with a handful of flips allowed for."""
    from oracle import dtoid_oracle
    g = torch.Generator().manual_seed(B * 100 + C + H)
    x = torch.randn(B, C, H, W, generator=g)
    k = torch.randn(kb, C, 3, 3, generator=g) * 0.3
    bn = torch.nn.BatchNorm2d(C).train()
    with torch.no_grad():
        bn.weight.copy_(1 + 0.3 * torch.randn(C, generator=g))
        bn.weight[0] = -0.7                                   # a negative gamma: relu(affine) is then decreasing in m
        bn.bias.copy_(0.3 * torch.randn(C, generator=g))
    rbn = copy.deepcopy(bn).double()
    xr, kr = x.double().requires_grad_(True), k.double().requires_grad_(True)
    yr = F.max_pool2d(F.relu(rbn(xr + dtoid_oracle.dw_xcorr(xr, kr.expand(B, -1, -1, -1)))), 3, 2, 1)
    go = torch.randn(yr.shape, generator=g)
    yr.backward(go.double())
    bn = bn.cuda()
    xm, km = cl(x).requires_grad_(True), k.cuda().requires_grad_(True)
    y = T.stem_tail(xm, km, bn)
    y.backward(cl(go))
    torch.cuda.synchronize()

    def l2(a, b):
        a, b = a.detach().double().cpu(), b.detach().double().cpu()
        return float((a - b).norm() / b.norm().clamp(min=1e-30))
    assert y.shape == yr.shape and rel(y, yr) < 1e-5
    assert l2(xm.grad, xr.grad) < 1e-4, l2(xm.grad, xr.grad)
    assert l2(km.grad, kr.grad) < 1e-4 and l2(bn.weight.grad, rbn.weight.grad) < 1e-4 and l2(bn.bias.grad, rbn.bias.grad) < 1e-4
    assert rel(bn.running_mean, rbn.running_mean) < 1e-5 and rel(bn.running_var, rbn.running_var) < 1e-5


@pytest.mark.parametrize("B,H,W,L", [(2, 30, 40, 3), (1, 7, 9, 2), (8, 29, 39, 4)])
def test_grouped_weight_gradients_of_a_dense_block_match_torch(hiplib, B, H, W, L):
    """ossid_conv_wgrad_group on the 2 L problems of a dense block as DenseBlockTrain hands them over: the 3x3 layers (128 -> 32,
    dy a 32-channel slice of the block's gradient buffer, BatchNorm + ReLU prologue) and the 1x1 layers (channel prefix of the
    block buffer -> 128) in one call -- two tiling variants, one grouped launch each -- against float64 autograd; and
    bit-reproducible."""
    g = torch.Generator().manual_seed(B * 100 + H)
    C0, Ct = 64, 64 + 32 * L
    buf = torch.randn(B, Ct, H, W, generator=g)
    G = torch.randn(B, Ct, H, W, generator=g)
    items, want = [], []
    bufd, Gd = cl(buf), cl(G)
    for li in range(L):
        c = C0 + 32 * li
        y1 = torch.randn(B, 128, H, W, generator=g)
        dz = torch.randn(B, 128, H, W, generator=g)
        ps2, pt2 = torch.randn(128, generator=g), torch.randn(128, generator=g)
        ps1, pt1 = torch.randn(c, generator=g), torch.randn(c, generator=g)
        # 3x3: x = relu(y1 * ps2 + pt2), dy = G[:, c:c+32]
        w2 = torch.zeros(32, 128, 3, 3, dtype=torch.float64, requires_grad=True)
        F.conv2d(F.relu(y1.double() * ps2.double().view(1, -1, 1, 1) + pt2.double().view(1, -1, 1, 1)), w2, padding=1).backward(
            G[:, c:c + 32].double())
        # 1x1: x = relu(buf[:, :c] * ps1 + pt1), dy = dz
        w1 = torch.zeros(128, c, 1, 1, dtype=torch.float64, requires_grad=True)
        F.conv2d(F.relu(buf[:, :c].double() * ps1.double().view(1, -1, 1, 1) + pt1.double().view(1, -1, 1, 1)), w1).backward(dz.double())
        y1d, dzd = cl(y1), cl(dz)
        dw2, dw1 = torch.empty(32, 128, 3, 3, device="cuda"), torch.empty(128, c, 1, 1, device="cuda")
        items.append(dict(x=y1d, dy=T.flat(Gd, c), B=B, H=H, W=W, cin=128, cout=32, taps=9, dw=dw2, pre=(ps2.cuda(), pt2.cuda()),
                          pre_relu=True, dy_cs=Ct))
        items.append(dict(x=bufd, dy=dzd, B=B, H=H, W=W, cin=c, cout=128, taps=1, dw=dw1, pre=(ps1.cuda(), pt1.cuda()),
                          pre_relu=True, in_cs=Ct))
        want += [w2.grad, w1.grad]
    T.wgrad_group(items)
    torch.cuda.synchronize()
    first = [it["dw"].clone() for it in items]
    for it, w in zip(items, want):
        assert rel(it["dw"], w) < 5e-5, (it["cin"], it["cout"], rel(it["dw"], w))
    T.wgrad_group(items)
    torch.cuda.synchronize()
    for it, f in zip(items, first):
        assert torch.equal(it["dw"], f)


def test_c_side_replay_of_a_dense_block_equals_the_python_loop_bit_for_bit(hiplib, monkeypatch):
    """ossid_seq_replay (csrc/seq.hip) re-issues a dense block's recorded forward and backward launches -- descriptor pointers,
    integers beyond the sixth on the stack, floats and doubles in xmm registers, the stream-order ops between the main and the
    weight-gradient stream as library-owned events -- exactly as the Python loop over ctypes calls does: two blocks with the same
    weights, one replayed by each, must agree in every bit of the output, the input gradient and every parameter gradient."""
    import copy
    from ossid_code_amd import _lib
    monkeypatch.setattr(T, "SEQ_REPLAY", True)
    torch.manual_seed(5)
    blk_c = backbones.DenseBlock(4, 64).cuda().train()
    blk_p = copy.deepcopy(blk_c)
    res = {}
    for name, blk, use_c in (("c", blk_c, True), ("py", blk_p, False)):
        monkeypatch.setattr(_lib, "SEQ_C", use_c)
        outs = []
        for rnd in range(3):                      # round 0 records, rounds 1 and 2 replay
            g = torch.Generator(device="cuda").manual_seed(10 + rnd)
            x = torch.randn(2, 64, 14, 18, device="cuda", generator=g).requires_grad_(True)
            go = torch.randn(2, 64 + 32 * 4, 14, 18, device="cuda", generator=g)
            for p in blk.parameters():
                p.grad = None
            y = T.dense_block_train(x, blk)
            y.backward(go)
            T.join_wgrad_stream()
            torch.cuda.synchronize()
            outs.append([y.detach().clone(), x.grad.clone()] + [p.grad.clone() for p in blk.parameters()] +
                        [b.clone() for b in blk.buffers() if b.dtype.is_floating_point])
        res[name] = outs
        (plan,) = blk.__dict__["_train_plans"].values()
        assert (plan.fwd._compiled is not None) == use_c and (plan.bwd._compiled is not None) == use_c
    for rnd in range(3):
        for a, b in zip(res["c"][rnd], res["py"][rnd]):
            assert torch.equal(a, b), rnd
