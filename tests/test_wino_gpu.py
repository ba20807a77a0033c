"""Winograd F(2x2,3x3) convolution (csrc/wino.hip) against torch's conv2d in float64 and against the direct kernel:
the layers it stands behind are the 3x3 convolutions of DTOID's head (network.py:102-110, :135-143, :288-326)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def T():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from ossid_code_amd.dtoid import train_ops
    return train_ops


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-12))


def nhwc(t):
    return t.contiguous(memory_format=torch.channels_last)


@pytest.mark.parametrize("B,cin,cout,H,W", [(2, 16, 32, 4, 4), (3, 64, 48, 7, 9), (1, 32, 64, 2, 2), (2, 48, 100, 1, 5),
                                            (8, 128, 32, 30, 40), (4, 640, 256, 29, 39), (2, 256, 96, 29, 39), (1, 32, 16, 61, 33)])
def test_wino_forward_matches_conv2d(T, B, cin, cout, H, W):
    g = torch.Generator().manual_seed(B * 1000 + cin + H)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5
    bias = torch.randn(cout, generator=g)
    ref = F.conv2d(x.double(), w.double(), bias.double(), padding=1)
    xd, wd = nhwc(x.cuda()), w.cuda()
    out = T.empty_nhwc(B, cout, H, W, "cuda")
    T.conv_raw(xd, T._pack(wd, "wino_fwd"), B, H, W, cin, cout, 9, out, bias=bias.cuda(), wino=True)
    assert rel(out, ref) < 2e-5
    direct = T.empty_nhwc(B, cout, H, W, "cuda")
    T.conv_raw(xd, T._pack(wd, "fwd"), B, H, W, cin, cout, 9, direct, bias=bias.cuda())
    assert rel(out, direct) < 2e-5 and rel(direct, ref) < 2e-5


@pytest.mark.parametrize("B,cin,cout,H,W", [(21, 256, 256, 29, 39),    # 197 tile groups x 2 channel groups = 400 workgroups on
                                                                        # 256 slots: the tail (144) is cut along the reduction
                                            (21, 64, 512, 29, 39),     # four channel groups, reduction too short to cut
                                            (16, 48, 160, 32, 32)])    # five channel tiles: the second group is one tile wide
def test_wino_128_channel_workgroups_match_conv2d(T, B, cin, cout, H, W):
    """Layers with >= 128 output channels and >= 128 tile groups run four channel tiles per (512-thread) workgroup -- the
    test-time head at 21 templates; incl. the tail split with its finishing launch, and against the 64-channel form
    (OSSID_WINO_CT is read once per process, so that comparison is against the direct kernel and float64 only)."""
    g = torch.Generator().manual_seed(B + cin + cout)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5
    bias = torch.randn(cout, generator=g)
    ps, pt = torch.rand(cin, generator=g) + 0.5, torch.randn(cin, generator=g) * 0.1
    ref = F.conv2d(F.relu(x.double() * ps.double()[None, :, None, None] + pt.double()[None, :, None, None]), w.double(),
                   bias.double(), padding=1)
    ref = F.elu(ref)
    xd, wd = nhwc(x.cuda()), w.cuda()
    out = T.empty_nhwc(B, cout, H, W, "cuda")
    T.conv_raw(xd, T._pack(wd, "wino_fwd"), B, H, W, cin, cout, 9, out, bias=bias.cuda(), pre=(ps.cuda(), pt.cuda()),
               pre_relu=True, act=1, wino=True)
    assert rel(out, ref) < 2e-5
    again = T.empty_nhwc(B, cout, H, W, "cuda")
    T.conv_raw(xd, T._pack(wd, "wino_fwd"), B, H, W, cin, cout, 9, again, bias=bias.cuda(), pre=(ps.cuda(), pt.cuda()),
               pre_relu=True, act=1, wino=True)
    assert torch.equal(out, again)                                  # fixed-order sums in the finishing launch


def test_wino_pair_launch_at_128_channel_workgroups(T):
    """The pair entry with four channel tiles per workgroup (the classification / regression trunks at 21 templates) against
    two single launches. Not bit-equal here: the combined grid's tail -- the workgroups whose reduction is cut into slices and
    summed by the finishing launch -- is a different set of workgroups than each single launch's, so some outputs are summed
    in a different (still fixed) order: equal to f32 reordering, 2e-6, and both within the kernel's bound of float64."""
    import ctypes
    from ossid_code_amd import _lib
    from ossid_code_amd.dtoid import ops
    g = torch.Generator().manual_seed(33)
    B, cin, cout, H, W = 21, 128, 256, 29, 39
    descs, outs, keep = [], [], []
    for _ in range(2):
        x = nhwc(torch.randn(B, cin, H, W, generator=g).cuda())
        w = (torch.randn(cout, cin, 3, 3, generator=g) * 0.05).cuda()
        bias = torch.randn(cout, generator=g).cuda()
        single = T.empty_nhwc(B, cout, H, W, "cuda")
        T.conv_raw(x, T._pack(w, "wino_fwd"), B, H, W, cin, cout, 9, single, bias=bias, act=1, wino=True)
        out = T.empty_nhwc(B, cout, H, W, "cuda")
        d = _lib.ConvDesc()
        d.x, d.wpk, d.bias, d.out = x.data_ptr(), T._pack(w, "wino_fwd").data_ptr(), bias.data_ptr(), out.data_ptr()
        d.in_batch_stride, d.batch, d.height, d.width, d.cin, d.cout, d.taps, d.act = -1, B, H, W, cin, cout, 9, 1
        descs.append(d)
        ref = F.elu(F.conv2d(x.double(), w.double(), bias.double(), padding=1))
        outs.append((single, out, ref))
        keep += [x, w, bias]
    ops.wino_workspace(descs, "cuda:0")
    assert _lib.fn("ossid_conv3x3_wino_fwd_pair")(ctypes.byref(descs[0]), ctypes.byref(descs[1]), _lib.stream()) == 0
    torch.cuda.synchronize()
    for single, out, ref in outs:
        assert rel(out, single) < 2e-6 and rel(out, ref) < 2e-5 and rel(single, ref) < 2e-5


@pytest.mark.parametrize("act", [0, 1, 2])
def test_wino_prologue_epilogue_strides(T, act):
    """Input affine + ReLU on real pixels only, bias -> activation -> output affine, channel strides and an output
    offset (a dense block's resident buffer), as ossid_conv_nhwc_fwd."""
    g = torch.Generator().manual_seed(5 + act)
    B, cin, cout, H, W, cs_in, cs_out, off = 3, 32, 40, 9, 11, 48, 64, 8
    xfull = torch.randn(B, cs_in, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) * 0.1
    bias, ps, pt = torch.randn(cout, generator=g), torch.rand(cin, generator=g) + 0.5, torch.randn(cin, generator=g) * 0.3
    qs, qt = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g)
    xin = torch.relu(xfull[:, :cin].double() * ps.double().view(1, -1, 1, 1) + pt.double().view(1, -1, 1, 1))
    y = F.conv2d(xin, w.double(), bias.double(), padding=1)
    y = F.elu(y) if act == 1 else (torch.relu(y) if act == 2 else y)
    ref = y * qs.double().view(1, -1, 1, 1) + qt.double().view(1, -1, 1, 1)
    out = torch.full((B, cs_out, H, W), 7.0).cuda().contiguous(memory_format=torch.channels_last)
    T.conv_raw(nhwc(xfull.cuda()), T._pack(w.cuda(), "wino_fwd"), B, H, W, cin, cout, 9, out, bias=bias.cuda(),
               pre=(ps.cuda(), pt.cuda()), pre_relu=True, act=act, in_cs=cs_in, out_cs=cs_out, out_coff=off, wino=True,
               post=(qs.cuda(), qt.cuda()))
    assert rel(out[:, off:off + cout], ref) < 2e-5
    assert float((out[:, :off] - 7).abs().max()) == 0 and float((out[:, off + cout:] - 7).abs().max()) == 0


def test_wino_data_gradient_layout(T):
    """The data gradient = the same kernel on ossid_conv_pack_weights_wino(dgrad = 1)."""
    g = torch.Generator().manual_seed(11)
    B, cin, cout, H, W = 4, 48, 64, 13, 10
    x = torch.randn(B, cin, H, W, generator=g, dtype=torch.float64, requires_grad=True)
    w = torch.randn(cout, cin, 3, 3, generator=g) * 0.1
    dy = torch.randn(B, cout, H, W, generator=g)
    F.conv2d(x, w.double(), padding=1).backward(dy.double())
    dx = T.empty_nhwc(B, cin, H, W, "cuda")
    T.conv_raw(nhwc(dy.cuda()), T._pack(w.cuda(), "wino_dgrad"), B, H, W, cout, cin, 9, dx, wino=True)
    assert rel(dx, x.grad) < 2e-5


def test_wino_rejects_what_it_does_not_do(T):
    from ossid_code_amd import _lib
    x = torch.zeros(1, 24, 4, 4).cuda().contiguous(memory_format=torch.channels_last)
    out = T.empty_nhwc(1, 32, 4, 4, "cuda")
    wpk = torch.zeros(1 << 16, device="cuda")
    with pytest.raises(Exception):
        T.conv_raw(x, wpk, 1, 4, 4, 24, 32, 9, out, wino=True)           # Cin % 16
    with pytest.raises(Exception):
        T.conv_raw(x, wpk, 1, 4, 4, 16, 32, 1, out, wino=True)           # not a 3x3
    with pytest.raises(Exception):
        T.conv_raw(x, wpk, 1, 4, 4, 16, 32, 9, out, wino=True, src_hw=(2, 2))   # fused up-sampling


def test_wino_pair_launch_equals_two_launches(T):
    """ossid_conv3x3_wino_fwd_pair: two independent layers in one grid, bit-equal to launching them one by one."""
    import ctypes
    from ossid_code_amd import _lib
    g = torch.Generator().manual_seed(21)
    outs = []
    descs = []
    keep = []
    for (B, cin, cout, H, W) in ((3, 64, 96, 9, 7), (2, 32, 64, 12, 5)):
        x = nhwc(torch.randn(B, cin, H, W, generator=g).cuda())
        w = (torch.randn(cout, cin, 3, 3, generator=g) * 0.1).cuda()
        bias = torch.randn(cout, generator=g).cuda()
        single = T.empty_nhwc(B, cout, H, W, "cuda")
        T.conv_raw(x, T._pack(w, "wino_fwd"), B, H, W, cin, cout, 9, single, bias=bias, act=1, wino=True)
        out = T.empty_nhwc(B, cout, H, W, "cuda")
        d = _lib.ConvDesc()
        d.x, d.wpk, d.bias, d.out = x.data_ptr(), T._pack(w, "wino_fwd").data_ptr(), bias.data_ptr(), out.data_ptr()
        d.in_batch_stride, d.batch, d.height, d.width, d.cin, d.cout, d.taps, d.act = -1, B, H, W, cin, cout, 9, 1
        descs.append(d)
        outs.append((single, out))
        keep += [x, w, bias]
    rc = _lib.fn("ossid_conv3x3_wino_fwd_pair")(ctypes.byref(descs[0]), ctypes.byref(descs[1]), _lib.stream())
    assert rc == 0
    torch.cuda.synchronize()
    for single, out in outs:
        assert torch.equal(single, out)
    descs[1].cin = 24                                            # one bad descriptor rejects the pair
    assert _lib.fn("ossid_conv3x3_wino_fwd_pair")(ctypes.byref(descs[0]), ctypes.byref(descs[1]), _lib.stream()) != 0


def test_pack_table_winograd_rows_equal_the_standalone_pack(T):
    """ossid_conv_pack_weights_table kinds 2 / 3 (the one-launch pack of a training step) write the same bytes as
    ossid_conv_pack_weights_wino."""
    class Conv:
        pass
    g = torch.Generator().manual_seed(8)
    convs = []
    for cout, cin in ((64, 64), (96, 128)):      # both layouts exist for these (the plan skips layouts the step never asks for)
        c = Conv()
        c.weight = (torch.randn(cout, cin, 3, 3, generator=g) * 0.1).cuda()
        convs.append(c)
    want = {(i, k): T._pack(c.weight, k).clone() for i, c in enumerate(convs) for k in ("wino_fwd", "wino_dgrad")}
    for c in convs:
        for k in ("wino_fwd", "wino_dgrad"):
            T._Packed.get(c.weight, k).zero_()
    plan = T.PackPlan(convs, {c: ("wino_fwd", "wino_dgrad") for c in convs})
    plan.run()
    for i, c in enumerate(convs):
        for k in ("wino_fwd", "wino_dgrad"):
            assert torch.equal(T._Packed.get(c.weight, k), want[(i, k)]), (i, k)
    T.end_step()
