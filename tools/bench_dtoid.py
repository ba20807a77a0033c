"""DTOID timing on one MI355X (not the headline bench; its numbers go to profiles/ and DESIGN.md):
  forward   forward_all_templates semantics, 1 image x n_t local templates, images/s
  finetune  DtoidNet.forward + loss + backward + fused AMSGrad step at batch B (cfg-4 per-GPU share), steps/s
  convs     the head's dominant 3x3 convolution shapes alone (MIOpen fp32 baseline for the hand-written MFMA conv)
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def timeit(fn, warm=2, reps=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nt", type=int, default=21)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--what", default="forward,finetune,convs")
    a = ap.parse_args()
    from ossid_code_amd import dtoid
    from ossid_code_amd.dtoid import finetune
    from test_dtoid_gpu import _batch
    torch.manual_seed(0)
    cfg = dtoid.DtoidConfig()
    out = {}
    if "convs" in a.what:
        shapes = [("corr 640->256 3x3", 640, 256, 3), ("cf 768->512 3x3", 768, 512, 3), ("cls/reg 512->256 3x3", 512, 256, 3),
                  ("trunk 256->256 3x3", 256, 256, 3), ("c1 tmpl 640->640 3x3 7x7", 640, 640, 3)]
        for name, ci, co, k in shapes:
            for B in (a.nt, a.batch):
                hw = (7, 7) if "7x7" in name else (29, 39)
                x = torch.randn(B, ci, *hw, device="cuda")
                w = torch.randn(co, ci, k, k, device="cuda")
                pad = 0 if "7x7" in name else 1
                t = timeit(lambda: F.conv2d(x, w, padding=pad))
                oh, ow = (hw[0] - 2, hw[1] - 2) if pad == 0 else hw
                fl = 2.0 * B * oh * ow * co * ci * k * k
                out["conv %s B=%d" % (name, B)] = {"ms": t * 1e3, "TFLOPs": fl / t / 1e12}
                if pad == 1:
                    from ossid_code_amd.dtoid import ops
                    cv = torch.nn.Conv2d(ci, co, 3, padding=1).cuda()
                    pk = ops.PackedConv3x3(cv)
                    xl = x.contiguous(memory_format=torch.channels_last)
                    t = timeit(lambda: pk(xl))
                    out["conv %s B=%d" % (name, B)].update({"hip_ms": t * 1e3, "hip_TFLOPs": fl / t / 1e12})
    if "bwd" in a.what:   # backward passes of the head shapes: MIOpen vs the hand-written kernels
        from ossid_code_amd.dtoid import ops
        for name, ci, co in (("corr 640->256", 640, 256), ("cf 768->512", 768, 512), ("cls/reg 512->256", 512, 256),
                             ("trunk 256->256", 256, 256)):
            B = a.batch
            x = torch.randn(B, ci, 29, 39, device="cuda").contiguous(memory_format=torch.channels_last)
            w = torch.randn(co, ci, 3, 3, device="cuda") * 0.01
            gy = torch.randn(B, co, 29, 39, device="cuda").contiguous(memory_format=torch.channels_last)
            fl = 2.0 * B * 29 * 39 * co * ci * 9
            t_wm = timeit(lambda: torch.nn.grad.conv2d_weight(x, w.shape, gy, padding=1))
            t_dm = timeit(lambda: torch.nn.grad.conv2d_input(x.shape, w, gy, padding=1))
            xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)

            def hip_bwd():
                xr.grad = wr.grad = None
                ops.conv3x3(xr, wr).backward(gy)
            def hip_fwd():
                with torch.no_grad():
                    ops.conv3x3(xr, wr)
            t_f = timeit(hip_fwd)
            t_all = timeit(hip_bwd)
            out["bwd %s B=%d" % (name, B)] = {"miopen_wgrad_ms": t_wm * 1e3, "miopen_dgrad_ms": t_dm * 1e3,
                                            "hip_fwd_ms": t_f * 1e3, "hip_fwd+dgrad+wgrad_ms": t_all * 1e3,
                                            "miopen_wgrad_TF": fl / t_wm / 1e12, "miopen_dgrad_TF": fl / t_dm / 1e12,
                                            "hip_bwd_TF(2 passes)": 2 * fl / (t_all - t_f) / 1e12}
    if "backbone" in a.what:
        net = dtoid.Network(img_size=(480, 640), heatmap_size=(29, 39)).cuda().eval()
        img = torch.rand(1, 3, 480, 640, device="cuda")
        tg = torch.randn(1, 64, 3, 3, device="cuda") * 0.1

        def graphed(fn):
            s_ = torch.cuda.Stream()
            s_.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s_), torch.no_grad():
                for _ in range(2):
                    fn()
            torch.cuda.current_stream().wait_stream(s_)
            g_ = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g_), torch.no_grad():
                fn()
            return g_
        with torch.no_grad():
            fb = net._fused_backbone()
            for name, fn in (("module(MIOpen)", lambda: net.image_feature_extractor(img, tg)), ("fused(HIP)", lambda: fb(img, tg))):
                te = timeit(fn, 2, 5)
                g_ = graphed(fn)
                tg_ = timeit(g_.replay, 2, 10)
                out["backbone %s" % name] = {"eager_ms": te * 1e3, "graph_ms": tg_ * 1e3, "TFLOPs_graph": 39.7e9 / tg_ / 1e12}
    if "forward" in a.what:
        m = dtoid.DtoidNet(cfg).cuda().eval()
        b = _batch(cfg, 1, "cuda")
        test = {"img": b["img"], "obj_id": torch.tensor([1]), "limg": torch.rand(1, a.nt, 3, 124, 124).cuda(),
                "lmask": (torch.rand(1, a.nt, 1, 124, 124) > 0.5).float().cuda()}
        t = timeit(lambda: m.forwardTestTime(test))
        out["forward n_t=%d" % a.nt] = {"ms": t * 1e3, "imgs_per_s": 1.0 / t,
                                        "TFLOPs": (39.7e9 + 46.0e9 * a.nt) / t / 1e12}
    if "finetune_cl" in a.what:   # experiment: whole model + inputs in channels_last memory format
        m = dtoid.DtoidNet(cfg).cuda().train().to(memory_format=torch.channels_last)
        flat = finetune.FlatParams(m)
        opt = finetune.FusedAMSGrad(flat)
        b = _batch(cfg, a.batch, "cuda")
        b = {k: (v.contiguous(memory_format=torch.channels_last) if v.dim() == 4 else v) for k, v in b.items()}
        t = timeit(lambda: finetune.finetune_step(m, b, opt), warm=3, reps=4)
        out["finetune channels_last B=%d" % a.batch] = {"ms": t * 1e3, "samples_per_s": a.batch / t}
    if "finetune_miopen" in a.what:   # the nn.Module path (MIOpen) for comparison; the default is the hip training path
        m = dtoid.DtoidNet(cfg).cuda().train()
        m.model.use_hip_training = False
        flat = finetune.FlatParams(m)
        opt = finetune.FusedAMSGrad(flat)
        b = _batch(cfg, a.batch, "cuda")
        t = timeit(lambda: finetune.finetune_step(m, b, opt), warm=3, reps=4)
        out["finetune module-path B=%d" % a.batch] = {"ms": t * 1e3, "samples_per_s": a.batch / t}
    if "finetune" in a.what.split(","):
        m = dtoid.DtoidNet(cfg).cuda().train()
        flat = finetune.FlatParams(m)
        opt = finetune.FusedAMSGrad(flat)
        b = _batch(cfg, a.batch, "cuda")
        t = timeit(lambda: finetune.finetune_step(m, b, opt), warm=2, reps=3)
        out["finetune B=%d" % a.batch] = {"ms": t * 1e3, "samples_per_s": a.batch / t, "TFLOPs": 258e9 * a.batch / t / 1e12}
        t2 = timeit(lambda: opt.step(), warm=1, reps=5)
        out["amsgrad_step"] = {"ms": t2 * 1e3, "GBps": flat.n_used * 4 * 9 / t2 / 1e9}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
