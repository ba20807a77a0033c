"""Scratch diagnostics for the hip training path: (1) gradient agreement table vs the module path, (2) which scopes can be
captured in a hipGraph (each in its own subprocess: a host crash in one does not stop the rest)."""
import json
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def batch_of(cfg, B, seed=5):
    from test_dtoid_gpu import _batch
    return _batch(cfg, B, "cuda", seed=seed)


def grads():
    import copy
    from ossid_code_amd import dtoid
    cfg = dtoid.DtoidConfig()
    torch.manual_seed(21)
    m = dtoid.DtoidNet(cfg).cuda().train()
    with torch.no_grad():
        for conv in (m.model.classification.output, m.model.regression.output, m.model.correlation_model.seg_final,
                     m.model.correlation_model.corr_conv_heatmap):
            conv.weight.normal_(0, 0.02)
    ref = copy.deepcopy(m)
    ref2 = copy.deepcopy(m)
    batch = batch_of(cfg, 2)
    m.model.use_hip_training, ref.model.use_hip_training, ref2.model.use_hip_training = True, False, False
    m(batch)["loss"].backward()
    ref(batch)["loss"].backward()
    torch.backends.cudnn.deterministic = True
    ref2(batch)["loss"].backward()
    rows = []
    tot_d = tot_r = tot_dd = 0.0
    for (n, p), q, q2 in zip(m.named_parameters(), ref.parameters(), ref2.parameters()):
        if q.grad is None:
            continue
        d = (p.grad.double() - q.grad.double())
        dd = (q2.grad.double() - q.grad.double())
        tot_d += float((d ** 2).sum()); tot_r += float((q.grad.double() ** 2).sum()); tot_dd += float((dd ** 2).sum())
        rows.append((float(d.abs().max() / q.grad.double().abs().max().clamp(min=1e-30)), n, float(q.grad.abs().max()),
                     float(d.abs().max()), float(dd.abs().max()), float(d.norm() / q.grad.double().norm().clamp(min=1e-30))))
    rows.sort(reverse=True)
    print("global rel L2 hip-vs-module %.3e ; module-vs-module (run to run) %.3e" % ((tot_d / tot_r) ** 0.5, (tot_dd / tot_r) ** 0.5))
    for r in rows[:40]:
        print("%.3e  %-90s max|ref| %.3e  max|d| %.3e  run-to-run %.3e  relL2 %.3e" % r)


def capture(scope):
    from ossid_code_amd import dtoid
    from ossid_code_amd.dtoid import backbones, finetune
    from ossid_code_amd.dtoid import train_ops as T
    torch.manual_seed(0)
    dev = "cuda"

    def run_graph(fn):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn(); fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fn()
        g.replay()
        torch.cuda.synchronize()
    if scope == "conv_fwd":
        conv = torch.nn.Conv2d(64, 32, 3, padding=1).cuda()
        x = torch.randn(2, 64, 20, 24, device=dev)
        with torch.no_grad():
            run_graph(lambda: T.fused_conv(x, conv, act_elu=True))
    elif scope == "conv_fwd_bwd":
        conv = torch.nn.Conv2d(64, 32, 3, padding=1).cuda()
        x = torch.randn(2, 64, 20, 24, device=dev, requires_grad=True)

        def fn():
            conv.weight.grad = None
            T.fused_conv(x, conv, act_elu=True).sum().backward()
        run_graph(fn)
    elif scope == "torch_fwd_bwd":
        conv = torch.nn.Conv2d(64, 32, 3, padding=1).cuda()
        x = torch.randn(2, 64, 20, 24, device=dev, requires_grad=True)

        def fn():
            conv.weight.grad = None
            torch.nn.functional.elu(conv(x)).sum().backward()
        run_graph(fn)
    elif scope == "dense":
        blk = backbones.DenseBlock(3, 64).cuda().train()
        x = torch.randn(2, 64, 12, 16, device=dev, requires_grad=True)

        def fn():
            for p in blk.parameters():
                p.grad = None
            T.dense_block_train(x, blk).sum().backward()
        run_graph(fn)
    elif scope in ("net_fwd", "net_hip", "net_miopen"):
        cfg = dtoid.DtoidConfig()
        m = dtoid.DtoidNet(cfg).cuda().train()
        m.model.use_hip_training = scope != "net_miopen"
        batch = batch_of(cfg, 2)
        if scope == "net_fwd":
            run_graph(lambda: m(batch)["loss"])
        else:
            flat = finetune.FlatParams(m)
            finetune.GraphedForwardBackward(m, flat, batch)
    print("CAPTURE_OK", scope, flush=True)


if __name__ == "__main__":
    what = sys.argv[1]
    if what == "grads":
        grads()
    elif what == "capture":
        capture(sys.argv[2])
    elif what == "bisect":
        for scope in ("conv_fwd", "torch_fwd_bwd", "conv_fwd_bwd", "dense", "net_fwd", "net_miopen", "net_hip"):
            r = subprocess.run([sys.executable, "-X", "faulthandler", os.path.abspath(__file__), "capture", scope],
                               capture_output=True, text=True, timeout=280)
            ok = "CAPTURE_OK" in r.stdout
            print("%-14s rc=%d %s" % (scope, r.returncode, "ok" if ok else "FAILED"), flush=True)
            if not ok:
                print("\n".join((r.stdout + r.stderr).splitlines()[-25:]), flush=True)
