"""Diagnostic: where does the dense-block training path's input-gradient error against the nn.Module path come from --
spread-out rounding or a few flipped ReLU decisions? Prints L2 error, the share of elements off by more than 1e-4 of the
rms, and the same against a float64 module reference."""
import copy
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from ossid_code_amd.dtoid import backbones, train_ops as T


def run(L, C0, B, H, W, seed=2):
    torch.manual_seed(seed)
    blk = backbones.DenseBlock(L, C0).cuda().train()
    with torch.no_grad():
        for m in blk.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.normal_(1, 0.2)
                m.bias.normal_(0, 0.2)
    ref = copy.deepcopy(blk)
    ref64 = copy.deepcopy(blk).double()
    x = torch.randn(B, C0, H, W, device="cuda")
    go = torch.randn(B, C0 + 32 * L, H, W, device="cuda")

    def module(mod, xx, g):
        xr = xx.clone().requires_grad_(True)
        feats = [xr]
        for layer in mod.values():
            feats.append(layer(torch.cat(feats, 1)))
        y = torch.cat(feats, 1)
        y.backward(g)
        return y.detach(), xr.grad
    y32, g32 = module(ref, x, go)
    y64, g64 = module(ref64, x.double(), go.double())
    xm = x.clone().requires_grad_(True)
    y = T.dense_block_train(xm, blk)
    y.backward(go)
    for name, a, b in (("ours vs f64", xm.grad.double(), g64), ("module f32 vs f64", g32.double(), g64),
                       ("ours vs module f32", xm.grad.double(), g32.double())):
        d = (a - b)
        rms = b.pow(2).mean().sqrt()
        print("%-20s L2 %.2e  share>1e-4rms %.4f  share>1e-3rms %.4f  max/rms %.3f" % (
            name, float(d.norm() / b.norm()), float((d.abs() > 1e-4 * rms).double().mean()),
            float((d.abs() > 1e-3 * rms).double().mean()), float(d.abs().max() / rms)))
    print("forward: ours vs f64 max %.2e, module f32 vs f64 max %.2e (scale %.2f)" % (
        float((y.double() - y64).abs().max()), float((y32.double() - y64).abs().max()), float(y64.abs().max())))


if __name__ == "__main__":
    for cfg in ((3, 64, 2, 12, 16), (6, 64, 2, 30, 40)):
        print(cfg)
        run(*cfg)
