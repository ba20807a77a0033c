"""Per-parameter gradient differences of TemplateEncoderTrain vs the nn.Module path, and run-to-run repeatability."""
import copy
import sys
import os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ossid_code_amd import dtoid  # noqa: E402
from ossid_code_amd.dtoid import train_encoders as TE, train_ops  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "local"
train_ops.SEQ_REPLAY = (sys.argv[2] != "0") if len(sys.argv) > 2 else True
torch.manual_seed(17)
net = dtoid.Network(img_size=(480, 640), heatmap_size=(29, 39)).cuda().train()
mod = net.template_feature_extractor if which == "local" else net.template_feature_extractor_global
with torch.no_grad():
    for m_ in mod.modules():
        if isinstance(m_, torch.nn.Conv2d):
            torch.nn.init.kaiming_normal_(m_.weight, nonlinearity="relu")
            m_.bias.normal_(0, 0.1)
ref = copy.deepcopy(mod)


def l2(a, b):
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).norm() / b.norm().clamp(min=1e-30))


prev = None
for rnd in range(3):
    torch.manual_seed(100 + (rnd if rnd < 2 else 1))          # rounds 1 and 2 use the SAME data: repeatability
    img = torch.rand(3, 4, 124, 124, device="cuda")
    for m in (mod, ref):
        for p in m.parameters():
            p.grad = None
    y_ref = ref(img)
    go = torch.randn_like(y_ref)
    y_ref.backward(go)
    y = TE.template_encoder_train(mod, img)
    y.backward(go)
    torch.cuda.synchronize()
    print("round", rnd, "out", l2(y, y_ref))
    cur = {}
    used = {id(p) for p in TE.encoder_params(mod)}
    for (n, p), q in zip(mod.named_parameters(), ref.parameters()):
        if id(p) not in used:
            continue
        cur[n] = p.grad.clone()
        e = l2(p.grad, q.grad)
        rep = "" if (prev is None or rnd != 2) else " repeat %.2e" % l2(p.grad, prev[n])
        if e > 1e-3 or rep:
            print("  %-45s %.3e%s" % (n, e, rep))
    prev = cur
