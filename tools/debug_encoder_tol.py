"""Diagnostic: per round, the output error and the four worst parameter-gradient errors (relative L2) of the global template
encoder's training node against the nn.Module path -- the numbers behind the tolerance of
tests/test_dtoid_gpu.py::test_template_encoder_training_node_matches_module_path. Run it on the default build and on a
-DOSSID_CONV_F32 build to separate arithmetic from flipped max-pool / ReLU decisions."""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import copy, torch
from ossid_code_amd import dtoid
from ossid_code_amd.dtoid import train_encoders as TE
torch.manual_seed(17)
net = dtoid.Network(img_size=(480, 640), heatmap_size=(29, 39)).cuda().train()
mod = net.template_feature_extractor_global
with torch.no_grad():
    for m in mod.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.weight.normal_(1, 0.2); m.bias.normal_(0, 0.2)
        elif isinstance(m, torch.nn.Conv2d):
            torch.nn.init.kaiming_normal_(m.weight, nonlinearity="relu"); m.bias.normal_(0, 0.1)
ref = copy.deepcopy(mod)
def l2(a, b):
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).norm() / b.norm().clamp(min=1e-30))
for rnd in range(3):
    img = torch.rand(3, 4, 124, 124, device="cuda") * (0.5 + 0.5 * rnd)
    for m in (mod, ref):
        for p in m.parameters(): p.grad = None
    y_ref = ref(img); go = torch.randn_like(y_ref); y_ref.backward(go)
    y = TE.template_encoder_train(mod, img); y.backward(go)
    torch.cuda.synchronize()
    errs = sorted(((l2(p.grad, q.grad), n) for (n, p), q in zip(mod.named_parameters(), ref.parameters()) if p.grad is not None), reverse=True)
    print(rnd, "y", "%.2e" % l2(y, y_ref), " ".join("%s %.2e" % (n.replace("backbone.features.", "f"), e) for e, n in errs[:4]))
