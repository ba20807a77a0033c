"""Per-parameter gradient of ONE step from the same state: eager multi-stream (twice) vs hipGraph replay (debug)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from ossid_code_amd import dtoid  # noqa: E402
from ossid_code_amd.dtoid import finetune  # noqa: E402
from test_dtoid_gpu import _batch, _condition_encoders  # noqa: E402

cfg = dtoid.DtoidConfig()
torch.manual_seed(0)
m = _condition_encoders(dtoid.DtoidNet(cfg).cuda().train())
flat = finetune.FlatParams(m)
b = _batch(cfg, 2, "cuda", seed=0)


def eager():
    flat.detach_grads()
    m(b)["loss"].backward()
    flat.gather_grads()
    torch.cuda.synchronize()
    return flat.grad.clone()


g0, g1, g2 = eager(), eager(), eager()
graphed = finetune.GraphedForwardBackward(m, flat, b)
graphed(b)
torch.cuda.synchronize()
gg = flat.grad.clone()
g3 = eager()
tot = float(g0.double().norm())
rows = []
for name, p in flat.entries:
    off, n = flat.offsets[name]
    if off >= flat.n_used:
        continue
    a = g0[off:off + n].double()
    d = lambda x: float((x[off:off + n].double() - a).norm())      # noqa: E731
    rows.append((d(gg) / tot, d(g1) / tot, d(g2) / tot, d(g3) / tot, float(a.norm()) / tot, name))
rows.sort(reverse=True)
print("total grad norm %.4e; columns: |graph - eager0|, |eager1 - eager0|, |eager2 - eager0|, |eager3 - eager0|, |g| (all / total)" % tot)
for r in rows[:14]:
    print("  %.2e %.2e %.2e %.2e %.2e %s" % r)
