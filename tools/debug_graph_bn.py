"""running_mean of a head BatchNorm after each of three finetune steps: eager vs hipGraph replay (debug)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from ossid_code_amd import dtoid  # noqa: E402
from ossid_code_amd.dtoid import finetune  # noqa: E402
from test_dtoid_gpu import _batch, _condition_encoders  # noqa: E402

cfg = dtoid.DtoidConfig()
for graphed in (False, True):
    torch.manual_seed(0)
    m = _condition_encoders(dtoid.DtoidNet(cfg).cuda().train())
    flat = finetune.FlatParams(m)
    opt = finetune.FusedAMSGrad(flat, lr=1e-4, weight_decay=1e-6)
    batches = [_batch(cfg, 2, "cuda", seed=s) for s in (0, 1, 2)]
    g = finetune.GraphedForwardBackward(m, flat, batches[0]) if graphed else None
    mods = {"nf": m.model.correlation_model.nf, "ns3": m.model.correlation_model.ns3,
            "b4.l1.n1": m.model.image_feature_extractor.backdense_2[-2]["denselayer1"].norm1 if hasattr(m.model.image_feature_extractor.backdense_2[-2], "values") else None,
            "tfe.norm_2": m.model.template_feature_extractor.norm_2, "n1": m.model.image_feature_extractor.n1}
    print("graphed" if graphed else "eager", {k: [round(float(v), 6) for v in b.running_mean[:3]] for k, b in mods.items() if b is not None})
    for i, b in enumerate(batches):
        loss = float(finetune.finetune_step(m, b, opt, graphed=g))
        print("  step", i, "loss %.6f" % loss, {k: [round(float(v), 6) for v in bb.running_mean[:3]] for k, bb in mods.items() if bb is not None},
              int(m.model.correlation_model.nf.num_batches_tracked))
