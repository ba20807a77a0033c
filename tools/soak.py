"""A longer run of what the online loop does (scripts/online_learning.py:466-679): test-time frames on the current weights with a
finetune burst every few frames -- the packed test-time plans and the frame's hipGraph have to follow every optimizer step, the
training step's recorded sequences, learned packing plan and gradient hooks have to survive eval / train switches. Checks that
every loss and score stays finite, that a frame after an update differs from the one before it, that device memory does not
grow, and prints the mean times.   python tools/soak.py [--frames 120 --every 6 --steps 3]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ossid_code_amd import dtoid  # noqa: E402
from ossid_code_amd.dtoid import finetune  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=120)
    ap.add_argument("--every", type=int, default=6)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8)
    a = ap.parse_args()
    cfg = dtoid.DtoidConfig()
    torch.manual_seed(0)
    m = dtoid.DtoidNet(cfg).cuda().eval()
    flat = finetune.FlatParams(m)
    opt = finetune.FusedAMSGrad(flat, lr=1e-4, weight_decay=1e-6)
    g = torch.Generator().manual_seed(1)
    B, nt = a.batch, 21
    test = {"img": torch.rand(1, 3, 480, 640, generator=g).cuda(), "obj_id": torch.tensor([1]),
            "limg": torch.rand(1, nt, 3, 124, 124, generator=g).cuda(),
            "lmask": (torch.rand(1, nt, 1, 124, 124, generator=g) > 0.5).float().cuda()}
    mask = torch.zeros(B, 1, 480, 640)
    mask[:, :, 120:240, 160:320] = 1

    def batch():
        b = {"img": torch.rand(B, 3, 480, 640, generator=g), "limg": torch.rand(B, 3, 124, 124, generator=g),
             "lmask": (torch.rand(B, 1, 124, 124, generator=g) > 0.5).float(), "gimg": torch.rand(B, 3, 124, 124, generator=g),
             "gmask": (torch.rand(B, 1, 124, 124, generator=g) > 0.5).float(),
             "bbox_gt": torch.tensor([[[160.0, 120.0, 320.0, 240.0, 1.0]]]).repeat(B, 1, 1),
             "heatmap": torch.rand(B, 1, 29, 39, generator=g).double(), "mask": mask}
        return {k: v.cuda() for k, v in b.items()}
    losses, top, mem, t_frame, t_step = [], [], [], [], []
    last = None
    changed = 0
    for f in range(a.frames):
        test["img"] = torch.rand(1, 3, 480, 640, generator=g).cuda()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = m.forwardTestTime(test)
        s = out["pred_scores"]
        torch.cuda.synchronize()
        t_frame.append(time.perf_counter() - t0)
        assert bool(torch.isfinite(s).all()) and s.numel() >= 1, f
        top.append(float(s[0]))
        if (f + 1) % a.every == 0:
            same_img = test["img"].clone()
            before = m.forwardTestTime(dict(test, img=same_img))["pred_scores"][:1].clone()
            m.clearCache()                       # the templates' features follow the encoders' weights (reference: clearCache after finetuning)
            m.train()
            for _ in range(a.steps):
                bt = batch()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                loss = finetune.finetune_step(m, bt, opt)
                torch.cuda.synchronize()
                t_step.append(time.perf_counter() - t0)
                losses.append(float(loss))
                assert losses[-1] == losses[-1] and abs(losses[-1]) < 1e6, (f, losses[-1])
            m.eval()
            after = m.forwardTestTime(dict(test, img=same_img))["pred_scores"][:1]
            changed += int(not torch.equal(before, after))
            mem.append(torch.cuda.memory_allocated() / 2 ** 20)
    n_bursts = a.frames // a.every
    print(json.dumps({"frames": a.frames, "finetune_steps": len(losses), "bursts": n_bursts,
                      "frames_that_changed_after_an_update": changed,
                      "loss_first_last": [round(losses[0], 4), round(losses[-1], 4)],
                      "top_score_first_last": [round(top[0], 5), round(top[-1], 5)],
                      "frame_ms_mean_of_last_half": round(1e3 * sum(t_frame[len(t_frame) // 2:]) / (len(t_frame) - len(t_frame) // 2), 3),
                      "step_ms_mean_of_last_half (synchronised per step)": round(1e3 * sum(t_step[len(t_step) // 2:]) / (len(t_step) - len(t_step) // 2), 3),
                      "step_ms_by_position_in_burst": [round(1e3 * sum(t_step[a.steps * (n_bursts // 2) + k::a.steps]) /
                                                             max(1, len(t_step[a.steps * (n_bursts // 2) + k::a.steps])), 3) for k in range(a.steps)],
                      "frame_ms_by_position_after_burst": [round(1e3 * sum(t_frame[a.every * (n_bursts // 2) + k::a.every]) /
                                                                 max(1, len(t_frame[a.every * (n_bursts // 2) + k::a.every])), 3) for k in range(a.every)],
                      "allocated_MiB_after_each_burst_first_last_max": [round(mem[0], 1), round(mem[-1], 1), round(max(mem), 1)]}))
    assert changed == n_bursts, "a frame did not follow its weights"
    assert mem[-1] <= mem[len(mem) // 2] * 1.02 + 64, "device memory grows"


if __name__ == "__main__":
    main()
