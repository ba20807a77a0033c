"""Synthetic online stream on one MI355X: detect -> score N hypotheses -> pseudo-label -> finetune every
`--interval` confident frames. Prints one JSON line with frames/s and the per-stage wall time. Random-init networks and
synthetic frames: this measures the loop, not accuracy.
  python tools/stream_demo.py --frames 24 --hypos 1000 --interval 8
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ossid_code_amd import dtoid, pipeline, synth, zephyr            # noqa: E402
from ossid_code_amd.dtoid import finetune                             # noqa: E402
from ossid_code_amd.stream import OnlineStream                        # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=24)
    ap.add_argument("--hypos", type=int, default=1000)
    ap.add_argument("--points", type=int, default=2048)
    ap.add_argument("--templates", type=int, default=21)
    ap.add_argument("--interval", type=int, default=8)
    ap.add_argument("--epochs", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8)
    a = ap.parse_args()
    t_start = time.perf_counter()
    from ossid_code_amd import parallel
    from ossid_code_amd.stream import run_speculative
    rank, world, local, dist = parallel.init_from_env(os.environ.get("OSSID_BENCH_BACKEND", "nccl"))
    local %= max(1, torch.cuda.device_count())       # gloo rehearsal: ranks may share a device
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    torch.manual_seed(0)
    det = dtoid.DtoidNet(dtoid.DtoidConfig()).to(dev).eval()
    flat = finetune.FlatParams(det)
    opt = finetune.FusedAMSGrad(flat, lr=1e-4, weight_decay=1e-6)
    sync = finetune.GradSync(flat, model=det) if dist is not None else None
    if sync is not None:
        sync.broadcast_params(0)
    class Args:
        dataset, no_valid_proj, no_valid_depth, inconst_ratio_th, interp = "HSVD_diff_uv_norm", True, True, 100, 0

    dataset = zephyr.ScoreDataset([], "", "lmo", Args(), mode="test")
    scorer = synth.random_pn2_state(zephyr.PointNet2SSG(dataset.dim_point, Args(), num_class=1), 0).to(0).eval()
    g = torch.Generator().manual_seed(1)
    limg = torch.rand(a.templates, 3, 124, 124, generator=g)
    lmask = (torch.rand(a.templates, 1, 124, 124, generator=g) > 0.5).float()
    bank = pipeline.TemplateBank(n_local_test=a.templates)
    bank.add(1, limg, lmask)
    frames = []
    for f in range(a.frames):
        d = synth.make_scoring_inputs(a.hypos, a.points, seed=100 + f)
        d.update(limg=limg, lmask=lmask, obj_id=1, pose_gt=d["pose_hypos"][0].copy())
        frames.append(d)
    ft_time, ft_steps = [0.0], [0]

    def note(msg):
        print("[stream_demo rank %d %.1fs] %s" % (rank, time.perf_counter() - t_start, msg), file=sys.stderr, flush=True)

    def finetune_fn(samples):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ps = pipeline.PseudoLabelSet(bank, mode="train", seed=len(samples))
        for i, (fr, smp) in enumerate(samples):
            ps.add(1, 0, i, fr["img"], fr["depth"], fr["cam_K"], smp["mask"][0], 0.0)
        items = [ps[i] for i in range(len(ps))]
        det.train()
        rng = np.random.default_rng(len(items))
        for _ in range(a.epochs):
            order = rng.permutation(len(items))
            for b0 in range(0, len(order), a.batch):
                mine = order[b0:b0 + a.batch][rank::world]       # this rank's slice of the global batch
                if len(mine) == 0:
                    mine = order[b0:b0 + 1]                      # every rank must take part in the gradient all-reduce
                batch = pipeline.collate([items[i] for i in mine])
                if os.environ.get("OSSID_STREAM_DEBUG"):
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    out = det(batch)
                    torch.cuda.synchronize()
                    t2 = time.perf_counter()
                    flat.detach_grads()
                    out["loss"].backward()
                    flat.gather_grads()
                    torch.cuda.synchronize()
                    t3 = time.perf_counter()
                    if sync is not None:
                        sync.sync()
                    torch.cuda.synchronize()
                    t4 = time.perf_counter()
                    opt.step()
                    if sync is not None:
                        sync.sync_buffers()
                    torch.cuda.synchronize()
                    note("step B=%d: fwd %.2f s, bwd %.2f s, sync %.2f s, opt %.3f s" %
                         (len(mine), t2 - t1, t3 - t2, t4 - t3, time.perf_counter() - t4))
                elif os.environ.get("OSSID_STREAM_STEPTIMES"):
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    finetune.finetune_step(det, batch, opt, sync)
                    torch.cuda.synchronize()
                    note("finetune step %d: %.1f ms (B=%d)" % (ft_steps[0], 1e3 * (time.perf_counter() - t1), len(mine)))
                else:
                    finetune.finetune_step(det, batch, opt, sync)
                ft_steps[0] += 1
        det.eval()
        det.clearCache()                       # template features depend on the weights that just changed
        torch.cuda.synchronize()
        ft_time[0] += time.perf_counter() - t0

    stream = OnlineStream(det, scorer, dataset, confident_threshold=-1e30, finetune_fn=finetune_fn)
    note("models and %d frames ready" % len(frames))
    r0 = stream.process(frames[0])             # warm-up: graph capture, workspace allocation ...
    note("warm-up frame done")
    snap_p, snap_b = flat.param.clone(), [b.clone() for b in det.buffers()]
    epochs, a.epochs = a.epochs, 1                # ... and MIOpen's per-shape algorithm search for the training convs
    for nb in sorted({a.batch, a.interval % a.batch or a.batch}):
        finetune_fn([(frames[0], r0["sample"])] * nb)
    a.epochs = epochs
    note("warm-up finetune done")
    with torch.no_grad():
        flat.param.copy_(snap_p)
        for b, s0 in zip(det.buffers(), snap_b):
            b.copy_(s0)
    for t in (opt.exp_avg, opt.exp_avg_sq, opt.max_exp_avg_sq):
        t.zero_()
    opt.step_count, ft_time[0], ft_steps[0] = 0, 0.0, 0
    det.clearCache()
    for k in stream.times:
        stream.times[k] = 0.0
    stream.n_processed = 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if dist is None:
        results, win = stream.run(frames, a.interval)
    else:
        samples = []

        def process_fn(frame):
            r = stream.process(frame)
            if stream.n_processed % 4 == 0:
                note("processed %d frames" % stream.n_processed)
            return r["confident"], r["sample"]

        def ft(train):          # identical frame-ordered (frame id, sample) list on every rank
            note("finetune on %d samples" % len(train))
            finetune_fn([(frames[f], smp) for f, smp in train])
        committed, win = run_speculative(frames, process_fn, ft, a.interval, dist)
    torch.cuda.synchronize()
    total = time.perf_counter() - t0
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return
    per = {k: 1e3 * v / max(1, stream.n_processed) for k, v in stream.times.items()}   # frames THIS rank processed
    print(json.dumps({"metric": "online stream frames/sec", "value": a.frames / total, "unit": "frames/s", "n_gpus": world,
                      "speculated_frames_discarded": win.discarded,
                      "frames": a.frames, "hypotheses_per_frame": a.hypos, "points": a.points,
                      "templates": a.templates, "finetunes": len(win.train_set) // a.interval,
                      "finetune_steps": ft_steps[0], "finetune_ms_per_step": 1e3 * ft_time[0] / max(1, ft_steps[0]),
                      "ms_per_frame_excl_finetune": 1e3 * (total - ft_time[0]) / a.frames,
                      "frames_processed_by_rank0": stream.n_processed,
                      "stage_ms_per_frame": per, "hyp_per_sec_in_stream": a.hypos / (per["score"] * 1e-3),
                      "data": "synthetic", "weights": "random-init"}))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
