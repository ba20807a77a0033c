"""bench.py -- OSSID hot-path benchmark on MI355X (contract: see the task statement / DESIGN.md section 6).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one frame of Zephyr hypothesis scoring, BASELINE.json configs[1]: 1000 pose hypotheses x 2048 model
points against a synthetic 640x480 RGB-D frame -- stage the frame (5x5 blur, /255, RGB-D interleave), build the
model table, project + gather + featurize every (hypothesis, point), and score with PointNet2SSG; inputs are
already resident in HBM when the timed region starts, scores stay on the device (one sync after the K steps).
Frames shard across ranks with no data-path collective ("weak" scaling): every rank scores its own frame.

The JSON line also carries
  roofline      the dominant kernel's achieved rate (HIP events recorded around it inside the timed steps),
  cpu_baseline  the CPU oracle (oracle/, a port of the same algorithm: kind "port") timed on the host cores on a
                bounded sample of the same workload, rank 0 / N=1 only.
"""
import argparse
import gc
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_HYP, N_PTS, IMG_H, IMG_W = 1000, 2048, 480, 640
PEAK_F32_MATRIX_TFLOPS = 157.3   # MI355X_MICROARCH.md, Chip-level parameters
PEAK_HBM_GBS = 8000.0

# algorithmic flops per hypothesis of each MFMA stage (2 x MACs)
SA1_FLOPS = 2.0 * 512 * 64 * (8 * 64 + 64 * 64 + 64 * 128)
P2_FLOPS = 2.0 * 512 * 128 * 128
SA2_FLOPS = 2.0 * 128 * 64 * (3 * 128 + 128 * 128 + 128 * 256)
SA3_FLOPS = 2.0 * 128 * (259 * 256 + 256 * 512 + 512 * 1024)
FC_FLOPS = 2.0 * (1024 * 512 + 512 * 256 + 256)
STAGE_FLOPS = {"sa1": SA1_FLOPS, "p2": P2_FLOPS, "sa2": SA2_FLOPS, "sa3": SA3_FLOPS, "fc": FC_FLOPS}


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-sample", type=int, default=192, help="hypotheses in the CPU-oracle sample")
    p.add_argument("--no-dtoid", action="store_true", help="skip the secondary DTOID measurements")
    p.add_argument("--streams", type=int, default=2,
                   help="frames in flight per GPU in the timed region, on alternating HIP streams (the product form: "
                        "scoring.networkInferenceMany keeps two in flight). The kernels of one frame fill the launch tails of "
                        "the other's (+3.5 %% measured). Per-kernel HIP-event durations for the roofline block are taken in a "
                        "second pass with ONE frame in flight, so that they do not include another frame's kernels")
    p.add_argument("--dtoid-templates", type=int, default=21)
    p.add_argument("--dtoid-images", type=int, default=32, help="images per batch of the configs[2] leg")
    p.add_argument("--dtoid-timeout", type=int, default=420, help="watchdog (s) for the secondary DTOID leg")
    p.add_argument("--dtoid-batch", type=int, default=8, help="finetune batch per GPU (cfg-4: 64 over 8 GPUs)")
    return p.parse_args()


class Args:
    dataset, no_valid_proj, no_valid_depth, inconst_ratio_th, extra_bottleneck_dim = \
        "HSVD_diff_uv_norm", True, True, 100, 0


def cpu_baseline(d, model, sample):
    """The oracle on the host cores, same workload, `sample` hypotheses (about 10-30 s of CPU work)."""
    from oracle import zephyr_oracle as ozr
    from ossid_code_amd.zephyr.pointnet2 import fold_pn2
    # the GPU box hands one GPU a 16-core CPU share; use at most that many OpenMP threads (and say how many)
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count()
    cores = ozr.set_threads(min(avail, 16))
    w = fold_pn2(model)
    T = d["pose_hypos"][:sample].astype(np.float32)
    t0 = time.time()
    rgbd = ozr.pack_rgbd(ozr.u8_to_unit(ozr.blur5_u8(d["img"])), d["depth"])
    tab = ozr.prep_model(d["model_points"], d["model_normals"], d["model_colors"])
    px, uv = ozr.featurize(rgbd, T, tab, d["cam_K"])
    t1 = time.time()
    scores = ozr.pn2_score(px, w)
    t2 = time.time()
    return {"value": sample / (t2 - t0), "unit": "hyp/s", "cores": int(cores), "kind": "port",
            "sample": "%d of the %d hypotheses of the same frame (x %d points): blur+featurize %.2f s, PointNet2SSG "
                      "%.2f s; C/OpenMP oracle, AVX2 fmaf chains" % (sample, N_HYP, N_PTS, t1 - t0, t2 - t1)}, scores


def dtoid_flops(nt, hw=(29, 39), img=(480, 640)):
    """(nominal, executed) f32 flops (2 x MAC) of one test-time frame with nt templates. Nominal = the reference's forward
    (39.7 G backbone + 45.96 G per (image, template) pair). Executed = what this build's kernels do after
    the exact reassociations of DESIGN.md 5: conv(image - avg_t) once per frame instead of per template; conv(image *
    avg_t) as G once per frame + a [nt x 640] GEMM from 40 templates on; the three decoder convs behind a 2x nearest
    up-sampling as four 2x2 phase convs (4/9); the decoder tail's first conv with merged kernel rows (2/3)."""
    px = hw[0] * hw[1]
    conv640 = 2.0 * px * 256 * 640 * 9                      # one 640->256 3x3 conv at 29x39: 3.336 G
    nominal = 39.7e9 + 45.96e9 * nt
    dec = 3 * 2.0 * (2 * hw[0]) * (2 * hw[1]) * 128 * 256 * 9          # s2, s3, s4: equal cost, 2.668 G each
    s5 = 2.0 * img[0] * img[1] * 16 * 32 * 9
    per_t = 45.96e9 - conv640 - dec * (5.0 / 9.0) - s5 / 3.0
    per_frame = 39.7e9 + conv640
    if nt >= 40:
        per_t -= conv640 - 2.0 * 640 * px * 256
        per_frame += conv640
    return nominal, per_frame + per_t * nt


def dtoid_leg(a, dev, dist, world):
    """Secondary metric of BASELINE.json ("DTOID imgs/sec"): test-time forward, 1 image x n_t templates per rank
    (cfg-3 (i), frames sharded), and the finetune step (cfg-4: batch per GPU, gradient mean over ranks on RCCL).
    Same barrier + synchronize + max-over-ranks timing as the main metric."""
    from ossid_code_amd import dtoid
    from ossid_code_amd.dtoid import finetune

    def timed(fn, warm, reps):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        # what the legs before this one left behind (CPU-baseline models, captured graphs, plans) goes to the permanent
        # generation: a full collection inside the timed region would walk all of it (measured: finetune leg 26.1 -> 23.9 ms
        # on a box where it hit one); nothing of the step itself is skipped
        gc.collect()
        gc.freeze()
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        t = time.perf_counter() - t0
        if dist is not None:
            tt = torch.tensor([t], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            t = float(tt.item())
        return t / reps

    cfg = dtoid.DtoidConfig()
    torch.manual_seed(0)
    m = dtoid.DtoidNet(cfg).to(dev).eval()
    g = torch.Generator().manual_seed(1)
    nt, B = a.dtoid_templates, a.dtoid_batch
    test = {"img": torch.rand(1, 3, 480, 640, generator=g).to(dev), "obj_id": torch.tensor([1]),
            "limg": torch.rand(1, nt, 3, 124, 124, generator=g).to(dev),
            "lmask": (torch.rand(1, nt, 1, 124, 124, generator=g) > 0.5).float().to(dev)}
    t_fwd = timed(lambda: m.forwardTestTime(test), 5, 40)      # (40 frames: 10 were 40 ms of wall, at the mercy of one slow frame)
    # BASELINE configs[2] as stated: batch = 32 images x 21 templates. (i) forward_all_templates semantics per image
    # through the additive batched API (backbone once on the batch, head graph per image); (ii) Network.forward on 32
    # (image, template) pairs
    B32 = a.dtoid_images
    test32 = dict(test, img=torch.rand(B32, 3, 480, 640, generator=g).to(dev))
    t_b32 = timed(lambda: m.forwardTestTimeBatch(test32), 1, 3)
    pairs = [torch.rand(B32, 3, 480, 640, generator=g), torch.rand(B32, 3, 124, 124, generator=g),
             (torch.rand(B32, 1, 124, 124, generator=g) > 0.5).float(), torch.rand(B32, 3, 124, 124, generator=g),
             (torch.rand(B32, 1, 124, 124, generator=g) > 0.5).float()]
    pairs = [dtoid.normalizeImageRange(p.to(dev)) if p.shape[1] == 3 else p.to(dev) for p in pairs]

    def run_pairs():
        with torch.no_grad():
            m.model(*pairs)
    t_pairs = timed(run_pairs, 2, 10)
    # matrix-core work each leg actually issues (ossid_code_amd._lib.count_mfma: summed at the launch sites over ONE eager,
    # graph-free pass -- a graph replay makes no host calls): Winograd layers count their 16 multiplies per tile, the
    # reassociated layers what they run. Layers still on MIOpen (7x7 stem in training, the 1-channel output convs) are not in it.
    from ossid_code_amd import _lib as oslib

    def counted(fn):
        old = m.model.__dict__.get("use_graph")
        m.model.use_graph = False
        try:
            with torch.no_grad(), oslib.count_mfma() as c:
                fn()
        finally:
            if old is None:
                m.model.__dict__.pop("use_graph", None)
            else:
                m.model.use_graph = old
        torch.cuda.synchronize()
        return c.flops, c.pipe
    mf_fwd = counted(lambda: m.forwardTestTime(test))
    mf_b32 = counted(lambda: m.forwardTestTimeBatch(test32))
    mf_pairs = counted(run_pairs)
    del test32, pairs
    torch.cuda.empty_cache()
    cpu_fwd = cpu_ft = None
    if dist is None and not a.no_cpu_baseline:
        # the same network through torch's CPU kernels (the nn.Module path of this build = the reference's structure),
        # like for like: 1 image x the same n_t templates (1.0 TFLOP at 21), all host cores; template features cached by a
        # first call as on the GPU
        torch.manual_seed(0)
        mc = dtoid.DtoidNet(cfg).eval()
        tc = {"img": test["img"].cpu(), "obj_id": torch.tensor([1]), "limg": test["limg"].cpu(), "lmask": test["lmask"].cpu()}
        from oracle import dtoid_oracle     # the checker's CPU restatements of the three HIP ops (baseline leg only)
        with dtoid_oracle.cpu_ops():
            mc.forwardTestTime(tc)
            t0 = time.perf_counter()
            mc.forwardTestTime(tc)
            t_cpu = time.perf_counter() - t0
        cpu_fwd = {"value": 1.0 / t_cpu, "unit": "img/s at n_t = %d" % nt, "cores": torch.get_num_threads(), "kind": "port",
                   "sample": "1 image x %d local templates (%.0f GFLOP) through torch CPU kernels: %.2f s, %.3f TFLOP/s"
                             % (nt, (39.7e9 + 45.96e9 * nt) / 1e9, t_cpu, (39.7e9 + 45.96e9 * nt) / t_cpu / 1e12)}
        # ... and the finetune step: DtoidNet.forward + loss + backward of 2 samples (BatchNorm in training mode needs
        # more than one value per channel), nn.Module path on the CPU
        g2 = torch.Generator().manual_seed(2)
        mk2 = torch.zeros(2, 1, 480, 640)
        mk2[:, :, 120:240, 160:320] = 1
        bc = {"img": torch.rand(2, 3, 480, 640, generator=g2), "limg": torch.rand(2, 3, 124, 124, generator=g2),
              "lmask": (torch.rand(2, 1, 124, 124, generator=g2) > 0.5).float(),
              "gimg": torch.rand(2, 3, 124, 124, generator=g2),
              "gmask": (torch.rand(2, 1, 124, 124, generator=g2) > 0.5).float(),
              "bbox_gt": torch.tensor([[[160.0, 120.0, 320.0, 240.0, 1.0]]]).repeat(2, 1, 1),
              "heatmap": torch.rand(2, 1, 29, 39, generator=g2).double(), "mask": mk2}
        mc.train()
        with dtoid_oracle.cpu_ops():
            t0 = time.perf_counter()
            mc(bc)["loss"].backward()
            t_cpu_ft = time.perf_counter() - t0
        cpu_ft = {"value": 2.0 / t_cpu_ft, "unit": "sample/s", "cores": torch.get_num_threads(), "kind": "port",
                  "sample": "DtoidNet.forward + 4-term loss + backward of 2 samples (516 GFLOP nominal) through torch CPU "
                            "kernels, no optimizer step: %.2f s, %.3f TFLOP/s" % (t_cpu_ft, 2 * 258e9 / t_cpu_ft / 1e12)}
        del mc, bc
    flat = finetune.FlatParams(m)
    opt = finetune.FusedAMSGrad(flat, lr=1e-4, weight_decay=1e-6)
    sync = finetune.GradSync(flat, model=m) if dist is not None else None
    if sync is not None:
        sync.broadcast_params(0)
    mask = torch.zeros(B, 1, 480, 640)
    mask[:, :, 120:240, 160:320] = 1
    batch = {"img": torch.rand(B, 3, 480, 640, generator=g), "limg": torch.rand(B, 3, 124, 124, generator=g),
             "lmask": (torch.rand(B, 1, 124, 124, generator=g) > 0.5).float(),
             "gimg": torch.rand(B, 3, 124, 124, generator=g),
             "gmask": (torch.rand(B, 1, 124, 124, generator=g) > 0.5).float(),
             "bbox_gt": torch.tensor([[[160.0, 120.0, 320.0, 240.0, 1.0]]]).repeat(B, 1, 1),
             "heatmap": torch.rand(B, 1, 29, 39, generator=g).double(), "mask": mask}
    batch = {k: v.to(dev) for k, v in batch.items()}
    m.train()
    # the product path: eager launches, weight gradients and the independent branches of the head on side HIP streams
    # (train_ops.WGRAD_SIDE, Network.use_train_streams). (The one-stream hipGraph replay of the same step -- slower, not the
    # product path -- is timed by tools/bench_finetune.py, not here.)
    t_ft = timed(lambda: finetune.finetune_step(m, batch, opt, sync), 5, 20)
    with oslib.count_mfma() as c_ft:
        finetune.finetune_step(m, batch, opt, sync)
    torch.cuda.synchronize()
    mf_ft = (c_ft.flops, c_ft.pipe)
    # the nn.Module path (MIOpen convolutions / BatchNorm, torch elementwise) on the same batch, for comparison
    m.model.use_hip_training = False
    try:
        t_ft_module = timed(lambda: finetune.finetune_step(m, batch, opt, sync), 2, 3)
    finally:
        m.model.use_hip_training = True
    nominal, executed = dtoid_flops(nt)
    note = ("frac_mfma: matrix-core multiply-adds the leg ISSUES (counted at the launch sites of one eager pass, "
            "ossid_code_amd._lib.count_mfma: Winograd layers at 16 multiplies per 2x2 tile, reassociated layers at what they "
            "run, weight gradients included), each launch in units of f32-pipe time (v_mfma_f32_32x32x2_f32 launches 1:1, "
            "split-bf16 launches 3/16, three-way-split launches 6/16) / wall / 157.3 TFLOP/s, over the whole call incl. "
            "top-k / NMS / host latency: <= 1 by construction. hbm_frac: (2 x FETCH_SIZE + WRITE_SIZE) per call from "
            "profiles/r04_dtoid_traffic.json / wall / 8 TB/s. achieved_mfma: the same multiply-adds as f32 arithmetic / wall")

    # bytes beyond L2 and launches per call of each leg, from the committed rocprofv3 passes (separate --pmc FETCH_SIZE /
    # WRITE_SIZE runs, tools/pmc_dtoid_traffic.py; read = 2 x FETCH_SIZE on gfx950, Infinity-Cache hits included: an upper
    # bound on HBM bytes)
    try:
        leg_pmc = json.load(open(os.path.join(ROOT, "profiles", "r04_dtoid_traffic.json")))["legs"]
    except Exception:
        leg_pmc = {}
    SPLIT = "f32 tensors, f32 accumulate; each f32 product = 3 bf16 matrix-core products (v_mfma_f32_32x32x16_bf16)"

    def roof(leg, flops_nom, t, counted_pair, dtype, flops_exec=None, calls_per_t=1):
        """What bounds a DTOID leg. frac_mfma: the matrix pipe's busy fraction (counted launches in f32-pipe time / wall /
        157.3 TFLOP/s); hbm_frac: counter bytes beyond L2 per call / wall / 8 TB/s; bound = the larger of the two, or
        "latency" when both are under one half (then the launch count beside it is the number to look at).
        nominal_over_f32_peak: the reference's nominal flops / wall / peak -- a throughput in the reference's units, NOT a
        fraction of anything these kernels run against (Winograd and the reassociations execute fewer multiplies, split-bf16
        products run at 16/3 of the f32 instruction's rate): it can exceed 1."""
        arith, pipe = counted_pair
        frac_mfma = pipe / t / 1e12 / PEAK_F32_MATRIX_TFLOPS
        pm = leg_pmc.get(leg) or {}
        traffic = pm.get("traffic_bytes_per_call")
        launches = pm.get("launches_per_call")
        hbm_frac = None if traffic is None else traffic * calls_per_t / t / 1e9 / PEAK_HBM_GBS
        top = max(frac_mfma, hbm_frac or 0.0)
        bound = "latency" if top < 0.5 else ("mfma" if frac_mfma >= (hbm_frac or 0.0) else "hbm")
        r = {"bound": bound, "frac_mfma": frac_mfma, "hbm_frac": hbm_frac, "launches_per_call": launches,
             "traffic": traffic, "achieved_mfma": arith / t / 1e12, "mfma_gflop_per_call": arith / 1e9,
             "mfma_pipe_gflop_per_call": pipe / 1e9, "peak": PEAK_F32_MATRIX_TFLOPS, "unit": "TFLOP/s",
             "peak_hbm_gbs": PEAK_HBM_GBS, "achieved_nominal": flops_nom / t / 1e12,
             "nominal_over_f32_peak": flops_nom / t / 1e12 / PEAK_F32_MATRIX_TFLOPS, "dtype": dtype, "note": note}
        if flops_exec is not None:
            r["executed_over_f32_peak"] = flops_exec / t / 1e12 / PEAK_F32_MATRIX_TFLOPS
        return r
    pair_flops = (39.7e9 + 45.96e9 + 0.36e9) * B32          # + the two template encoders per pair
    # per-pair images: only the decoder reassociations apply (phase convs 4/9, tail rows 2/3)
    pair_saved = B32 * (3 * 2.0 * 58 * 78 * 128 * 256 * 9 * (5.0 / 9.0) + 2.0 * 480 * 640 * 16 * 32 * 9 / 3.0)
    return {"forward": {"metric": "DTOID imgs/sec", "value": world / t_fwd, "unit": "img/s", "ms_per_image": 1e3 * t_fwd,
                        "config": "forward_all_templates, 1 image x %d local templates per rank, 480x640, topk 500, f32; "
                                  "hand-written MFMA conv head + hipGraph" % nt,
                        "tflops": world * nominal / t_fwd / 1e12,
                        "roofline": roof("forward", nominal, t_fwd, mf_fwd, SPLIT, executed),
                        "cpu_baseline": cpu_fwd},
            "forward_batch": {"metric": "DTOID imgs/sec", "value": world * B32 / t_b32, "unit": "img/s",
                              "ms_per_batch": 1e3 * t_b32, "ms_per_image": 1e3 * t_b32 / B32,
                              "config": "BASELINE configs[2]: batch=%d images 640x480 x %d templates per rank "
                                        "(forwardTestTimeBatch: forward_all_templates semantics per image, backbone "
                                        "batched, head graph per image), topk 500, f32" % (B32, nt),
                              "roofline": roof("forward_batch", B32 * nominal, t_b32, mf_b32, SPLIT, B32 * executed)},
            "forward_pairs": {"metric": "DTOID (image, template) pairs/sec", "value": world * B32 / t_pairs,
                              "unit": "pair/s", "ms_per_batch": 1e3 * t_pairs,
                              "config": "Network.forward on %d (image, template) pairs per rank, eval, f32 (template "
                                        "encoders + backbone + head, dense outputs)" % B32,
                              "roofline": roof("forward_pairs", pair_flops, t_pairs, mf_pairs, SPLIT, pair_flops - pair_saved)},
            "finetune": {"metric": "DTOID finetune samples/sec", "value": world * B / t_ft, "unit": "sample/s",
                         "ms_per_step": 1e3 * t_ft,
                         "ms_per_step_module_path_miopen": 1e3 * t_ft_module, "global_batch": world * B,
                         "config": "DtoidNet.forward + 4-term loss + backward + fused AMSGrad on the hand-written training "
                                   "kernels (channels-last; csrc/conv.hip fwd/dgrad, csrc/train.hip wgrad / BatchNorm fold / "
                                   "generic passes / fused loss; csrc/wino.hip for the head's 3x3 fwd/dgrad), eager launches on "
                                   "HIP streams (weight gradients and independent head branches beside the critical path), "
                                   "batch %d per GPU, BatchNorm batch statistics per rank, gradient mean over %d rank(s)%s" %
                                   (B, world,
                                    " (RCCL all-reduce of the flat 136 MB buffer)" if world > 1 else ""),
                         "tflops": world * B * 258e9 / t_ft / 1e12,
                         "cpu_baseline": cpu_ft,
                         "roofline": roof("finetune", B * 258e9, t_ft, mf_ft,
                                          SPLIT + " in data / weight gradients and the ELU head; the training forward of the ReLU / "
                                          "max-pool networks at f32 level (six products of a three-way split; exact-f32 instruction for "
                                          "the stem and the template encoders)")}}


def resolve_world(a, environ=None, spawn=None):
    """How many ranks ACTUALLY run, and what to do when --gpus disagrees (pure host logic, tests/test_bench_contract.py).
    Every reported number (value, n_gpus, the parallelism string) is computed from the returned world size, never
    from --gpus. Returns (world, rank, local_rank), or calls `spawn(n)` -- which must not return -- when `--gpus N > 1`
    was given to a bare `python bench.py` (no torchrun environment): the N ranks are then launched as children through
    torch.distributed.run BEFORE this process touches the GPU. A torchrun world that differs from --gpus is an error."""
    environ = os.environ if environ is None else environ
    launched = "WORLD_SIZE" in environ and "RANK" in environ
    world = int(environ.get("WORLD_SIZE", "1"))
    if not launched and a.gpus > 1:
        (spawn or spawn_ranks)(a.gpus)
        raise SystemExit("bench.py: spawn_ranks returned")
    if a.gpus != world:
        raise SystemExit("bench.py: --gpus %d but %d rank(s) were launched (WORLD_SIZE); refusing to report a "
                         "throughput for GPUs that did no work" % (a.gpus, world))
    return world, int(environ.get("RANK", "0")), int(environ.get("LOCAL_RANK", "0"))


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N ranks (one per GPU, RCCL) as a child job and leave with
    its exit code. The ranks are CHILD processes (subprocess, never an exec of this one), so whether counting the devices
    touched the HIP runtime here does not matter to them."""
    import subprocess
    have = torch.cuda.device_count()
    if have < n:
        raise SystemExit("bench.py: --gpus %d but only %d GPU(s) are visible" % (n, have))
    port = 29500 + os.getpid() % 2000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd))


def make_line(names, stage_ms, feat_ms, elapsed, world, steps, warmup, n_streams, top1, base, dtoid_out, extra=None):
    """THE json line (pure host logic: tests/test_bench_contract.py checks its schema and arithmetic on synthetic stage
    times). names / stage_ms: the scorer's stage names and their mean HIP-event durations (ms) per frame with every kernel
    alone on the chip (one stream, one launch per stage); feat_ms: the featurize kernel's; elapsed: seconds for `steps`
    frames per rank (max over ranks) with n_streams frames in flight. extra: reserved."""
    traffic = None
    try:   # HBM-side bytes per launch from the committed rocprofv3 PMC passes (tools/pmc_traffic.py)
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))["kernels"]
    except Exception:
        pmc = {}
    names = list(names)
    dom = max(STAGE_FLOPS, key=lambda k: stage_ms[names.index(k)])
    if dom + "_kernel" in pmc:
        traffic = pmc[dom + "_kernel"]["traffic_bytes"]
    dom_ms = float(stage_ms[names.index(dom)])
    achieved = STAGE_FLOPS[dom] * N_HYP / (dom_ms * 1e-3) / 1e12
    feat_bytes = N_HYP * N_PTS * (32 + 8) + IMG_H * IMG_W * 16 + N_PTS * 48 + N_HYP * 64
    return {
        "metric": "hypotheses scored/sec", "value": world * steps * N_HYP / elapsed, "unit": "hyp/s",
        "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * elapsed / steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "zephyr_score: %d hypotheses x %d model points, %dx%d RGB-D frame, "
                               "HSVD_diff_uv_norm features (D=8) + PointNet2SSG (SA 512/0.2/64 [8,64,64,128], "
                               "SA 128/0.4/64 [131,128,128,256], SA all [259,256,512,1024], FC 512-256-1); "
                               "BASELINE.json configs[1]" % (N_HYP, N_PTS, IMG_W, IMG_H),
                   "frames_per_step_per_gpu": 1, "frames_in_flight_per_gpu": n_streams,
                   "parallelism": "frames sharded, %d rank(s)" % world, "top1": top1},
        "roofline": {"bound": "mfma", "kernel": dom + "_kernel", "achieved": achieved,
                     "peak": PEAK_F32_MATRIX_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_F32_MATRIX_TFLOPS,
                     "traffic": traffic, "avg_launch_ms": dom_ms,
                     "flops_per_launch": STAGE_FLOPS[dom] * N_HYP,
                     "measured": "HIP events on the launch stream around the kernel, one launch over all %d hypotheses, alone "
                                 "on the chip: a second pass over the timed region's frames with one frame in flight (the timed "
                                 "region keeps %d in flight, whose kernels share the chip: rocprofv3's per-kernel average over the whole "
                                 "command mixes both passes, tools/stats_by_pass.py splits the trace -- "
                                 "profiles/r04_kernel_stats_by_pass.txt)" % (N_HYP, n_streams),
                     "whole_step_frac": sum(STAGE_FLOPS.values()) * N_HYP / (elapsed / steps) / 1e12 / PEAK_F32_MATRIX_TFLOPS},
        "stage_ms": {n: round(float(v), 4) for n, v in zip(names, stage_ms)},
        "stage_ms_sum": round(float(sum(stage_ms)), 4),
        "featurize": {"bound": "hbm", "avg_launch_ms": feat_ms, "achieved": feat_bytes / (feat_ms * 1e-3) / 1e9,
                      "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": feat_bytes / (feat_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                      "bytes_per_launch": feat_bytes,
                      "measured": "HIP events around the launch in the one-frame-in-flight pass, like the scorer's stages"},
        "cpu_baseline": base,
        "dtoid": dtoid_out,
    }


def main():
    a = parse()
    world, rank, local = resolve_world(a)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL ("nccl") is the product path. OSSID_BENCH_BACKEND=gloo exists only to rehearse the multi-rank control
        # flow on a box with fewer GPUs than ranks (ranks then share devices; the numbers mean nothing).
        backend = os.environ.get("OSSID_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
            local %= max(1, torch.cuda.device_count())
    assert torch.cuda.is_available(), "bench.py measures the GPU path; no GPU is visible"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from ossid_code_amd import _build, _lib, synth, zephyr
    if rank == 0:
        _build.build_lib()
    if dist is not None:
        dist.barrier()

    # every rank gets its own frame + hypothesis set (frames shard across GPUs)
    d = synth.make_scoring_inputs(N=N_HYP, M=N_PTS, seed=42 + rank, H=IMG_H, W=IMG_W)
    model = synth.random_pn2_state(zephyr.PointNet2SSG(8, Args(), num_class=1), 0).eval()

    base, base_scores = None, None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        base, base_scores = cpu_baseline(d, model, a.cpu_sample)

    model = model.to(dev)
    img = torch.from_numpy(d["img"]).to(dev)
    depth = torch.from_numpy(d["depth"]).to(dev)
    T = torch.from_numpy(d["pose_hypos"].astype(np.float32)).to(dev)
    pts, nrm, col = (torch.from_numpy(d[k].astype(np.float32)).to(dev) for k in
                     ("model_points", "model_normals", "model_colors"))
    K = d["cam_K"]
    cam = tuple(float(np.float32(v)) for v in (K[0, 0], K[1, 1], K[0, 2], K[1, 2]))
    names = _lib.StageEvents.names()
    # The timed region runs the product form of the loop: `--streams` frames in flight on alternating HIP streams
    # (scoring.networkInferenceMany). Per-kernel HIP events are taken in a second pass (below), one frame at a time.
    feat_evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]

    def step(events=None, feat_pair=None):
        rgbd = zephyr.stage_frame(img, depth, dev, blur=True)
        tab = zephyr.stage_model(pts, nrm, col, dev)
        if feat_pair is not None:
            feat_pair[0].record()
        px, uv = zephyr.featurize(rgbd, T, tab, cam, want_uv=True)
        if feat_pair is not None:
            feat_pair[1].record()
        scores = model.score(px, stage_events=events)
        return scores, scores.argmax()

    streams = [torch.cuda.Stream(device=dev) for _ in range(max(1, a.streams))]
    for k in range(max(a.warmup, len(streams))):
        with torch.cuda.stream(streams[k % len(streams)]):
            scores, top = step()
    torch.cuda.synchronize()

    # timed region: EXACTLY K steps (frames) between barrier+synchronize pairs; consecutive frames alternate streams
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        with torch.cuda.stream(streams[i % len(streams)]):
            scores, top = step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # The roofline leg: the same K frames once more, ONE frame in flight, HIP events recorded on the launch stream around
    # every stage (ossid_pn2_score's stage_events_host) and around the featurize kernel -- each kernel alone on the chip,
    # which is what a per-kernel fraction of peak means. (rocprofv3's per-kernel averages over this command mix both
    # passes: launches of the timed region run beside another frame's kernels and take a few per cent longer.)
    clean = [_lib.StageEvents() for _ in range(a.steps)]
    for i in range(a.steps):
        step(clean[i], feat_evs[i])
    torch.cuda.synchronize()
    feat_ms = float(np.mean([s0.elapsed_time(s1) for s0, s1 in feat_evs]))
    stage_ms = np.mean([e.elapsed_ms() for e in clean], axis=0)
    for e in clean:
        e.close()

    top1 = int(top.item())
    got_sample = scores[: len(base_scores)].cpu().numpy() if base_scores is not None else None

    def emit(dtoid_out):
        """rank 0 prints THE json line (everything above is already measured; dtoid_out may be an error note)"""
        if rank != 0:
            return
        if base_scores is not None:   # the GPU scores of the sampled hypotheses are the oracle's, bit for bit
            assert np.array_equal(got_sample, base_scores), "GPU scores differ from the CPU oracle"
        extra = None
        print(json.dumps(make_line(names, stage_ms, feat_ms, elapsed, world, a.steps, a.warmup, len(streams), top1, base,
                                   dtoid_out, extra)), flush=True)

    dtoid_out = None
    if not a.no_dtoid:
        # The secondary measurement must never take the headline number down with it: any exception becomes a note,
        # and a watchdog prints the line and leaves if a collective of the multi-rank finetune leg should ever hang.
        import threading

        def on_timeout():   # a thread, not a signal: the main thread may be blocked inside a collective
            # the headline line is already measured: print it with the error note, then leave NON-ZERO on every rank --
            # a hung collective or GPU must not read as a successful run (no retry, no restart in-process)
            emit({"error": "DTOID leg timed out after %d s" % a.dtoid_timeout})
            os._exit(3)
        watchdog = threading.Timer(a.dtoid_timeout, on_timeout)
        watchdog.daemon = True
        watchdog.start()
        try:
            del scores, top
            torch.cuda.empty_cache()
            dtoid_out = dtoid_leg(a, dev, dist, world)
        except Exception as exc:
            dtoid_out = {"error": repr(exc)[:300]}
        watchdog.cancel()
    emit(dtoid_out)
    failed = isinstance(dtoid_out, dict) and "error" in dtoid_out
    if failed:
        # the headline line is out (it was measured before this leg); a failed secondary leg must still not read as a
        # successful run. Leave at once on this rank -- no teardown handshake -- so that peers blocked in a collective fail
        # fast on the broken connection instead of waiting for their watchdog.
        sys.stdout.flush()
        os._exit(4)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
