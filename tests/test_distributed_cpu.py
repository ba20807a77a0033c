"""world_size-2 gloo tests (CPU): the data-parallel gradient averaging of the finetune step and the frame sharding
of the scoring path -- the N>1 logic of bench.py / finetune.GradSync, exercised without GPUs."""
import os
import socket
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _BNNet(torch.nn.Module):
    """A detector-shaped stand-in: conv -> BatchNorm (train mode: per-rank batch statistics + running buffers) -> ReLU
    -> conv, returning the {"loss": ...} dict finetune_step expects."""

    def __init__(self):
        super().__init__()
        self.c1 = torch.nn.Conv2d(3, 6, 3, padding=1)
        self.bn = torch.nn.BatchNorm2d(6)
        self.c2 = torch.nn.Conv2d(6, 2, 3, padding=1)
        self.unused = torch.nn.Linear(3, 3)          # never runs: gets no gradient (like the SqueezeNet classifier)

    def forward(self, batch):
        y = self.c2(torch.relu(self.bn(self.c1(batch["x"]))))
        return {"loss": (y - batch["t"]).square().mean()}


def _worker(rank, world, port, q, overlap):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from ossid_code_amd.dtoid import finetune
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)                     # replicas start DIFFERENT on purpose (weights AND BN buffers)
    net = _BNNet().train()
    with torch.no_grad():
        net.bn.running_mean.add_(float(rank))
    flat = finetune.FlatParams(net, unused_filter=lambda n: n.startswith("unused"))
    sync = finetune.GradSync(flat, bucket_mb=1e-4, model=net, overlap=overlap)    # tiny buckets: several in flight
    assert len(sync.bounds) > 1 and len(sync._buckets) > 1
    sync.broadcast_params(0)
    start = {k: v.clone() for k, v in net.state_dict().items()}
    opt = torch.optim.Adam([p for _, p in flat.entries], lr=1e-2, amsgrad=True)
    g = torch.Generator().manual_seed(7 + rank)       # each rank its own slice of the global batch
    batch = {"x": torch.randn(4, 3, 8, 8, generator=g) * (1 + rank), "t": torch.randn(4, 2, 8, 8, generator=g)}
    # local gradient of this rank, for the mean check (a plain backward on a copy)
    import copy
    ref = copy.deepcopy(net)
    ref(batch)["loss"].backward()
    local = torch.cat([p.grad.reshape(-1) for n, p in ref.named_parameters() if not n.startswith("unused")])
    losses = [float(finetune.finetune_step(net, batch, opt, sync))]
    g_first = flat.grad[: local.numel()].clone()      # the exchanged gradient of step 1 (same weights as `ref`)
    losses += [float(finetune.finetune_step(net, batch, opt, sync)) for _ in range(2)]
    npy = lambda d: {k: v.numpy().copy() for k, v in d.items()}  # noqa: E731  (plain pickles: no shared-memory handles)
    q.put((rank, npy(start), local.numpy().copy(), g_first.numpy().copy(), npy(net.state_dict()), losses))
    dist.barrier()
    dist.destroy_process_group()


def _run_ddp(overlap):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, overlap)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return res


def test_grad_sync_averages_over_ranks_and_replicas_stay_identical():
    """Data-parallel finetune steps on per-rank batch slices through a train-mode BatchNorm: parameters AND buffers
    (running_mean / running_var / num_batches_tracked) are bit-identical on both ranks before and after (VERDICT r1:
    BN buffers were never synchronised), with the hook-driven overlapped exchange and with the plain one, and both
    exchanges give the same numbers."""
    out = {}
    for overlap in (True, False):
        (_, s0, l0, g0, e0, loss0), (_, s1, l1, g1, e1, loss1) = _run_ddp(overlap)
        for k in s0:
            assert np.array_equal(s0[k], s1[k]), ("start", k)            # broadcast_params: weights and buffers
        for k in e0:
            assert np.array_equal(e0[k], e1[k]), ("end", k, overlap)       # after 3 steps: still the same detector
        assert np.abs(e0["bn.running_mean"]).sum() > 0 and int(e0["bn.num_batches_tracked"]) == 3
        assert not np.array_equal(l0, l1)                                # the ranks really saw different data
        assert np.allclose(g0, (l0 + l1) / 2, rtol=1e-5, atol=1e-7)      # mean of the per-rank gradients
        assert np.array_equal(g0, g1)                                    # and identical on both ranks
        assert np.abs(e0["unused.weight"] - s0["unused.weight"]).max() == 0   # no gradient -> untouched, as Adam does
        out[overlap] = (e0, g0)
    for k in out[True][0]:
        assert np.array_equal(out[True][0][k], out[False][0][k]), k      # overlapped == non-overlapped, bit for bit


def test_frame_sharding_is_a_partition():
    """bench.py gives rank r the frames r, r+N, ...; together the ranks cover every frame exactly once."""
    from ossid_code_amd.parallel import shard_frames
    for n_frames in (0, 1, 7, 8, 100):
        for world in (1, 2, 8):
            parts = [shard_frames(n_frames, r, world) for r in range(world)]
            flat = sorted(i for p in parts for i in p)
            assert flat == list(range(n_frames))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def _sequential_reference(n_frames, interval, confident_of):
    """The reference's single-process loop: the detector version a frame sees = number of finetunes before it."""
    version, train, nxt, seen = 0, 0, interval, []
    for f in range(n_frames):
        c = confident_of(f, version)
        seen.append((f, version, c))
        if c:
            train += 1
            if train == nxt:
                version += 1
                nxt += interval
    return seen


def test_speculative_window_reproduces_sequential_semantics():
    """8 'GPUs' scoring windows of frames with frozen weights, in-order commit, re-issue after a finetune: every frame
    ends up scored with exactly the detector version the sequential loop would have used."""
    import random
    from ossid_code_amd.stream import SpeculativeWindow
    for seed in range(20):
        rng = random.Random(seed)
        table = {}

        def confident_of(f, version):
            return table.setdefault((f, version), rng.random() < 0.45)
        n_frames, interval = 97, 5
        for world in (1, 2, 8):
            win = SpeculativeWindow(n_frames, world, interval)
            version, seen = 0, []
            while not win.done:
                frames = win.window()
                flags = [confident_of(f, version) for f in frames]          # all ranks use the CURRENT weights
                before = len(win.committed)
                fired = win.commit(flags)
                for f, c in win.committed[before:]:
                    seen.append((f, version, c))
                if fired is not None:
                    version += 1                                            # DDP finetune on all ranks
            assert seen == _sequential_reference(n_frames, interval, confident_of), (seed, world)
            assert [f for f, _, _ in seen] == list(range(n_frames))          # every frame committed once, in order


def _stream_worker(rank, world, port, n_frames, interval, q):
    import torch
    import torch.distributed as dist
    from ossid_code_amd.stream import run_speculative
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    version = torch.zeros(1)                                   # stands for the replicated detector weights
    seen_by_me = []

    def process(frame):
        conf = ((frame * 7 + int(version.item()) * 3) % 5) < 2
        seen_by_me.append((frame, int(version.item())))
        return conf, {"x": torch.full((2, 3), float(frame)), "v": version.clone()}

    sets = []

    def finetune(train):
        # every rank must hold the same frame-ordered set, with the payload produced by whichever rank scored the frame
        sets.append([(f, float(s["x"][0, 0]), float(s["v"][0])) for f, s in train])
        g = torch.ones(1)
        dist.all_reduce(g)                                     # the data-parallel step: everyone takes part
        version.add_(g / world)

    committed, win = run_speculative(list(range(n_frames)), process, finetune, interval, dist)
    q.put((rank, committed, sets, seen_by_me, win.discarded))
    dist.destroy_process_group()


def test_speculative_stream_gloo_world2_matches_sequential():
    import torch.multiprocessing as mp
    n_frames, interval = 41, 4
    # sequential truth
    version, train, nxt, truth, sets = 0, [], interval, [], []
    for f in range(n_frames):
        c = ((f * 7 + version * 3) % 5) < 2
        truth.append((f, c))
        if c:
            train.append((f, float(f), float(version)))
            if len(train) == nxt:
                sets.append(list(train))
                version += 1
                nxt += interval
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_stream_worker, args=(r, 2, port, n_frames, interval, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(30)
    for rank, committed, rsets, seen, discarded in got:
        assert committed == truth, rank
        assert rsets == sets, rank
    assert got[0][4] == got[1][4] and got[0][4] > 0           # some speculated frames were thrown away and re-issued


def _bn_frame(f):
    g = torch.Generator().manual_seed(1000 + f)
    return {"x": torch.randn(1, 3, 8, 8, generator=g), "t": torch.randn(1, 2, 8, 8, generator=g)}


def _bn_confident(net, frame):
    net.eval()
    with torch.no_grad():
        return float(net(frame)["loss"]) < 1.45


def _bn_collate(frames):
    return {k: torch.cat([f[k] for f in frames]) for k in ("x", "t")}


def _bn_stream_worker(rank, world, port, n_frames, interval, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from ossid_code_amd.dtoid import finetune
    from ossid_code_amd.stream import run_speculative
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    torch.manual_seed(5)
    net = _BNNet()
    flat = finetune.FlatParams(net, unused_filter=lambda n: n.startswith("unused"))
    sync = finetune.GradSync(flat, bucket_mb=1e-4, model=net)
    sync.broadcast_params(0)
    opt = torch.optim.Adam([p for _, p in flat.entries], lr=3e-2, amsgrad=True)

    def process(frame):
        return _bn_confident(net, frame), {"x": frame["x"], "t": frame["t"]}

    def ft(train):
        net.train()
        items = [s for _, s in train][-4:]                      # the last 4 pseudo-labelled frames = one global batch
        finetune.finetune_step(net, _bn_collate(items[rank::world]), opt, sync)
        net.eval()

    committed, win = run_speculative([_bn_frame(f) for f in range(n_frames)], process, ft, interval, dist)
    q.put((rank, committed, {k: v.numpy().copy() for k, v in net.state_dict().items()}, win.discarded))
    dist.destroy_process_group()


def test_speculative_stream_with_batchnorm_detector_matches_sequential():
    """The real detector has BatchNorm: after a data-parallel finetune every rank must hold the SAME eval-mode network
    (parameters and running statistics), or the speculative stream commits frames scored by different detectors.
    Two gloo ranks vs. the sequential loop running the same finetune function (the two per-rank slices evaluated one
    after the other with per-slice batch statistics, gradient mean, rank 0's buffers): same confident flags, same
    final state_dict, bit for bit."""
    import copy
    n_frames, interval, world = 30, 4, 2
    torch.manual_seed(5)
    net = _BNNet()
    params = [p for n, p in net.named_parameters() if not n.startswith("unused")]
    opt = torch.optim.Adam(params, lr=3e-2, amsgrad=True)
    truth, train, nxt = [], [], interval
    for f in range(n_frames):
        fr = _bn_frame(f)
        c = _bn_confident(net, fr)
        truth.append((f, c))
        if c:
            train.append(fr)
            if len(train) == nxt:
                nxt += interval
                items = train[-4:]
                grads, bufs = [], None
                for r in range(world):                         # what rank r computes on its slice
                    rep = copy.deepcopy(net).train()
                    rep(_bn_collate(items[r::world]))["loss"].backward()
                    grads.append([p.grad for n, p in rep.named_parameters() if not n.startswith("unused")])
                    if r == 0:
                        bufs = [b.clone() for b in rep.buffers()]
                for p, g0, g1 in zip(params, *grads):
                    p.grad = g0 * (1.0 / world) + g1 * (1.0 / world)
                opt.step()
                with torch.no_grad():
                    for b, v in zip(net.buffers(), bufs):
                        b.copy_(v)
    assert sum(c for _, c in truth) >= 2 * interval            # at least two finetunes really happened
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bn_stream_worker, args=(r, world, port, n_frames, interval, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    want = {k: v.numpy() for k, v in net.state_dict().items()}
    for rank, committed, sd, discarded in got:
        assert committed == truth, rank
        for k in want:
            assert np.array_equal(sd[k], want[k]), (rank, k)
    assert int(want["bn.num_batches_tracked"]) >= 2


def _top1_worker(rank, world, port, scores, q):
    import torch.distributed as dist
    from ossid_code_amd import parallel
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    out = []
    for s in scores:
        lo, hi = parallel.shard_hypotheses(len(s), rank, world)
        out.append(parallel.reduce_top1(s[lo:hi], lo, dist))
    q.put((rank, out))
    dist.destroy_process_group()


def test_within_frame_hypothesis_sharding_top1_matches_unsharded_argmax():
    """Split a frame's hypotheses over 3 ranks, exchange 8 bytes per rank: same (max, argmax) as the unsharded array,
    including ties (lowest index wins), fewer hypotheses than ranks, and an empty frame."""
    import torch.multiprocessing as mp
    from ossid_code_amd import parallel
    rng = np.random.default_rng(0)
    cases = [rng.normal(size=1000).astype(np.float32), np.array([1.0, 5.0, 5.0, 2.0, 5.0], np.float32),
             np.array([3.0, 7.0], np.float32), np.zeros(0, np.float32), rng.normal(size=7).astype(np.float32)]
    assert [parallel.shard_hypotheses(10, r, 3) for r in range(3)] == [(0, 4), (4, 7), (7, 10)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_top1_worker, args=(r, 3, port, cases, q)) for r in range(3)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(30)
    want = [(float(s.max()), int(s.argmax())) if len(s) else (float("-inf"), -1) for s in cases]
    for rank, out in got:
        assert out == want, (rank, out, want)
