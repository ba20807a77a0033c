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


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from ossid_code_amd.dtoid import finetune
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)                     # replicas start DIFFERENT on purpose
    net = torch.nn.Sequential(torch.nn.Linear(13, 7), torch.nn.ReLU(), torch.nn.Linear(7, 3))
    flat = finetune.FlatParams(net, unused_filter=lambda n: False)
    sync = finetune.GradSync(flat, bucket_mb=1e-4)    # tiny buckets: several all-reduces in flight
    assert len(sync.bounds) > 1
    sync.broadcast_params(0)
    x = torch.randn(5, 13, generator=torch.Generator().manual_seed(7 + rank))   # each rank its own shard
    flat.zero_grad()
    net(x).square().mean().backward()
    local = flat.grad.clone()
    sync.sync()
    q.put((rank, flat.param.clone(), local, flat.grad.clone()))
    dist.barrier()
    dist.destroy_process_group()


def test_grad_sync_averages_over_ranks():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, p0, l0, g0), (_, p1, l1, g1) = res
    assert torch.equal(p0, p1)                                   # broadcast made the replicas identical
    assert torch.allclose(g0, (l0 + l1) / 2, rtol=1e-6, atol=1e-7)   # mean of the per-rank gradients
    assert torch.equal(g0, g1)                                   # and identical on both ranks


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
