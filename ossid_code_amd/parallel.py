"""Multi-GPU layout of the hot path: one process per GPU (torch.distributed, backend "nccl" = RCCL
over xGMI). Zephyr scoring and DTOID test-time forward shard by FRAME with no data-path collective; the DTOID
finetune step is data-parallel with one gradient all-reduce per step (dtoid.finetune.GradSync)."""
import os

import torch


def shard_frames(n_frames, rank, world):
    """Frame indices of `rank`: round-robin, so consecutive (time-adjacent) frames land on different GPUs."""
    return list(range(rank, n_frames, world))


def init_from_env(backend="nccl"):
    """(rank, world, local_rank, dist-or-None) from torchrun's environment; single-process when WORLD_SIZE is unset."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank, local = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1:
        return rank, world, local, None
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend == "nccl":
        torch.cuda.set_device(local)
        dist.init_process_group(backend, device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(backend)
    return rank, world, local, dist


def shard_hypotheses(n, rank, world):
    """[lo, hi) of rank's contiguous slice of a frame's n hypotheses (sizes differ by at most one)."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def reduce_top1(local_scores, lo, dist=None, group=None):
    """Within-frame sharding: every rank scored hypotheses [lo, lo + len(local_scores)) of one frame; the only
    exchange is one all_gather of (max score, global argmax) = 8 bytes per rank. Returns (best_score, best_index), the
    same on every rank; ties go to the lowest index, as numpy/torch argmax over the unsharded array would.
    An empty shard (more ranks than hypotheses, or everything filtered out) takes no part."""
    s = torch.as_tensor(local_scores, dtype=torch.float32).reshape(-1)
    if s.numel():
        m, i = torch.max(s, 0)
        mine = torch.stack([m.double(), (i + lo).double()])
    else:
        mine = torch.tensor([float("-inf"), float("inf")], dtype=torch.float64, device=s.device)
    if dist is None or not dist.is_initialized() or dist.get_world_size(group) == 1:
        allr = mine[None]
    else:
        if dist.get_backend(group) == "nccl":
            mine = mine.cuda()
        parts = [torch.zeros_like(mine) for _ in range(dist.get_world_size(group))]
        dist.all_gather(parts, mine, group=group)
        allr = torch.stack(parts).cpu()
    allr = allr.cpu()
    best = allr[:, 0].max()
    idx = allr[allr[:, 0] == best, 1].min()
    return float(best), (int(idx) if idx != float("inf") else -1)
