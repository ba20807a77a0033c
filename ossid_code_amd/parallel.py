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
