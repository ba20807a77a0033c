"""HIP side streams that really run beside the main one (shared by the DTOID finetune step and the Zephyr scorer).

HIP maps streams onto a few hardware queues in order of creation and two streams on one queue serialise, so whether a side
stream created "now" overlaps with the main stream depends on how many streams the process created before (measured in round
2: the same finetune step 42.0 or 48.4 ms). The streams are therefore CHOSEN by a probe, once per device.
"""
import os

import torch

# Which HIP streams actually run beside the main one. HIP maps streams onto a few hardware queues (4 by default) in order of
# creation, and two streams on one queue serialise: whether a side stream created "now" shares the main stream's queue depends
# on how many streams the process created before (measured: the same step 42.0 or 48.4 ms depending on whether a test-time graph
# had been captured first). So the side streams are CHOSEN, once per device: a handful of streams created back to back spread over the
# queues; each is probed -- a long kernel on the main stream, a short one on the candidate, did the short one finish
# first? -- and three that run beside the main stream AND beside each other become the weight-gradient stream and the two branch slots.
_side_pools = {}
N_STREAM_CANDIDATES = int(os.environ.get("OSSID_STREAM_CANDIDATES", "8"))


def side_streams(device):
    """{"wgrad": Stream, "b0": Stream, "b1": Stream} for `device`, chosen by the probe above (cached)."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    pool = _side_pools.get(idx)
    if pool is not None:
        return pool
    if torch.cuda.is_current_stream_capturing() or not hasattr(torch.cuda, "_sleep"):
        # no probing inside a capture (or without torch's spin kernel): plain streams. Cached as well -- scratch buffers are
        # keyed by stream handle, so a fresh set per call would leak a stream and >= 1 MB of scratch per call -- but under
        # its own key, so that a later call outside the capture still probes.
        pool = _side_pools.get((idx, "plain"))
        if pool is None:
            pool = _side_pools[(idx, "plain")] = {k: torch.cuda.Stream(device=idx) for k in ("wgrad", "b0", "b1")}
        return pool
    with torch.cuda.device(idx):
        main = torch.cuda.current_stream(idx)
        cands = [torch.cuda.Stream(device=idx) for _ in range(N_STREAM_CANDIDATES)]
        small = torch.zeros(64, dtype=torch.float32, device=device)
        torch.cuda._sleep(1000)                              # (loads the spin kernel)
        torch.cuda.synchronize(idx)
        def runs_beside(busy, cand):
            """A ~2 ms one-thread spin kernel on `busy` (occupies its hardware queue and nothing else), a tiny kernel on
            `cand` with no dependency on it: is the tiny one done while the spin still runs?"""
            ev_c, ev_b = torch.cuda.Event(), torch.cuda.Event()
            with torch.cuda.stream(busy):
                torch.cuda._sleep(4000000)
                ev_b.record(busy)
            with torch.cuda.stream(cand):
                small.add_(1.0)
                ev_c.record(cand)
            ev_c.synchronize()
            beside = not ev_b.query()
            torch.cuda.synchronize(idx)
            return beside

        good = [c for c in cands if runs_beside(main, c)]
        chosen = []
        for c in good:                                       # ... and beside each other
            if all(runs_beside(x, c) for x in chosen):
                chosen.append(c)
            if len(chosen) == 3:
                break
        good = chosen + [c for c in good if c not in chosen]
    rest = [c for c in cands if c not in good]
    order = good + rest                                      # fewer than three concurrent ones: take what there is
    pool = {"wgrad": order[0], "b0": order[1], "b1": order[2], "concurrent": len(good), "mutual": len(chosen)}
    _side_pools[idx] = pool
    return pool
