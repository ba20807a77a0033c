"""Per-kernel durations of the Zephyr scorer's launches in a rocprofv3 --kernel-trace csv of `bench.py`, split by the pass
they belong to. bench.py launches every scorer kernel W (warm-up) + K (timed region: `--streams` frames in flight, so a
launch shares the chip with another frame's kernels) + K (roofline pass: one frame in flight, each kernel alone -- what
`roofline.avg_launch_ms` of the JSON line is measured on) times; rocprofv3's --stats average mixes the three.
  python tools/stats_by_pass.py trace.csv --steps 20 --warmup 5"""
import argparse
import collections
import csv
import re

ap = argparse.ArgumentParser()
ap.add_argument("csv")
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--warmup", type=int, default=5)
a = ap.parse_args()
names = ("fps_reg_kernel", "ball_query_kernel", "sa1_kernel", "p2_kernel", "sa2_kernel", "sa3_kernel", "fc_head_kernel", "featurize_kernel")
rows = collections.defaultdict(list)
with open(a.csv) as f:
    for r in csv.DictReader(f):
        m = re.search(r"::(\w+)[<(]", r["Kernel_Name"])
        if m and m.group(1) in names:
            rows[m.group(1)].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
print("%-20s %6s | %-28s | %-28s | %-28s" % ("kernel", "calls", "warm-up  avg / min / max ms", "timed region (frames in flight)", "roofline pass (one frame)"))
for n in names:
    v = sorted(rows.get(n, []))
    per = len(v) // (a.warmup + 2 * a.steps) if v else 0          # launches per frame (fps / ball query run twice)
    if not per:
        continue
    d = [x[1] * 1e-6 for x in v]
    segs = (d[: a.warmup * per], d[a.warmup * per: (a.warmup + a.steps) * per], d[(a.warmup + a.steps) * per:])
    fmt = lambda s: "%7.3f / %7.3f / %7.3f" % (sum(s) / len(s) * per, min(s), max(s)) if s else "-"      # noqa: E731
    print("%-20s %6d | %-28s | %-28s | %-28s" % (n, len(v), fmt(segs[0]), fmt(segs[1]), fmt(segs[2])))
print("(avg = per frame: summed over the kernel's launches of one frame; min / max per launch)")
