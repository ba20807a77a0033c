"""Per-kernel time of ONE steady-state step from a rocprofv3 --kernel-trace csv: the launches between the last two
occurrences of a marker kernel (default: the AMSGrad step, which ends every finetune iteration). Avoids MIOpen's
find-phase kernels, which dominate --stats for short runs.
  python tools/trace_step.py gpurun_out/prof/*/*_kernel_trace.csv [--marker amsgrad] [--top 40]
"""
import argparse
import collections
import csv
import json
import re


def category(name):
    n = name.lower()
    rules = [("wgrad", r"wrw|wgrad|bwdwrw|backward_weight|bwd_weights"), ("dgrad", r"bwd|backward_data|dgrad"),
             ("conv_fwd", r"conv|igemm|winograd|gemm|cijk|sp3|xdlops"), ("batchnorm", r"batchnorm|batch_norm|bn"),
             ("pool/upsample", r"pool|upsample|interp"), ("cat/copy", r"cat|copy|fill|memset|memcpy"),
             ("elementwise", r"elementwise|vectorized|unrolled"), ("reduce", r"reduce|softmax|sum"),
             ("ossid", r"ossid|dw_xcorr|amsgrad|nms|decode")]
    for cat, pat in rules:
        if re.search(pat, n):
            return cat
    return "other"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--marker", default="amsgrad")
    ap.add_argument("--top", type=int, default=40)
    ap.add_argument("--min-launches", type=int, default=100, help="ignore marker pairs closer than this (timing loops)")
    a = ap.parse_args()
    rows = []
    with open(a.csv) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if a.marker in r[2].lower()]
    if len(marks) < 2:
        raise SystemExit("marker kernel seen %d times" % len(marks))
    pairs = [(a0, a1) for a0, a1 in zip(marks[:-1], marks[1:]) if a1 - a0 > a.min_launches]
    if not pairs:
        raise SystemExit("no two markers more than %d launches apart" % a.min_launches)
    step = rows[pairs[-1][0] + 1: pairs[-1][1] + 1]
    wall = (step[-1][1] - step[0][0]) * 1e-6
    by, cat = collections.Counter(), collections.Counter()
    cnt = collections.Counter()
    for s, e, n in step:
        short = re.sub(r"<.*", "", n)[:90]
        if "elementwise" in short or "reduce_kernel" in short:      # keep the functor: it says which torch op this is
            inner = re.findall(r"at::native::(?:\(anonymous namespace\)::)?([A-Za-z_0-9]+)", n[n.find("<"):])
            short = short.split("::")[-1] + " : " + ",".join(inner[:3])
        by[short] += (e - s) * 1e-6
        cnt[short] += 1
        cat[category(n)] += (e - s) * 1e-6
    busy = sum(by.values())
    print(json.dumps({"launches": len(step), "wall_ms": wall, "busy_ms": busy,
                      "by_category_ms": dict(cat.most_common())}, indent=1))
    for n, t in by.most_common(a.top):
        print("%8.3f ms %5d x  %s" % (t, cnt[n], n))


if __name__ == "__main__":
    main()
