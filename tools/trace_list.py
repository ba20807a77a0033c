"""The launches of ONE steady-state call, in start order, from a rocprofv3 --kernel-trace csv: offset from the call's first
launch, duration, gap since the previous launch ended (on any stream), stream, grid, LDS, kernel name. The call = the launches
between the last two occurrences of a marker kernel (tools/trace_step.py's convention).
  python tools/trace_list.py gpurun_out/fw/*/*_kernel_trace.csv --marker nms_scan
"""
import argparse
import csv
import re


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--marker", default="amsgrad")
    ap.add_argument("--min-launches", type=int, default=100)
    a = ap.parse_args()
    rows = []
    with open(a.csv) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", "?"),
                         r.get("Grid_Size_X", "?"), r.get("Workgroup_Size_X", "?"), r.get("LDS_Block_Size", "?"),
                         r.get("VGPR_Count", "?")))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if a.marker in r[2].lower()]
    pairs = [(a0, a1) for a0, a1 in zip(marks[:-1], marks[1:]) if a1 - a0 > a.min_launches]
    if not pairs:
        raise SystemExit("no two markers more than %d launches apart" % a.min_launches)
    step = rows[pairs[-1][0] + 1: pairs[-1][1] + 1]
    t0 = step[0][0]
    last_end = t0
    print("%9s %8s %7s %3s %9s %6s %5s  kernel" % ("at us", "dur us", "gap us", "st", "grid", "lds", "vgpr"))
    for s, e, n, st, gx, wx, lds, vg in step:
        short = re.sub(r"\(anonymous namespace\)::", "", n)
        short = re.sub(r"^void ", "", short)[:100]
        print("%9.1f %8.1f %7.1f %3s %9s %6s %5s  %s" % ((s - t0) * 1e-3, (e - s) * 1e-3, (s - last_end) * 1e-3, st, gx, lds, vg, short))
        last_end = max(last_end, e)
    print("span %.1f us, busy %.1f us" % ((step[-1][1] - t0) * 1e-3, sum(e - s for s, e, *_ in step) * 1e-3))


if __name__ == "__main__":
    main()
