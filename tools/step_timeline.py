"""Coarse timeline of ONE finetune step from a rocprofv3 --kernel-trace csv: per 1-ms bin the busy fraction of every HIP
stream and the main stream's largest kernel -- where the critical path waits.  python tools/step_timeline.py trace.csv"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "amsgrad" in r["Kernel_Name"]]
fr = rows[idx[-2] + 1: idx[-1] + 1]
t0 = int(fr[0]["Start_Timestamp"])
t1 = int(fr[-1]["End_Timestamp"])
streams = sorted({r["Stream_Id"] for r in fr}, key=int)
nb = (t1 - t0) // 1000000 + 1
busy = {s: [0.0] * nb for s in streams}
top = [collections.Counter() for _ in range(nb)]
for r in fr:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    b = s // 1000000
    while b * 1000000 < e and b < nb:
        lo, hi = max(s, b * 1000000), min(e, (b + 1) * 1000000)
        busy[r["Stream_Id"]][b] += (hi - lo) / 1e6
        if r["Stream_Id"] == streams[0]:
            n = re.sub(r"\(anonymous namespace\)::|^void ", "", r["Kernel_Name"]).split("(")[0][:34]
            top[b][n] += (hi - lo) / 1e6
        b += 1
print("span %.1f ms, streams %s (first = main)" % ((t1 - t0) / 1e6, streams))
for b in range(nb):
    print("%3d ms | " % b + " ".join("%4.2f" % busy[s][b] for s in streams) + " | " +
          ", ".join("%s %.2f" % kv for kv in top[b].most_common(2)))
