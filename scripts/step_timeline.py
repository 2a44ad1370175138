"""Dev aid: the kernels of ONE replayed step in launch order, from a rocprofv3 --kernel-trace csv.

usage: step_timeline.py <x_kernel_trace.csv> [which]      (which: index of the step among the adamw_kernel-delimited spans, default: the median-length one)
Prints start offset, duration and the idle gap before every kernel, then the totals.
"""
import csv, sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if "adamw_kernel" in r["Kernel_Name"]]
spans = [(ends[i] + 1, ends[i + 1] + 1) for i in range(len(ends) - 1)]
spans = [s for s in spans if s[1] - s[0] > 20]
length = lambda s: int(rows[s[1] - 1]["End_Timestamp"]) - int(rows[s[0]]["Start_Timestamp"])
if len(sys.argv) > 2:
    span = spans[int(sys.argv[2])]
else:
    span = sorted(spans, key=length)[len(spans) // 4]       # a fast (replayed) one, not the fastest
t0 = int(rows[span[0]]["Start_Timestamp"])
prev_end = t0
busy = gaps = 0
for r in rows[span[0]:span[1]]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"].replace("void dmp::(anonymous namespace)::", "").replace("dmp::(anonymous namespace)::", "").replace("void at::native::", "at::")
    n = n.split("(")[0][:70]
    gap = s - prev_end
    q = "q%s/s%s" % (r.get("Queue_Id", "?"), r.get("Stream_Id", "?"))
    print("%8.1f  %7.1f us  gap %6.1f  %-8s grid %-8s %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap / 1e3, q, r.get("Grid_Size", r.get("Grid_Size_X", "")), n))
    busy += e - s
    gaps += max(gap, 0)
    prev_end = max(prev_end, e)
print("kernels %d  busy %.1f us  gaps %.1f us  span %.1f us  (spans seen: %d)" % (span[1] - span[0], busy / 1e3, gaps / 1e3, (prev_end - t0) / 1e3, len(spans)))
