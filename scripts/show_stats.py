"""Dev aid: per-step view of a rocprofv3 kernel-stats csv (calls / step, avg us, us per step)."""
import csv, sys
path = sys.argv[1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else None
rows = list(csv.DictReader(open(path)))
if steps is None:      # adamw_kernel runs once per step
    steps = next(float(r["Calls"]) for r in rows if "adamw_kernel" in r["Name"])
tot = sum(float(r["TotalDurationNs"]) for r in rows) / steps / 1e3
calls = sum(float(r["Calls"]) for r in rows) / steps
small = sum(float(r["TotalDurationNs"]) for r in rows if float(r["AverageNs"]) < 15000) / steps / 1e3
nsmall = sum(float(r["Calls"]) for r in rows if float(r["AverageNs"]) < 15000) / steps
print("steps %d  kernel time per step %.1f us  launches per step %.1f  (< 15 us: %.1f launches, %.1f us)" % (steps, tot, calls, nsmall, small))
for r in rows:
    c = float(r["Calls"]) / steps
    if c < 0.05:
        continue
    n = r["Name"].replace("void dmp::(anonymous namespace)::", "").replace("dmp::(anonymous namespace)::", "")
    print("%-90s x%5.2f avg %7.1f us  = %7.1f us/step" % (n[:90], c, float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / steps / 1e3))
