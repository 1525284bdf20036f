"""Probe: kernel timeline of the last iterations of a rocprofv3 --kernel-trace run (csv) -- per kernel: launches per iteration, mean
duration, mean start offset within the iteration.    python tools/probes/pass_timeline.py <kernel_trace.csv> [iterations from the end]"""
import csv, sys, collections
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
last = int(sys.argv[2]) if len(sys.argv) > 2 else 20
# an iteration starts at the flow pass's table kernel (its idle direct-evaluation fallback follows it)
starts = [i for i, r in enumerate(rows) if "ff_ode_fwd_kernel" in r[2] and ", 0, true>" in r[2]]
starts = starts[-last - 1:]
acc = collections.OrderedDict()
for a, b in zip(starts[:-1], starts[1:]):
    t0 = rows[a][0]
    for s, e, nm in rows[a:b]:
        k = nm.split("(")[0][:70]
        d = acc.setdefault(k, [0, 0.0, 0.0, 0.0])
        d[0] += 1; d[1] += (e - s) / 1e3; d[2] += (s - t0) / 1e3; d[3] += (e - t0) / 1e3
n = len(starts) - 1
span = (rows[starts[-1]][0] - rows[starts[0]][0]) / 1e3 / n
print(f"{n} iterations, {span:.1f} us each")
for k, (c, d, s, e) in acc.items():
    print(f"{c / n:5.2f} x  dur {d / c:8.1f} us  start {s / c:8.1f}  end {e / c:8.1f}  {k}")
