"""names and counts of the kernels of the LAST step in a rocprofv3 kernel trace (steps are delimited by the optimizer launch):  python tools/probes/last_step_kernels.py t_kernel_trace.csv [needle]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "adamw" in r["Kernel_Name"].lower()]
step = rows[idx[-2] + 1: idx[-1] + 1]
c = collections.Counter()
t = collections.Counter()
for r in step:
    n = r["Kernel_Name"][:70]
    c[n] += 1
    t[n] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print(len(step), "launches in the last step")
for n, k in c.most_common(60):
    if len(sys.argv) < 3 or sys.argv[2] in n:
        print("%5d %9.1f us  %s" % (k, t[n] / 1e3, n))
if len(sys.argv) > 2:
    for i, r in enumerate(step):
        if sys.argv[2] in r["Kernel_Name"]:
            print(i, step[i - 1]["Kernel_Name"][:60], "->", r["Kernel_Name"][:40], r["Grid_Size_X"], "->", step[i + 1]["Kernel_Name"][:60] if i + 1 < len(step) else "")
