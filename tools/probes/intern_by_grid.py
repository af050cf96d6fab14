"""per (kernel, grid size) averages of the last step of a rocprofv3 kernel trace -- which level of InternImage a kernel's time sits in:
python tools/probes/intern_by_grid.py t_kernel_trace.csv [needle ...]"""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
whole = "--all" in sys.argv        # every launch of the trace instead of the last step
top = "--top" in sys.argv          # also: the 40 longest single launches of the step that are not GEMMs / DCNv3
args = [a for a in sys.argv[2:] if a not in ("--all", "--top", "--gemms")]
idx = [i for i, r in enumerate(rows) if "adamw" in r["Kernel_Name"].lower()]
step = rows if whole else rows[idx[-2] + 1: idx[-1] + 1]
needles = args or ["dcnv3", "ln_res", "dwconv", "softmax_groups"]
c, t = collections.Counter(), collections.Counter()
for r in step:
    n = r["Kernel_Name"]
    if any(k in n for k in needles):
        m = re.search(r"(\w+)(<[^(]*>)?\(", n.replace("(anonymous namespace)::", ""))
        key = ((m.group(1) + (m.group(2) or "")) if m else n[:60], int(r["Grid_Size_X"]), int(r["Workgroup_Size_X"]), r.get("VGPR_Count"), r.get("LDS_Block_Size"))
        c[key] += 1
        t[key] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in step)
print("last step: %d launches, kernel time %.2f ms" % (len(step), tot / 1e6))
for key in sorted(c, key=lambda k: (k[0], -k[1])):
    print("%-44s grid %9d wg %4d vgpr %4s lds %6s  x%3d  avg %8.1f us  sum %7.3f ms" % (key + (c[key], t[key] / c[key] / 1e3, t[key] / 1e6)))

if top:
    def short(n):
        m = re.search(r"(\w+)(<[^(]*>)?\(", n.replace("(anonymous namespace)::", ""))
        return ((m.group(1) + (m.group(2) or "")) if m else n)[:70]
    rest = [r for r in step if ("gemm_nt" not in r["Kernel_Name"] and "gemm_tn" not in r["Kernel_Name"] and "dcnv3" not in r["Kernel_Name"]) or "--gemms" in sys.argv]
    rest.sort(key=lambda r: int(r["Start_Timestamp"]) - int(r["End_Timestamp"]))
    print("longest single launches (GEMMs and DCNv3 aside):")
    for r in rest[:40]:
        print("  %8.1f us  grid %9s  %s" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Grid_Size_X"], short(r["Kernel_Name"])))
