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


# ---- `--xl`: InternImage-XL at 512^2, batch 8, bf16 -- algorithmic bytes of the HBM-bound families per level (every operand and result of a launch once) against their
# time: the table of profiles/r06_internimage_hbm_bound_kernels.txt.  Levels are told apart by the time-ordered position of a family's launches in the step (the schedule is
# fixed: forward levels 0 -> 3 then backward 3 -> 0), not by grid size (the LayerNorm grids are capped).
if "--xl" in sys.argv:
    B, S = 8, 512
    L = [dict(rows=B * (S // (4 << i)) ** 2, C=192 << i, G=12 << i, depth=d) for i, d in enumerate((5, 5, 24, 5))]
    fam = collections.defaultdict(list)
    for r in step:
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "")
        m = re.search(r"(\w+)(<[^(]*>)?\(", n)
        fam[(m.group(1) + (m.group(2) or "")) if m else n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)

    def split(times, per_layer, forward):
        """launch times of one family -> {level: [us]}: per_layer launches per layer, levels in schedule order"""
        out, k = {}, 0
        order = range(4) if forward else range(3, -1, -1)
        for i in order:
            n = per_layer * L[i]["depth"]
            out[i] = times[k:k + n]
            k += n
        return out, k
    rowsfmt = "%-44s level %d  %6d x %4d  x%3d  %8.1f us  %8.1f MB  %5.2f TB/s  %4.2f"
    print("# InternImage-XL 512^2 B = 8 bf16: HBM-bound families per level -- launches, us per launch, algorithmic MB per launch, TB/s, fraction of 8 TB/s")
    spec = [   # (kernel-name prefix, launches per layer, forward?, bytes per launch as f(level))
        ("ln_res_fwd_kernel", 2, True, lambda l: l["rows"] * l["C"] * 12),          # h bf16 + x f32 + out f32 + out bf16
        ("ln_res_bwd_kernel", 2, False, lambda l: l["rows"] * l["C"] * 8),          # dout f32 + h bf16 + dh bf16
        ("dcnv3_fwd9_kernel", 1, True, lambda l: l["rows"] * (l["C"] * 4 + l["G"] * 27 * 2)),
        ("dcnv3_bwd_om_kernel", 1, False, lambda l: l["rows"] * (l["C"] * 4 + l["G"] * 27 * 2 + l["G"] * 27 * 4 + l["G"] * 18 * 2)),
        ("dcnv3_bwd_input_kernel", 1, False, lambda l: l["rows"] * (l["C"] * 2 + l["G"] * 27 * 2 + l["C"] * 4)),
        ("softmax_groups_fwd_kernel", 1, True, lambda l: l["rows"] * l["G"] * 9 * 4),
        ("softmax_groups_bwd_kernel", 1, False, lambda l: l["rows"] * l["G"] * 9 * (2 + 4 + 2)),
        ("dwconv3x3_p8_kernel<bf16_t, false>", 1, True, lambda l: l["rows"] * l["C"] * 4),
        ("dwconv3x3_p8_kernel<float, true>", 1, False, lambda l: l["rows"] * l["C"] * (2 + 8)),      # dy bf16, f32 accumulate (read + write)
    ]
    for prefix, per_layer, fwd, nbytes in spec:
        # (families with a width template parameter come as several names: merge them back into schedule order by sorting on start time)
        sel = [r for r in step if prefix.split("<")[0] in r["Kernel_Name"] and (("<" not in prefix) or prefix.split("<")[1].rstrip(">") in r["Kernel_Name"])]
        sel.sort(key=lambda r: int(r["Start_Timestamp"]))
        times = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in sel]
        need = per_layer * sum(l["depth"] for l in L)
        if len(times) < need:
            print("# %s: %d launches in the step, %d expected -- skipped" % (prefix, len(times), need))
            continue
        times = times[:need] if fwd else times[-need:]
        lv, _ = split(times, per_layer, fwd)
        for i in range(4):
            t = sum(lv[i]) / len(lv[i])
            mb = nbytes(L[i]) / 1e6
            print(rowsfmt % (prefix, i, L[i]["rows"], L[i]["C"], len(lv[i]), t, mb, mb / t, mb / t / 8.0))      # (1 MB per us = 1 TB/s)
