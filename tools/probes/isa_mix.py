"""static instruction mix of one kernel in a `hipcc -S` listing: python tools/probes/isa_mix.py file.s kernel_substring"""
import collections
import re
import sys

txt = open(sys.argv[1]).read()
name = sys.argv[2]
m = re.search(r"^(_Z\S*%s\S*):" % re.escape(name), txt, re.M)
start = m.start()
end = txt.index("s_endpgm", start)
body = txt[start:end]
c = collections.Counter()
for ln in body.splitlines():
    ln = ln.strip()
    if not ln or ln.startswith((";", ".", "_Z")) or ln.endswith(":"):
        continue
    op = ln.split()[0]
    c[op] += 1
tot = sum(c.values())
def grp(pred): return sum(v for k, v in c.items() if pred(k))
valu = grp(lambda k: k.startswith("v_") and not k.startswith("v_mfma"))
print("%s: %d instructions, VALU %d, MFMA %d, SALU %d, LDS %d, VMEM %d" % (m.group(1)[:60], tot, valu, grp(lambda k: k.startswith("v_mfma")), grp(lambda k: k.startswith("s_")),
      grp(lambda k: k.startswith("ds_")), grp(lambda k: k.startswith(("global_", "buffer_", "flat_")))))
def show(title, pred):
    items = sorted(((k, v) for k, v in c.items() if pred(k)), key=lambda kv: -kv[1])
    print("  %s %d: %s" % (title, sum(v for _, v in items), ", ".join("%s %d" % kv for kv in items[:14])))
show("VALU", lambda k: k.startswith("v_") and not k.startswith("v_mfma"))
show("LDS", lambda k: k.startswith("ds_"))
show("VMEM", lambda k: k.startswith(("global_", "buffer_")))
