"""Per-dispatch averages (in millions) of rocprofv3 --pmc counter_collection.csv files, one line per kernel."""
import csv, collections, sys
for path in sys.argv[1:]:
    rows = list(csv.DictReader(open(path)))
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for r in rows:
        k = r['Kernel_Name'].replace('(anonymous namespace)::', '')[:28]
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); n[k].add(r['Dispatch_Id'])
    for k, v in agg.items():
        c = len(n[k])
        print(k, c, {a: round(b / c / 1e6, 2) for a, b in sorted(v.items())})
