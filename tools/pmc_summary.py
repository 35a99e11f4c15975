"""Per-kernel averages of every counter in a rocprofv3 --pmc counter_collection.csv (tools only).
usage: pmc_summary.py counter_collection.csv [name-filter] [min-launches]
Kernel names are shortened; launches are grouped by (kernel, grid, workgroup)."""
import collections
import csv
import re
import sys

csv.field_size_limit(1 << 30)
flt = sys.argv[2] if len(sys.argv) > 2 else ""
minl = int(sys.argv[3]) if len(sys.argv) > 3 else 1
agg = collections.OrderedDict()
names = []
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if flt and not re.search(flt, k):
        continue
    k = k.replace("(anonymous namespace)::", "").replace("void ", "")
    k = re.sub(r"\(.*$", "", k)[:60]
    key = (k, r["Grid_Size"], r["Workgroup_Size"])
    c = r["Counter_Name"]
    if c not in names:
        names.append(c)
    a = agg.setdefault(key, {})
    v = a.setdefault(c, [0, 0.0])
    v[0] += 1
    v[1] += float(r["Counter_Value"])
print("kernel | grid | wg | n | " + " | ".join(names))
for (k, g, w), a in agg.items():
    n = max(v[0] for v in a.values())
    if n < minl:
        continue
    print(f"{k} | {g} | {w} | {n} | " + " | ".join(f"{a[c][1] / a[c][0]:.4g}" if c in a else "-" for c in names))
