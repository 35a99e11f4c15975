"""Aggregate one replayed sample-step from a rocprofv3 kernel trace CSV (tools only; not part of the product)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
idx = [i for i, r in enumerate(rows) if 'mse_kernel' in r['Kernel_Name']]
a, b = idx[-6], idx[-5]
seg = rows[a + 1:b + 1]
agg = collections.OrderedDict()
for r in seg:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    n = r['Kernel_Name'].replace('_ZN12_GLOBAL__N_1', '').replace('void at::native::', '')[:64]
    key = (n, r['Grid_Size_X'], r['Workgroup_Size_X'])
    agg.setdefault(key, [0, 0.0]); agg[key][0] += 1; agg[key][1] += d
tot = sum(v[1] for v in agg.values())
span = (int(seg[-1]['End_Timestamp']) - int(seg[0]['Start_Timestamp'])) / 1e3
top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{v[1]:8.1f} us  n={v[0]:3d} avg={v[1]/v[0]:7.1f}  grid={k[1]:>8s} wg={k[2]:>4s} {k[0]}")
small = sum(v[1] for k, v in agg.items() if v[1] / v[0] < 6.0)
nsmall = sum(v[0] for k, v in agg.items() if v[1] / v[0] < 6.0)
print(f"busy {tot:.0f} us, span {span:.0f} us, kernels {len(seg)}; kernels <6us: {nsmall} totalling {small:.0f} us")
