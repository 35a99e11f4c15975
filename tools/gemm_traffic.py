"""profiles/r1_gemm_traffic.json from the two rocprofv3 --pmc passes of bench.py (tools only).
usage: gemm_traffic.py fetch_counter_collection.csv write_counter_collection.csv config_name steps_profiled"""
import csv, sys, json, collections
csv.field_size_limit(1 << 30)


def per_kernel(path, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and "gemm_bf16_p3_kernel" in r["Kernel_Name"]:
            a = agg[(r["Kernel_Name"][:90], r["Grid_Size"])]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return agg


f = per_kernel(sys.argv[1], "FETCH_SIZE")
w = per_kernel(sys.argv[2], "WRITE_SIZE")
# only launch groups that belong to calibration steps (the teacher pass launches other grids, fewer times than steps)
steps = int(sys.argv[4])
f = {k: v for k, v in f.items() if v[0] >= steps}
w = {k: v for k, v in w.items() if v[0] >= steps}
nf = sum(v[0] for v in f.values())
fetch = sum(v[1] for v in f.values()) * 2048.0 / nf        # KiB, doubled (gfx950 tallies 128-B requests at 64 B)
nw = sum(v[0] for v in w.values())
write = sum(v[1] for v in w.values()) * 1024.0 / nw
out = {sys.argv[3]: {"kernel": "gemm_bf16_p3_kernel", "launches_profiled": nf, "fetch_bytes_per_launch": fetch,
                     "write_bytes_per_launch": write, "bytes_per_launch": fetch + write,
                     "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of bench.py; "
                               "FETCH_SIZE x2 (gfx950), KiB -> bytes; average over the kernel's launches inside calibration steps"}}
print(json.dumps(out, indent=1))
