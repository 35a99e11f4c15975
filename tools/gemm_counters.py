"""profiles/r3_gemm_counters.json from rocprofv3 --pmc passes of bench.py (tools only).

usage: gemm_counters.py CONFIG STEPS HEAD mfma_counter_collection.csv [fetch_counter_collection.csv write_counter_collection.csv]
(HEAD = the commit the passes were collected at; bench.py reports it as the provenance of roofline.traffic / mfma_busy_frac)

Per launch of the linear-layer GEMM kernel inside calibration steps (launch groups that occur at least STEPS times):
  mfma_busy_frac = sum SQ_VALU_MFMA_BUSY_CYCLES / sum (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs)   (MI355X_MICROARCH.md:
                   the matrix pipe is per SIMD; rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs)
  bytes_per_launch = FETCH_SIZE x 2 (gfx950 tallies 128-B requests at 64 B) + WRITE_SIZE, KiB -> bytes
"""
import collections
import csv
import json
import sys

csv.field_size_limit(1 << 30)
KERNELS = ("gemm_bf16_p3_kernel", "gemm_bf16_w256_kernel")      # dgrad / wgrad of the fake-quant linears
KERNEL = " + ".join(KERNELS)


def load(path, counters):
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for r in csv.DictReader(open(path)):
        if any(k in r["Kernel_Name"] for k in KERNELS) and r["Counter_Name"] in counters:
            a = agg[(r["Kernel_Name"][:90], r["Grid_Size"])][r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return agg


cfg, steps, head = sys.argv[1], int(sys.argv[2]), sys.argv[3]
sys.argv = sys.argv[:3] + sys.argv[4:]
m = load(sys.argv[3], {"SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_INSTS_MFMA"})
m = {k: v for k, v in m.items() if v["SQ_VALU_MFMA_BUSY_CYCLES"][0] >= steps}
busy = sum(v["SQ_VALU_MFMA_BUSY_CYCLES"][1] for v in m.values())
act = sum(v["GRBM_GUI_ACTIVE"][1] for v in m.values())
n = sum(v["SQ_VALU_MFMA_BUSY_CYCLES"][0] for v in m.values())
out = {"kernel": KERNEL, "launches_profiled": n, "mfma_busy_frac": busy / (act / 8.0 * 1024.0),
       "mfma_insts_per_launch": sum(v["SQ_INSTS_MFMA"][1] for v in m.values()) / n,
       "per_group": {f"{k[0][20:60]} grid={k[1]}": {"launches": v["SQ_VALU_MFMA_BUSY_CYCLES"][0],
                                                   "mfma_busy_frac": v["SQ_VALU_MFMA_BUSY_CYCLES"][1] / (v["GRBM_GUI_ACTIVE"][1] / 8.0 * 1024.0)}
                     for k, v in m.items()},
       "method": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE (one pass) of bench.py; "
                 "busy cycles / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs), summed over the kernel's launches inside calibration steps"}
if len(sys.argv) > 5:
    f = load(sys.argv[4], {"FETCH_SIZE"})
    w = load(sys.argv[5], {"WRITE_SIZE"})
    f = {k: v for k, v in f.items() if v["FETCH_SIZE"][0] >= steps}
    w = {k: v for k, v in w.items() if v["WRITE_SIZE"][0] >= steps}
    nf = sum(v["FETCH_SIZE"][0] for v in f.values())
    nw = sum(v["WRITE_SIZE"][0] for v in w.values())
    fetch = sum(v["FETCH_SIZE"][1] for v in f.values()) * 2048.0 / nf
    write = sum(v["WRITE_SIZE"][1] for v in w.values()) * 1024.0 / nw
    out.update({"fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write, "bytes_per_launch": fetch + write})
    out["method"] += "; FETCH_SIZE and WRITE_SIZE in two further separate passes, FETCH_SIZE x2 (gfx950), KiB -> bytes"
print(json.dumps({"_head": head, cfg: out}, indent=1))
