"""What the host and the device do around the end of an adaptation step: HIP API calls, kernels and copies in a window
around the 5th `sgd_multi_kernel` of a rocprofv3 trace.

    rocprofv3 --kernel-trace --hip-runtime-trace --memory-copy-trace --output-format csv -d DIR -o t -- python3 bench.py ...
    python tools/boundary_timeline.py DIR
"""
import csv
import glob
import os
import sys


def load(pattern):
    files = glob.glob(pattern, recursive=True)
    return list(csv.DictReader(open(files[0]))) if files else []


def main(d, nth=5, before_us=300, after_us=2500):
    kern = load(os.path.join(d, "**", "*kernel_trace.csv"))
    api = load(os.path.join(d, "**", "*hip_api_trace.csv"))
    cop = load(os.path.join(d, "**", "*memory_copy_trace.csv"))
    kern.sort(key=lambda r: int(r["Start_Timestamp"]))
    sgd = [r for r in kern if "sgd_multi_kernel" in r["Kernel_Name"]]
    t0 = int(sgd[nth]["End_Timestamp"])
    lo, hi = t0 - before_us * 1000, t0 + after_us * 1000
    ev = []
    for r in kern:
        if lo <= int(r["Start_Timestamp"]) <= hi:
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "KERNEL", r["Kernel_Name"].replace("(anonymous namespace)::", "")[:70]))
    for r in api:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if lo <= s <= hi or (s < lo and e > lo):
            ev.append((s, e, "api", r.get("Function", r.get("Name", "?"))))
    for r in cop:
        if lo <= int(r["Start_Timestamp"]) <= hi:
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY", r.get("Direction", "") + " " + r.get("Name", "")))
    ev.sort()
    last_api = None
    n_api = 0
    for s, e, kind, name in ev:
        if kind == "api" and e - s < 20000 and name in ("hipGetLastError", "hipGetDevice", "hipPeekAtLastError", "hipSetDevice", "hipStreamIsCapturing", "__hipPushCallConfiguration", "__hipPopCallConfiguration"):
            n_api += 1
            continue
        print(f"{(s - t0) / 1e3:9.1f} us  {(e - s) / 1e3:8.1f} us  {kind:6s} {name}")
    print(f"({n_api} short bookkeeping API calls not shown)")


if __name__ == "__main__":
    main(sys.argv[1])
