"""Idle time between kernels, from a rocprofv3 --kernel-trace CSV: where the device waits for the host.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -o t -- python3 bench.py --steps 6 --warmup 3 --no-eager ...
    python tools/trace_gaps.py gpurun_out/trace/**/t_kernel_trace.csv

Steps are delimited by `sgd_multi_kernel` (one launch per adaptation step); the first `skip` steps are dropped.
"""
import collections
import csv
import os
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "").replace("at::native::", "")
    if name.startswith("vectorized_elementwise_kernel") or name.startswith("reduce_kernel") or name.startswith("elementwise_kernel"):
        return name[:90]
    for stop in "(<":
        if stop in name:
            name = name[:name.index(stop)]
    return name[:60]


def main(path, skip=4):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ends = [i for i, r in enumerate(rows) if "sgd_multi_kernel" in r["Kernel_Name"]]
    if len(ends) < skip + 2:
        raise SystemExit(f"only {len(ends)} steps in the trace")
    rs = rows[ends[skip] + 1:ends[-1] + 1]
    steps = len(ends) - 1 - skip
    span = int(rs[-1]["End_Timestamp"]) - int(rs[0]["Start_Timestamp"])
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs)
    sites = collections.defaultdict(lambda: [0, 0])
    hist = collections.Counter()
    for a, b in zip(rs, rs[1:]):
        g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
        if g <= 0:
            continue
        hist["<1us" if g < 1000 else "<5us" if g < 5000 else "<20us" if g < 20000 else "<100us" if g < 100000 else ">=100us"] += g
        if g >= 5000:
            key = (short(a["Kernel_Name"]), short(b["Kernel_Name"]))
            sites[key][0] += g
            sites[key][1] += 1
    print(f"{steps} steps, {len(rs) / steps:.0f} launches per step; per step: span {span / steps / 1e6:.2f} ms, kernels {busy / steps / 1e6:.2f} ms, "
          f"idle {(span - busy) / steps / 1e6:.2f} ms")
    print("idle time per step by gap length (ms):", {k: round(v / steps / 1e6, 3) for k, v in sorted(hist.items())})
    if os.environ.get("KERNELS"):  # per-step kernel table (steady state only: set-up launches are not in it)
        tab = collections.defaultdict(lambda: [0, 0])
        for r in rs:
            k = short(r["Kernel_Name"])
            tab[k][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            tab[k][1] += 1
        print("kernels (ms per step, launches per step, mean us):")
        for k, (d, n) in sorted(tab.items(), key=lambda kv: -kv[1][0])[:int(os.environ["KERNELS"])]:
            print(f"  {d / steps / 1e6:7.3f} {n / steps:7.1f} {d / n / 1e3:8.1f}  {k}")
    print("gaps >= 5 us by site (ms per step, count per step):")
    for (a, b), (g, n) in sorted(sites.items(), key=lambda kv: -kv[1][0])[:25]:
        print(f"  {g / steps / 1e6:7.3f} {n / steps:6.1f}  {a} -> {b}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 4)
