"""Where do the `__amd_rocclr_copyBuffer` (device copy) launches of a step come from?  For every copy kernel in a
rocprofv3 --kernel-trace CSV, the names of the non-copy kernels before and after it on the same stream; counts per step.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -o t -- python3 bench.py --steps 8 --warmup 3 ...
    python tools/copy_neighbours.py gpurun_out/trace/**/t_kernel_trace.csv
"""
import collections
import csv
import sys

from trace_gaps import short


def main(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    steps = sum("sgd_multi_kernel" in r["Kernel_Name"] for r in rows)
    is_copy = lambda r: "copyBuffer" in r["Kernel_Name"] or "fillBuffer" in r["Kernel_Name"]
    hits = collections.Counter()
    for i, r in enumerate(rows):
        if not is_copy(r):
            continue
        j = i - 1
        while j >= 0 and is_copy(rows[j]):
            j -= 1
        k = i + 1
        while k < len(rows) and is_copy(rows[k]):
            k += 1
        before = short(rows[j]["Kernel_Name"]) if j >= 0 else "-"
        after = short(rows[k]["Kernel_Name"]) if k < len(rows) else "-"
        hits[(short(r["Kernel_Name"]), before, after)] += 1
    print(f"{steps} steps; copy / fill launches by neighbourhood (count per step):")
    for (name, before, after), n in hits.most_common(40):
        print(f"{n / max(steps, 1):7.1f}  {name:28s} after {before:40s} before {after}")


if __name__ == "__main__":
    main(sys.argv[1])
