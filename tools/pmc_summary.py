"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (collected separately, as
MI355X_MICROARCH.md prescribes) into per-kernel HBM traffic per launch.

gfx950 corrections from the guide: FETCH_SIZE and WRITE_SIZE are in KiB; FETCH_SIZE reports half
the bytes of wide coalesced streaming reads (16 B per lane) -> doubled; WRITE_SIZE is exact.

    python tools/pmc_summary.py <fetch_dir> <write_dir> <out.json> [workload description]
"""
import collections, csv, glob, json, sys


def agg(d, counter):
    a = collections.defaultdict(lambda: [0.0, 0, 0.0])
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] != counter:
                continue
            k = r["Kernel_Name"]
            a[k][0] += float(r["Counter_Value"])
            a[k][1] += 1
            a[k][2] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return a


def main():
    fetch, write = agg(sys.argv[1], "FETCH_SIZE"), agg(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k, (f, n, ns) in fetch.items():
        w = write.get(k, [0.0, 0, 0.0])[0]
        rd, wr = f * 1024 * 2, w * 1024
        out[k] = {"launches": n, "read_bytes_per_launch": rd / n, "write_bytes_per_launch": wr / n,
                  "hbm_bytes_per_launch": (rd + wr) / n, "avg_launch_us_profiled": ns / n / 1e3,
                  "gbps_profiled": (rd + wr) / ns}
    out = dict(sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"]))
    blob = {"note": "FETCH_SIZE x2 (gfx950 wide-read correction) + WRITE_SIZE, KiB -> bytes; separate --pmc passes",
            "kernels": out}
    if len(sys.argv) > 4:
        blob["workload"] = sys.argv[4]
    json.dump(blob, open(sys.argv[3], "w"), indent=1)
    for k, v in list(out.items())[:8]:
        print(f"{k[:70]:70s} n={v['launches']:5d} {v['hbm_bytes_per_launch']/1e6:9.1f} MB/launch {v['gbps_profiled']:7.0f} GB/s")


if __name__ == "__main__":
    main()
