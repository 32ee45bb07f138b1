"""Summarise a rocprofv3 --pmc pass of SQ counters per kernel (issue/wait split, MFMA busy, LDS conflicts).

    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES \
        SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d DIR -o sq -- python3 tools/one_pass.py
    python tools/sq_summary.py DIR/sq_counter_collection.csv [top_n]
"""
import collections, csv, re, sys


def main():
    a = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    for r in csv.DictReader(open(sys.argv[1])):
        k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])[:64]
        a[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVE_CYCLES":
            n[k] += 1
            a[k]["ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    print(f"{'kernel':64s} {'n':>5s} {'ms':>7s} {'parked':>7s} {'stall':>6s} {'issue':>6s} {'mfma/simd-cyc':>13s} {'lds-confl':>9s}")
    for k, v in sorted(a.items(), key=lambda kv: -kv[1]["ns"])[:top]:
        wc = max(v["SQ_WAVE_CYCLES"], 1.0)
        simd_cycles = v["GRBM_GUI_ACTIVE"] / 8 * 1024  # GUI_ACTIVE sums the 8 XCDs; 1024 SIMDs
        print(f"{k:64s} {n[k]:5d} {v['ns']/1e6:7.2f} {v['SQ_WAIT_ANY']/wc:7.2f} {v['SQ_WAIT_INST_ANY']/wc:6.2f} "
              f"{v['SQ_ACTIVE_INST_ANY']/wc:6.2f} {v['SQ_VALU_MFMA_BUSY_CYCLES']/max(simd_cycles,1):13.3f} "
              f"{v['SQ_LDS_BANK_CONFLICT']/max(v['SQ_LDS_IDX_ACTIVE'],1):9.3f}")


if __name__ == "__main__":
    main()
