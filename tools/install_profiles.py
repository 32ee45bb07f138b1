"""Move one evidence run of tools/profile_round.sh from gpurun_out/ (scratch) into profiles/ (tracked): bench lines of every
configuration, kernel stats (default command and one-stream), PMC traffic, SQ counters, per-shape conv rates; stamps each file
with the library source hash the run used and points the bench lines' roofline.traffic at the PMC summary of the same run.
usage: python tools/install_profiles.py r04_c [old_tag_to_remove]"""
import glob, json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tag = sys.argv[1]
old = sys.argv[2] if len(sys.argv) > 2 else None
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
src = json.load(open(os.path.join(G, f"{tag}_bench_line.json")))["config"]["library"]["library_src"]
if old:
    for f in glob.glob(os.path.join(P, f"{old}_*")):
        subprocess.run(["git", "rm", "-q", "-f", f], cwd=ROOT)
t = json.load(open(os.path.join(G, f"{tag}_hbm_traffic.json")))
t["library_src"] = src
t["command"] = ("rocprofv3 --pmc FETCH_SIZE (and, separately, WRITE_SIZE) --kernel-trace --output-format csv -- "
                + t.get("workload", "python3 tools/one_pass.py") + "; tools/pmc_summary.py")
json.dump(t, open(os.path.join(P, f"{tag}_hbm_traffic.json"), "w"), indent=1)
import bench  # noqa: E402  (committed_traffic reads the file just written)
lines = sorted(glob.glob(os.path.join(G, f"{tag}_bench_line*.json"))) + sorted(glob.glob(os.path.join(G, f"{tag}_f32_bench_line.json"))) + [os.path.join(G, f"{tag}_profiled_bench_line.json"),
                                                                         os.path.join(G, f"{tag}_profiled_bench_line_one_stream.json")]
for f in lines:
    j = json.load(open(f))
    if j.get("roofline"):
        j["roofline"].update(bench.committed_traffic(j["roofline"]["kernel"]))
        if j["roofline"].get("traffic") and j["roofline"].get("avg_launch_algorithmic_bytes"):
            j["roofline"]["traffic_over_algorithmic_bytes"] = round(j["roofline"]["traffic"] / j["roofline"]["avg_launch_algorithmic_bytes"], 3)
    j["library_src"] = src
    open(os.path.join(P, os.path.basename(f)), "w").write(json.dumps(j) + "\n")
    print(os.path.basename(f), j["ms_per_step"], j["value"], (j.get("roofline") or {}).get("frac"))
for n in ("bench_kernel_stats.csv", "bench_kernel_stats_one_stream.csv", "sq_counters.txt", "conv_shapes.txt", "hbm_traffic_top.txt",
          "step_kernel_table.txt", "power_summary.txt", "f32_sq_counters.txt", "f32_conv_shapes.txt", "f32_power_summary.txt"):
    if os.path.exists(os.path.join(G, f"{tag}_{n}")):
        shutil.copy(os.path.join(G, f"{tag}_{n}"), os.path.join(P, f"{tag}_{n}"))
print("library_src", src)
