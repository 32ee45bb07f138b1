"""Move one evidence run from gpurun_out/ (scratch) into profiles/ (tracked): bench lines of every configuration, kernel
stats, PMC traffic, SQ counters, per-shape conv rates; stamps each file with the library source hash the run used and
points the bench lines' roofline.traffic at the PMC summary of the same run.
usage: python tools/install_profiles.py r02_n <library_src> [old_tag_to_remove]"""
import glob, json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tag, src = sys.argv[1], sys.argv[2]
old = sys.argv[3] if len(sys.argv) > 3 else None
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
names = {"default": "bench_line.json", "static": "bench_line_static_branch.json", "c1": "bench_line_config1.json",
         "c2": "bench_line_config2.json", "c5": "bench_line_config5.json", "gb32": "bench_line_strong_gb32_n1.json"}
t = json.load(open(os.path.join(G, f"{tag}_hbm_traffic.json")))
t["library_src"] = src
t["command"] = ("rocprofv3 --pmc FETCH_SIZE (and, separately, WRITE_SIZE) --kernel-trace --output-format csv -- python3 "
                "tools/one_pass.py; tools/pmc_summary.py")
if old:
    for f in glob.glob(os.path.join(P, f"{old}_*")):
        subprocess.run(["git", "rm", "-q", "-f", f], cwd=ROOT)
json.dump(t, open(os.path.join(P, f"{tag}_hbm_traffic.json"), "w"), indent=1)
import bench  # noqa: E402  (committed_traffic reads the file just written)
for k, n in names.items():
    line = open(os.path.join(G, f"m_{k}.log")).read().strip().splitlines()[-1]
    j = json.loads(line)
    j["roofline"].update(bench.committed_traffic(j["roofline"]["kernel"]))
    j["library_src"] = src
    open(os.path.join(P, f"{tag}_{n}"), "w").write(json.dumps(j) + "\n")
    print(k, j["value"], j["ms_per_step"], j["roofline"]["frac"], j["roofline"]["traffic_source"])
for a, b in (("kstats.csv", "bench_kernel_stats.csv"), ("sq_counters.txt", "sq_counters.txt"), ("conv_shapes.txt", "conv_shapes.txt")):
    shutil.copy(os.path.join(G, f"{tag}_{a}"), os.path.join(P, f"{tag}_{b}"))
