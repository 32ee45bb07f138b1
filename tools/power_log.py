"""Sample the GPU's board power and shader clock from sysfs while a command runs (a measurement aid: is a bench run held
at the board's power cap?).

    python3 tools/power_log.py out.csv -- python3 bench.py --steps 20 ...
Columns: seconds, power W (hwmon power1_average or power1_input), cap W (power1_cap), sclk MHz (hwmon freq1_input).
The sampled command is a CHILD process; this parent never touches the GPU."""
import glob, os, subprocess, sys, time


def first(pattern):
    hits = sorted(glob.glob(pattern))
    return hits[0] if hits else None


def read(path, scale):
    try:
        return float(open(path).read().strip()) / scale
    except Exception:
        return float("nan")


def main():
    out, cmd = sys.argv[1], sys.argv[sys.argv.index("--") + 1:]
    cands = []  # every amdgpu hwmon that reports power: a node may show several boards while the job runs on one of them
    for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        for f in ("power1_average", "power1_input"):
            if os.path.exists(os.path.join(d, f)):
                cands.append((d, os.path.join(d, f)))
                break
    child = subprocess.Popen(cmd)
    series, t0 = {d: [] for d, _ in cands}, time.time()
    while cands and child.poll() is None:
        now = time.time() - t0
        for d, pw in cands:
            series[d].append((now, read(pw, 1e6), read(os.path.join(d, "power1_cap"), 1e6), read(os.path.join(d, "freq1_input"), 1e6)))
        time.sleep(0.02)
    # the board the job ran on = the one that drew the most
    hw = max(series, key=lambda d: max((r[1] for r in series[d]), default=0.0)) if series else None
    rows = series[hw] if hw else []
    rc = child.wait()
    with open(out, "w") as f:
        f.write("# hwmon: %s\nseconds,power_w,cap_w,sclk_mhz\n" % hw)
        for r in rows:
            f.write("%.3f,%.1f,%.1f,%.0f\n" % r)
    if rows:
        busy = [r for r in rows if r[1] > 0.5 * max(x[1] for x in rows)]
        print("power_log: %d samples; max %.0f W, cap %.0f W; mean over the busy half %.0f W at %.0f MHz" % (
            len(rows), max(r[1] for r in rows), rows[0][2], sum(r[1] for r in busy) / len(busy), sum(r[3] for r in busy) / len(busy)), file=sys.stderr)
    sys.exit(rc)


if __name__ == "__main__":
    main()
