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
    hw = None
    for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        if os.path.exists(os.path.join(d, "power1_average")) or os.path.exists(os.path.join(d, "power1_input")):
            hw = d
            break
    child = subprocess.Popen(cmd)
    rows, t0 = [], time.time()
    if hw:
        pw = os.path.join(hw, "power1_average") if os.path.exists(os.path.join(hw, "power1_average")) else os.path.join(hw, "power1_input")
        while child.poll() is None:
            rows.append((time.time() - t0, read(pw, 1e6), read(os.path.join(hw, "power1_cap"), 1e6), read(os.path.join(hw, "freq1_input"), 1e6)))
            time.sleep(0.02)
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
