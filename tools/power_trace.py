#!/usr/bin/env python3
"""Sample board power and shader clock while a command runs (GPU box).

    python tools/power_trace.py <out.csv> [--period 0.02] -- <command ...>

Reads the amdgpu hwmon files of card 0 directly (power1_input / power1_average in microwatts, power1_cap, freq1_input =
sclk in Hz, temp1_input) every `period` seconds -- rocm-smi takes ~0.3 s per call, too slow to resolve a 120 ms launch --
and falls back to `rocm-smi --showpower --showclocks --json` when the files are absent.  Writes a CSV and prints a summary
line (JSON): mean / max power, the cap, mean / min sclk over the samples whose power is within 10 % of the maximum (the
busy phase)."""
import glob
import json
import os
import subprocess
import sys
import time


def hwmon_dir():
    for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        if any(os.path.exists(os.path.join(d, f)) for f in ("power1_input", "power1_average")):
            return d
    return None


def read(path):
    try:
        with open(path) as f:
            return float(f.read().strip())
    except Exception:
        return float("nan")


def main():
    argv = sys.argv[1:]
    out = argv[0]
    period = 0.02
    if "--period" in argv:
        period = float(argv[argv.index("--period") + 1])
    cmd = argv[argv.index("--") + 1:]
    d = hwmon_dir()
    rows = []
    proc = subprocess.Popen(cmd)
    t0 = time.perf_counter()
    if d is not None:
        pfile = os.path.join(d, "power1_input") if os.path.exists(os.path.join(d, "power1_input")) else os.path.join(d, "power1_average")
        cap = read(os.path.join(d, "power1_cap")) * 1e-6
        while proc.poll() is None:
            rows.append((time.perf_counter() - t0, read(pfile) * 1e-6, read(os.path.join(d, "freq1_input")) * 1e-6,
                         read(os.path.join(d, "temp1_input")) * 1e-3))
            time.sleep(period)
        source = pfile
    else:
        cap = float("nan")
        while proc.poll() is None:
            try:
                j = json.loads(subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True,
                                              timeout=10).stdout)
                c = j[sorted(j)[0]]
                pw = next((float(v) for k, v in c.items() if "ower" in k and "(W)" in k), float("nan"))
                sclk = next((float(str(v).strip("()Mhz ")) for k, v in c.items() if k.startswith("sclk clock speed")), float("nan"))
                rows.append((time.perf_counter() - t0, pw, sclk, float("nan")))
            except Exception:
                pass
        source = "rocm-smi"
    rc = proc.wait()
    with open(out, "w") as f:
        f.write("t_s,power_W,sclk_MHz,temp_C\n")
        for r in rows:
            f.write("%.4f,%.1f,%.0f,%.1f\n" % r)
    pw = [r[1] for r in rows if r[1] == r[1]]
    summ = {"source": source, "samples": len(rows), "period_s": period, "power_cap_W": cap, "rc": rc}
    if pw:
        pmax = max(pw)
        busy = [r for r in rows if r[1] == r[1] and r[1] >= 0.9 * pmax]
        summ.update({"power_max_W": pmax, "power_mean_busy_W": sum(r[1] for r in busy) / len(busy),
                     "sclk_mean_busy_MHz": sum(r[2] for r in busy) / len(busy), "sclk_min_busy_MHz": min(r[2] for r in busy),
                     "sclk_max_MHz": max(r[2] for r in rows), "busy_samples": len(busy)})
    print("POWER_TRACE " + json.dumps(summ), flush=True)
    sys.exit(rc)


if __name__ == "__main__":
    main()
