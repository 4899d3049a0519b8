#!/usr/bin/env python3
"""Sample board power and shader clock while a command runs (GPU box).

    python tools/power_trace.py <out.csv> [--period 0.02] -- <command ...>

Reads the amdgpu hwmon files of the container's own GPU directly (found by the PCI address of HIP device 0: the host's sysfs
shows every GPU of the node, other tenants' included) (power1_input / power1_average in microwatts, power1_cap, freq1_input =
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


def my_gpu_pci_address():
    """PCI address of HIP device 0 of THIS container (the host's sysfs shows every GPU of the node, other tenants' included)"""
    code = ("import torch; p = torch.cuda.get_device_properties(0); "
            "print('%04x:%02x:%02x.0' % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id))")
    try:
        return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120).stdout.strip().splitlines()[-1]
    except Exception:
        return None


def hwmon_dirs():
    """the hwmon directory of this container's GPU (by PCI address); every amdgpu hwmon directory if that cannot be resolved"""
    addr = my_gpu_pci_address()
    mine = sorted(glob.glob(f"/sys/bus/pci/devices/{addr}/hwmon/hwmon*")) if addr else []
    mine = [d for d in mine if any(os.path.exists(os.path.join(d, f)) for f in ("power1_input", "power1_average"))]
    if mine:
        return mine
    return [d for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))
            if any(os.path.exists(os.path.join(d, f)) for f in ("power1_input", "power1_average"))]


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
    dirs = hwmon_dirs()
    rows = []
    proc = subprocess.Popen(cmd)
    t0 = time.perf_counter()
    if dirs:
        pname = lambda d: os.path.join(d, "power1_input") if os.path.exists(os.path.join(d, "power1_input")) else os.path.join(d, "power1_average")
        per = {d: [] for d in dirs}
        while proc.poll() is None:
            now = time.perf_counter() - t0
            for d in dirs:
                per[d].append((now, read(pname(d)) * 1e-6, read(os.path.join(d, "freq1_input")) * 1e-6, read(os.path.join(d, "temp1_input")) * 1e-3))
            time.sleep(period)
        # the GPU this command ran on = the card whose power rose the most above its own minimum
        swing = lambda r: (max(x[1] for x in r) - min(x[1] for x in r)) if r else 0.0
        d = max(dirs, key=lambda k: swing(per[k]))
        rows = per[d]
        cap = read(os.path.join(d, "power1_cap")) * 1e-6
        source = pname(d) + (f" (PCI address of HIP device 0)" if len(dirs) == 1 else f" (of {len(dirs)} cards, the one with the largest power swing)")
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
                     "sclk_max_MHz": max(r[2] for r in rows), "busy_samples": len(busy), "busy_seconds": len(busy) * period,
                     "power_idle_W": min(pw)})
    print("POWER_TRACE " + json.dumps(summ), flush=True)
    sys.exit(rc)


if __name__ == "__main__":
    main()
