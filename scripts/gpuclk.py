"""Shader clock and socket power of GPU 0 from sysfs (hwmon freq1_input / power1_average|power1_input, else pp_dpm_sclk), with
rocm-smi as the fallback; a background sampler for "what did the chip hold while this loop ran".  Profiling scripts only."""
import glob
import re
import subprocess
import threading
import time


def _paths():
    best = None
    for dev in sorted(glob.glob("/sys/class/drm/card*/device")):
        hw = sorted(glob.glob(dev + "/hwmon/hwmon*"))
        if not hw:
            continue
        h = hw[0]
        cand = {"freq": h + "/freq1_input", "pavg": h + "/power1_average", "pin": h + "/power1_input", "dpm": dev + "/pp_dpm_sclk"}
        try:
            open(cand["freq"]).read()
        except OSError:
            cand["freq"] = None
        best = best or cand
        if cand["freq"]:
            return cand
    return best


_P = None


def read():
    """-> (sclk MHz or None, watts or None)"""
    global _P
    if _P is None:
        _P = _paths() or {}
    mhz = watts = None
    try:
        if _P.get("freq"):
            mhz = int(open(_P["freq"]).read()) / 1e6
        elif _P.get("dpm"):
            for line in open(_P["dpm"]).read().splitlines():
                if "*" in line:
                    mhz = float(re.search(r"(\d+)\s*Mhz", line, re.I).group(1))
    except (OSError, ValueError, AttributeError):
        pass
    for k in ("pavg", "pin"):
        try:
            if _P.get(k):
                watts = int(open(_P[k]).read()) / 1e6
                break
        except (OSError, ValueError):
            pass
    if mhz is None or watts is None:
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=20).stdout
            for line in out.splitlines():
                if "GPU[0]" not in line:
                    continue
                m = re.search(r"sclk clock level.*\((\d+)Mhz\)", line)
                if m and mhz is None:
                    mhz = float(m.group(1))
                m = re.search(r"Power \(W\):\s*([\d.]+)", line)
                if m and watts is None:
                    watts = float(m.group(1))
        except (OSError, subprocess.SubprocessError):
            pass
    return mhz, watts


class Sampler:
    """with Sampler() as s: <gpu work>;  s.summary() -> 'sclk 1724 MHz (1719-1730), 1393 W (1390-1396), n=12'"""

    def __init__(self, period=0.15, skip=0.6):
        self.period, self.skip, self.rows, self._stop = period, skip, [], threading.Event()

    def _run(self):
        t0 = time.time()
        while not self._stop.is_set():
            r = read()
            if time.time() - t0 >= self.skip:
                self.rows.append(r)
            self._stop.wait(self.period)

    def __enter__(self):
        self._t = threading.Thread(target=self._run, daemon=True)
        self._t.start()
        return self

    def __exit__(self, *a):
        self._stop.set()
        self._t.join()

    def stats(self):
        f = [r[0] for r in self.rows if r[0] is not None]
        w = [r[1] for r in self.rows if r[1] is not None]
        return (sum(f) / len(f) if f else None, sum(w) / len(w) if w else None)

    def summary(self):
        f = [r[0] for r in self.rows if r[0] is not None]
        w = [r[1] for r in self.rows if r[1] is not None]
        fs = f"sclk {sum(f) / len(f):.0f} MHz ({min(f):.0f}-{max(f):.0f})" if f else "sclk n/a"
        ws = f"{sum(w) / len(w):.0f} W ({min(w):.0f}-{max(w):.0f})" if w else "power n/a"
        return f"{fs}, {ws}, n={len(self.rows)}"


if __name__ == "__main__":
    print(_paths())
    print(read())
