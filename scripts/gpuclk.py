"""Shader clock and socket power of GPU 0 from sysfs (hwmon freq1_input / power1_average|power1_input, else pp_dpm_sclk), with
rocm-smi as the fallback; a background sampler for "what did the chip hold while this loop ran".  Profiling scripts only."""
import glob
import re
import subprocess
import threading
import time


def _paths():
    """One entry per card that has a hwmon directory (a 1-GPU box may still list every card of the node in sysfs)."""
    cards = []
    for dev in sorted(glob.glob("/sys/class/drm/card*/device")):
        hw = sorted(glob.glob(dev + "/hwmon/hwmon*"))
        if not hw:
            continue
        h = hw[0]
        cards.append({"freq": h + "/freq1_input", "pavg": h + "/power1_average", "pin": h + "/power1_input", "dpm": dev + "/pp_dpm_sclk"})
    return cards


_P = None


def _read_card(c):
    mhz = watts = None
    try:
        mhz = int(open(c["freq"]).read()) / 1e6
    except (OSError, ValueError):
        try:
            for line in open(c["dpm"]).read().splitlines():
                if "*" in line:
                    mhz = float(re.search(r"(\d+)\s*Mhz", line, re.I).group(1))
        except (OSError, ValueError, AttributeError):
            pass
    for k in ("pavg", "pin"):
        try:
            watts = int(open(c[k]).read()) / 1e6
            break
        except (OSError, ValueError):
            pass
    return mhz, watts


def read():
    """-> (sclk MHz or None, watts or None) of the BUSIEST card (highest socket power): the one this process is running on"""
    global _P
    if _P is None:
        _P = _paths()
    best = (None, None)
    for c in _P:
        mhz, watts = _read_card(c)
        if watts is not None and (best[1] is None or watts > best[1]):
            best = (mhz, watts)
    mhz, watts = best
    if mhz is None or watts is None:
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=20).stdout
            rows = {}
            for line in out.splitlines():
                m0 = re.match(r"GPU\[(\d+)\]", line)
                if not m0:
                    continue
                r = rows.setdefault(int(m0.group(1)), [None, None])
                m = re.search(r"sclk clock level.*\((\d+)Mhz\)", line)
                if m:
                    r[0] = float(m.group(1))
                m = re.search(r"Power \(W\):\s*([\d.]+)", line)
                if m:
                    r[1] = float(m.group(1))
            if rows:
                mhz, watts = max(rows.values(), key=lambda r: r[1] or 0.0)
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
    cs = _paths()
    print(len(cs), "cards with hwmon;", [(_read_card(c)) for c in cs])
    print(read())
