#!/usr/bin/env python3
"""Per-launch duration of the first N step launches after the process starts (config 3 by default), WITH what the chip
was doing meanwhile: a sampler thread reads the GPU's clock levels / power / busy counters from sysfs while the launches
run in segments (synchronised after each, so a sample can be attributed to a range of launches).  Then the same world is
reset and played again in the same, now warm, process: if the fast-slow-fast pattern belonged to the world (its content
at turns 10-100) it would repeat; if it belongs to the process start (clocks, power management, caches) it does not.
Run on the GPU box: python tools/ramp.py [N]"""
import glob, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
import torch
from sorrel_amd.engine import GridEngine
from sorrel_amd.spec import treasurehunt_spec

N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
SEG = 10


def find_device_dir():
    """The sysfs directory of the GPU this process computes on (a box shows every card of the host; only one is ours):
    matched by PCI address."""
    pr = torch.cuda.get_device_properties(0)
    want = None
    if hasattr(pr, "pci_bus_id"):
        want = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{getattr(pr, 'pci_device_id', 0):02x}"
    cands = []
    for d in sorted(glob.glob("/sys/class/drm/card*/device")):
        if os.path.exists(os.path.join(d, "pp_dpm_sclk")):
            real = os.path.realpath(d)
            cands.append((d, real))
            if want and want in real:
                return d
    print(f"(no card matches PCI address {want}; candidates: {[r for _, r in cands][:4]} ...)")
    return None


DEV = find_device_dir()
HWMON = (glob.glob(os.path.join(DEV, "hwmon", "hwmon*")) or [None])[0] if DEV else None


def read(path):
    try:
        with open(path) as fh:
            return fh.read()
    except OSError:
        return None


def active_level(text):
    if not text:
        return None
    for ln in text.splitlines():
        if ln.strip().endswith("*"):
            return ln.split(":")[1].strip().rstrip("*").strip()
    return None


def sample():
    out = {"t": time.perf_counter()}
    if DEV:
        out["sclk"] = active_level(read(os.path.join(DEV, "pp_dpm_sclk")))
        out["mclk"] = active_level(read(os.path.join(DEV, "pp_dpm_mclk")))
        out["fclk"] = active_level(read(os.path.join(DEV, "pp_dpm_fclk")))
        out["busy"] = (read(os.path.join(DEV, "gpu_busy_percent")) or "").strip() or None
    if HWMON:
        for key, name in (("power_uW", "power1_average"), ("power_in_uW", "power1_input"), ("freq1_Hz", "freq1_input"), ("temp_mC", "temp1_input")):
            v = read(os.path.join(HWMON, name))
            if v is not None:
                out[key] = v.strip()
    return out


samples, stop = [], False


def sampler():
    while not stop:
        samples.append(sample())
        time.sleep(0.0005)


eng = GridEngine(treasurehunt_spec(32, 32, 8, 3, spawn_prob=0.005, seed=0), 65536, device="cuda:0")
print(eng.launch_info())
print(f"sysfs device dir: {DEV}  hwmon: {HWMON}")
print("one sample before any launch:", {k: v for k, v in sample().items() if k != "t"})


def play(n, label):
    global stop, samples
    eng.reset(0)
    torch.cuda.synchronize()
    samples, stop = [], False
    th = threading.Thread(target=sampler, daemon=True)
    th.start()
    eng.set_timing(True)
    marks = []                                   # (launch index after the segment, host time when the segment had finished)
    for i in range(0, n, SEG):
        for _ in range(min(SEG, n - i)):
            eng.step(random_actions=True)
        torch.cuda.synchronize()
        marks.append((min(i + SEG, n), time.perf_counter()))
    stop = True
    th.join()
    ms = eng.step_times_ms()
    eng.set_timing(False)
    print(f"\n== {label}: {n} launches in segments of {SEG} (synchronised after each), {len(samples)} sysfs samples")
    print("launches      mean_us   min_us   max_us   | sclk levels seen (count)            mclk        power W (mean)  busy %")
    edges = [0, 10, 20, 50, 100, 200, 400, 800, 1600, n]
    t_of = dict(marks)
    t_start = marks[0][1] - 0.002
    for lo, hi in zip(edges, edges[1:]):
        if lo >= n:
            break
        hi = min(hi, n)
        seg = ms[lo:hi]
        t0 = t_start if lo == 0 else t_of.get(lo, t_start)
        t1 = t_of.get(hi, marks[-1][1])
        ss = [s for s in samples if t0 <= s["t"] <= t1]
        def hist(key):
            h = {}
            for s_ in ss:
                h[s_.get(key)] = h.get(s_.get(key), 0) + 1
            return ", ".join(f"{k} ({v})" for k, v in sorted(h.items(), key=lambda kv: -kv[1])[:3]) or "-"
        pw = [float(s_["power_uW"]) / 1e6 for s_ in ss if s_.get("power_uW")] or [float(s_["power_in_uW"]) / 1e6 for s_ in ss if s_.get("power_in_uW")]
        busy = [float(s_["busy"]) for s_ in ss if s_.get("busy")]
        print(f"{lo:5d}-{hi:5d}  {sum(seg) / len(seg) * 1e3:8.1f} {min(seg) * 1e3:8.1f} {max(seg) * 1e3:8.1f}   | {hist('sclk'):34s}  {hist('mclk'):10s}  "
              f"{(sum(pw) / len(pw)) if pw else float('nan'):8.1f}       {(sum(busy) / len(busy)) if busy else float('nan'):5.1f}")
    return ms


first = play(N, "fresh process")
second = play(min(N, 800), "same process, world reset to epoch 0 again (same turns 1.., warm chip)")
both = min(len(first), len(second))
for lo, hi in ((0, 10), (10, 100), (100, 400)):
    if hi <= both:
        a = sum(first[lo:hi]) / (hi - lo) * 1e3
        b = sum(second[lo:hi]) / (hi - lo) * 1e3
        print(f"launches {lo:3d}-{hi:3d}: fresh process {a:7.1f} us, after a reset in the warm process {b:7.1f} us")
