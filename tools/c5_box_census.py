"""BASELINE config 5 at its per-GPU share (2 048 envs) on THIS box: the walking variant with the dynamic tail / the purely static
split / one env per workgroup, with the card's clock and power telemetry sampled while each runs; then (with a -DSGW_STAMPS build:
SGW_LIB=tools/libsgw_stamps.so) how long a workgroup lives on each XCD.  One call per box: boxes differ (round 3: "one of two speeds
per box").  GPU only."""
import ctypes as C, glob, os, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch
from sorrel_amd import _native as N
from sorrel_amd.engine import GridEngine
from sorrel_amd.spec import treasurehunt_spec
from _warm import timed_us

E = 2048
spec = treasurehunt_spec(128, 128, 64, 5, spawn_prob=0.05, seed=0, dense_prob=0.25)
by = spec.algorithmic_bytes_per_env_step() * E
STAMPS = "stamps" in os.path.basename(N.LIB_PATH)


def dev_dir():
    pr = torch.cuda.get_device_properties(0)
    want = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{getattr(pr, 'pci_device_id', 0):02x}" if hasattr(pr, "pci_bus_id") else None
    for d in sorted(glob.glob("/sys/class/drm/card*/device")):
        if os.path.exists(os.path.join(d, "pp_dpm_sclk")) and (not want or want in os.path.realpath(d)):
            return d
    return None


DEV = dev_dir()
HW = (glob.glob(os.path.join(DEV, "hwmon", "hwmon*")) or [None])[0] if DEV else None


def rd(p):
    try:
        with open(p) as fh:
            return fh.read().strip()
    except OSError:
        return None


def telemetry():
    out = {}
    if HW:
        for k, f in (("sclk_MHz", "freq1_input"), ("power_W", "power1_average"), ("power_in_W", "power1_input"), ("temp_C", "temp1_input")):
            v = rd(os.path.join(HW, f))
            if v:
                out[k] = float(v) / (1e6 if "W" in k or "MHz" in k else 1e3)
    if DEV:
        for k in ("pp_dpm_sclk", "pp_dpm_mclk", "pp_dpm_fclk"):
            t = rd(os.path.join(DEV, k)) or ""
            cur = [ln.split(":")[1].strip().rstrip("*").strip() for ln in t.splitlines() if ln.strip().endswith("*")]
            out[k[7:]] = cur[0] if cur else None
    return out


def measure(label, opts):
    with N.options(**opts):
        eng = GridEngine(spec, E, device="cuda:0")
    eng.reset(0)
    for _ in range(700):
        eng.step(random_actions=True)
    samples, stop = [], [False]

    def sampler():
        while not stop[0]:
            samples.append(telemetry())
            time.sleep(0.002)

    th = threading.Thread(target=sampler)
    th.start()
    us = [timed_us(lambda: eng.step(random_actions=True), 300) for _ in range(3)]
    stop[0] = True
    th.join()
    keys = sorted({k for s in samples for k, v in s.items() if isinstance(v, float)})
    tele = {k: round(float(np.median([s[k] for s in samples if k in s])), 1) for k in keys}
    lv = {k: samples[len(samples) // 2].get(k) for k in ("sclk", "mclk", "fclk")} if samples else {}
    print(f"{label:44s} {min(us):6.1f} us ({by / min(us) / 1e3 / 8000:.3f})  {tele} {lv}  [{eng.launch_info().split(' group')[0]}]", flush=True)
    return eng


print("box:", rd("/proc/sys/kernel/hostname"), "| card:", (subprocess.run("rocm-smi --showserial 2>/dev/null | grep -i serial", shell=True, capture_output=True, text=True).stdout.strip()[-14:]),
      "| idle:", telemetry())
opts = {"jit": 0} if STAMPS else {}
measure("walking workgroups, dynamic tail (default)", dict(opts))
measure("walking workgroups, first env static, rest dynamic", dict(opts, big_walk_share=1))
measure("walking workgroups, static split", dict(opts, big_walk_share=1 << 20))
measure("1 024 walking workgroups, all dynamic", dict(opts, big_walk_share=1, big_walk_blocks=1024))
eng = measure("one env per workgroup", dict(opts, big_walk=0))
if STAMPS:
    torch.cuda.synchronize()
    buf = np.zeros((65536, 8), np.uint64)
    N.load().sgw_debug_stamps(buf.ctypes.data_as(C.c_void_p))
    life = buf[:E, :5].astype(np.float64).sum(axis=1) / 100.0
    seg = buf[:E, :5].astype(np.float64) / 100.0
    start = buf[:E, 6].astype(np.int64)
    start -= start.min()
    xcc = ((buf[:E, 7] >> 32) & 0xF).astype(np.int64)
    print("one env per workgroup, per XCD: workgroups | life of a workgroup (us): mean, of which load+sweep / moves / windows | last end (us)")
    for x in np.unique(xcc):
        m = xcc == x
        print(f"  xcc {x}: {int(m.sum()):4d} | {life[m].mean():5.1f} = {seg[m, 0].mean():4.1f} / {seg[m, 1].mean():4.1f} / {seg[m, 2].mean():4.1f} | {(start[m] / 100.0 + life[m]).max():5.1f}")
