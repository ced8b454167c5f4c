"""Which kind of box is this?  Device properties next to a quick timing of the config-3 step (staged, cap 8 vs 6)."""
import json, os, subprocess, sys
import torch
p = torch.cuda.get_device_properties(0)
print("name", p.name, "| CUs", p.multi_processor_count, "| mem GiB", round(p.total_memory / 2**30, 1), "| gcnArch", getattr(p, "gcnArchName", "?"),
      "| L2", getattr(p, "L2_cache_size", "?"), "| clock kHz", getattr(p, "clock_rate", "?"), "| mem clock kHz", getattr(p, "memory_clock_rate", "?"),
      "| bus", getattr(p, "memory_bus_width", "?"))
for cmd in (["rocm-smi", "--showmemorypartition", "--showcomputepartition"], ["rocm-smi", "--showclocks"], ["rocm-smi", "--showpower", "--showperflevel"]):
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=20).stdout
        print("\n".join(l for l in out.splitlines() if l.strip() and "=====" not in l)[:1500])
    except Exception as e:
        print(cmd, "failed:", e)
