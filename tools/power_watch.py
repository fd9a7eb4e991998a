#!/usr/bin/env python
"""Samples rocm-smi (power, sclk, temperature) every ~0.25 s while the bench loop runs: is the step clock/power limited?"""
import json, os, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
samples, stop = [], False


def poll():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp", "--json"], capture_output=True, text=True, timeout=5).stdout
            d = json.loads(out)
            c = d.get("card0", {})
            samples.append({k: v for k, v in c.items() if any(t in k.lower() for t in ("power", "sclk", "mclk", "fclk", "junction", "edge", "hbm"))})
        except Exception as e:  # noqa: BLE001
            samples.append({"err": repr(e)})
        time.sleep(0.25)


idle = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp", "--showmaxpower", "--json"], capture_output=True, text=True).stdout
t = threading.Thread(target=poll); t.start()
p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "120", "--warmup", "10", "--no-cpu-baseline", "--no-batch2",
                    "--no-fp8", "--no-dropin", "--no-hostfed"], capture_output=True, text=True)
stop = True; t.join()
line = p.stdout.strip().splitlines()[-1] if p.stdout.strip() else p.stderr[-300:]
try:
    bench = json.loads(line)
except Exception:  # noqa: BLE001
    bench = {"raw": line[:300]}
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump({"idle": idle[:2000], "bench": {k: bench.get(k) for k in ("value", "ms_per_step", "steps")}, "n_samples": len(samples), "samples": samples},
          open(os.path.join(ROOT, "gpurun_out", "power.json"), "w"), indent=0)
print("IDLE", idle[:1500])
print("BENCH", line[:200])
print("N", len(samples))
for s in samples[:: max(1, len(samples) // 24)]:
    print(s)
