"""Power and clocks of the GPU while the headline loop runs (device-resident, no samples to the host), for one or more library
builds on ONE box:  python tools/power_trace.py [seconds] libA.so libB.so ...   (BNMTF_OPERAND etc. are taken from the environment)
Each library runs in its own child process (BNMTF_LIB); this process samples `rocm-smi --showpower --showclocks` a few times a
second meanwhile.  Prints per library: iterations/s, and the median / min / max of the socket power and the shader clock.
(is the iteration bound by the chip's power budget?)"""
import json, os, re, statistics, subprocess, sys, time

CHILD = r'''
import sys, time, numpy as np
from bnmtf_amd import bnmf_gibbs_optimised
from bnmtf_amd.synthetic import generate_bnmf
secs = float(sys.argv[1])
R, M, _, _ = generate_bnmf(8192, 8192, 64, 0.1, seed_data=1, seed_mask=2)
b = bnmf_gibbs_optimised(R, M, 64, dict(alpha=1., beta=1., lambdaU=0.1, lambdaV=0.1), verbose=False, seed=3)
b.initialise("random"); b.run(200, store_samples=False)
print("READY", flush=True)
n = 0; t0 = time.perf_counter()
while time.perf_counter() - t0 < secs:
    b.run(1000, store_samples=False); n += 1000
dt = time.perf_counter() - t0
print("RATE %.1f" % (n / dt), flush=True)
b.close()
'''

def sample():
    try:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=10).stdout
        d = json.loads(out)
        card = d[sorted(d)[0]]
        p = s = m = None
        for k, v in card.items():
            kl = k.lower()
            if "power" in kl and p is None:
                try: p = float(v)
                except (TypeError, ValueError): pass
            if kl.startswith("sclk") and "mhz" in str(v).lower():
                s = float(re.sub(r"[^0-9.]", "", str(v)))
            if kl.startswith("mclk") and "mhz" in str(v).lower():
                m = float(re.sub(r"[^0-9.]", "", str(v)))
        return p, s, m
    except Exception as e:      # noqa: BLE001
        return None, None, None

def main():
    args = sys.argv[1:]
    secs = 12.0
    if args and re.fullmatch(r"[0-9.]+", args[0]):
        secs = float(args.pop(0))
    for lib in args:
        env = dict(os.environ, BNMTF_LIB=os.path.realpath(lib))
        ch = subprocess.Popen([sys.executable, "-c", CHILD, str(secs)], stdout=subprocess.PIPE, text=True, env=env)
        line = ch.stdout.readline()
        assert line.startswith("READY"), line
        time.sleep(1.0)
        rows = []
        t_end = time.time() + secs - 2.5
        while time.time() < t_end:
            rows.append(sample()); time.sleep(0.1)
        rate = None
        for line in ch.stdout:
            if line.startswith("RATE"):
                rate = float(line.split()[1])
        ch.wait()
        def st(i):
            v = [r[i] for r in rows if r[i] is not None]
            return None if not v else (round(statistics.median(v), 1), min(v), max(v), len(v))
        print(lib, "it/s", rate, "| power W (median, min, max, n)", st(0), "| sclk MHz", st(1), "| mclk MHz", st(2), flush=True)

if __name__ == "__main__":
    main()
