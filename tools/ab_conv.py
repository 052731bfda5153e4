"""A/B timing of conv kernel variants selected through SEGSDE_TUNE (each variant needs its own process)."""
import os, subprocess, sys
variants = ["", "bk64=1", "wplan=1", "nos2=1", "adjfix=1"]   # knobs parsed by tune() in csrc/conv_igemm.hip
for v in variants:
    env = dict(os.environ, SEGSDE_TUNE=v, BENCH_B=os.environ.get("BENCH_B", "8"), BENCH_ONLY_CONV="1")
    print("=== SEGSDE_TUNE=%r" % v, flush=True)
    out = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "bench_kernels.py")], env=env,
                         capture_output=True, text=True).stdout
    for line in out.splitlines():
        if " TF" in line:
            print(line[:44], " ".join(line.split("fwd")[1].split()[2:4]) if "fwd" in line else "", "| dgrad",
                  " ".join(line.split("dgrad")[1].split()[2:4]) if "dgrad" in line else "")
