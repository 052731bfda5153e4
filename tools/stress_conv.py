"""Random-geometry stress of the convolution kernels on the GPU box (forward, fused statistics, both data gradients, weight
gradient against torch autograd on the CPU, through tests/kernel_cases.run_conv_case): odd sizes, partial tiles, several
images per tile, both padding modes, strides, dilations, upsample + concat -- the shapes the fixed test list does not hold.
usage: python tools/stress_conv.py [ncases] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import kernel_cases as KC  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.RandomState(seed)
    bad, t0 = 0, time.time()
    for i in range(n):
        case = KC.random_conv_case(rng, i)
        try:
            KC.run_conv_case(case, "cuda", seed=seed + i)
            KC.run_dgrad_epilogue_variants(case, "cuda", seed=seed + i)
            print("ok   %s" % (case,), flush=True)
        except AssertionError as e:
            bad += 1
            print("FAIL %s: %s" % (case, str(e)[:300]), flush=True)
    print("%d cases, %d failed, %.0f s" % (n, bad, time.time() - t0))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
