"""profiles/traffic_rNN.json from the FETCH_SIZE / WRITE_SIZE tables tools/gpu_pmc.sh leaves (tools/pmc_table.py format).
HBM-side bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (KB x 1024): on gfx950 FETCH_SIZE tallies 64 B per 128-B request of
a wide coalesced read (MI355X_MICROARCH.md, HBM section); counts are fabric-side, Infinity-Cache hits included.
usage: traffic_json.py <fetch.txt> <write.txt> <out.json>"""
import json
import sys


def table(path):
    out = {}
    for line in open(path):
        if line.startswith("#") or line.startswith("kernel"):
            continue
        parts = line.rstrip().rsplit(None, 2)
        if len(parts) == 3:
            try:
                out[parts[0].strip()] = (int(parts[1]), float(parts[2]))
            except ValueError:
                pass
    return out


def main():
    f, w = table(sys.argv[1]), table(sys.argv[2])
    groups = {"conv_igemm_kernel (forward + data-gradient, all tile shapes / modes)": "conv_igemm_kernel",
              "conv_wgrad_kernel": "conv_wgrad_kernel", "bn_bwd_apply_kernel": "bn_bwd_apply_kernel",
              "bn_apply_kernel": "bn_apply_kernel", "colreduce_kernel<BnBwdOp": "colreduce_kernel<BnBwdOp",
              "photometric_bwd_kernel": "photometric_bwd2_kernel", "photometric_fwd_kernel": "photometric_fwd2_kernel",
              "wino_fused_kernel (one-kernel Winograd convolution)": "wino_fused_kernel",
              "wino_in_kernel (Winograd input transform)": "wino_in_kernel", "wino_out_kernel (Winograd output transform)": "wino_out_kernel",
              "wino_grad_kernel (Winograd output-gradient transform)": "wino_grad_kernel",
              "wino_wgrad_fused_kernel (one-kernel Winograd weight gradient)": "wino_wgrad_fused_kernel",
              "wino_wgrad_finish_kernel (slab fold + G^T dU G)": "wino_wgrad_finish_kernel",
              "reflect_borders_kernel (mirrored-padding terms of the Winograd data-gradient)": "reflect_borders_kernel"}
    res = {"command": "rocprofv3 --kernel-trace --pmc <FETCH_SIZE | WRITE_SIZE> -- python bench.py --steps 2 --warmup 1 "
                      "--no-cpu-baseline --no-kernel-timing (tools/gpu_pmc.sh: one pass per counter)",
           "correction": "gfx950: read bytes = 2 x FETCH_SIZE KB x 1024 (64 B tallied per 128-B request on wide coalesced reads, "
                         "MI355X_MICROARCH.md HBM section); WRITE_SIZE as reported; fabric-side counts (Infinity-Cache hits included)",
           "steps_profiled": 3,      # --steps 2 --warmup 1: the counters cover warm-up and timed steps alike
           "kernels": {}}
    for label, prefix in groups.items():
        nf = sum(v[0] for k, v in f.items() if k.startswith(prefix))
        kf = sum(v[1] for k, v in f.items() if k.startswith(prefix))
        kw = sum(v[1] for k, v in w.items() if k.startswith(prefix))
        if nf:
            res["kernels"][label] = {"launches": nf, "FETCH_SIZE_KB": kf, "WRITE_SIZE_KB": kw,
                                     "bytes_per_launch": (2 * kf + kw) * 1024.0 / nf}
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import csrc_sha256
    res["csrc_sha256"] = csrc_sha256()     # bench.py refuses this file once the kernel sources change
    json.dump(res, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
