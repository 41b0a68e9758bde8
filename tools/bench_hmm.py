#!/usr/bin/env python3
"""Time of the device HMM recursion (vgmi_hmm_recursion) at the shape of a chr20-scale sample: 60 windows x 2 directions,
120 genotypes, `steps` nodes per window."""
import json, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from varigraph_amd import vgmi
LD = np.longdouble
def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    n_windows, n, ploidy = (int(sys.argv[2]) if len(sys.argv) > 2 else 60), 120, 2
    rng = np.random.default_rng(1)
    keep = rng.integers(0, 3, size=(n_windows, n, n), dtype=np.uint8)
    n_rows = n_windows * steps
    obs = (rng.random((n_rows, n)).astype(LD) + LD(0.01)) * np.power(LD(10), rng.integers(-300, 0, size=(n_rows, n)).astype(LD))
    row, restart, chains = [], [], []
    for w in range(n_windows):
        for d in (1, -1):
            rows = np.arange(w * steps, (w + 1) * steps, dtype=np.uint32)[::d]
            chains.append((len(row) * 0 + sum(c[1] for c in chains), steps, w))
            row.append(rows)
            r = np.zeros(steps, dtype=np.uint8); r[0] = 1
            restart.append(r)
    row = np.concatenate(row); restart = np.concatenate(restart)
    pw = np.empty((row.size, 2, ploidy + 1), dtype=LD)
    pw[:, 0, :] = np.array([1, 0.9999, 0.9998], dtype=LD)
    pw[:, 1, :] = np.array([1, 1e-5, 1e-10], dtype=LD)
    ctx = vgmi.Context(0, buffer_mib=16)
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        out = ctx.hmm_recursion(keep, obs, row, restart, pw, LD(1) / LD(n), chains, ploidy)
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    ctx.close()
    print(json.dumps({"windows": n_windows, "steps_per_chain": steps, "genotypes": n, "call_s": best,
                      "us_per_node_and_pass_if_kernel_bound": best / steps * 1e6, "bytes_in": obs.nbytes + pw.nbytes, "bytes_out": out.nbytes}))
if __name__ == "__main__":
    main()
