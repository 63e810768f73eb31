#!/usr/bin/env python3
"""What the host of the GPU box really gives the CPU baseline: logical CPUs, affinity, cgroup quota, and how the baseline's
multi-exponentiation (oracle/cpubase.hpp) scales with OpenMP threads.  Output is kept under profiles/ to back the `cores` field
of bench.py's cpu_baseline."""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np  # noqa: E402
import oracle as O  # noqa: E402


def read(path):
    try:
        return open(path).read().strip()
    except OSError as e:
        return f"<{e.__class__.__name__}>"


def main():
    print("os.cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "omp default threads", O.threads())
    print("cgroup cpu.max:", read("/sys/fs/cgroup/cpu.max"), "| cpu.stat:", read("/sys/fs/cgroup/cpu.stat").replace("\n", " "))
    for line in read("/proc/cpuinfo").split("\n"):
        if line.startswith("model name"):
            print(line)
            break
    print("loadavg", read("/proc/loadavg"))
    omp = ctypes.CDLL("libgomp.so.1")
    k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    n = 1 << k
    sc = O.fr_random(1, n)
    base = O.fr_random(2, 4096)
    p1 = np.tile(O.g1_from_scalars(base), (n // 4096 + 1, 1))[:n].copy()
    a = O.fr_random(3, n); b = O.fr_random(4, n); c = O.fr_mul(a, b)
    O.fast_g1_msm(p1[:1024], sc[:1024])
    t1 = None
    for nt in (1, 4, 16, 64, 128, 256):
        if nt > (os.cpu_count() or 1):
            break
        omp.omp_set_num_threads(nt)
        t0 = time.time(); O.fast_g1_msm(p1, sc); dt = time.time() - t0
        t0 = time.time(); O.fast_compute_h(a, b, c, k); dh = time.time() - t0
        t1 = t1 or dt
        print(f"threads {nt:4d}: G1 MultiExp 2^{k} {dt:7.3f}s (speed-up {t1 / dt:5.1f}), computeH incl. tables {dh:7.3f}s")


if __name__ == "__main__":
    main()
