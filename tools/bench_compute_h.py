"""computeH alone on the device at 2^log2 (zkpor_compute_h_dev, device-resident a, b, c): the six-transform schedule ("ntt_h" 1, the default) against gnark's
seven ("ntt_h" 0), wall clock around a synchronised batch, nothing else on the GPU.  Prints one JSON line.  usage: python tools/bench_compute_h.py [log2] [reps]"""
import ctypes, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "zkmerkle-proof-of-solvency_amd"))
import zkpor

log2 = int(sys.argv[1]) if len(sys.argv) > 1 else 26
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
ctx = zkpor.Context(0)
N = 1 << log2
bufs = [ctx.alloc(32 * N) for _ in range(3)]
L = ctx.lib
run = lambda: ctx._ck(L.zkpor_compute_h_dev(ctx.h, ctypes.c_int(log2), *[ctypes.c_void_p(b.ptr) for b in bufs]))
out = {"log2": log2, "reps": reps, "what": "zkpor_compute_h_dev alone on one MI355X, ms per computeH"}
for h in (1, 0, 1, 0):
    ctx.set_param("ntt_h", h)
    for i, b in enumerate(bufs):
        ctx.fill_fr(b, N, 5 + i, 0)
    run(); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        run()
    ctx.sync()
    out.setdefault("six_transforms_ms" if h else "seven_transforms_ms", []).append(round((time.perf_counter() - t0) / reps * 1e3, 2))
ctx.set_param("ntt_h", 1)
# 7 transforms x (read + write) x 32 B per element + the pointwise step's 4 x 32 B = 18 x 32 B x D algorithmic bytes per computeH
alg = 18 * 32 * N
out["algorithmic_bytes"] = alg
out["algorithmic_GBps_six"] = round(alg / (min(out["six_transforms_ms"]) * 1e-3) / 1e9, 1)
out["frac_of_8TBps_six"] = round(out["algorithmic_GBps_six"] / 8000.0, 4)
print(json.dumps(out))
