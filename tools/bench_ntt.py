"""times zkpor_fft_dev / zkpor_compute_h_dev at 2^log2 for the NTT variants (device-resident input, HIP-event-free wall clock
around a synchronised batch).  usage: python tools/bench_ntt.py [log2] """
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "zkmerkle-proof-of-solvency_amd"))
import zkpor

log2 = int(sys.argv[1]) if len(sys.argv) > 1 else 26
ctx = zkpor.Context(0)
N = 1 << log2
bufs = [ctx.alloc(32 * N) for _ in range(3)]
for i, b in enumerate(bufs):
    ctx.fill_fr(b, N, 5 + i, 0)
L = ctx.lib


def t_fft(reps=4):
    ctx._ck(L.zkpor_fft_dev(ctx.h, ctypes.c_void_p(bufs[0].ptr), ctypes.c_int(log2), 0, 1, 0)); ctx.sync()
    t0 = time.time()
    for _ in range(reps):
        ctx._ck(L.zkpor_fft_dev(ctx.h, ctypes.c_void_p(bufs[0].ptr), ctypes.c_int(log2), 0, 1, 0))
    ctx.sync()
    return (time.time() - t0) / reps * 1e3


def t_h(reps=2):
    ctx._ck(L.zkpor_compute_h_dev(ctx.h, ctypes.c_int(log2), *[ctypes.c_void_p(b.ptr) for b in bufs])); ctx.sync()
    t0 = time.time()
    for _ in range(reps):
        ctx._ck(L.zkpor_compute_h_dev(ctx.h, ctypes.c_int(log2), *[ctypes.c_void_p(b.ptr) for b in bufs]))
    ctx.sync()
    return (time.time() - t0) / reps * 1e3


for variant, tiles in ((0, [10]), (1, [9, 10, 11, 12])):
    ctx.set_param("ntt_variant", variant)
    for tl in tiles:
        if variant == 1:
            ctx.set_param("ntt_tile_log", tl)
        print(f"variant {variant} tile_log {tl}: fft {t_fft():.2f} ms  computeH {t_h():.2f} ms", flush=True)
