"""round 4: the compiled BatchCreateUserCircuit solved and proved on the device, first contact (run through gpurun)."""
import ctypes, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
for p in (os.path.join(ROOT, "zkmerkle-proof-of-solvency_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import zkpor, circuit as C
import oracle as O, trapdoor as T

def run(shape, compare, seed=0x5A4B504F52, variant=1):
    t0 = time.time()
    inp = C.synth_inputs(*shape)
    t1 = time.time()
    cir = C.Circuit(*shape)
    t2 = time.time()
    print(shape, "synth %.2fs compile %.2fs" % (t1 - t0, t2 - t1), {k: cir.dims[k] for k in ("n_wires", "n_constraints", "n_instructions", "n_levels", "n_committed")}, flush=True)
    ctx = zkpor.Context(0)
    ctx.set_param("solver_poseidon", variant)
    print("  solver_poseidon =", variant)
    log2 = max(4, int(np.ceil(np.log2(max(cir.n_constraints, 2)))))
    D = 1 << log2
    pk = zkpor.ProvingKey(ctx)
    pk.synth(log2, cir.n_wires, cir.n_public, cir.n_committed, seed)
    t3 = time.time()
    dc = C.DeviceCircuit(ctx, cir)
    t4 = time.time()
    print("  key %.2fs upload+create %.2fs" % (t3 - t2, t4 - t3), dc.solver.dims(), flush=True)
    bufs = [ctx.alloc(32 * n) for n in (cir.n_wires, D, D, D, cir.n_committed + 1)]
    try:
        rows = os.environ.get("E2E_ROWS", "0") == "1"
        if rows:
            dc.solver.set_abc_dev(bufs[1].ptr, bufs[2].ptr, bufs[3].ptr)
        for rep in range(3):
            tm = {}
            ctx.phase_reset()
            ta = time.perf_counter()
            com, pok, ch = C.solve_on_device(ctx, dc, pk, bufs[0].ptr, bufs[4].ptr, inp, tm)
            tb = time.perf_counter()
            if rows:
                dc.solver.eval_abc_dev(bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, bufs[3].ptr, D)
            else:
                dc.r1cs.eval_dev(bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, bufs[3].ptr, D)
            ctx.sync()
            tc = time.perf_counter()
            rr = O.fr_random(171 + rep, 1)[0]; ss = O.fr_random(272 + rep, 1)[0]
            proof = ctx.prove_tail_dev(pk, bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, bufs[3].ptr, rr, ss)
            td_ = time.perf_counter()
            print("  rep %d: solve %.1f ms (%s) abc %.1f ms tail %.1f ms  launches %d  solver_levels phase %.1f ms" % (
                rep, (tb - ta) * 1e3, {k: round(v, 1) for k, v in tm.items()}, (tc - tb) * 1e3, (td_ - tc) * 1e3, dc.solver.dims()["launches_last_run"], ctx.phase_ms("solver_levels")[0]), flush=True)
        bad = dc.r1cs.check_dev(bufs[0].ptr)
        print("  constraints failing on the device-solved wires:", bad, flush=True)
        w = bufs[0].download(np.uint64, (cir.n_wires, 4))
        h = bufs[1].download(np.uint64, (D, 4))
        cv = bufs[4].download(np.uint64, (cir.n_committed + 1, 4))[1:]
        assert np.array_equal(cv, w[cir.committed()]), "committed values are not the committed wires"
        ec, ek = T.expected_commitment(seed, cv)
        print("  commitment / pok vs trapdoor:", bool(np.array_equal(com, ec)), bool(np.array_equal(pok, ek)))
        tdr = T.SynthKeyTrapdoor(seed, cir.n_public, w, h[: D - 1])
        print("  proof vs trapdoor:", tdr.check(proof, rr, ss), flush=True)
        if compare:
            cir2 = C.Circuit(*shape, inputs=inp, commitment=ch)
            same = np.array_equal(w, cir2.values())
            print("  device w == interpreter (same challenge):", bool(same))
            if not same:
                badw = np.nonzero((w != cir2.values()).any(axis=1))[0]; print("   first differing wires", badw[:8], len(badw))
            wh = cir.solve_host(inp, ch, threads=8)
            print("  device w == host executor:", bool(np.array_equal(w, wh)))
            cir2.close()
    finally:
        for b in bufs: b.free()
        dc.close(); pk.close(); ctx.close(); cir.close()

if __name__ == "__main__":
    shapes = [((3, 6, 3), True, 1), ((3, 6, 3), True, 0), ((5, 20, 6), True, 1), ((50, 500, 8), True, 1), ((50, 500, 8), False, 0), ((500, 500, 2), True, 1)]
    if len(sys.argv) > 1:
        shapes = [(tuple(int(x) for x in sys.argv[1:4]), len(sys.argv) > 5, int(sys.argv[4]) if len(sys.argv) > 4 else 1)]
    for sh, cmp_, var in shapes:
        run(sh, cmp_, variant=var)
