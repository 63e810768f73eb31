"""round-4 probe: where k_r1cs_eval's 31 ms go once the Poseidon rows are skipped — the production matrices with classes of rows emptied
(timing only: the wire vector is whatever the buffer holds)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
for p in (os.path.join(ROOT, "zkmerkle-proof-of-solvency_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np
import zkpor, circuit as C

shape = tuple(int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (50, 500, 1380)
cir = C.Circuit(*shape)
ctx = zkpor.Context(0)
D = 1 << int(np.ceil(np.log2(cir.n_constraints)))
mats = [cir.matrix(m) for m in range(3)]
bufs = [ctx.alloc(32 * n) for n in (cir.n_wires, D, D, D)]
ctx.fill_fr(bufs[0], cir.n_wires, 3, 0)

def timed(r1cs, solver, label):
    if solver is not None: solver.set_abc_dev(bufs[1].ptr, bufs[2].ptr, bufs[3].ptr)
    best = 1e9
    for _ in range(3):
        ctx.phase_reset()
        if solver is not None: solver.eval_abc_dev(bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, bufs[3].ptr, D)
        else: r1cs.eval_dev(bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, bufs[3].ptr, D)
        ctx.sync()
        best = min(best, ctx.phase_ms("r1cs_eval")[0])
    print(f"{label}: {best:.2f} ms", flush=True)

def variant(keep):
    """keep(m, lengths) -> bool mask of the rows that keep their terms"""
    r = zkpor.R1CS(ctx, cir.n_constraints, cir.n_wires, cir.coeff())
    tot = 0
    for m, (rp, cid, wid) in enumerate(mats):
        ln = np.diff(rp.astype(np.int64))
        k = keep(m, ln)
        tm = np.repeat(k, ln)
        nl = np.where(k, ln, 0)
        nrp = np.concatenate([[0], np.cumsum(nl)]).astype(np.uint64)
        r.set_matrix(m, nrp, cid[tm], wid[tm])
        tot += int(nl.sum())
    return r, tot

container = cir.solver_container()
r0, t0 = variant(lambda m, ln: np.ones(len(ln), bool))
s0 = zkpor.Solver(r0, container, ctx=ctx)
timed(r0, None, f"all rows, {t0} terms")
timed(r0, s0, "Poseidon rows skipped")
s0.close(); r0.close()
for cap in (256, 16, 3):
    r, t = variant(lambda m, ln: ln <= cap)
    s = zkpor.Solver(r, container, ctx=ctx)
    timed(r, None, f"rows of more than {cap} terms emptied ({t} terms left), nothing skipped")
    timed(r, s, f"rows of more than {cap} terms emptied, Poseidon rows skipped")
    s.close(); r.close()
for only in (0, 1, 2):
    r, t = variant(lambda m, ln: np.full(len(ln), m == only))
    s = zkpor.Solver(r, container, ctx=ctx)
    timed(r, s, f"matrix {only} only ({t} terms), Poseidon rows skipped")
    s.close(); r.close()
r, t = variant(lambda m, ln: np.zeros(len(ln), bool))
timed(r, None, "every row empty (the launch, the row pointers and the zero stores)")
r.close()
