"""Test-side WRITER of the flat constraint-system container (layout: go/export_r1cs/main.go header).  The product only reads
it (host/r1cs_file.hpp); on a box with Go the file comes from go/export_r1cs run over gnark's .r1cs."""
import struct

import numpy as np


def write(n_constraints, n_wires, n_public, n_secret, coeff_table, mats, commitments=()):
    """mats: [(row_ptr u64[n+1], coeff_ids u32[nnz], wire_ids u32[nnz])] x 3; commitments: [(index, private[], public[])]"""
    coeff_table = np.ascontiguousarray(coeff_table, dtype=np.uint64).reshape(-1, 4)
    out = bytearray(b"ZKPR1CS\x01")
    out += struct.pack("<9Q", n_constraints, n_wires, n_public, n_secret, coeff_table.shape[0],
                       *(len(m[1]) for m in mats), len(commitments))
    for idx, priv, pub in commitments:
        out += struct.pack("<3Q", idx, len(priv), len(pub))
        out += np.asarray(priv, dtype="<u4").tobytes() + np.asarray(pub, dtype="<u4").tobytes()
    out += b"\0" * (-len(out) % 8)
    out += coeff_table.astype("<u8").tobytes()
    for row_ptr, cid, wid in mats:
        out += np.asarray(row_ptr, dtype="<u8").tobytes() + np.asarray(cid, dtype="<u4").tobytes() + np.asarray(wid, dtype="<u4").tobytes()
        out += b"\0" * (-len(out) % 8)
    return bytes(out)


def from_synth(S, commitments=()):
    table, mats = S.r1cs()
    return write(S.n_cons, S.n_wires, S.n_public, S.n_wires - S.n_public, table, mats, commitments)
