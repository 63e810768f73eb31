"""-m gpu: CEX asset-list commitments and batch commitments on the device (SURVEY.md §8 a12 / f3;
src/witness/witness/witness.go:159-198, src/utils/utils.go:26-88,779-800) bit-exact with the oracle."""
import numpy as np
import pytest

import cex_cases as C
import oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_assets,n_states", [(1, 1), (9, 3), (500, 2), (37, 130)])
def test_cex_commitments_match_oracle(zk, n_assets, n_states):
    consts = C.make_assets(n_assets, seed=10 + n_assets)
    totals = C.make_totals(n_states, n_assets, seed=20 + n_states)
    got = zk.cex_commitments(consts, totals)
    ref = O.fr_to_be(O.cex_commitments(consts, totals))
    assert np.array_equal(got, ref)


def test_commitment_depends_on_every_field(zk):
    consts = C.make_assets(5, seed=1)
    totals = C.make_totals(1, 5, seed=2)
    base = zk.cex_commitments(consts, totals)[0].tobytes()
    for name in O.CEX_TOTALS_DTYPE.names:
        t2 = totals.copy(); t2[0, 3][name] ^= np.uint64(1)
        assert zk.cex_commitments(consts, t2)[0].tobytes() != base
    c2 = consts.copy(); c2[4]["margin"][11]["ratio"] ^= 1
    assert zk.cex_commitments(c2, totals)[0].tobytes() != base
    c3 = consts.copy(); c3[0]["base_price"] += np.uint64(1)
    assert zk.cex_commitments(c3, totals)[0].tobytes() != base


def test_batch_commitments_match_oracle(zk):
    n = 70
    roots = O.fr_to_be(O.fr_random(1, n)); before = O.fr_to_be(O.fr_random(2, n)); after = O.fr_to_be(O.fr_random(3, n))
    mn = np.arange(n, dtype=np.uint32) * 1380; mx = mn + 1379
    mn[0] = 0                                                     # the [0x00] case of witness.go:186-193
    mx[-1] = 0xFFFFFFFF
    got = zk.batch_commitments(roots, before, after, mn, mx)
    for i in (0, 1, n // 2, n - 1):
        el = np.concatenate([O.fr_from_be(roots[i:i + 1]), O.fr_from_be(before[i:i + 1]), O.fr_from_be(after[i:i + 1]),
                             O.fr_from_ints([int(mn[i]), int(mx[i])])])
        assert got[i].tobytes() == O.fr_to_be(O.poseidon_hash(el)[None, :])[0].tobytes()
