"""assign2D (shortestPathCPP.hpp:144-149, cpp:735-762) and shortestPathCPP (hpp:178-182, cpp:119-238) against golden
vectors recorded from the unmodified reference (tests/golden/assign_golden.npz, gen_assign_golden.py): rectangular,
maximise, infeasible, numCol4Gain < numCol -- assignments, gain AND the dual variables u, v bit for bit."""
import os

import numpy as np
import pytest

import oracle_lib as ol

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "assign_golden.npz")


def _cases():
    z = np.load(GOLD)
    for name in z["names"]:
        name = str(name)
        N, M, maximize, shift, gc, ok = (int(x) for x in z[name + "/meta"])
        yield dict(name=name, N=N, M=M, maximize=bool(maximize), shift=bool(shift), gc=gc, ok=ok, cost=z[name + "/cost"],
                   r4c=z[name + "/row4col"].astype(np.int64), c4r=z[name + "/col4row"].astype(np.int64),
                   gain=float(z[name + "/gain"][0]), u=z[name + "/u"], v=z[name + "/v"])


CASES = list(_cases())


def _check(c, ok, r4c, c4r, g, u, v):
    assert int(ok) == c["ok"], c["name"]
    assert np.float64(g).view(np.int64) == np.float64(c["gain"]).view(np.int64), (c["name"], g, c["gain"])
    if c["ok"]:
        assert np.asarray(r4c).tolist() == c["r4c"].tolist(), c["name"]
        assert np.asarray(c4r).tolist() == c["c4r"].tolist(), c["name"]
        assert np.asarray(u).view(np.int64).tolist() == c["u"].view(np.int64).tolist(), c["name"]
        assert np.asarray(v).view(np.int64).tolist() == c["v"].view(np.int64).tolist(), c["name"]


def test_oracle_assign_matches_reference_golden():
    for c in CASES:
        _check(c, *ol.orc_assign2d_ex(c["cost"], c["N"], c["M"], c["maximize"], c["shift"], c["gc"]))


@pytest.mark.skipif(not ol.have_ref(), reason="oracle/_ref not built (needs /root/reference)")
def test_golden_is_what_the_compiled_reference_returns():
    for c in CASES:
        _check(c, *ol.ref_assign2d_ex(c["cost"], c["N"], c["M"], c["maximize"], c["shift"], c["gc"]))


@pytest.mark.gpu
def test_gpu_assign_matches_reference_golden():
    import probabilisticsemslam_amd as pk
    eng = pk.KBestEngine(0)
    for c in CASES:
        ok, r4c, c4r, g, u, v = eng.assign(c["cost"], c["N"], c["M"], c["maximize"], c["shift"], c["gc"])
        _check(c, ok[0], r4c[0], c4r[0], g[0], u[0], v[0])
    # a batch of equal shapes in one launch
    rng = np.random.default_rng(3)
    costs = rng.random((40, 24 * 9)) * 5 - 1
    ok, r4c, c4r, g, u, v = eng.assign(costs, 24, 9, maximize=True)
    for b in range(40):
        o = ol.orc_assign2d_ex(costs[b], 24, 9, True, True, 0)
        assert ok[b] == o[0] and r4c[b].tolist() == o[1].tolist() and c4r[b].tolist() == o[2].tolist()
        assert g[b] == o[3] and u[b].tolist() == o[4].tolist() and v[b].tolist() == o[5].tolist()


@pytest.mark.gpu
def test_gpu_to_probs_matches_reference_golden():
    import probabilisticsemslam_amd as pk
    eng = pk.KBestEngine(0)
    z = np.load(os.path.join(os.path.dirname(GOLD), "weights_golden.npz"))
    for name in z["names"]:
        got = eng.to_probs(z[str(name) + "/cond"])
        np.testing.assert_allclose(got, z[str(name) + "/toProbs"], rtol=1e-14, atol=0, err_msg=str(name))
