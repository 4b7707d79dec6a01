"""CPU: the oracle restatement (oracle/kbest_oracle.c) against the golden
vectors produced by the unmodified reference solver, the SURVEY 8(c)
known-answer vectors, and -- where oracle/_ref exists -- the compiled
reference itself on fresh random inputs."""
import numpy as np
import pytest

import oracle_lib as ol
from probabilisticsemslam_amd import workloads as wl


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.int64)


def test_generator_kat():
    # first inputs of seed 12345
    c = wl.splitmix64_u01(12345, 4)
    assert c.tolist() == [0.13307966866142729, 0.20481663336165912, 0.11954258300911547, 0.17611780724496118]


def test_survey_kat_c1():
    cs, N, M, k = wl.dense_config("c1")
    nf, r4c, c4r, g = ol.orc_kbest(cs[0], N, M, k)
    assert nf == 10
    want = ["0x1.0bd6f90d82018p+0", "0x1.16a04244a80aep+0", "0x1.2297fc8bb94f4p+0", "0x1.2547e0bbea402p+0",
            "0x1.297a0204a0b0cp+0", "0x1.2a3797477bb64p+0", "0x1.2d6145c2df588p+0", "0x1.301129f310497p+0",
            "0x1.3500e07ea1bfap+0", "0x1.4031bb872b758p+0"]
    assert [float.hex(x) for x in g] == want
    assert r4c[0].tolist() == [0, 2, 7, 5, 3, 4, 6, 1]
    assert r4c[1].tolist() == [0, 2, 7, 5, 1, 4, 6, 3]
    assert r4c[9].tolist() == [5, 2, 7, 0, 3, 4, 6, 1]


@pytest.mark.parametrize("name,g0,glast,r4c0", [
    ("c2", "0x1.6935446741033p+0", "0x1.a077c1a7b2d2dp+0", [12, 9, 1, 2]),
    ("c3", "0x1.6b3e0fc692ea4p+0", "0x1.855a55211d816p+0", [14, 17, 16, 3]),
    ("c4", "0x1.33cde2c7b780cp+0", "0x1.3e1e2d1a163acp+0", [8, 46, 2, 48]),
])
def test_survey_kat_streams(name, g0, glast, r4c0):
    # first problem of each dense stream
    cs, N, M, k = wl.dense_config(name, B=1)
    nf, r4c, c4r, g = ol.orc_kbest(cs[0], N, M, k)
    assert nf == k
    assert float.hex(g[0]) == g0 and float.hex(g[k - 1]) == glast
    assert r4c[0][:4].tolist() == r4c0


def test_all_golden_cases(golden):
    for name in golden.names:
        c = golden.case(name)
        nf, r4c, c4r, g = ol.orc_kbest(c["cost"], c["N"], c["M"], c["k"], c["maximize"], c["cutoff"])
        assert nf == c["nf"], name
        assert (r4c[:nf] == c["row4col"]).all(), name
        assert (c4r[:nf] == c["col4row"]).all(), name          # verbatim, not even canonicalised
        assert (bits(g[:nf]) == bits(c["gain"])).all(), name


@pytest.mark.skipif(not ol.have_ref(), reason="oracle/_ref not built (no /root/reference here)")
def test_oracle_vs_compiled_reference_random():
    rng = np.random.default_rng(7)
    for trial in range(120):
        N = int(rng.integers(1, 20))
        M = int(rng.integers(1, N + 1))
        k = int(rng.integers(1, 40))
        cost = rng.random(N * M) * 10 - 3
        if trial % 3 == 0:
            cost[rng.random(N * M) < 0.3] = np.inf
        maximize = trial % 5 == 0
        if maximize:
            cost = np.where(np.isinf(cost), -np.inf, cost)
        cutoff = [None, 0.5, 3.0][trial % 3]
        a = ol.orc_kbest(cost, N, M, k, maximize, cutoff)
        r = ol.ref_kbest(cost, N, M, k, maximize, cutoff)
        assert a[0] == r[0], trial
        nf = a[0]
        assert (a[1][:nf] == r[1][:nf]).all() and (a[2][:nf] == r[2][:nf]).all(), trial
        assert (bits(a[3][:nf]) == bits(r[3][:nf])).all(), trial


@pytest.mark.skipif(not ol.have_ref(), reason="oracle/_ref not built (no /root/reference here)")
def test_oracle_vs_compiled_reference_ties():
    # integer costs: massive exact ties.  The oracle restates libstdc++'s heap
    # sift rules, so even the order among equal gains matches.
    rng = np.random.default_rng(11)
    for trial in range(60):
        N = int(rng.integers(2, 9))
        M = int(rng.integers(1, N + 1))
        k = int(rng.integers(1, 60))
        cost = rng.integers(0, 4, N * M).astype(np.float64)
        a = ol.orc_kbest(cost, N, M, k)
        r = ol.ref_kbest(cost, N, M, k)
        assert a[0] == r[0]
        nf = a[0]
        assert (a[1][:nf] == r[1][:nf]).all() and (bits(a[3][:nf]) == bits(r[3][:nf])).all()


def test_properties_vs_permutation_brute_force():
    # exhaustive small cases equal brute force over all injections cols -> rows
    import itertools
    rng = np.random.default_rng(3)
    for N, M in ((4, 4), (5, 3), (6, 2), (3, 3), (5, 5)):
        cost = rng.random(N * M)
        allg = []
        for rows in itertools.permutations(range(N), M):
            allg.append((sum(cost[c * N + r] for c, r in enumerate(rows)), rows))
        allg.sort()
        k = len(allg) + 5
        nf, r4c, c4r, g = ol.orc_kbest(cost, N, M, k)
        assert nf == len(allg)
        assert [tuple(x) for x in r4c[:nf].tolist()] == [rows for _, rows in allg]
        assert np.all(np.diff(g[:nf]) >= 0)
        np.testing.assert_allclose(g[:nf], [x for x, _ in allg], rtol=1e-13)


def test_assign2d_matches_first_of_kbest():
    rng = np.random.default_rng(5)
    for N, M in ((7, 7), (9, 4), (12, 12)):
        cost = rng.random(N * M)
        c4r = np.zeros(N, np.int32)
        r4c = np.zeros(M, np.int32)
        g = np.zeros(1)
        assert ol.oracle().orc_assign2d(N, M, 0, cost, c4r, r4c, g) == 1
        nf, R, Cc, G = ol.orc_kbest(cost, N, M, 1)
        assert (r4c == R[0]).all()
        assert g[0] == G[0]
