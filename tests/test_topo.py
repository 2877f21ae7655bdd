"""Config "tools/topo_expl 8-GPU ring all-reduce schedule synthesis" — CPU only.

Fixtures: the reference's own topology model file (tools/topo_expl/models/topo_8p_940.xml, copied
as data) and the ring set its model table holds for that topology (rome_models.cc:558).  The
library's synthesiser is not a table lookup, so it need not return the same 14 rings; it must return
a set with the same defining properties, which are asserted on the reference's set first.
"""
import os

import pytest

import rccl_amd as ra

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _golden_rings():
    rings = []
    for line in open(os.path.join(GOLD, "rome_model_79_rings.txt")):
        if line.strip() and not line.startswith("#"):
            rings.append([int(x) for x in line.split()])
    return rings


def _coverage(rings, n):
    cov = [[0] * n for _ in range(n)]
    for r in rings:
        assert sorted(r) == list(range(n)), r               # Hamiltonian: every rank once
        for i in range(n):
            cov[r[i]][r[(i + 1) % n]] += 1
    return cov


def _check_balanced(rings, n, adj):
    cov = _coverage(rings, n)
    uses = {cov[u][v] for u in range(n) for v in range(n) if u != v and adj[u][v]}
    assert len(uses) == 1 and 0 not in uses, cov            # every directed link used equally often
    for u in range(n):
        for v in range(n):
            if u == v or not adj[u][v]:
                assert cov[u][v] == 0                       # only real links are used
    return uses.pop()


def test_reference_ring_set_properties():
    rings = _golden_rings()
    assert len(rings) == 14
    full = [[int(i != j) for j in range(8)] for i in range(8)]
    assert _check_balanced(rings, 8, full) == 2
    # rank 0's next hops over rings 0-13 cover all 7 neighbours twice each
    nxt = sorted(r[(r.index(0) + 1) % 8] for r in rings)
    assert nxt == sorted(list(range(1, 8)) * 2)
    # second half = first half reversed
    for f, b in zip(rings[:7], rings[7:]):
        i = b.index(f[0])
        assert (b[i:] + b[:i])[1:] == f[1:][::-1]


def test_xml_adjacency_8p_and_4p():
    adj = ra.topo_parse_xml(open(os.path.join(GOLD, "topo_8p_940.xml")).read())
    assert len(adj) == 8
    assert all(adj[i][j] == int(i != j) for i in range(8) for j in range(8))     # all-to-all xGMI
    adj4 = ra.topo_parse_xml(open(os.path.join(GOLD, "topo_4p_940.xml")).read())
    assert len(adj4) == 4 and all(adj4[i][j] == int(i != j) for i in range(4) for j in range(4))


def test_synthesis_matches_reference_properties_8p():
    adj = ra.topo_parse_xml(open(os.path.join(GOLD, "topo_8p_940.xml")).read())
    rings = ra.topo_rings(8, adj)
    assert len(rings) == 14                                                       # as the reference: 7 + 7 reversed
    assert _check_balanced(rings, 8, adj) == 2
    assert _check_balanced(rings[:7], 8, adj) == 1                                # forward half alone is a decomposition
    for f, b in zip(rings[:7], rings[7:]):
        assert b[0] == 0 and b[1:] == f[1:][::-1]
    nxt = sorted(r[1] for r in rings)
    assert nxt == sorted(list(range(1, 8)) * 2)


@pytest.mark.parametrize("n", [2, 3, 4, 5, 6, 7, 8])
def test_synthesis_all_to_all(n):
    adj = [[int(i != j) for j in range(n)] for i in range(n)]
    rings = ra.topo_rings(n, adj)
    assert rings and all(r[0] == 0 for r in rings)
    if n > 2:
        _check_balanced(rings, n, adj)


def test_synthesis_on_a_plain_ring_topology():
    # 6 GPUs wired as one bidirectional ring: the only Hamiltonian cycles are the ring and its reverse
    n = 6
    adj = [[int((i - j) % n in (1, n - 1)) for j in range(n)] for i in range(n)]
    rings = ra.topo_rings(n, adj)
    assert sorted(map(tuple, rings)) == sorted([tuple(range(n)), (0, 5, 4, 3, 2, 1)])


def test_synthesis_rejects_bad_arguments():
    with pytest.raises(ra.NcclError):
        ra.topo_rings(0, [])


def test_ring_orders_match_the_committed_fixture():
    """The ring orders the kernels walk on all-to-all nodes of 2..16 ranks are pinned: tests/expected.py builds its
    ring- and chain-order expectations from this fixture, so the search cannot drift without a test saying so."""
    import json
    with open(os.path.join(GOLD, "ring_orders_all_to_all.json")) as f:
        fixture = json.load(f)
    for n in range(2, 17):
        adj = [[0 if i == j else 1 for j in range(n)] for i in range(n)]
        assert ra.topo_rings(n, adj) == fixture[str(n)], n
