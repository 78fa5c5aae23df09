"""The simplex fan regenerated from the closed-form rule must equal the tables the reference
builds at run time (mesh/simplicial_regular_mesh.hh:891-927), dumped by oracle/_ref/ftk_ref_driver tables."""
import os
import re

from common import GOLDEN


def parse_tables():
    out = {}
    cur = None
    for line in open(os.path.join(GOLDEN, "unit_simplex_tables.txt")):
        m = re.match(r"mesh (\d+) dim (\d+) ntypes (\d+) ordinal (\d+) interval (\d+)", line)
        if m:
            cur = dict(n=int(m[1]), ntypes=int(m[3]), n_ord=int(m[4]), n_int=int(m[5]), types={}, side_of={}, sides={})
            out[cur["n"]] = cur
            continue
        m = re.match(r"type (\d+) ordinal (\d) :((?: \d+)+) \| side_of:(.*)", line)
        if m:
            t = int(m[1])
            cur["types"][t] = (bool(int(m[2])), [[int(ch) for ch in v] for v in m[3].split()])
            cur["side_of"][t] = [(int(a), tuple(int(x) for x in b.split())) for a, b in re.findall(r"\((\d+);([^)]*)\)", m[4])]
            continue
        m = re.match(r"cell (\d+) sides:(.*)", line)
        if m:
            cur["sides"][int(m[1])] = [(int(a), tuple(int(x) for x in b.split())) for a, b in re.findall(r"\((\d+);([^)]*)\)", m[2])]
    return out


def test_unit_simplex_tables_match_reference(oracle):
    ref = parse_tables()
    for n in (3, 4):
        verts, is_ord = oracle.unit_simplices(n)
        r = ref[n]
        assert len(verts) == r["ntypes"] == {3: 12, 4: 60}[n]
        assert int(is_ord.sum()) == r["n_ord"] == {3: 2, 4: 6}[n]
        for t in range(r["ntypes"]):
            assert bool(is_ord[t]) == r["types"][t][0]
            assert verts[t].tolist() == r["types"][t][1], f"n={n} type {t}"
            assert oracle.side_of(n, t) == r["side_of"][t], f"n={n} side_of type {t}"
        for c, s in r["sides"].items():
            assert oracle.sides(n, c) == s, f"n={n} sides of cell {c}"


def test_ordinal_types_are_the_documented_ones(oracle):
    _, o3 = oracle.unit_simplices(3)
    _, o4 = oracle.unit_simplices(4)
    assert [i for i, v in enumerate(o3) if v] == [4, 8]
    assert [i for i, v in enumerate(o4) if v] == [16, 20, 30, 34, 46, 50]
