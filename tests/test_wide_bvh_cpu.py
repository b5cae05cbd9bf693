"""Host-side checks of the 8-ary quantised node set (jtx_mi_wide_build; layout in jtx_scene_dev.hpp):
every quantised child box contains the exact one, and a walk over the wide nodes reaches exactly the leaves,
in exactly the order, that the reference's binary traversal (scene.cpp:10-55) reaches.  No GPU needed."""
from fractions import Fraction

import numpy as np
import pytest

import jtx_pathtracer_amd as jtx
from jtx_pathtracer_amd import api, scenes

f32 = np.float32


def plane(q, s, p):
    """origin + q * cell, exactly"""
    return Fraction(int(q)) * Fraction(float(s)) + Fraction(float(p))


def slab_q(nd, s, o, inv, tmin, tmax):
    """the wide-node test of traverseWide on child slot s, float32, with its outward slack mu"""
    t0, t1 = f32(tmin), f32(tmax)
    for k in range(3):
        a = f32(nd["cell"][k] * inv[k]); b = f32(f32(nd["origin"][k] - o[k]) * inv[k])
        mu = f32(abs(b) * f32(2.0 ** -21) + f32(abs(a) * f32(2.0 ** -14) + f32(2.0 ** -100)))   # >= the kernel's fma form
        qn, qf = (nd["hi"][k][s], nd["lo"][k][s]) if inv[k] < 0 else (nd["lo"][k][s], nd["hi"][k][s])
        t0 = max(t0, f32(f32(f32(qn) * a) + f32(b - mu))); t1 = min(t1, f32(f32(f32(qf) * a) + f32(b + mu)))
    return t0 <= t1


def decode(w, a):
    """wide node at granule a -> dict"""
    n0, n1, n2, n3, n4 = (w[a + i] for i in range(5))
    origin = n0[:3].view(np.float32)
    cell = [np.array([((int(n0[3]) >> (8 * k)) & 0xff) << 23], np.uint32).view(np.float32)[0] for k in range(3)]
    byts = lambda u0, u1: [(int(u0) >> (8 * i)) & 0xff for i in range(4)] + [(int(u1) >> (8 * i)) & 0xff for i in range(4)]
    lo = [byts(n2[0], n2[1]), byts(n2[2], n2[3]), byts(n3[0], n3[1])]
    hi = [byts(n3[2], n3[3]), byts(n4[0], n4[1]), byts(n4[2], n4[3])]
    order = [(int(n1[2 + (o >> 2)]) >> (8 * (o & 3))) & 0xff for o in range(8)]
    return dict(origin=origin, cell=cell, base=int(n1[0]), imask=int(n1[1]) & 0xff, lmask=(int(n1[1]) >> 8) & 0xff,
                lo=lo, hi=hi, order=order)


def child_addr(nd, slot):
    below = (1 << slot) - 1
    if nd["imask"] >> slot & 1:
        return nd["base"] + 5 * bin(nd["imask"] & below).count("1"), False
    return nd["base"] + 5 * bin(nd["imask"]).count("1") + 2 * bin(nd["lmask"] & below).count("1"), True


def slot_of(k, B):
    b2 = (k >> 2) ^ ((B >> 3) & 1)
    b1 = ((k >> 1) & 1) ^ ((B >> (1 + 4 * b2)) & 1)
    b0 = (k & 1) ^ ((B >> (4 * b2 + 2 * b1)) & 1)
    return 4 * b2 + 2 * b1 + b0


def slab(pmin, pmax, o, inv, tmin, tmax):
    """slabRegular: AABB::hit (aabb.hpp:66-81) for a regular ray, float32 throughout"""
    t0, t1 = f32(tmin), f32(tmax)
    for k in range(3):
        a = f32(f32(pmin[k] - o[k]) * inv[k]); b = f32(f32(pmax[k] - o[k]) * inv[k])
        t0 = max(t0, min(a, b)); t1 = min(t1, max(a, b))
    return t0 <= t1


def binary_leaves(nodes, o, inv, neg, tmin, tmax):
    out, stack = [], [0]
    while stack:
        i = stack.pop()
        n = nodes[i]
        if not slab(n["pmin"], n["pmax"], o, inv, tmin, tmax):
            continue
        if n["num_prims"]:
            out.append((int(n["offset"]), int(n["num_prims"])))
        else:
            first, second = i + 1, int(n["offset"])
            if neg[n["axis"]]:
                stack += [first, second]          # second child visited first (scene.cpp:40-46)
            else:
                stack += [second, first]
    return out


def wide_leaves(w, o, inv, neg, tmin, tmax):
    octant = neg[0] | neg[1] << 1 | neg[2] << 2
    out = []

    def visit(a):
        nd = decode(w, a)
        hits = []
        for s in range(8):
            if not ((nd["imask"] | nd["lmask"]) >> s & 1):
                continue
            if slab_q(nd, s, o, inv, tmin, tmax):
                hits.append(s)
        for k in range(8):
            s = slot_of(k, nd["order"][octant])
            if s not in hits:
                continue
            addr, is_leaf = child_addr(nd, s)
            if is_leaf:
                la, lb = w[addr].view(np.float32), w[addr + 1]
                pmin = [la[0], la[2], lb[:2].view(np.float32)[0]]; pmax = [la[1], la[3], lb[:2].view(np.float32)[1]]
                if slab(pmin, pmax, o, inv, tmin, tmax):
                    out.append((int(lb[2]), int(lb[3])))
            else:
                visit(addr)

    visit(0)
    return out


@pytest.fixture(scope="module")
def small_atrium():
    data = scenes.atrium(target_tris=3000)
    nodes, refs, depth = api.bvh_build_host(data)
    w, wdepth = api.wide_build_host(nodes)
    return nodes, w, wdepth, depth


def test_wide_nodes_contain_their_children(small_atrium):
    nodes, w, wdepth, depth = small_atrium
    assert 2 <= wdepth <= (depth + 3) // 3 + 1
    # walk binary and wide trees together
    seen_leaves, todo = 0, [(0, 0)]
    while todo:
        b, a = todo.pop()
        nd = decode(w, a)
        child = {}
        c1 = [b + 1, int(nodes[b]["offset"])]
        for i in range(2):
            if nodes[c1[i]]["num_prims"]:
                child[i << 2] = c1[i]; continue
            c2 = [c1[i] + 1, int(nodes[c1[i]]["offset"])]
            for j in range(2):
                if nodes[c2[j]]["num_prims"]:
                    child[i << 2 | j << 1] = c2[j]; continue
                child[i << 2 | j << 1] = c2[j] + 1
                child[i << 2 | j << 1 | 1] = int(nodes[c2[j]]["offset"])
        assert nd["imask"] | nd["lmask"] == sum(1 << s for s in child)
        assert (nd["origin"] == nodes[b]["pmin"]).all()
        for s, c in child.items():
            for k in range(3):
                lo = plane(nd["lo"][k][s], nd["cell"][k], nd["origin"][k]); hi = plane(nd["hi"][k][s], nd["cell"][k], nd["origin"][k])
                cmin, cmax, cell = Fraction(float(nodes[c]["pmin"][k])), Fraction(float(nodes[c]["pmax"][k])), Fraction(float(nd["cell"][k]))
                assert lo <= cmin and hi >= cmax, (b, s, k)                     # contains the exact box -- in exact arithmetic
                assert cmin - lo < cell and hi - cmax < cell, (b, s, k)         # and is the tightest such box on the grid
            addr, is_leaf = child_addr(nd, s)
            assert is_leaf == bool(nodes[c]["num_prims"])
            if is_leaf:
                seen_leaves += 1
                lb = w[addr + 1]
                assert int(lb[2]) == nodes[c]["offset"] and int(lb[3]) == nodes[c]["num_prims"]
                assert (w[addr].view(np.float32) == [nodes[c]["pmin"][0], nodes[c]["pmax"][0], nodes[c]["pmin"][1], nodes[c]["pmax"][1]]).all()
            else:
                todo.append((c, addr))
    assert seen_leaves == int((nodes["num_prims"] > 0).sum())


def test_wide_walk_reaches_the_reference_leaves_in_order(small_atrium):
    nodes, w, _, _ = small_atrium
    rs = np.random.RandomState(5)
    lo, hi = nodes[0]["pmin"], nodes[0]["pmax"]
    nonempty = 0
    for i in range(120):
        o = (lo + (hi - lo) * rs.uniform(0.05, 0.95, 3)).astype(np.float32)
        d = rs.normal(size=3).astype(np.float32)
        inv = (f32(1.0) / d).astype(np.float32)
        neg = [int(inv[k] < 0) for k in range(3)]
        tmax = f32(np.inf) if i % 3 == 0 else f32(rs.uniform(1.0, 60.0))
        a = binary_leaves(nodes, o, inv, neg, f32(0.001), tmax)
        b = wide_leaves(w, o, inv, neg, f32(0.001), tmax)
        assert a == b, f"ray {i}"
        nonempty += bool(a)
    assert nonempty > 60


def test_wide_build_small_and_degenerate_inputs():
    nodes, _, _ = api.bvh_build_host(scenes.cornell())
    w, depth = api.wide_build_host(nodes)
    n_leaves = int((nodes["num_prims"] > 0).sum())
    assert depth == 2 and (len(w) - 2 * n_leaves) % 5 == 0
    nodes, _, _ = api.bvh_build_host(scenes.quad_scene())  # a single leaf: nothing to collapse
    assert len(nodes) == 1
    w, depth = api.wide_build_host(nodes)
    assert depth == 0 and len(w) == 0
    w, depth = api.wide_build_host(nodes[:0])
    assert depth == 0 and len(w) == 0
