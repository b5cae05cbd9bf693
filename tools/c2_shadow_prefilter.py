#!/usr/bin/env python3
"""local (CPU, uses the oracle as a ray caster): VERDICT r3 next 5b -- how many of the 28 phase-A leaf-box tests of a C2 shadow traversal are decided by
SIGNS alone (the box lies entirely behind the ray origin, or entirely beyond the light, along some axis), and for how many (wave, leaf) pairs that
holds for ALL 64 lanes of a wave, so that a wave-uniform prefilter could skip the leaf's 24 VALU instructions?
Waves: 8x8 pixel blocks of primary hits (the most coherent shadow rays the frame has: bounce 1, one stratum), and the same rays in random groups of
64 (what a wave holds once dynamic path assignment has mixed bounce depths)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle_lib as ol
import jtx_pathtracer_amd as jtx
from jtx_pathtracer_amd import api, scenes
data = scenes.cornell()
osc = ol.OracleScene(data)
nodes, _, _ = api.bvh_build_host(data)
leaves = [n for n in nodes if n["num_prims"] > 0]
lo = np.array([n["pmin"] for n in leaves], np.float64); hi = np.array([n["pmax"] for n in leaves], np.float64)
W, H = 1920, 1080
cam = data.camera_desc(W, H, 8, 8, 8)
rs = np.random.RandomState(1)
blocks = [(int(r), int(c)) for r, c in zip(rs.randint(0, H // 8, 160) * 8, rs.randint(0, W // 8, 160) * 8)]
rows = np.concatenate([np.repeat(np.arange(r, r + 8), 8) for r, c in blocks]).astype(np.int32)
cols = np.concatenate([np.tile(np.arange(c, c + 8), 8) for r, c in blocks]).astype(np.int32)
o, d = ol.camera_rays(cam, rows, cols, np.zeros(len(rows), np.int32))
hit = osc.closestHit(o, d)
ok = hit["hit"].astype(bool)
P = hit["point"].astype(np.float64) + 1e-4 * hit["normal"].astype(np.float64)
L = np.array(data.lights[0]["position"] if isinstance(data.lights[0], dict) else [278.0, 500.0, 279.5], np.float64)
def decided(P):
    """[ray, leaf]: the leaf box lies entirely behind the origin or entirely beyond the light along SOME axis (the ray runs from P to L)"""
    a = np.minimum(P, L)[:, None, :]; b = np.maximum(P, L)[:, None, :]
    return ((hi[None] < a) | (lo[None] > b)).any(axis=2)
dec = decided(P)
dec[~ok] = True                                                   # lanes without a shadow ray do not hold the wave back
nw = len(blocks)
per_ray = dec[ok].mean()
coherent = dec.reshape(nw, 64, -1).all(axis=1).mean()
perm = rs.permutation(len(P))
mixed = dec[perm].reshape(nw, 64, -1).all(axis=1).mean()
live = ok.reshape(nw, 64).any(axis=1)
print(f"{ok.sum()} shadow rays of {len(ok)} primary samples in {nw} pixel blocks; leaves {len(leaves)}")
print(f"per ray: {per_ray:.3f} of the leaf-box tests are decided by signs alone")
print(f"per wave, 8x8 pixel blocks at bounce 1 (coherent): {coherent:.3f} of the (wave, leaf) pairs are decided for ALL lanes -> skippable")
print(f"per wave, the same rays in random groups of 64 (mixed): {mixed:.3f}")
