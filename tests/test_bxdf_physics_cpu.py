"""Physics of the BxDF restatement, checked against closed forms computed HERE in float64 numpy -- not against another
transcription of the reference's code (VERDICT r1 "parity" 1: oracle and kernels are near-twin restatements, a shared
misreading would pass every GPU-vs-oracle test).  The GPU kernels equal the oracle bit for bit (test_bxdf_all_materials),
so what holds for the oracle holds for them.  Conductor / dielectric Fresnel terms, Snell's law, total internal reflection,
GGX sample / eval / pdf consistency, reciprocity, pdf normalisation, energy bounds.  No GPU."""
import numpy as np
import pytest

import jtx_pathtracer_amd as jtx
import oracle_lib as ol

N = 20000
# materials of scenes.mixed(): 3 metal-rough (metallic 1, roughness .3), 5 (metallic 0, roughness .8), 6 smooth glass, 7 rough glass,
# 8 rough gold, 9 smooth gold
MR_METAL, MR_PLASTIC, GLASS, ROUGH_GLASS, ROUGH_GOLD, GOLD = 3, 5, 6, 7, 8, 9


@pytest.fixture(scope="module")
def scene():
    data = jtx.scenes.mixed()
    return data, ol.OracleScene(data)


def up(n=N):
    return np.tile(np.array([[0, 0, 1]], np.float32), (n, 1))


def at(theta, n=N, below=False):
    d = np.tile(np.array([[np.sin(theta), 0, np.cos(theta)]], np.float32), (n, 1))
    if below:
        d[:, 2] *= -1
    return d


def fresnel_conductor(cosi, eta, k):
    e = eta + 1j * k
    cost = np.sqrt(1 - (1 - cosi ** 2) / e ** 2 + 0j)
    rp = (e * cosi - cost) / (e * cosi + cost); rs = (cosi - e * cost) / (cosi + e * cost)
    return (abs(rp) ** 2 + abs(rs) ** 2) / 2


def fresnel_dielectric(cosi, eta):
    sin2t = (1 - cosi ** 2) / eta ** 2
    if sin2t >= 1:
        return 1.0
    cost = np.sqrt(1 - sin2t)
    rpar = (eta * cosi - cost) / (eta * cosi + cost); rper = (cosi - eta * cost) / (cosi + eta * cost)
    return (rpar ** 2 + rper ** 2) / 2


@pytest.mark.parametrize("deg", [0.0, 20.0, 60.0, 85.0])
def test_smooth_conductor_is_a_mirror_weighted_by_the_complex_fresnel_term(scene, deg):
    data, o = scene
    th = np.deg2rad(deg)
    rs = np.random.RandomState(1)
    sm = o.sampleBxdf(GOLD, up(), at(th), rs.rand(N).astype(np.float32), rs.rand(N, 2).astype(np.float32))
    assert (sm["ok"] > 0).all()
    assert np.allclose(sm["wi"], at(th) * np.array([-1, -1, 1], np.float32), atol=1e-6)
    w = sm["f"] * np.abs(sm["wi"][:, 2:3]) / sm["pdf"][:, None]
    m = data.materials[GOLD]
    want = [fresnel_conductor(np.cos(th), m["ior"][c], m["k"][c]) for c in range(3)]
    assert np.allclose(w, want, rtol=2e-5)


@pytest.mark.parametrize("deg,below", [(0.0, False), (35.0, False), (60.0, False), (80.0, False), (30.0, True), (41.0, True), (60.0, True)])
def test_smooth_dielectric_obeys_fresnel_snell_and_total_internal_reflection(scene, deg, below):
    data, o = scene
    th = np.deg2rad(deg)
    eta = 1.5 if not below else 1 / 1.5                        # relative index seen from the side of wo
    rs = np.random.RandomState(2)
    sm = o.sampleBxdf(GLASS, up(), at(th, below=below), rs.rand(N).astype(np.float32), rs.rand(N, 2).astype(np.float32))
    assert (sm["ok"] > 0).all()
    side = -1.0 if below else 1.0
    refl = sm["wi"][:, 2] * side > 0
    R = fresnel_dielectric(np.cos(th), eta)
    assert abs(refl.mean() - R) < 4 * np.sqrt(R * (1 - R) / N) + 1e-9         # reflection is chosen with probability R
    assert np.allclose(sm["pdf"][refl], R, rtol=2e-5) and np.allclose(sm["pdf"][~refl], 1 - R, rtol=2e-5)
    assert np.allclose(sm["wi"][refl], at(th, 1, below=below) * np.array([-1, -1, 1], np.float32), atol=1e-6)
    if R < 1:
        t = sm["wi"][~refl]
        assert np.allclose(np.hypot(t[:, 0], t[:, 1]), np.sin(th) / eta, atol=2e-6)      # Snell
        assert (t[:, 0] < 1e-7).all() and (t[:, 2] * side < 0).all()                     # across the surface, away from wo
    else:
        assert refl.all()                                                                # beyond the critical angle
    w = sm["f"] * np.abs(sm["wi"][:, 2:3]) / sm["pdf"][:, None]
    assert np.allclose(w, 1.0, rtol=1e-5)                      # lossless interface: every sample carries weight 1


@pytest.mark.parametrize("mat", [ROUGH_GOLD, ROUGH_GLASS, MR_METAL, MR_PLASTIC])
def test_rough_lobes_sample_eval_and_pdf_agree(scene, mat):
    """the f and pdf a sample returns are what evalBxdf / pdfBxdf give for that pair of directions -- with the two
    exceptions the reference's own code makes (reproduced, and pinned here):
      * rough glass: evaluate / pdf take absCosTheta of both directions (dielectric.hpp:14-16, 122-127), so `reflect` is
        always true and a TRANSMITTED pair is evaluated with the reflection half-vector: only reflected pairs agree;
      * metal-rough with a diffuse share: sample() returns the pdf of the lobe it chose, pdf() the mixture
        (gltf.hpp:54-69 against 87-110): f agrees, pdf does not."""
    data, o = scene
    rs = np.random.RandomState(3)
    wo = rs.normal(size=(N, 3)).astype(np.float32); wo[:, 2] = np.abs(wo[:, 2]) + 0.05
    wo /= np.linalg.norm(wo, axis=1, keepdims=True).astype(np.float32)
    sm = o.sampleBxdf(mat, up(), wo, rs.rand(N).astype(np.float32), rs.rand(N, 2).astype(np.float32))
    ok = sm["ok"] > 0
    assert ok.mean() > 0.7
    assert (sm["pdf"][ok] > 0).all() and np.isfinite(sm["f"][ok]).all()
    assert np.allclose(np.linalg.norm(sm["wi"][ok], axis=1), 1.0, atol=1e-4)
    f = o.evalBxdf(mat, up()[ok], wo[ok], sm["wi"][ok]); pdf = o.pdfBxdf(mat, up()[ok], wo[ok], sm["wi"][ok])
    same_side = sm["wi"][ok][:, 2] > 0
    if mat == ROUGH_GLASS:
        assert 0.02 < same_side.mean() < 0.5 and (~same_side).sum() > 1000
        assert np.allclose(f[same_side], sm["f"][ok][same_side], rtol=1e-4, atol=1e-6)
        assert np.allclose(pdf[same_side], sm["pdf"][ok][same_side], rtol=1e-4, atol=1e-6)
        assert not np.allclose(pdf[~same_side], sm["pdf"][ok][~same_side], rtol=1e-2)       # the reference's abs() bug, reproduced
        return
    assert same_side.all()
    assert np.allclose(f, sm["f"][ok], rtol=1e-4, atol=1e-6)
    if mat == MR_PLASTIC:
        assert not np.allclose(pdf, sm["pdf"][ok], rtol=1e-2)                               # lobe pdf against mixture pdf
        cosine = np.abs(sm["wi"][ok][:, 2]) / np.pi
        diffuse_pick = np.isclose(sm["pdf"][ok], cosine, rtol=1e-5)
        assert 0.5 < diffuse_pick.mean() < 0.999                                            # most samples took the cosine lobe and say so
    else:
        assert np.allclose(pdf, sm["pdf"][ok], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("mat", [ROUGH_GOLD, MR_METAL])
def test_rough_reflection_is_reciprocal(scene, mat):
    data, o = scene
    rs = np.random.RandomState(4)
    a = rs.normal(size=(N, 3)).astype(np.float32); a[:, 2] = np.abs(a[:, 2]) + 0.1
    b = rs.normal(size=(N, 3)).astype(np.float32); b[:, 2] = np.abs(b[:, 2]) + 0.1
    a /= np.linalg.norm(a, axis=1, keepdims=True).astype(np.float32); b /= np.linalg.norm(b, axis=1, keepdims=True).astype(np.float32)
    fab = o.evalBxdf(mat, up(), a, b); fba = o.evalBxdf(mat, up(), b, a)
    assert (fab > 0).any()
    assert np.allclose(fab, fba, rtol=2e-4, atol=1e-7)


@pytest.mark.parametrize("mat,deg", [(MR_METAL, 20.0), (MR_METAL, 70.0), (MR_PLASTIC, 45.0), (ROUGH_GOLD, 30.0)])
def test_pdf_integrates_to_at_most_one_and_energy_is_bounded(scene, mat, deg):
    data, o = scene
    th = np.deg2rad(deg)
    rs = np.random.RandomState(5)
    n = 400000
    wi = rs.normal(size=(n, 3)).astype(np.float32)
    wi /= np.linalg.norm(wi, axis=1, keepdims=True).astype(np.float32)
    pdf = o.pdfBxdf(mat, up(n), at(th, n), wi).astype(np.float64)
    total = pdf.mean() * 4 * np.pi                              # uniform directions over the sphere
    err = 4 * pdf.std() / np.sqrt(n) * 4 * np.pi
    assert 0.80 < total < 1.0 + err + 0.01, total               # sampling visible normals: all but the mass that lands below the horizon
    sm = o.sampleBxdf(mat, up(N), at(th), rs.rand(N).astype(np.float32), rs.rand(N, 2).astype(np.float32))
    ok = sm["ok"] > 0
    w = np.where(ok[:, None], sm["f"] * np.abs(sm["wi"][:, 2:3]) / np.maximum(sm["pdf"][:, None], 1e-30), 0.0)
    assert (w.mean(axis=0) < 1.02).all() and w.mean() > 0.2     # a passive surface: albedo below one, and not black
