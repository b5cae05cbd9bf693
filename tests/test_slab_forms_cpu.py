"""The two forms of AABB::hit for a regular ray (aabb.hpp:66-81; jtx_scene_dev.hpp: slabRegular, slabRegularSel), in numpy float32, no GPU:
the leaf list picks an axis' near / far plane by the SIGN of 1/d through fma(a, ip, b * in) -- ip = 1/d where positive, in = 1/d where
negative, else 0 -- instead of min / max of the two products.  One term is an exact zero, so the values are the min / max bit for bit up
to the sign of a zero, and the verdict t0 <= t1 is the same -- as long as plane - o is finite, which LEAF_RANGE (2^60 for planes and ray
origins) guarantees.  Also the 8-ary node test's three differences against the clamped form it replaced."""
import numpy as np

f32 = np.float32
LEAF_RANGE = f32(2.0 ** 60)


def fma32(a, b, c):
    """fp32 fma where one addend is known to be a zero or the product is exact in float64 (24 x 24 bits): one rounding"""
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)


def forms(lo, hi, o, d, tmin, tmax):
    with np.errstate(all="ignore"):
        inv = (f32(1.0) / d).astype(f32)
        a, b = (lo - o).astype(f32), (hi - o).astype(f32)
        # slabRegular: min / max of the two products per axis
        ta, tb = (a * inv).astype(f32), (b * inv).astype(f32)
        n_ref, f_ref = np.minimum(ta, tb), np.maximum(ta, tb)
        # slabRegularSel
        ip = np.where(inv > 0, inv, f32(0.0)).astype(f32); in_ = np.where(inv > 0, f32(0.0), inv).astype(f32)
        zn, zf = (b * in_).astype(f32), (a * in_).astype(f32)
        n_sel, f_sel = fma32(a, ip, zn), fma32(b, ip, zf)
        t0r = np.maximum(n_ref.max(axis=1), tmin); t1r = np.minimum(f_ref.min(axis=1), tmax)
        t0s = np.maximum(n_sel.max(axis=1), tmin); t1s = np.minimum(f_sel.min(axis=1), tmax)
    return inv, (a, b), (n_ref, f_ref), (n_sel, f_sel, zn, zf), (t0r <= t1r), (t0s <= t1s)


def rays(n, seed):
    rs = np.random.RandomState(seed)
    lo = rs.uniform(-600, 600, (n, 3)).astype(f32)
    hi = (lo + rs.choice([0.0, 1e-5, 0.5, 30.0, 555.0], (n, 3)) * rs.uniform(0, 1, (n, 3))).astype(f32)      # flat, thin and fat boxes
    o = rs.uniform(-900, 900, (n, 3)).astype(f32)
    d = rs.normal(size=(n, 3)).astype(f32)
    k = n // 8
    d[:k] *= f32(1e-20); d[k:2 * k, 0] = f32(1e-37)                                # huge 1/d: products overflow to +-inf (still no NaN)
    o[2 * k:3 * k] = lo[2 * k:3 * k]                                               # origin ON a plane: a == +0
    o[3 * k:4 * k, 1] = hi[3 * k:4 * k, 1]
    sel = slice(4 * k, 5 * k)
    o[sel] = (o[sel] * f32(1e15)).astype(f32); lo[sel] = (lo[sel] * f32(1e14)).astype(f32); hi[sel] = (hi[sel] * f32(1e14)).astype(f32)   # near LEAF_RANGE
    hi[sel] = np.maximum(hi[sel], lo[sel])
    tmin = np.where(rs.rand(n) < 0.5, f32(0.001), f32(0.0)).astype(f32)
    tmax = np.where(rs.rand(n) < 0.5, f32(np.inf), rs.uniform(0.1, 2000, n).astype(f32)).astype(f32)
    return lo, hi, o, d, tmin, tmax


def test_sign_select_equals_min_max():
    lo, hi, o, d, tmin, tmax = rays(400000, 5)
    inv, (a, b), (n_ref, f_ref), (n_sel, f_sel, zn, zf), pass_ref, pass_sel = forms(lo, hi, o, d, tmin, tmax)
    regular = np.isfinite(inv).all(axis=1) & (inv != 0).all(axis=1) & (np.abs(o) <= LEAF_RANGE).all(axis=1) & \
              (np.abs(lo) <= LEAF_RANGE).all(axis=1) & (np.abs(hi) <= LEAF_RANGE).all(axis=1)
    assert regular.mean() > 0.9
    r = regular
    assert np.isfinite(a[r]).all() and np.isfinite(b[r]).all()                     # what the range buys
    with np.errstate(all="ignore"):
        ip = np.where(inv > 0, inv, f32(0.0)).astype(f32)
        van_n = np.where(inv > 0, zn, (a * ip).astype(f32)); van_f = np.where(inv > 0, zf, (b * ip).astype(f32))
    assert (van_n[r] == 0).all() and (van_f[r] == 0).all()                         # the vanishing terms ARE zeros (either sign), never NaN
    # values: equal as numbers everywhere (+0 == -0), bitwise wherever the value is not a zero
    for ref, sel in ((n_ref, n_sel), (f_ref, f_sel)):
        assert not np.isnan(ref[r]).any() and not np.isnan(sel[r]).any()
        assert (ref[r] == sel[r]).all()
        nz = ref[r] != 0
        assert (ref[r][nz].view(np.uint32) == sel[r][nz].view(np.uint32)).all()
    assert (pass_ref[r] == pass_sel[r]).all()
    assert 0.02 < pass_ref[r].mean() < 0.9                                         # both verdicts occur
    assert np.isinf(n_ref[r]).any()                                                # ... also with overflowed products


def test_beyond_the_range_the_forms_may_part():
    """why LEAF_RANGE exists: with plane - o overflowed to inf the vanishing term is inf * 0 = NaN"""
    lo = np.array([[3e38, 0, 0]], f32); hi = np.array([[3.2e38, 1, 1]], f32); o = np.array([[-3e38, 0.5, 0.5]], f32); d = np.array([[1, 1e-3, 1e-3]], f32)
    inv, (a, b), _, (n_sel, f_sel, zn, zf), pass_ref, pass_sel = forms(lo, hi, o, d, f32(0.001), f32(np.inf))
    assert np.isinf(a[0, 0]) and np.isnan(zf[0, 0])


def test_node_interval_as_three_differences():
    """wideNodePend (round 5): a child is missed iff far < near or t.max < near or far < t.min -- the sign of min(far - near, t.max - near,
    far - t.min) -- against the sign of min(far, t.max) - max(near, t.min) it replaced: equal whenever t.min <= t.max, and where they
    differ (an empty interval) the new form only passes MORE children."""
    rs = np.random.RandomState(9)
    n = 500000
    near = rs.uniform(-50, 50, n).astype(f32); far = (near + rs.uniform(-5, 40, n)).astype(f32)
    tmin = np.where(rs.rand(n) < 0.5, f32(0.001), f32(0.0)).astype(f32)
    tmax = np.where(rs.rand(n) < 0.3, f32(np.inf), rs.uniform(-1, 60, n).astype(f32)).astype(f32)
    k = n // 10
    far[:k] = near[:k]; tmax[k:2 * k] = near[k:2 * k]; far[2 * k:3 * k] = tmin[2 * k:3 * k]          # equalities: +0 differences pass
    with np.errstate(all="ignore"):
        old = (np.minimum(far, tmax) - np.maximum(near, tmin)).astype(f32)
        new = np.minimum(np.minimum((far - near).astype(f32), (tmax - near).astype(f32)), (far - tmin).astype(f32))
    miss_old, miss_new = np.signbit(old), np.signbit(new)
    ok = tmin <= tmax
    assert (miss_old[ok] == miss_new[ok]).all()
    assert (~miss_new[~ok] | miss_old[~ok]).all() and miss_old[~ok].all()          # empty interval: the old form misses everything, the new one may pass
