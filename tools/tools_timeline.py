# diagnostic: occupancy over time of the C2 launch from per-wave wall-clock stamps (build with -DJTX_PROFILE_TIMELINE)
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import jtx_pathtracer_amd as jtx
lib = jtx._capi.load()
data = jtx.scenes.cornell(); sc = jtx.Scene(data); sc.buildBVH()
cam = jtx.StaticCamera(1920, 1080, data.camera, 8, 8, 8)
cam.render(sc, count_rays=False, integrator=1)
cam.render(sc, count_rays=False, integrator=1)
n = int(os.environ.get('TL_WAVES', '7168'))    # persistent grid of k_render_paths; 32640 with JTX_DYNAMIC_PATHS=0 (one wave per pixel block)
buf = (C.c_uint64 * (2 * n))()
f = lib.jtx_mi_debug_timeline; f.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]
assert f(sc.handle, buf, n) == 0
t = np.array(buf, dtype=np.int64).reshape(n, 2).astype(np.float64)
t0 = t[:, 0].min(); t -= t0
tick = 1e-5                                     # wall_clock64: 100 MHz
dur = (t[:, 1] - t[:, 0]) * tick
total = t[:, 1].max() * tick
print(f"launch {total:.2f} ms; wave duration mean {dur.mean():.2f} ms, p10 {np.percentile(dur, 10):.2f}, p50 {np.percentile(dur, 50):.2f}, p90 {np.percentile(dur, 90):.2f}, max {dur.max():.2f}")
edges = np.linspace(0, t[:, 1].max(), 41)
occ = [((t[:, 0] <= (a + b) / 2) & (t[:, 1] > (a + b) / 2)).sum() for a, b in zip(edges[:-1], edges[1:])]
print("resident waves over time (40 bins):", " ".join(str(int(x)) for x in occ))
work = dur.sum()
print(f"wave-ms of work {work:.0f}; at 7168 resident waves that is {work / 7168:.2f} ms -> schedule efficiency {work / 7168 / total:.3f}")
# by tile row: where the long waves are
if n == 32640:
    rows = (np.arange(n) // 16) // 60
    print("mean wave duration by tile row (34 rows, bottom to top):", " ".join(f"{dur[rows == r].mean():.1f}" for r in range(34)))
late = t[:, 1] > 0.85 * t[:, 1].max()
print(f"waves ending in the last 15 % of the launch: {late.sum()}; their start (ms) p10/p50/p90 = "
      + "/".join(f"{np.percentile(t[late, 0] * tick, q):.1f}" for q in (10, 50, 90))
      + "; duration p10/p50/p90 = " + "/".join(f"{np.percentile(dur[late], q):.1f}" for q in (10, 50, 90)))
st = t[:, 0] * tick
for a, b in ((0, 5), (5, 15), (15, 25), (25, 35), (35, 50)):
    m = (st >= a) & (st < b)
    if m.sum():
        print(f"  waves started in [{a},{b}) ms: {m.sum():6d}, duration mean {dur[m].mean():5.2f} p90 {np.percentile(dur[m], 90):5.2f} max {dur[m].max():5.2f}")
