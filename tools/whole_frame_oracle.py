#!/usr/bin/env python3
"""ON THE GPU BOX: a timed workload against a WHOLE-FRAME run of the CPU oracle -- every pixel, every stratum (the GPU suite does this for
C2 and C5 and for 16 of C3's 64 strata; this is the full C3 frame, ~80 s of oracle on the host's cores).
usage: python3 tools/whole_frame_oracle.py [workload]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bench
import jtx_pathtracer_amd as jtx
import oracle_lib as ol
wl = sys.argv[1] if len(sys.argv) > 1 else "atrium_1920x1080_64spp_d8"
name, data, (W, H, xs, ys, depth) = bench.load_workload(jtx, wl)
sc = jtx.Scene(data); sc.buildBVH()
cam = data.camera_desc(W, H, xs, ys, depth)
t = time.perf_counter(); acc, img, cnt = ol.OracleScene(data).render(cam, threads=min(os.cpu_count() or 1, 64)); t_o = time.perf_counter() - t
g = jtx.StaticCamera(W, H, data.camera, xs, ys, depth)
t = time.perf_counter(); g.render(sc, count_rays=False); t_g = time.perf_counter() - t
diff = (np.asarray(g.acc_).view(np.uint32) != acc.view(np.uint32)).any(-1)
same_img = np.array_equal(g.img_, img)
g.render(sc, count_rays=True)
print(f"{name}: {W}x{H}x{xs * ys} spp, {cnt['n_closest'] + cnt['n_any']} rays; oracle {t_o:.1f} s on {min(os.cpu_count() or 1, 64)} threads, GPU {t_g * 1e3:.1f} ms (uncounted, film to host)")
print(f"pixels whose film words differ: {int(diff.sum())} of {W * H}; RGB8 image identical: {same_img}; all ray counters (incl. per-class tallies) identical: {g.counters == cnt}")
sys.exit(0 if (not diff.any() and same_img and g.counters == cnt) else 1)
