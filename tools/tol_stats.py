#!/usr/bin/env python3
"""ON THE GPU BOX: image statistics of SURVEY 8d(ii) for the library JTX_MI_LIB names (a tolerance build, or the product: all zeros)
against the CPU oracle -- RMSE of acc/spp, % identical RGB8 bytes, % within 1 LSB, max byte difference -- on C1 (Cornell 512x512,
4x4 spp, depth 4: the config the tolerance is stated on) and on small frames of the atrium and the mixed scene (the 8-ary traversal,
all BxDFs).  SURVEY's gate at 16 spp: RMSE <= 4e-3, >= 97 % identical bytes, >= 99 % within 1 LSB."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import jtx_pathtracer_amd as jtx
import oracle_lib as ol

CASES = [("C1 cornell 512x512x16 d4", lambda: jtx.scenes.cornell(), 512, 512, 4, 4, 4),
         ("atrium 262k 480x270x16 d8", lambda: jtx.scenes.atrium(), 480, 270, 4, 4, 8),
         ("mixed 480x270x16 d8", lambda: jtx.scenes.mixed(), 480, 270, 4, 4, 8)]
print("library:", os.environ.get("JTX_MI_LIB", "product"))
for name, make, W, H, xs, ys, depth in CASES:
    data = make()
    sc = jtx.Scene(data); sc.buildBVH()
    cam = jtx.StaticCamera(W, H, data.camera, xs, ys, depth)
    cam.render(sc, count_rays=False)
    acc, img, _ = ol.OracleScene(data).render(data.camera_desc(W, H, xs, ys, depth), count=False)
    spp = xs * ys
    a = np.asarray(cam.acc_, np.float64).reshape(-1, 3) / spp; b = np.asarray(acc, np.float64).reshape(-1, 3) / spp
    l2 = np.sqrt(((a - b) ** 2).sum(1))
    rmse = float(np.sqrt((l2 ** 2).mean()))
    di = np.abs(np.asarray(cam.img_, np.int32).reshape(-1) - np.asarray(img, np.int32).reshape(-1))
    print(f"{name}: rmse {rmse:.3e}  identical bytes {100.0 * (di == 0).mean():.4f} %  within 1 LSB {100.0 * (di <= 1).mean():.4f} %  "
          f"max byte diff {int(di.max())}  pixels differing {int((l2 > 0).sum())} of {len(l2)}", flush=True)
    sc.destroy()
