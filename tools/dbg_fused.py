#!/usr/bin/env python3
"""GPU box: the fused-commit path kernel against the counting kernel on small frames; prints where they differ."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import jtx_pathtracer_amd as jtx
jtx._capi.check(jtx._capi.load().jtx_mi_set_device(0))
data = jtx.scenes.cornell()
sc = jtx.Scene(data); sc.buildBVH()
for (W, H, xs, ys, d) in [(96, 64, 2, 2, 4), (96, 64, 4, 2, 4), (200, 120, 4, 4, 4), (512, 512, 4, 4, 4)]:
    a = jtx.StaticCamera(W, H, data.camera, xs, ys, d); a.render(sc, count_rays=True)
    b = jtx.StaticCamera(W, H, data.camera, xs, ys, d); b.render(sc, count_rays=False)
    same = (a.acc_.view(np.uint32) == b.acc_.view(np.uint32)).all(axis=2)
    print(W, H, xs * ys, "differing pixels:", int((~same).sum()), "of", W * H, "currentSample", b.currentSample_, "img same:", bool((a.img_ == b.img_).all()))
    if not same.all():
        bad = np.argwhere(~same)
        print("  rows", bad[:, 0].min(), bad[:, 0].max(), "cols", bad[:, 1].min(), bad[:, 1].max())
        blocks = sorted(set((int(r) // 8, int(c) // 8) for r, c in bad))
        print("  blocks with differences:", len(blocks), blocks[:20])
        for r, c in bad[:6]:
            print("   ", r, c, a.acc_[r, c], b.acc_[r, c])
        # is b a prefix (fewer strata)?
        for n in range(1, xs * ys):
            p = jtx.StaticCamera(W, H, data.camera, xs, ys, d); p.render(sc, count_rays=True, sample_begin=0, sample_end=n)
            m = (p.acc_.view(np.uint32) == b.acc_.view(np.uint32)).all(axis=2)
            print("   pixels equal to the", n, "-strata film:", int((m & ~same).sum()))
