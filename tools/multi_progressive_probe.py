# ON THE GPU BOX: a C2 frame through jtx_mi_multi_render with a callback per pass, N shards on this one card: the progressive launch per shard
# (round 6) against the pass-by-pass loop (JTX_PROGRESSIVE_LAUNCH=0).  One card: the shards' kernels share it -- the times say what the
# host-side loop costs, not what N devices would take.
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jtx_pathtracer_amd as jtx
jtx._capi.check(jtx._capi.load().jtx_mi_set_device(0))
data = jtx.scenes.cornell()
for shards in [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["1", "2", "8"])]:
    ms = jtx.MultiScene(data, [0] * shards)
    for spp_pass in (1, 8):
        cam = jtx.StaticCamera(1920, 1080, data.camera, 8, 8, 8); cam._pin()
        n = [0]
        ms.render(cam, progress=lambda c, t: n.__setitem__(0, n[0] + 1) and False, samples_per_tick=spp_pass)
        best = 1e9
        for _ in range(3):
            n[0] = 0; t = time.perf_counter(); ms.render(cam, progress=lambda c, t: n.__setitem__(0, n[0] + 1) and False, samples_per_tick=spp_pass); best = min(best, time.perf_counter() - t)
        print(f"{shards} shard(s), samplesPerPass {spp_pass}: {best * 1e3:8.2f} ms per 64-spp frame, {n[0]} callbacks (progressive launch {os.environ.get('JTX_PROGRESSIVE_LAUNCH', '1')})", flush=True)
    ms.destroy()
