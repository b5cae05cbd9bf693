# ON THE GPU BOX: host-side phase times of jtx_mi_render's pass loop (JTX_TRACE_RENDER=1), averaged over the passes of one C2 frame
# usage: [JTX_PASSES_IN_FLIGHT=k] python3 tools/progressive_trace.py [samplesPerPass]
import os, sys, time, collections, re, subprocess
if os.environ.get("JTX_TRACE_RENDER") != "1":
    env = dict(os.environ, JTX_TRACE_RENDER="1")
    r = subprocess.run([sys.executable, __file__] + sys.argv[1:], env=env, capture_output=True, text=True)
    agg = collections.defaultdict(list)
    for l in r.stderr.splitlines():
        m = re.match(r"\[jtx_mi_render\] (.*?)\s+([0-9.]+) ms", l)
        if m:
            agg[m.group(1)].append(float(m.group(2)))
    print(r.stdout.strip())
    for k, v in agg.items():
        v2 = v[len(v) // 2:]                 # the second (warm) frame
        print(f"  {k:20s} n={len(v2):3d} mean {sum(v2) / len(v2):7.3f} ms  max {max(v2):7.3f}  total {sum(v2):8.2f} ms")
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jtx_pathtracer_amd as jtx
spp_pass = int(sys.argv[1]) if len(sys.argv) > 1 else 1
data = jtx.scenes.cornell(); sc = jtx.Scene(data); sc.buildBVH()
cam = jtx.StaticCamera(1920, 1080, data.camera, 8, 8, 8)
cam.samplesPerPass_ = spp_pass
for _ in range(2):
    t = time.perf_counter(); cam.render(sc, progress=lambda c, t: None); dt = time.perf_counter() - t
print(f"samplesPerPass {spp_pass}: {dt * 1e3:.2f} ms per frame, passes in flight {os.environ.get('JTX_PASSES_IN_FLIGHT', 'default')}")
