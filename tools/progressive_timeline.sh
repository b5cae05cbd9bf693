#!/bin/bash
# ON THE GPU BOX: kernel timeline of a progressive C2 frame (one stratum per pass): start / end of every path kernel and resolve
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
k=${1:-2}
rm -rf gpurun_out/prog_tl_$k
JTX_PASSES_IN_FLIGHT=$k timeout -k 10 200 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/prog_tl_$k -- python3 tools/progressive_trace_run.py > gpurun_out/prog_tl_$k.log 2>&1
python3 - <<PY
import csv, glob
rows = []
for f in glob.glob("gpurun_out/prog_tl_$k/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40], r.get("Queue_Id", "")))
for f in glob.glob("gpurun_out/prog_tl_$k/*/*memory_copy_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") , ""))
rows.sort()
# the last frame: take the last 64 k_render_paths launches
idx = [i for i, r in enumerate(rows) if "k_render_paths" in r[2]]
first = idx[-64]
t0 = rows[first][0]
print("passes in flight $k: timeline of the last frame (us from its first path kernel)")
for r in rows[first:first + 60]:
    print(f"{(r[0] - t0) / 1e3:9.1f} {(r[1] - t0) / 1e3:9.1f}  {(r[1] - r[0]) / 1e3:8.1f}  q{r[3]:4s} {r[2]}")
PY
rm -rf gpurun_out/prog_tl_$k
