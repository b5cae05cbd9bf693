"""bench.py's roofline arithmetic on recorded inputs (no GPU): the useful-work fraction is reference-algorithm lane-operations over
the fp32 lane peak, the issue model prices the instruction mix with the measured per-class rates, stale
counters are withheld -- recomputed here from profiles/r03_bench.json and profiles/r03_pmc.json the way a reader would."""
import json
import os

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_useful_ops_table():
    c = dict(n_camera=2, n_closest=3, n_any=5, n_nodes_closest=7, n_tri_closest=11, n_accept=13, n_nodes_any=17, n_tri_any=19, n_shade=23)
    want = 6 * (3 + 5) + 25 * (7 + 17) + 55 * (11 + 19) + 242 * 23 + 40 * 2
    assert bench.useful_lane_ops(c) == want
    assert bench.USEFUL_OPS == {"ray": 6, "node": 25, "tri": 55, "shade": 242, "camera": 40}


def test_recorded_bench_line_is_reproducible_from_profiles():
    line = json.loads(open(os.path.join(ROOT, "profiles", "r03_bench.json")).read().strip().splitlines()[-1])
    r = line["roofline"]
    t = r["kernel_ms"] * 1e-3
    cus = r["num_cus"]
    # useful_frac: lane-ops / time / (CUs x 128 lanes x 2.4 GHz)
    assert abs(r["useful"]["lane_ops_per_launch"] / t / (cus * 128 * 2.4e9) - r["useful_frac"]) < 2e-4
    pmc = json.load(open(os.path.join(ROOT, "profiles", "r03_pmc.json")))["workloads"][line["config"]["workload"]]["counters"]
    # frac: SQ_INSTS_VALU / time / (CUs x 4 SIMDs x 2.4 GHz / 2)
    assert abs(pmc["SQ_INSTS_VALU"] / t / (cus * 4 * 2.4e9 / 2) - r["frac"]) < 2e-3
    fast = pmc["SQ_INSTS_VALU_ADD_F32"] + pmc["SQ_INSTS_VALU_MUL_F32"] + pmc["SQ_INSTS_VALU_FMA_F32"]
    trans = pmc["SQ_INSTS_VALU_TRANS_F32"]
    need = fast * 2.4 + (pmc["SQ_INSTS_VALU"] - fast - trans) * 4.4 + trans * 8.4
    assert abs(need / (cus * 4 * t * 2.4e9) - r["issue_model"]["busy"]) < 2e-3
    assert r["issue_model"]["busy"] > 1.0 and 0.15 < r["useful_frac"] < 0.3
    assert abs((2 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024 - r["traffic"]) < 1e6
    assert line["value"] == round(line["config"]["rays_per_frame"] * line["steps"] / (line["ms_per_step"] * line["steps"] * 1e-3) / 1e6, 2) or \
        abs(line["value"] - line["config"]["rays_per_frame"] / (line["ms_per_step"] * 1e-3) / 1e6) / line["value"] < 1e-3


def test_stale_counters_are_withheld(monkeypatch):
    monkeypatch.setattr(bench, "source_hash", lambda: "not the hash in the file")
    c = dict(n_camera=10, n_closest=10, n_any=10, n_nodes_closest=100, n_tri_closest=20, n_accept=5, n_nodes_any=80, n_tri_any=10, n_shade=8)
    r = bench.roofline_block("cornell_1920x1080_64spp_d8", {"lds_resident": True, "lds_bytes": 1000, "workgroups": 1792}, c, "k_render_paths", 27.4, 1, 256)
    assert r["pmc_stale"] is True and r["frac"] is None and r["traffic"] is None and "issue_model" not in r and "lane_util" not in r
    assert r["useful_frac"] is not None                                        # needs no profile
