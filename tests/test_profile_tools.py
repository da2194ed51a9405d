"""profiles/derive.py on the committed round-6 profile: the tool that turns the rocprofv3 summaries into the figures DESIGN.md and the bench line
quote must reproduce them from the committed text (no GPU needed)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_derived_counters_reproduce_from_the_committed_profile():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "profiles", "derive.py"), os.path.join(ROOT, "profiles", "r06_final.txt")],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-1000:]
    got = json.loads(r.stdout)["kernels"]
    ref = json.load(open(os.path.join(ROOT, "profiles", "r06_final_derived.json")))["kernels"]
    assert set(got) == set(ref)
    for k, e in ref.items():
        for name, v in e.items():
            if isinstance(v, float):
                assert abs(got[k][name] - v) <= 1e-9 * max(1.0, abs(v)), (k, name)
    # the structure-channel launch of the metric step: its own duration cluster, the counters DESIGN.md section 4.0 quotes
    s = next(v for k, v in got.items() if k.startswith("sc_match_e_kernel<true, 4, 4, false> @ 262144") and 15 < v["ms_per_launch"] < 25)
    assert 0.35 < s["matrix_pipe_busy"] < 0.50 and 5.5 < s["valu_per_mfma"] < 6.5 and s["lds_bank_conflict_cycles"] < 1e5
    assert 1.7 < s["sustained_ghz"] < 2.45 and 3.5e9 < s["hbm_bytes"] < 5e9
    b = next(v for k, v in got.items() if k.startswith("sc_match_e_kernel<false, 8, 8, true> @ 524288"))
    assert b["lds_bank_conflict_cycles"] < 1e5                    # round 6: the compact query image without the row shift (round 5: 7.9e8 per launch)
    m = next(v for k, v in got.items() if k.startswith("m2dp_match_h_kernel<4, true>"))
    assert 0.70 < m["matrix_pipe_busy"] < 0.82 and m["sustained_ghz"] < 1.8
    tr = json.load(open(os.path.join(ROOT, "profiles", "r06_traffic.json")))
    st = [v for k, v in tr.items() if isinstance(v, dict) and v.get("launch") == "structure channel"]
    assert len(st) == 1 and abs(st[0]["hbm_bytes_per_launch"] - s["hbm_bytes"]) < 1.0
