"""tools/issue_model.py (the issue-bound model of the SC matcher as code) on the kernel as it is in the tree: the
unit loop it finds in the gfx950 assembly must be the one the model was validated on (282 MFMAs of the split-f16 form, 94 of the
single-product form), and its prediction must stay where two boxes measured the kernel (profiles/r04_issue_model_box{A,B}.json) - a change
of the kernel that moves either shows up here, without a GPU (hipcc cross-compiles)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(kernel):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "issue_model.py"), "--kernel", kernel, "--json"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


def test_split_f16_unit_and_its_bound():
    d = _run("split")
    h = d["histogram"]
    assert h["mfma16"] == 186 and h["mfma32"] == 96
    assert 1500 < d["instructions_per_unit"] < 1900
    box = json.load(open(os.path.join(ROOT, "profiles", "r04_issue_model_boxA.json")))["f16x2"]
    measured = box["model"]["measured_cycles_per_unit"]
    assert 0.88 < d["model_cycles_per_unit"] / measured < 1.0          # the model is a bound the measured kernel sits within ~7 % of
    assert d["matrix_pipe_cycles"] > d["cycles_per_unit_for_0.60"]["at_2.4GHz"] * 0.85   # the executed MFMAs alone nearly fill what 0.60 would allow


def test_single_product_unit():
    d = _run("single")
    h = d["histogram"]
    assert h["mfma16"] == 62 and h["mfma32"] == 32
    box = json.load(open(os.path.join(ROOT, "profiles", "r04_issue_model_boxA.json")))["f16"]
    assert 0.88 < d["model_cycles_per_unit"] / box["model"]["measured_cycles_per_unit"] < 1.02
