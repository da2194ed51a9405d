"""tools/issue_model.py (the issue-bound model of the SC matcher as code) on the kernel as it is in the tree: the
unit loop it finds in the gfx950 assembly must be the one the model was run on (220 MFMAs of the split-f16 form since round 5's two
operand pairs per frequency - 282 before -, 94 of the single-product form), and its prediction must stay where the boxes measured the kernel
(profiles/r05_issue_model_box.json, profiles/r04_issue_model_box{A,B}.json) - a change of the kernel that moves either shows up here, without
a GPU (hipcc cross-compiles)."""
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
    assert h["mfma16"] == 124 and h["mfma32"] == 96
    assert 1500 < d["instructions_per_unit"] < 1900
    box = json.load(open(os.path.join(ROOT, "profiles", "r05_issue_model_box.json")))["f16x2"]
    measured = box["model"]["measured_cycles_per_unit"]
    # the model is a lower bound; with 186 stage-1 MFMAs the measured kernel sat within 7 % of it, with 124 it sits 13 % above: a third of the
    # stage-1 MFMAs gone took 7.8 % off the issue chain and 1.2 % off the measured cycles (the launch is 4.2 % shorter because the clock rose
    # 1.97 -> 2.03 GHz) - what the unit waits for now is not in the wave's own instruction stream
    assert 0.84 < d["model_cycles_per_unit"] / measured < 1.0
    # the executed MFMAs occupy the matrix pipe for 72 % of what 0.60 of the peak would allow a unit (88 % with 186 stage-1 MFMAs)
    assert 0.65 < d["matrix_pipe_cycles"] / d["cycles_per_unit_for_0.60"]["at_2.4GHz"] < 0.80


def test_single_product_unit():
    d = _run("single")
    h = d["histogram"]
    assert h["mfma16"] == 62 and h["mfma32"] == 32
    box = json.load(open(os.path.join(ROOT, "profiles", "r04_issue_model_boxA.json")))["f16"]
    assert 0.88 < d["model_cycles_per_unit"] / box["model"]["measured_cycles_per_unit"] < 1.02
