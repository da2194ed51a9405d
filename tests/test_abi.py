"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol the header declares,
and fails loudly (no CPU fallback) when no gfx950 device is present."""
import ctypes as C
import os
import re

import pytest

from so_dso_place_recognition_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "place_recognition.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(pr_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    declared = _header_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), name
    assert sorted(_lib.SYMBOLS) == declared
    assert b"gfx950" in lib.pr_version()


def test_no_silent_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present; the loud-failure path is for CPU-only hosts")
    lib = _lib.load()
    h = C.c_void_p()
    rc = lib.pr_create(0, C.byref(h))
    assert rc == _lib.PR_EHIP and not h
    assert b"no CPU fallback" in lib.pr_last_error(None)
    from so_dso_place_recognition_amd import api
    with pytest.raises(_lib.PRError):
        api.Context(0)


def test_product_does_not_link_the_oracle():
    so = os.path.join(ROOT, "so_dso_place_recognition_amd", "libpr_amd.so")
    data = open(so, "rb").read()
    assert b"pr_ref_" not in data and b"libpr_ref" not in data
    for dirpath, _, files in os.walk(os.path.join(ROOT, "so_dso_place_recognition_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle_lib" not in src and "libpr_ref" not in src and "np_checker" not in src, f


def test_inline_asm_mfma_operands_are_not_written_right_before_use():
    """hipcc pads nothing for inline asm: a compiler-generated VALU write (spill reload, copy) of an operand within two
    instructions of a hand-written MFMA is silent corruption (seen once during development).  tools/audit_asm_hazards.py
    scans the gfx950 assembly of the SC matchers that contain asm MFMAs for it."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = [os.path.join(root, "so_dso_place_recognition_amd", "csrc", f) for f in ("sc_match_h.hip", "sc_match.hip", "sc_match_e.hip")]
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "audit_asm_hazards.py")] + src, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    import re
    counts = [int(x) for x in re.findall(r"(\d+) inline-asm MFMAs checked, 0 finding", r.stdout)]
    # sc_match_h.hip: 2 instantiations x 31 frequencies x 4 (two operand pairs x Re | Im); sc_match_e.hip: 94 per single-product instantiation + 220
    assert len(counts) == 3 and counts[0] == 248 and counts[2] >= 470 + 220, r.stdout
    # the audit itself: a reload in front of an asm MFMA and a copy of its result right behind it are both reported
    sys.path.insert(0, os.path.join(root, "tools"))
    import audit_asm_hazards as aud
    bad_before = ["v_accvgpr_read_b32 v5, a7", ";;#ASMSTART", "v_mfma_f32_16x16x32_f16 v[0:3], v[4:7], v[8:11], 0", ";;#ASMEND"]
    bad_after = [";;#ASMSTART", "v_mfma_f32_16x16x32_f16 v[0:3], v[4:7], v[8:11], 0", ";;#ASMEND", "v_mov_b32_e32 v20, v2"]
    good = ["v_mov_b32_e32 v5, v30", "s_nop 1", ";;#ASMSTART", "v_mfma_f32_16x16x32_f16 v[0:3], v[4:7], v[8:11], 0", ";;#ASMEND",
            ";;#ASMSTART", "v_mfma_f32_16x16x32_f16 v[0:3], v[4:7], v[12:15], v[0:3]", ";;#ASMEND", "s_nop 9", "v_add_f32_e32 v20, v2, v3"]
    assert aud.audit_lines("x", bad_before) == 1 and aud.audit_lines("x", bad_after) == 1 and aud.audit_lines("x", good) == 0


def _run_py(code):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    return subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=600)


def test_loader_leaves_one_hip_runtime_whatever_the_import_order():
    """torch bundles its own libamdhip64.so.7; the loader maps that copy first when torch is installed, so that torch still sees its
    runtime after libpr_amd.so was loaded (CPU box: only that both load and that the mapped copy is torch's)."""
    r = _run_py("from so_dso_place_recognition_amd import _lib; _lib.load(); import torch\n"
                "maps = open('/proc/self/maps').read()\n"
                "hip = sorted({l.split()[-1] for l in maps.splitlines() if 'libamdhip64' in l})\n"
                "print(hip); assert len(hip) == 1 and '/torch/lib/' in hip[0], hip")
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.gpu
def test_torch_finds_its_gpu_after_the_library_was_used_first():
    r = _run_py("import numpy as np\n"
                "from so_dso_place_recognition_amd import api, synth\n"
                "db = synth.sc_database(45, 64); q, planted = synth.sc_queries(46, db, 8)\n"
                "idx, sc = api.match_topk('sc', q, db)\n"
                "import torch\n"
                "assert torch.cuda.is_available()\n"
                "x = torch.arange(8, device='cuda', dtype=torch.float64).sum().item(); assert x == 28.0\n"
                "from so_dso_place_recognition_amd.matcher import Matcher\n"
                "mt = Matcher('sc', 8, 64); mt.pack_database(torch.from_numpy(db).cuda())\n"
                "i2, s2 = mt.match(torch.from_numpy(q).cuda())\n"
                "assert np.array_equal(i2.cpu().numpy(), idx), (i2, idx)\n"
                "print('ok')")
    assert r.returncode == 0, r.stdout + r.stderr


def test_split_f16_operand_pairs_cover_the_three_products_once(tmp_path):
    """The SC matchers run the three split products of a frequency (q_hi d_hi + q_hi d_lo + q_lo d_hi, 20 rings each) as two 16x16x32 MFMAs
    whose 16-byte lane pieces come from fixed places of the packed query rows and DB tiles (csrc/kernels.hpp).  tests/sch_layout_check.cpp
    walks those places with the header's own functions and the pack's layout rules: 60 products, each once, nothing else."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "sch_layout_check")
    r = subprocess.run(["/opt/rocm/bin/hipcc", "-x", "hip", "--cuda-host-only", "-std=c++17", "-I", os.path.join(root, "so_dso_place_recognition_amd", "csrc"),
                        os.path.join(root, "tests", "sch_layout_check.cpp"), "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout
