"""The torch-device samplers of synth.py draw the same bits as the numpy samplers (they exist so that full-size inputs
can be drawn in HBM on the GPU box; here they run on the CPU device)."""
import numpy as np
import torch

from so_dso_place_recognition_amd import synth


def test_uniform_and_sc_database_identical():
    s = np.array([0, 5, 123456789, 2 ** 40 + 3], dtype=np.uint64)
    a = synth.uniform(45, s, 77, offset=9)
    b = synth.uniform_torch(45, torch.from_numpy(s.astype(np.int64)), 77, offset=9).numpy()
    assert np.array_equal(a, b)
    assert np.array_equal(synth.sc_database(45, 300, first=99_990), synth.sc_database_torch(45, 300, first=99_990, device="cpu", chunk=128).numpy())


def test_m2dp_database_matches_to_the_last_ulp():
    a = synth.m2dp_database(43, 70, first=11)
    b = synth.m2dp_database_torch(43, 70, first=11, device="cpu").numpy()
    assert np.abs(a - b).max() < 4e-16


def test_scene_clouds_identical():
    xa, ia, oa = synth.scene_clouds(42, 5, 3000, first=7)
    xb, ib, ob = synth.scene_clouds_torch(42, 5, 3000, first=7, device="cpu", chunk=2)
    assert np.array_equal(oa, ob.numpy()) and np.array_equal(ia, ib.numpy())
    assert np.array_equal(xa, xb.numpy())
