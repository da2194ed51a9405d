"""Generates tests/golden/synthetic_v1.npz: small seeded inputs (reproducible from synth.py) with expected
outputs from the independent numpy/LAPACK checker (tests/np_checker.py).  Run: python tests/make_golden.py
The reference cannot run in this image (Eigen3/roscpp/MATLAB absent), so these are checker outputs, not
reference outputs ("parity unpinned")."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import np_checker  # noqa: E402
from so_dso_place_recognition_amd import synth  # noqa: E402


def main():
    out = {}
    P = 2000
    sc, m2 = [], []
    for c in range(3):
        xyz, it = synth.scene_cloud(42, c, P)
        sc.append(np_checker.sc_signature(xyz, it))
        m2.append(np_checker.m2dp_signature(xyz, it))
    out["cloud_seed"] = np.array([42]); out["cloud_P"] = np.array([P])
    out["sc_sig"] = np.stack(sc); out["m2dp_sig"] = np.concatenate(m2)
    db = synth.sc_database(45, 16); q, et = synth.sc_queries(46, db, 8)
    dp, di = np_checker.sc_distance(q, db)
    a, v, _ = np_checker.fuse_top1(dp, di, 0)
    out["sc_dp"], out["sc_di"], out["sc_top1"], out["sc_score"], out["sc_planted"] = dp, di, a, v, et
    mdb = synth.m2dp_database(43, 16); mq, met = synth.m2dp_queries(44, mdb, 8)
    mp, mi = np_checker.m2dp_distance(mq, mdb)
    a, v, _ = np_checker.fuse_top1(mp, mi, 0)
    out["m2dp_dp"], out["m2dp_di"], out["m2dp_top1"], out["m2dp_score"], out["m2dp_planted"] = mp, mi, a, v, met
    np.savez_compressed(os.path.join(HERE, "golden", "synthetic_v1.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
