"""Copies the reference's own (poses_history_file.txt, incoming_id_file.txt) pairs - all 13 sequences it holds under
place_recognition/results/{KITTI,RobotCar}/ - into tests/golden/ref_sequences/<name>/ (poses gzip-compressed).  These are DATA
files of the reference (MIT-licensed), the known answer of pts_preprocess.h:187-215: which pose ids become clouds.
usage (in the build container, where /root/reference exists): python tests/make_ref_sequences.py"""
import glob
import gzip
import os
import shutil

REF = "/root/reference/place_recognition/results"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_sequences")
for d in sorted(glob.glob(REF + "/*/*")):
    ids = os.path.join(d, "incoming_id_file.txt")
    poses = os.path.join(d, "poses_history_file.txt")
    if not (os.path.exists(ids) and os.path.exists(poses)):
        continue
    name = os.path.basename(os.path.dirname(d)).lower() + "_" + os.path.basename(d)
    o = os.path.join(OUT, name)
    os.makedirs(o, exist_ok=True)
    with open(poses, "rb") as f, gzip.GzipFile(os.path.join(o, "poses_history_file.txt.gz"), "wb", mtime=0) as g:
        shutil.copyfileobj(f, g)
    shutil.copyfile(ids, os.path.join(o, "incoming_id_file.txt"))
    # ground truth of the evaluation drivers (test_kitti.m:24-25 gt.txt, test_robotcar.m:33-36 gps.txt) for the sequences
    # tests/test_eval.py runs them on
    if name in ("kitti_seq06", "kitti_seq07", "robotcar_2015-05-19-14-06-38", "robotcar_2015-05-22-11-14-30"):
        for gname in ("gt.txt", "gps.txt"):
            src = os.path.join(d, gname)
            if os.path.exists(src):
                with open(src, "rb") as f, gzip.GzipFile(os.path.join(o, gname + ".gz"), "wb", mtime=0) as g:
                    shutil.copyfileobj(f, g)
    print(name, os.path.getsize(os.path.join(o, "poses_history_file.txt.gz")))
