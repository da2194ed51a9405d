"""Shared test helpers: the config-1 synthetic points file: the reference's real pose files come
without their points (missing blobs), so points are synthesised around every pose and written in PosesPts format."""
import numpy as np

from so_dso_place_recognition_amd import api, synth


def read_poses(path):
    ids, W = [], []
    for line in open(path):
        t = line.split()
        if len(t) >= 13:
            ids.append(int(t[0]))
            W.append(np.array(t[1:13], np.float64).reshape(3, 4))
    return np.array(ids, np.int32), np.stack(W)


def write_synthetic_points(poses_path, pts_path, per_pose=120, seed=41, max_poses=None):
    ids, W = read_poses(poses_path)
    if max_poses:
        ids, W = ids[:max_poses], W[:max_poses]
    pid, pts, its = [], [], []
    for i, w in zip(ids, W):
        p, it = synth.scene_cloud(seed, int(i), per_pose)           # camera frame, ||p|| < 45
        R, t = w[:, :3], w[:, 3]
        world = (p - t) @ np.linalg.inv(R).T                         # camToWorld
        pid.append(np.full(per_pose, i, np.int32)); pts.append(world); its.append(it)
    api.write_points(pts_path, np.concatenate(pid), np.concatenate(pts), np.concatenate(its))
    return ids
