"""SC generation timing (HIP events around pr_sc_generate_dev): PR_SC_GEN / PR_SC_FUSED_WGS from the environment; also compares the
result with the default two-pass path of a second library handle loaded in another process (run once per setting)."""
import ctypes as C
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from so_dso_place_recognition_amd import synth
from so_dso_place_recognition_amd.api import Context

P = lambda t: C.c_void_p(t.data_ptr())
dev = torch.device("cuda", 0)
cur = torch.cuda.current_stream().cuda_stream
ctx = Context(0, stream=cur)
for N, PTS in ((64, 50_000), (1024, 50_000), (5000, 50_000), (4096, 2_000)):
    xyz, it, offs = synth.scene_clouds_torch(42, N, PTS, device=dev)
    sig = torch.empty((N, 2400), dtype=torch.float64, device=dev)
    run = lambda: ctx.check(ctx.lib.pr_sc_generate_dev(ctx.h, P(xyz), P(it), P(offs), N, 45.0, P(sig)))
    run(); run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ms = float(np.median(ts))
    by = N * (28 * PTS + 19200)
    print(f"{os.environ.get('PR_SC_GEN', 'two-pass'):9s} wgs={os.environ.get('PR_SC_FUSED_WGS', '-'):4s} N={N:5d} pts={PTS:6d}  {ms:8.3f} ms  {by / ms / 1e9:7.2f} TB/s algorithmic"
          f"  checksum {float(sig.sum()):.6f} {float(sig[:, 1200:].sum()):.1f}")
    del xyz, it, offs, sig
