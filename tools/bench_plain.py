"""Wall time of the GIST / BoW matchers (host-pointer API: includes the H2D copy of the signatures and the D2H copy of the distances).
usage: python tools/bench_plain.py"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from so_dso_place_recognition_amd import api, synth

ctx = api.Context(0)
for name, fn, a, b in (("gist 1024 x 20000 x 512", api.processGIST, synth.gist_signatures(1, 1024, 512), synth.gist_signatures(2, 20000, 512)),
                       ("bow 256 x 5000 x 4000 (150-400 words)", api.processBoW, synth.bow_signatures(3, 256, 4000, 20000, (150, 400)),
                        synth.bow_signatures(4, 5000, 4000, 20000, (150, 400)))):
    fn(a[:8], b[:8], ctx)
    t0 = time.perf_counter(); d = fn(a, b, ctx); t1 = time.perf_counter()
    print(f"{name}: {1e3 * (t1 - t0):.1f} ms wall, {d.shape}, {d.size / (t1 - t0) / 1e6:.1f} M pairs/s")
