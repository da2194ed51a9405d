#!/bin/bash
# usage: tools/online_trace.sh <m>   -> the last call's kernels in launch order with start offsets and durations (us)
m=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ot
rocprofv3 --kernel-trace -d /tmp/ot -o t -- python $root/tools/online_trace.py $m ${2:-f16x2} > /tmp/ot.log 2>&1
tail -1 /tmp/ot.log
python - <<PY
import sqlite3, glob
db = glob.glob("/tmp/ot/**/*_results.db", recursive=True)[0]
c = sqlite3.connect(db)
rows = list(c.execute("select name, start, end, grid_x, workgroup_x from kernels order by start"))
# the last call = the kernels after the last-but-one 'rerank' ... simply the last N where N = kernels per call (count between two sc_pack of queries)
names = [r[0] for r in rows]
idxs = [i for i, nme in enumerate(names) if "sc_pack_h" in nme and rows[i][3] < 200000]
a = idxs[-1]
t0 = rows[a][1]
for r in rows[a:]:
    nm = r[0].replace("pr::(anonymous namespace)::", "").replace("void ", "")[:70]
    print("%8.1f us  +%7.1f  %-70s grid=%d x %d" % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, nm, r[3], r[4]))
print("span of the call's kernels: %.1f us" % ((rows[-1][2] - t0) / 1e3))
PY
