#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_round6.sh <tag>     e.g. r06_final
# Like tools/profile_round.sh, but EVERY pass runs the `extra` workloads too (round 6: counter-backed fractions for M2DP matching and
# generation, SC generation, the pack and selection kernels - VERDICT r05 item 2):
#   gpurun_out/<tag>.txt            the un-profiled bench line, the rocprofv3 --kernel-trace --stats summary, one --pmc pass per counter group
#                                    (never combined with another trace domain), per (kernel, grid size)
#   gpurun_out/<tag>_traffic.json   HBM bytes per launch of every library kernel (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE, MI355X_MICROARCH.md)
# Copy what should be judged into profiles/.
set -u
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
f=$out/$tag.txt
keep="^==|^--|pr::"
{
  echo "# $tag: python bench.py --steps 5 --warmup 2 (un-profiled); kernel trace AND PMC passes: --steps 3 --warmup 1 --no-cpu-baseline --no-kitti-shape (the same launches in every pass) - all WITH the extra workloads (the PMC passes without the drive sampler's ~70 000 torch launches)"
  echo "# MI355X, rocprofv3 --kernel-trace --stats, then one --pmc group per run"
  echo
  echo "## bench.py JSON line (un-profiled run, with the CPU baseline)"
  python $root/bench.py --steps 5 --warmup 2 2> $out/${tag}_bench.err | tail -n 1
  echo
} > $f
tail -n 1 $f > /dev/null
rm -rf $out/${tag}_trace
rocprofv3 --kernel-trace --stats -d $out/${tag}_trace -o sc -- python $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kitti-shape > $out/${tag}_trace.log 2>&1
python $root/profiles/summarize.py $(find $out/${tag}_trace -name "*_results.db") | grep -E "$keep|rocclr" >> $f
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" ${PMC_MORE_GROUPS:-}; do
  i=$((i+1))
  d=$out/${tag}_p$i
  rm -rf $d
  rocprofv3 --pmc $grp -d $d -o sc -- python $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kitti-shape ${PMC_BENCH_ARGS:-} > $d.log 2>&1
  echo >> $f
  python $root/profiles/summarize.py $(find $d -name "*_results.db") | grep -E "$keep" >> $f
done
python - $out $tag <<'PY'
import glob, json, re, sqlite3, sys
out, tag = sys.argv[1], sys.argv[2]
res = {"source": f"gpurun_out/{tag}.txt: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes of `python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kitti-shape` (with the extra "
                 "workloads); KiB per launch; FETCH_SIZE x2 (gfx950 wide-read correction, MI355X_MICROARCH.md); keys: kernel @ grid size",
       "workload": {"db": 100000, "queries": 4096, "n_gpus": 1}}
for db in glob.glob(f"{out}/{tag}_p*/**/*_results.db", recursive=True):
    c = sqlite3.connect(db)
    per = {}
    for name, g, ctr, d, val in c.execute("select kernel_name, grid_size, counter_name, dispatch_id, sum(value) from counters_collection "
                                          "where counter_name in ('FETCH_SIZE', 'WRITE_SIZE') group by kernel_name, grid_size, counter_name, dispatch_id"):
        per.setdefault((name, g, ctr), []).append(val)
    for (name, g, ctr), vals in per.items():
        if "pr::" not in name:
            continue
        mm = re.search(r"([A-Za-z_]\w*(?:<[^()]*?>)?)\(", name.replace("(anonymous namespace)::", ""))
        short = (mm.group(1) if mm else name.strip()) + " @ " + str(g)
        big = [v for v in vals if v > 0.1 * max(vals)] or vals
        e = res.setdefault(short, {})
        e[ctr + "_KiB"] = sum(big) / len(big)
        e[ctr + "_dispatches"] = [len(big), len(vals)]
        if short.startswith("sc_match_e_kernel<true") and g == 262144:
            e["launch"] = "structure channel"
for k, v in res.items():
    if isinstance(v, dict) and "FETCH_SIZE_KiB" in v:
        v["fetch_correction"] = 2.0
        v["hbm_bytes_per_launch"] = (2.0 * v["FETCH_SIZE_KiB"] + v.get("WRITE_SIZE_KiB", 0.0)) * 1024
json.dump(res, open(f"{out}/{tag}_traffic.json", "w"), indent=1)
PY
rm -rf $out/${tag}_trace $out/${tag}_p[0-9]*
wc -l $f
