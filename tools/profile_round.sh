#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_round.sh <tag>     e.g. r03_mid
# -> gpurun_out/<tag>.txt: the un-profiled bench line, the rocprofv3 --kernel-trace --stats summary of the same command, and one --pmc pass per
# counter group (never combined with another trace domain); gpurun_out/<tag>_traffic.json: HBM bytes per launch of every kernel
# (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE, MI355X_MICROARCH.md).  Copy what should be judged into profiles/.
set -u
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
f=$out/$tag.txt
{
  echo "# $tag: python bench.py --steps 3 --warmup 1 (kernel trace: + --no-cpu-baseline; PMC passes: --steps 1 --warmup 0 --no-cpu-baseline --no-extra)"
  echo "# MI355X, rocprofv3 --kernel-trace --stats, then one --pmc group per run"
  echo
  echo "## bench.py JSON line (un-profiled run, with the CPU baseline)"
  python $root/bench.py --steps 5 --warmup 2 2> $out/${tag}_bench.err | tail -n 1
  echo
} > $f
rm -rf $out/${tag}_trace
rocprofv3 --kernel-trace --stats -d $out/${tag}_trace -o sc -- python $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/${tag}_trace.log 2>&1
python $root/profiles/summarize.py $(find $out/${tag}_trace -name "*_results.db") | grep -v "at::native\|elementwise_kernel" >> $f
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  d=$out/${tag}_p$i
  rm -rf $d
  rocprofv3 --pmc $grp -d $d -o sc -- python $root/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra ${PMC_BENCH_ARGS:-} > $d.log 2>&1
  echo >> $f
  python $root/profiles/summarize.py $(find $d -name "*_results.db") | grep -E "^==|^--|sc_match|sc_pack|fuse_select|row_moments|rerank_kernel" >> $f
done
python - $out $tag <<'PY'
import glob, json, re, sqlite3, sys
out, tag = sys.argv[1], sys.argv[2]
res = {"source": f"gpurun_out/{tag}.txt: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes of `python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra`; "
                 "KiB per launch; FETCH_SIZE x2 (gfx950 wide-read correction, MI355X_MICROARCH.md)",
       "workload": {"db": 100000, "queries": 4096, "n_gpus": 1}}
for db in glob.glob(f"{out}/{tag}_p*/**/*_results.db", recursive=True):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    disp = next((x for x in ("dispatch_id", "dispatch_index", "correlation_id") if x in cols), None)
    per = {}      # (kernel, counter) -> per-dispatch sums
    if disp:
        for name, ctr, d, val in c.execute(f"select kernel_name, counter_name, {disp}, sum(value) from counters_collection group by kernel_name, counter_name, {disp}"):
            per.setdefault((name, ctr), []).append(val)
    else:
        for name, ctr, val, n in c.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name"):
            per[(name, ctr)] = [val / n] * n
    for (name, ctr), vals in per.items():
        if ctr not in ("FETCH_SIZE", "WRITE_SIZE"):
            continue
        if "at::native" in name or "elementwise" in name:
            continue
        mm = re.search(r"([A-Za-z_]\w*(?:<[^()]*?>)?)\(", name.replace("(anonymous namespace)::", ""))
        short = mm.group(1) if mm else name.strip()
        # launches that left at once (the gated channel-1 launches of the binary path) are not the kernel's traffic:
        # the average is over the dispatches above a tenth of the largest
        big = [v for v in vals if v > 0.1 * max(vals)] or vals
        e = res.setdefault(short, {})
        e[ctr + "_KiB"] = sum(big) / len(big)
        e[ctr + "_dispatches"] = [len(big), len(vals)]
        if short.startswith("sc_match_e_kernel<true"):
            e["launch"] = "structure channel"      # (with the binary path on, the only launch of this kernel that does not leave at once)
for k, v in res.items():
    if isinstance(v, dict) and "FETCH_SIZE_KiB" in v:
        v["fetch_correction"] = 2.0
        v["hbm_bytes_per_launch"] = (2.0 * v["FETCH_SIZE_KiB"] + v.get("WRITE_SIZE_KiB", 0.0)) * 1024
json.dump(res, open(f"{out}/{tag}_traffic.json", "w"), indent=1)
PY
rm -rf $out/${tag}_trace $out/${tag}_p[0-9]*
tail -n 60 $f
