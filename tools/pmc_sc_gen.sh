#!/bin/bash
# usage (GPU box, repo root): tools/pmc_sc_gen.sh  -> gpurun_out/pmc_sc_gen.txt : PMC passes over tools/bench_sc_gen.py (SC generation kernels)
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
: > $out/pmc_sc_gen.txt
i=0
for grp in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD" "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1)); d=$out/scgen_p$i; rm -rf $d
  rocprofv3 --kernel-trace --pmc $grp -d $d -o r --output-format csv -- python $root/tools/bench_sc_gen.py > $d.log 2>&1
  f=$(find $d -name "*counter_collection.csv" | head -n 1)
  python - "$f" >> $out/pmc_sc_gen.txt <<'PY'
import sys, csv, collections
acc = collections.defaultdict(float); n = collections.defaultdict(int)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "sc_bin" in k or "cloud_frames" in k or "sc_gen_fused" in k or "ave_chain" in k:
        nm = [t for t in ("sc_bin_kernel", "cloud_frames_kernel", "sc_gen_fused_kernel", "ave_chain_kernel") if t in k][0]
        key = (nm + " grid " + r.get("Grid_Size", "?"), r["Counter_Name"]); acc[key] += float(r["Counter_Value"]); n[key] += 1
for (k, c) in sorted(acc): print(f"{k:42s} {c:28s} {acc[(k, c)] / n[(k, c)]:16.6g} per dispatch (n={n[(k, c)]})")
PY
  rm -rf $d
done
cat $out/pmc_sc_gen.txt
