#!/bin/bash
# usage (GPU box, repo root): tools/profile_round.sh <tag>   -> gpurun_out/<tag>.txt
# rocprofv3 kernel trace of bench.py (default arithmetic) + one PMC pass per counter group (never combined with traces),
# summarised by profiles/summarize.py.
set -u
tag=${1:-round}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
txt=$out/$tag.txt
{
echo "# $tag: python bench.py --steps 3 --warmup 1 --no-cpu-baseline   (PMC passes: --steps 1 --warmup 0)"
echo "# MI355X, ROCm 7.2, rocprofv3 --kernel-trace --stats, then one --pmc group per run"
echo
echo "## bench.py JSON line (un-profiled run, steps 5, with the CPU baseline)"
python $root/bench.py --steps 5 --warmup 2 2>/dev/null | tail -1
echo
echo "## bench.py JSON line, --sc-arith f32 (un-profiled)"
python $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra --sc-arith f32 2>/dev/null | tail -1
echo
} > $txt
rm -rf $out/${tag}_trace
rocprofv3 --kernel-trace --stats -d $out/${tag}_trace -o sc -- python $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $out/${tag}_trace.log 2>&1
python $root/profiles/summarize.py $(find $out/${tag}_trace -name "*_results.db") | sed "s#$out/##" >> $txt
rm -rf $out/${tag}_trace
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1)); d=$out/${tag}_pmc$i; rm -rf $d
  rocprofv3 --pmc $grp -d $d -o sc -- python $root/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra > $d.log 2>&1
  python $root/profiles/summarize.py $(find $d -name "*_results.db") | sed "s#$out/##" >> $txt
  rm -rf $d
done
cat $txt | cut -c1-220
