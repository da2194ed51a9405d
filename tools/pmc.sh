#!/bin/bash
# usage (on the GPU box, from the repo root): tools/pmc.sh <tag> "<counters of pass 1>" ["<counters of pass 2>" ...]
# one rocprofv3 --pmc pass per counter group over one bench step; summaries to gpurun_out/<tag>.txt
set -u
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
: > $out/$tag.txt
for grp in "$@"; do
  i=$((i+1))
  d=$out/${tag}_p$i
  rm -rf $d
  rocprofv3 --pmc $grp -d $d -o sc -- python $root/bench.py --steps 1 --warmup 0 --no-cpu-baseline ${PMC_BENCH_ARGS:-} > $d.log 2>&1
  python $root/profiles/summarize.py $(find $d -name "*_results.db") | grep -E "^==|sc_match" >> $out/$tag.txt
done
cat $out/$tag.txt
