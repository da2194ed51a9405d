mkdir -p gpurun_out
: > gpurun_out/r05_fuzz5.txt
for s in 101 102 103 104; do
  echo "## seed $s" >> gpurun_out/r05_fuzz5.txt
  timeout 700 python tools/fuzz_all.py $s 30 2>&1 | grep -E "^BAD|fuzz_all:" >> gpurun_out/r05_fuzz5.txt
done
echo "## big seed 111" >> gpurun_out/r05_fuzz5.txt
timeout 900 python tools/fuzz_all.py 111 12 match,group,matcher,big 2>&1 | grep -E "^BAD|fuzz_all:" >> gpurun_out/r05_fuzz5.txt
cat gpurun_out/r05_fuzz5.txt
