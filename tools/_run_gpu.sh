mkdir -p gpurun_out
: > gpurun_out/r05_fuzz4.txt
for s in 71 72 73 74 75 76; do
  echo "## seed $s" >> gpurun_out/r05_fuzz4.txt
  timeout 700 python tools/fuzz_all.py $s 30 2>&1 | grep -E "^BAD|fuzz_all:" >> gpurun_out/r05_fuzz4.txt
done
for s in 81 82; do
  echo "## big seed $s" >> gpurun_out/r05_fuzz4.txt
  timeout 900 python tools/fuzz_all.py $s 12 match,group,matcher,big 2>&1 | grep -E "^BAD|fuzz_all:" >> gpurun_out/r05_fuzz4.txt
done
cat gpurun_out/r05_fuzz4.txt
