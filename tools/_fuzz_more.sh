set -o pipefail
mkdir -p gpurun_out
: > gpurun_out/r05_fuzz2.txt
for s in 21 22 23 24 25 26 27 28; do
  echo "## seed $s" >> gpurun_out/r05_fuzz2.txt
  timeout 600 python tools/fuzz_all.py $s 30 2>&1 | tail -4 >> gpurun_out/r05_fuzz2.txt
done
for s in 31 32 33; do
  echo "## big seed $s" >> gpurun_out/r05_fuzz2.txt
  timeout 900 python tools/fuzz_all.py $s 12 match,group,matcher,big 2>&1 | tail -4 >> gpurun_out/r05_fuzz2.txt
done
python tools/bench_xrow.py 50000 m2dp > gpurun_out/r05_xrow_m2dp.txt 2>&1
grep -c "fuzz_all: ok" gpurun_out/r05_fuzz2.txt
