mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r05_gputest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r05_gputest.log
grep -E "passed|failed|rc=" gpurun_out/r05_gputest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
bash tools/profile_round.sh r05_final > /dev/null 2>&1
sed -n 5p gpurun_out/r05_final.txt | cut -c1-200
