mkdir -p gpurun_out
bash tools/profile_round.sh r05_final > /dev/null 2>&1
sed -n 5p gpurun_out/r05_final.txt | cut -c1-300
