mkdir -p gpurun_out
python tools/model_check.py r05kc > gpurun_out/r05kc_model.log 2>&1; tail -5 gpurun_out/r05kc_model.log
bash tools/profile_round.sh r05kc > /dev/null 2>&1
grep -E "sc_match_e|sc_pack_h_col|\"value\"" gpurun_out/r05kc.txt | cut -c1-220 | head -20
