bash tools/abn.sh 2 nopf | sed 's/select.*//'
for m in 1 8 16 32 64; do echo "m=$m new: $(python tools/latency_probe.py $m 2>/dev/null | tail -1)   old: $(PR_AMD_LIB=tools/expbuild/libpr_amd_nopf.so python tools/latency_probe.py $m 2>/dev/null | tail -1)"; done
