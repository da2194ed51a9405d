cd /root/repo
export TMPDIR=/tmp
for k in ${KERNELS:-p h}; do
  export PR_SC_KERNEL=$k
  i=0
  for ctr in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $ctr -d gpurun_out/pmc_$k$i -o r --output-format csv -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra > /dev/null 2>&1
    f=$(find gpurun_out/pmc_$k$i -name "*counter_collection.csv" | head -n 1)
    python - "$f" $k <<'PY'
import sys,csv,collections
acc=collections.defaultdict(float); n=collections.defaultdict(int)
for r in csv.DictReader(open(sys.argv[1])):
    if r["Kernel_Name"].startswith("sc_match") or "sc_match_" in r["Kernel_Name"]:
        acc[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]]+=1
for c in acc: print(sys.argv[2], c, acc[c], "dispatches", n[c])
PY
    rm -rf gpurun_out/pmc_$k$i
  done
done
