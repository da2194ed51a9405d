mkdir -p gpurun_out
bash tools/profile_round.sh r02_mid > /dev/null 2>&1
wc -l gpurun_out/r02_mid.txt
# M2DP generation: VALU busy / instruction counts (SURVEY 8-d asks for VALUBusy of this VALU-bound stage)
cd /tmp && export TMPDIR=/tmp
cat > /tmp/m2gen.py <<'PY'
import sys, ctypes as C; sys.path.insert(0,'/root/repo')
import torch
from so_dso_place_recognition_amd import synth
from so_dso_place_recognition_amd.api import Context
P=lambda t: C.c_void_p(t.data_ptr())
N=256
xyz,it,offs=synth.scene_clouds_torch(42,N,50000)
ctx=Context(0); out=torch.empty((4*N,384),dtype=torch.float64,device='cuda'); torch.cuda.synchronize()
for _ in range(2): ctx.check(ctx.lib.pr_m2dp_generate_dev(ctx.h,P(xyz),P(it),P(offs),N,45.0,P(out)))
PY
for grp in "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD"; do
  rm -rf /root/repo/gpurun_out/m2g; rocprofv3 --pmc $grp -d /root/repo/gpurun_out/m2g -o m2g -- python /tmp/m2gen.py > /dev/null 2>&1
  python /root/repo/profiles/summarize.py $(find /root/repo/gpurun_out/m2g -name "*_results.db") | grep -E "m2dp_bin|m2dp_svd" >> /root/repo/gpurun_out/r02_m2gen_pmc.txt; rm -rf /root/repo/gpurun_out/m2g
done
rm -rf /root/repo/gpurun_out/m2g; rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/m2g -o m2g -- python /tmp/m2gen.py > /dev/null 2>&1
python /root/repo/profiles/summarize.py $(find /root/repo/gpurun_out/m2g -name "*_results.db") | grep -E "m2dp_|frames|ave_chain" >> /root/repo/gpurun_out/r02_m2gen_pmc.txt; rm -rf /root/repo/gpurun_out/m2g
cat /root/repo/gpurun_out/r02_m2gen_pmc.txt | cut -c1-160
