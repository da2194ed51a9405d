mkdir -p gpurun_out
for v in base fm base fm; do
  if [ $v = base ]; then unset PR_AMD_LIB; else export PR_AMD_LIB=$PWD/tools/expbuild/libpr_amd_$v.so; fi
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['ms_per_launch'],3), d['parity']['planted_top1_correct'])"
done
unset PR_AMD_LIB
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra --force-exchange 2>gpurun_out/fx.err | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('force-exchange', round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['ms_per_launch'],3), d['parity']['planted_top1_correct'], d['config']['step'])"; tail -n 3 gpurun_out/fx.err
