mkdir -p gpurun_out
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra --force-exchange > gpurun_out/fx.out 2> gpurun_out/fx.err; echo "rc=$?"; tail -n 15 gpurun_out/fx.err | cut -c1-400; head -c 600 gpurun_out/fx.out
