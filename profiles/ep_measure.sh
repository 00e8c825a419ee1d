set -x
mkdir -p gpurun_out/ep
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-gather-bench --no-f32-line > gpurun_out/ep/n1.json 2> gpurun_out/ep/n1.err
for G in 2 4 8; do
  python bench.py --steps 20 --warmup 4 --ep-emulate $G > gpurun_out/ep/emu$G.json 2> gpurun_out/ep/emu$G.err
done
python bench.py --steps 20 --warmup 4 --parallel ep --force-dist --no-cpu-baseline --no-gather-bench --no-f32-line > gpurun_out/ep/force_ep.json 2> gpurun_out/ep/force_ep.err
python bench.py --steps 20 --warmup 4 --model fnn --ep-emulate 8 > gpurun_out/ep/emu8_fnn.json 2> gpurun_out/ep/emu8_fnn.err
tail -3 gpurun_out/ep/*.err
