mkdir -p gpurun_out/r3b
for wl in flow_only joint; do
for fuse in 0 1; do
CVFT_BLOCK_FUSE=$fuse python bench.py --workload $wl --steps 40 --warmup 8 --no-cpu-baseline --no-roofline 2>gpurun_out/r3b/${wl}_f${fuse}.err | tail -1 > gpurun_out/r3b/${wl}_f${fuse}.json
python -c "
import json;d=json.load(open('gpurun_out/r3b/${wl}_f${fuse}.json'));print('$wl fuse=$fuse', d['ms_per_step'], d['value'])"
done; done
