#!/bin/bash
# Regenerates, on the GPU box, everything profiles/ holds for the current build (run from the repo root):
#   gpurun_out/final/{bench_n1.json, kernel_stats.csv, counters.json, pmc_traffic*.json, tests.log}
# Copy the files into profiles/ afterwards (profiles/README.md names them).  rocprofv3 gets the program directly after `--`.
set -e -o pipefail
OUT=gpurun_out/final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $OUT/prof.log 2>&1
cp "$(find $OUT/prof -name '*kernel_stats.csv' | head -1)" $OUT/kernel_stats.csv
python tools/step_summary.py "$(find $OUT/prof -name '*kernel_trace.csv' | head -1)" $OUT/step_summary.json joint 16 500
python tools/cu_time.py "$(find $OUT/prof -name '*kernel_trace.csv' | head -1)" 5 40 $OUT/cu_time.json joint 16 500 > $OUT/cu_time.txt || true
rm -rf $OUT/prof
# (the bench line takes launches / kernel time per step from the newest profiles/*step_summary*.json, and WHICH kernel its roofline record
# is about from the newest profiles/*cu_time.json: this build's trace first)
cp $OUT/step_summary.json profiles/r4_step_summary.json
cp $OUT/cu_time.json profiles/r4_cu_time.json
python tools/counters.py collect $OUT/counters > $OUT/counters.log 2>&1
python tools/counters.py summarise $OUT/counters $OUT/counters.json > $OUT/counters.txt
F=$(find $OUT/counters/fetch -name '*counter_collection.csv' | head -1)
W=$(find $OUT/counters/write -name '*counter_collection.csv' | head -1)
python tools/pmc_traffic.py $F $W 'gemm_glds_kernelILi64ELi64ELi2ELi2ELi0ELi4ELb1ELb0ELb0' 'gemm_glds_kernel<bf16,64,64,2,2,ns4,regepi>' $OUT/pmc_traffic.json
python tools/pmc_traffic.py $F $W 'gemm_glds_kernelILi128ELi128ELi4ELi2ELi0ELi2ELb1ELb0ELb0' 'gemm_glds_kernel<bf16,128,128,4,2,ns2,regepi>' $OUT/pmc_traffic_128x128.json
python tools/pmc_traffic.py $F $W 'gemm_p256_kernel' 'gemm_p256_kernel<bf16,256,256,2,4,ring10>' $OUT/pmc_traffic_p256.json || true
python tools/pmc_traffic.py $F $W 'gemm_glds_kernelILi96ELi256ELi3ELi4ELi0ELi3ELb1ELb0ELb0' 'gemm_glds_kernel<bf16,96,256,3,4,ns3,regepi>' $OUT/pmc_traffic_96x256.json || true
python tools/pmc_traffic.py $F $W 'block_tail_fwd_kernel' 'block_tail_fwd' $OUT/pmc_traffic_block_tail_fwd.json || true
python tools/pmc_traffic.py $F $W 'block_link_fwd_kernel' 'block_link_fwd' $OUT/pmc_traffic_block_link_fwd.json || true
python tools/pmc_traffic.py $F $W 'block_link_bwd_kernel' 'block_link_bwd' $OUT/pmc_traffic_block_link_bwd.json || true
python tools/pmc_traffic.py $F $W 'block_qkv_wide_fwd_kernel' 'block_qkv_wide_fwd' $OUT/pmc_traffic_block_qkv_fwd.json || true
python tools/pmc_traffic.py $F $W 'block_tail_wide_bwd_kernel' 'block_tail_wide_bwd' $OUT/pmc_traffic_block_tail_wide_bwd.json || true
rm -rf $OUT/counters
# (the bench line quotes this build's counter records: they go into profiles/ BEFORE the line is made)
cp $OUT/counters.json profiles/r4_counters.json; cp $OUT/counters.txt profiles/r4_counters.txt
cp $OUT/pmc_traffic.json profiles/r4_pmc_traffic.json; cp $OUT/pmc_traffic_128x128.json profiles/r4_pmc_traffic_128x128.json
for k in p256 96x256 block_tail_fwd block_link_fwd block_link_bwd block_qkv_fwd block_tail_wide_bwd; do [ -f $OUT/pmc_traffic_$k.json ] && cp $OUT/pmc_traffic_$k.json profiles/r4_pmc_traffic_$k.json; done
CVFT_BENCH_DUMP_PROFILE=$OUT/insitu_gemm_table.txt python bench.py --steps 20 --warmup 5 > $OUT/bench_n1.log 2>&1
tail -1 $OUT/bench_n1.log > $OUT/bench_n1.json
python bench.py --workload flow_only --batch 8 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | tail -1 > $OUT/bench_flow_only_b8.json
python bench.py --workload flow_only --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | tail -1 > $OUT/bench_flow_only_b16.json
python bench.py --workload llm_only --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | tail -1 > $OUT/bench_llm_only_b16.json
python bench.py --ragged 2 --steps 120 --warmup 30 --no-cpu-baseline --no-roofline 2>&1 | tail -1 > $OUT/bench_ragged2.json
# eager launches through the same trainer (no hipGraph): what the captured micro-step buys
python bench.py --graph 0 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>&1 | tail -1 > $OUT/bench_eager.json
# BASELINE configs[4] shape on one GPU (r = 64, 1000-frame mel, B = 16): bf16 vs the fp8 frozen-W GEMM path
python bench.py --frames 1000 --rank-lora 64 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>&1 | tail -1 > $OUT/bench_cfg4_bf16.json
python bench.py --frames 1000 --rank-lora 64 --fp8 1 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>&1 | tail -1 > $OUT/bench_cfg4_fp8.json
# the driver's N = 2 launch line with both ranks on this one GPU over gloo (rehearsal of the launch path, not a scaling number)
CVFT_SINGLE_DEVICE=1 CVFT_DIST_BACKEND=gloo timeout -k 10 280 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 4 --warmup 2 --no-roofline > $OUT/bench_dp2.log 2>&1 || true
grep "^{" $OUT/bench_dp2.log | tail -1 > $OUT/bench_dp2_one_gpu_gloo.json || true      # (an empty file = the launch failed: bench_dp2.log says why)
echo refreshed
