#!/bin/bash
# round 3, first GPU call: configs[1] at N = 16 (tests + bench line), DDP probes
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests/test_baseline_configs.py -x -v --timeout 180 --timeout-method thread -k "config1" > gpurun_out/r3a_config1_tests.log 2>&1 || { tail -30 gpurun_out/r3a_config1_tests.log; exit 1; }
tail -5 gpurun_out/r3a_config1_tests.log
timeout -k 10 300 python bench.py --workload config2 --steps 20 --warmup 5 > gpurun_out/r3a_bench_config2.json 2> gpurun_out/r3a_bench_config2.err || { tail -20 gpurun_out/r3a_bench_config2.err; exit 1; }
cat gpurun_out/r3a_bench_config2.json
export MASTER_ADDR=127.0.0.1
timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 tools/ddp_probe.py --backend gloo --tag gloo2_head --model head > gpurun_out/r3a_probe_gloo2_head.out 2>&1
echo "probe gloo2 head rc=$?"
timeout -k 10 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29612 tools/ddp_probe.py --backend nccl --tag nccl1 > gpurun_out/r3a_probe_nccl1.out 2>&1
rc=$?; echo "probe nccl1 artemis rc=$rc"; [ $rc -eq 124 -o $rc -eq 137 ] && exit 1
timeout -k 10 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29613 tools/ddp_probe.py --backend gloo --tag gloo2 > gpurun_out/r3a_probe_gloo2.out 2>&1
echo "probe gloo2 artemis rc=$?"
tail -3 gpurun_out/ddp_probe_*_rank0.log
