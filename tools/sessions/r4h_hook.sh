#!/bin/bash
# Round 4: the CustomEncoder hook with several batches on the device at once (lanes) and the fused S2 LDS kernel.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4h
g++ -O2 -std=c++17 -I include tools/hook_bench.cpp -o /tmp/hook_bench -L compress_amd -lkcgpu -Wl,-rpath,$PWD/compress_amd -lpthread
for lanes in ${LANES:-1 2 3 4}; do timeout 120 /tmp/hook_bench $lanes; done 2>&1 | tee gpurun_out/r4h/hook_bench.jsonl
for lanes in 2 4; do echo "GPU_MAX_HW_QUEUES=8"; GPU_MAX_HW_QUEUES=8 timeout 120 /tmp/hook_bench $lanes; done 2>&1 | tee -a gpurun_out/r4h/hook_bench.jsonl
