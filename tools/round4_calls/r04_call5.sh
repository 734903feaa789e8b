#!/bin/bash
# round 4, GPU call 5: instruction-rate micro-benchmark, packed-fp32 standalone reproducer
mkdir -p gpurun_out/c5
timeout 120 tools/_pk/valu_rate > gpurun_out/c5/valu_rate.txt 2>&1
cat gpurun_out/c5/valu_rate.txt
timeout 900 tools/pk_f32_repro.sh run 400 6 > gpurun_out/c5/pk_f32_repro.txt 2>&1
cat gpurun_out/c5/pk_f32_repro.txt
