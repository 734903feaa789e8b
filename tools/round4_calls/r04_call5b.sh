#!/bin/bash
mkdir -p gpurun_out/c5
timeout 120 tools/_pk/valu_rate > gpurun_out/c5/valu_rate.txt 2>&1
cat gpurun_out/c5/valu_rate.txt
