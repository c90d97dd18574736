#!/bin/bash
# Scratch job runner for gpurun calls: scripts/gpu_job.sh <name> ; runs scripts/jobs/<name>.sh
mkdir -p gpurun_out
bash "scripts/jobs/$1.sh" > "gpurun_out/$1.log" 2>&1
tail -c 6000 "gpurun_out/$1.log"
