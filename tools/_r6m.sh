#!/bin/bash
python tools/gpu_ab.py --rounds 5 --batch 37 --shape 200,300,50 rl=build/ab/rl.so fq=build/ab/fq.so fq1=build/ab/fq1.so > gpurun_out/factor_q1.log 2>&1
python tools/gpu_ab.py --rounds 5 --batch 1 rl=build/ab/rl.so fq=build/ab/fq.so fq1=build/ab/fq1.so >> gpurun_out/factor_q1.log 2>&1
python tools/gpu_ab.py --rounds 5 --batch 1024 fq1=build/ab/fq1.so fq=build/ab/fq.so >> gpurun_out/factor_q1.log 2>&1
cat gpurun_out/factor_q1.log
