#!/bin/bash
mkdir -p gpurun_out
python tools/sweep_stream.py --precision f32 --steps 10 --out gpurun_out/sweep_1024_f32.json > gpurun_out/sweep_f32.log 2>&1; grep BEST gpurun_out/sweep_f32.log
python tools/sweep_stream.py --precision f64 --steps 10 --out gpurun_out/sweep_1024_f64.json > gpurun_out/sweep_f64.log 2>&1; grep BEST gpurun_out/sweep_f64.log
python tools/sweep_stream.py --precision f64 --n 256 --steps 200 --out gpurun_out/sweep_256_f64.json > gpurun_out/sweep_256.log 2>&1; grep BEST gpurun_out/sweep_256.log
