#!/bin/bash
mkdir -p gpurun_out
timeout 900 tools/stream_bench ${1:-1024} ${2:-20} > gpurun_out/stream_bench.jsonl 2>&1; echo "rc=$?"; grep -v '"copy' gpurun_out/stream_bench.jsonl
