#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest gpu"; timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu.log
echo "== stream_bench"; timeout 900 tools/stream_bench 1024 20 > gpurun_out/stream_bench_1024.jsonl 2>&1; echo "rc=$?"; cat gpurun_out/stream_bench_1024.jsonl
rocprofv3 -L > gpurun_out/counters_list.txt 2>&1
echo "== pmc pass 1"; timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d gpurun_out/pmc1 -o p -- tools/stream_bench 1024 5 prof > gpurun_out/pmc1.log 2>&1; echo "rc=$?"
echo "== pmc pass 2"; timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM --output-format csv -d gpurun_out/pmc2 -o p -- tools/stream_bench 1024 5 prof > gpurun_out/pmc2.log 2>&1; echo "rc=$?"
echo "== pmc pass 3"; timeout 600 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_REQ_sum TCC_READ_sum --output-format csv -d gpurun_out/pmc3 -o p -- tools/stream_bench 1024 5 prof > gpurun_out/pmc3.log 2>&1; echo "rc=$?"
find gpurun_out -name "*.csv" | head -20
