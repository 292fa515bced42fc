#!/bin/bash
# bench.py --gpus N as the driver launches it, on ONE GPU: the ranks are processes that share the device, tests/mock_rccl/mock_rccl_shm.cpp
# stands in for librccl (shared memory instead of links).  Rates mean nothing here (N ranks take turns on one chip); what it shows: the
# whole multi-process path -- parity check over the transport, agreement, three-step passes with three exchanges, the bench line -- at the
# driver's sizes.   tools/dry_run_world.sh <N> [bench.py args]
N=${1:-2}; shift
/opt/rocm/bin/hipcc -O2 -fPIC -shared -std=c++17 tests/mock_rccl/mock_rccl_shm.cpp -o /tmp/libwvmockrccl.so -lrt || exit 1
python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus $N \
    --rccl-library /tmp/libwvmockrccl.so --no-cpu-baseline --no-reference-on-gpu "$@" 2>/tmp/dry_run_world.err | tail -1
