#!/bin/bash
# end-to-end rates at a few sizes (bench.py, long runs): the engine's own choice of pass
for n in 256 384 512; do
python bench.py --nx $n --ny $n --nz $n --steps $((768000/n)) --warmup 300 --no-cpu-baseline --no-small --no-reference-on-gpu $@ 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; b=r.get('boundary',{}); print('$n', d['value'], r['kernel'], r['kernel_ms'], b.get('ms'), b.get('third_level_ms',{}).get('boundary_nodes_to_t3'), b.get('third_level_ms',{}).get('fixup_list_of_the_shell_nodes'))"
done
