mkdir -p gpurun_out/fp
run() { name=$1; shift; env "$@" python bench.py --no-cpu-baseline --no-alone --steps 4 --warmup 2 > gpurun_out/fp/$name.json 2> gpurun_out/fp/$name.err; python -c "
import json,sys; d=json.loads(open('gpurun_out/fp/$name.json').read().strip().splitlines()[-1]); s=d['device_stage_ms']; n=d['stage_events']['read_blocks_with_events']; print('$name', d['value'], d['config']['ms_per_read_block'], {k: round(s.get(k,0)/n,1) for k in ('chain_simple_groups','chain_lane_groups','merge_match','read_sort','merge_emit')})" || tail -5 gpurun_out/fp/$name.err; }
run sg512 DMH_SIMPLE_GRID=512
run sg1024 DMH_SIMPLE_GRID=1024
run sg4096 DMH_SIMPLE_GRID=4096
run sg512l8 DMH_SIMPLE_GRID=512 DMH_LANE_GRID_DIV=8
