mkdir -p gpurun_out/fp
run() { name=$1; shift; env "$@" DMH_BENCH_DUMP_TIMERS=1 python bench.py --no-cpu-baseline --steps 3 --warmup 2 > gpurun_out/fp/$name.json 2> gpurun_out/fp/$name.err; python -c "
import json,sys; d=json.loads(open('gpurun_out/fp/$name.json').read().strip().splitlines()[-1]); print('$name', d['value'], d['config']['ms_per_read_block'])"; grep "timer host" gpurun_out/fp/$name.err | grep "wait\|grow\|finish\|flush" | head -8; }
run ov10 DMH_X=1
run ov12 DMH_OVERLAP_PASSES=12
run ov6 DMH_OVERLAP_PASSES=6
run ov10sync DMH_SYNC_REPORT_JOBS=1
