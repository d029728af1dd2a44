mkdir -p gpurun_out/fp
run() { name=$1; shift; env "$@" DMH_BENCH_DUMP_TIMERS=1 python bench.py --no-cpu-baseline --steps 3 --warmup 2 > gpurun_out/fp/$name.json 2> gpurun_out/fp/$name.err; python -c "
import json,sys; d=json.loads(open('gpurun_out/fp/$name.json').read().strip().splitlines()[-1]); print('$name', d['value'], d['config']['ms_per_read_block'])"; grep "timer host" gpurun_out/fp/$name.err | grep "wait\|grow\|finish\|flush" | head -8; }
run base DMH_X=1
run base2 DMH_X=1
