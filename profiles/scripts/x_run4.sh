mkdir -p gpurun_out/fp
run() { name=$1; shift; python bench.py --no-cpu-baseline --steps 3 --warmup 2 "$@" > gpurun_out/fp/$name.json 2> gpurun_out/fp/$name.err; python -c "
import json,sys; d=json.loads(open('gpurun_out/fp/$name.json').read().strip().splitlines()[-1]); print('$name', d['value'], d['config']['ms_per_read_block'])"; }
run timed
run roofonly --roofline-only
run notiming --no-stage-timing
run timed2
