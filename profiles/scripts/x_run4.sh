mkdir -p gpurun_out/fp
run() { name=$1; shift; python bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" > gpurun_out/fp/$name.json 2> gpurun_out/fp/$name.err; python -c "
import json,sys; d=json.loads(open('gpurun_out/fp/$name.json').read().strip().splitlines()[-1]); print('$name', d['value'], d['config']['ms_per_read_block'])"; }
run lognormal --read-lengths lognormal
run fixed
