mkdir -p gpurun_out/fp
run() { name=$1; shift; env "$@" python bench.py --workload ecoli --no-cpu-baseline --no-alone --steps 6 --warmup 2 > gpurun_out/fp/$name.json 2> gpurun_out/fp/$name.err; python -c "
import json,sys; d=json.loads(open('gpurun_out/fp/$name.json').read().strip().splitlines()[-1]); print('$name', d['value'], d['config']['ms_per_read_block'])" || tail -5 gpurun_out/fp/$name.err; }
run e_def X=1
run e_nopf DMH_PREFETCH_OVERLAP=0
run e_def2 X=1
run e_nopf2 DMH_PREFETCH_OVERLAP=0
