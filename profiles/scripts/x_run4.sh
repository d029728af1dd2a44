mkdir -p gpurun_out/fp
run() { name=$1; shift; env "$@" python bench.py --no-cpu-baseline --steps 4 --warmup 2 > gpurun_out/fp/$name.json 2> gpurun_out/fp/$name.err; python -c "
import json,sys; d=json.loads(open('gpurun_out/fp/$name.json').read().strip().splitlines()[-1]); s=d['device_stage_ms']; n=d['stage_events']['read_blocks_with_events']; print('$name', d['value'], d['config']['ms_per_read_block'], d['roofline']['avg_ms'], d['roofline']['alone']['avg_ms'], {k: round(s.get(k,0)/n,1) for k in ('merge_match','merge_fixup','read_sort','merge_emit')})" || tail -5 gpurun_out/fp/$name.err; }
run b29 DMH_BUCKET_BITS=29
run b28 DMH_BUCKET_BITS=28
run b27 DMH_BUCKET_BITS=27
