#!/bin/bash
# one short bench line per argument set.  usage: ab_quick.sh <tag> "<bench args>" ["<bench args>" ...]; env in DMH_AB_ENV_<n>
set -u
cd "${GRAFT_REPO_ROOT:?not on a gpurun box}" || exit 1
export TMPDIR=/tmp
T="$1"; shift
O="gpurun_out/ab_$T"
rm -rf "$O"; mkdir -p "$O"
n=0
for a in "$@"; do
  n=$((n+1))
  envs=$(eval echo "\${DMH_AB_ENV_$n:-}")
  env $envs timeout -k 10 300 python3 bench.py $a --no-cpu-baseline > "$O/run$n.json" 2> "$O/run$n.err" || { tail -5 "$O/run$n.err"; exit 1; }
  python3 -c "
import json,sys
j=json.loads(open('$O/run$n.json').read().strip().splitlines()[-1]); a=j.get('device_stage_ms_alone_one_block',{})
print('run$n [$envs $a]', j['value'], j['config']['ms_per_read_block'], j.get('alignment_ladder'))
print('   alone:', {k:v for k,v in a.items() if v >= 2.0})"
done
