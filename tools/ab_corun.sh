#!/bin/bash
# A/B of the pipelined step on the GPU box: tools/ab_corun.sh <tag> "[ENV=v,ENV=v:]<bench flags with + for spaces>" ...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; mkdir -p "$O"; shift
n=0
for v in "$@"; do
  n=$((n+1))
  envs=""; flags=$v
  case $v in *:*) envs=${v%%:*}; flags=${v#*:};; esac
  env ${envs//,/ } timeout 600 python bench.py --steps 30 --no-extras --no-cpu-baseline --no-latency-probe ${flags//+/ } > "$O/bench_$n.json" 2> "$O/bench_$n.err"
  echo "[$v] rc $?"; python -c "
import json,sys
d=json.loads(open('$O/bench_$n.json').read().strip().splitlines()[-1])
print({k:d.get(k) for k in ('value','ms_per_step','serial_ms_per_step','value_serial')}, {k:v for k,v in d.get('stage_ms',{}).items()} if isinstance(d.get('stage_ms'),dict) else '')
"
done
