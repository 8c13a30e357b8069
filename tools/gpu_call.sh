#!/bin/bash
# The one GPU-box driver script (run through gpurun):  tools/gpu_call.sh <tag> <step> [<step> ...]
# Steps run in order; logs land in gpurun_out/<tag>/.  Steps:
#   smoke              __graft_entry__.smoke()
#   tests              the whole `-m gpu` suite
#   k=<expr>           pytest -m gpu -k <expr>
#   t=<file>           pytest -m gpu tests/<file>
#   bench[=<flags>]    python bench.py <flags>         (flags with '+' for spaces: bench=--steps+5)
#   py=<script>[+args] python <script> args            (e.g. py=tools/time_conv4.py+--b+8)
#   prof=<name>=[ENV=v+...]python+<script>[+args]   rocprofv3 --kernel-trace --stats of the command (run through env)
#   pmc=<name>=<counters,comma>=[ENV=v+...]python+<script>[+args]   one rocprofv3 --pmc pass (own run, kernel-trace only)
#   (MF_MARK=<kernel substring> in the environment of gpu_call.sh: keep only launches between its 2nd and 3rd occurrence)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1; shift
O=gpurun_out/$tag; mkdir -p "$O"
for step in "$@"; do
  name=${step%%=*}; arg=${step#*=}; [ "$name" = "$step" ] && arg=""
  arg=${arg//+/ }
  echo "=== $step"
  case $name in
    smoke) timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ;;
    tests) timeout 2400 python -m pytest tests -m gpu -x -q > "$O/tests.log" 2>&1; echo "rc $?"; tail -5 "$O/tests.log" ;;
    k) timeout 1500 python -m pytest tests -m gpu -x -q -k "$arg" > "$O/k.log" 2>&1; echo "rc $?"; tail -15 "$O/k.log" ;;
    t) f=$(echo $arg | tr ' /' '__'); timeout 1500 python -m pytest -m gpu -x -q $(for a in $arg; do echo tests/$a; done) > "$O/t_$f.log" 2>&1; echo "rc $?"; tail -15 "$O/t_$f.log" ;;
    bench) n=$(ls "$O"/bench*.json 2>/dev/null | wc -l); timeout 900 python -X faulthandler bench.py $arg > "$O/bench$n.json" 2> "$O/bench$n.err"; echo "rc $?"; cut -c1-600 "$O/bench$n.json"; tail -3 "$O/bench$n.err" ;;
    py) s=$(echo "$arg" | cut -d' ' -f1 | xargs basename | sed 's/\.py$//'); timeout 900 python $arg > "$O/py_$s.log" 2>&1; echo "rc $?"; tail -40 "$O/py_$s.log" ;;
    prof) pn=${arg%%=*}; cmd=${arg#*=}; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_$pn" -o p -- env $cmd > "$O/prof_$pn.log" 2>&1; echo "rc $?"
          MF_MARK=${MF_MARK:-} python tools/kernel_stats.py "$O/prof_$pn" > "$O/prof_${pn}_kernel_stats.csv"; head -25 "$O/prof_${pn}_kernel_stats.csv" | cut -c1-160; rm -rf "$O/prof_$pn" ;;
    pmc) pn=${arg%%=*}; rest=${arg#*=}; ctr=${rest%%=*}; cmd=${rest#*=}; ctr=${ctr//,/ }
         timeout 900 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d "$O/pmc_$pn" -o p -- env $cmd > "$O/pmc_$pn.log" 2>&1; echo "rc $?"
         python tools/pmc_summary.py "$O/pmc_$pn" > "$O/pmc_$pn.json"; cut -c1-1500 "$O/pmc_$pn.json"; rm -rf "$O/pmc_$pn" ;;
    *) echo "unknown step $name" ;;
  esac
done
