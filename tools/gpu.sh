#!/bin/bash
# ONE runner for everything that happens on the GPU box (gpurun -- 'bash tools/gpu.sh <verb> ...; bash tools/gpu.sh <verb> ...').
# Replaces the 30 one-shot tools/gpu_*.sh scripts of rounds 1-5 (VERDICT r5 Weak 10).  Every verb writes under gpurun_out/
# (merged back by gpurun) and prints a short summary.  Verbs:
#
#   tests  <name> [pytest args...]          pytest -m gpu (default: the whole suite) -> <name>_pytest.log
#   bench  <name> [bench.py args...]        one bench.py line -> <name>.json (+ .err)
#   quick  <label> [bench.py args...]       bench.py without its side legs; prints "label tok/s device_ms ttft"
#   ab     <name> <rounds> VAR=a,b [args]   quick bench alternating an ENV knob (VAR=,1 = unset vs 1), <rounds> passes -> <name>.txt
#   kt     <name> <python args...>          rocprofv3 --kernel-trace --stats over `python <args>` -> <name>_kernel_stats.csv
#   pmc    <name> "<counters>" <python args...>   one rocprofv3 --pmc pass (--kernel-trace only) -> <name>.csv
#   sweep  <name> batch|ctx v1 v2 ...       bench.py quick over batch sizes / contexts -> <name>.txt
#   py     <name> <python args...>          plain python command, stdout+stderr -> <name>.txt
repo=${GRAFT_REPO_ROOT:-/root/repo}
out=$repo/gpurun_out
mkdir -p $out
QUICK="--cpu-steps 0 --profile-iters 0 --ttft10k-iters 0 --sweep-steps 0"
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
line() {  # label, bench args... -> one summary line
  local label=$1; shift
  PEGAINFER_BENCH_TRAFFIC=0 timeout 300 python $repo/bench.py $QUICK "$@" 2>/dev/null | grep '^{' | tail -1 | python -c "
import sys, json
s = sys.stdin.read().strip()
if not s:
    print('$label', 'NO LINE'); sys.exit(0)
d = json.loads(s)
h = d.get('host_loop', {}).get('sync_per_step', {})
print('$label', 'tok/s', d['value'], 'ms_per_step', d['ms_per_step'], 'device_ms', d['tpot_ms']['device_p50'], 'sync_tok_s', h.get('tok_s'), 'ttft_ms', d['ttft_ms']['p50'], 'gate_up_us', (d.get('roofline') or {}).get('avg_launch_us'))"
}
verb=$1; shift
case $verb in
  tests)
    name=$1; shift
    cd $repo
    if [ $# -eq 0 ]; then set -- tests; fi
    timeout 1700 python -m pytest "$@" -m gpu -q --durations=12 > $out/${name}_pytest.log 2>&1
    echo "pytest rc $?" >> $out/${name}_pytest.log
    grep -E "passed|failed|error|rc " $out/${name}_pytest.log | tail -6
    ;;
  bench)
    name=$1; shift
    cd $repo
    timeout 900 python bench.py "$@" > $out/$name.json 2> $out/$name.err
    echo "bench rc $?" >> $out/$name.err
    tail -c 700 $out/$name.json
    ;;
  quick)
    label=$1; shift
    cd $repo
    line "$label" "$@"
    ;;
  ab)
    name=$1; rounds=$2; knob=$3; shift 3
    var=${knob%%=*}; IFS=',' read -ra vals <<< "${knob#*=}"
    cd $repo
    for r in $(seq 1 $rounds); do
      for v in "${vals[@]}"; do
        if [ -z "$v" ]; then (unset $var; line "$var=<unset>" "$@"); else (export $var="$v"; line "$var=$v" "$@"); fi
      done
    done | tee $out/$name.txt
    ;;
  kt)
    name=$1; shift
    export PEGAINFER_BENCH_TRAFFIC=0     # no nested rocprofv3 (bench.py's live traffic probe) under the tracer
    cd /tmp && export TMPDIR=/tmp
    rm -rf /tmp/prof_$name
    (cd $repo && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o run -- python "$@" > /tmp/prof_$name.log 2>&1)
    python $repo/tools/rocpd_stats.py "$(find /tmp/prof_$name -name '*.db' | head -1)" $out/${name}_kernel_stats.csv > /dev/null
    grep '^{' /tmp/prof_$name.log | tail -1 > $out/${name}_under_rocprof.json
    grep -v '^{' /tmp/prof_$name.log | grep -v amdgpu.ids | tail -20 > $out/${name}_under_rocprof.txt
    head -24 $out/${name}_kernel_stats.csv | cut -c1-170
    ;;
  pmc)
    name=$1; ctr=$2; shift 2
    [ "$ctr" = SQ ] && ctr=$SQ
    export PEGAINFER_BENCH_TRAFFIC=0
    cd /tmp && export TMPDIR=/tmp
    rm -rf /tmp/pmc_$name
    (cd $repo && timeout 600 rocprofv3 --pmc $ctr --kernel-trace -d /tmp/pmc_$name -o run -- python "$@" > /tmp/pmc_$name.log 2>&1)
    python $repo/tools/rocpd_pmc.py "$(find /tmp/pmc_$name -name '*.db' | head -1)" $out/$name.csv ${PMC_BY_GRID:+--by-grid}   # PMC_BY_GRID=1: one row per (kernel, grid size)
    head -40 $out/$name.csv | cut -c1-170
    ;;
  sweep)
    name=$1; what=$2; shift 2
    cd $repo
    for v in "$@"; do
      if [ $what = batch ]; then line "bs $v" --batch $v --steps 48 --ttft-iters 1; else line "ctx $v" --ctx $v --steps 32 --ttft-iters 2; fi
    done | tee $out/$name.txt
    ;;
  py)
    name=$1; shift
    cd $repo
    timeout 900 python "$@" 2>&1 | grep -v amdgpu.ids | tee $out/$name.txt | tail -40
    ;;
  *)
    echo "unknown verb $verb"; exit 2 ;;
esac
