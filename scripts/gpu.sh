#!/bin/bash
# The one GPU-box helper (the per-experiment gpu_*_r2.sh / gpu_r3_*.sh launchers of earlier rounds are gone: every run they made is a
# step list of this script -- e.g. the streaming-read ceiling probe is `kbench=kbench_stream`, a PMC pass is part of `profile`).  usage: scripts/gpu.sh <tag> <step> [<step> ...]
# Steps run in order and the chain stops at the first failure (no GPU step is started after one that failed or timed out).
#   tests[=<pytest -k expression>]   the -m gpu suite (or a selection)           -> gpurun_out/pytest_<tag>.log
#   file=<tests/x.py[::test]>        one test file / node id                      -> gpurun_out/pytest_<tag>.log (appended)
#   smoke                            __graft_entry__.smoke()                      -> gpurun_out/smoke_<tag>.log
#   bench[=<bench.py arguments>]     python bench.py ...                          -> gpurun_out/bench_<tag>[_n].json
#   profile                          scripts/gpu_profile.sh <tag>: bench + rocprofv3 stats + PMC passes -> profiles/<tag>_*
#   trace                            rocprofv3 --kernel-trace of a short headline run, one call launch by launch -> gpurun_out/timeline_<tag>.md
#   timeline                         scripts/call_timeline.py <tag>               -> profiles/<tag>_call_timeline.md
#   ab=<ENV=a>,<ENV=b>[,reps]        scripts/ab_env.py, both orders               -> gpurun_out/ab_<tag>.log
#   kbench=<name>[,args]             hipcc scripts/kbench/<name>.hip and run it   -> gpurun_out/kbench_<name>_<tag>.log
#   py=<script.py>[,args]            python <script> args                         -> gpurun_out/py_<tag>.log (appended)
#   sh=<script.sh>[,args]            bash <script> args                           -> gpurun_out/sh_<tag>.log (appended)
#        scripts/stats_of.sh,<bench.py args>            rocprofv3 per-kernel totals of one bench.py command line
#        scripts/trace_around.sh,<pattern> <args>       the launches around every kernel whose name holds <pattern>
#        scripts/config5_who_bounds.sh                  configs[4]'s streamed shard with and without the panel generator
#        scripts/kbench/sweep_gqd_short.sh              int8 K1 on short sample axes, k_gq_d against the skewed-round experiment
#        scripts/kbench/eig_ablate.sh                   the Jacobi eigensolver's ablations
tag=$1; shift
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
nb=0
fault() { if grep -q "Memory access fault" "$@" 2>/dev/null; then echo "GPU FAULT"; exit 1; fi; }
for step in "$@"; do
  name=${step%%=*}; arg=""; [ "$name" != "$step" ] && arg=${step#*=}
  echo "== $step"
  case $name in
    tests)
      if [ -n "$arg" ]; then timeout -k 10 1100 python -m pytest tests -m gpu -x -q --durations=10 -k "$arg" > gpurun_out/pytest_$tag.log 2>&1
      else timeout -k 10 1150 python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/pytest_$tag.log 2>&1; fi
      rc=$?; tail -25 gpurun_out/pytest_$tag.log; fault gpurun_out/pytest_$tag.log; if [ $rc -ne 0 ]; then exit $rc; fi ;;
    file)
      timeout -k 10 900 python -m pytest $arg -m gpu -x -q --durations=6 >> gpurun_out/pytest_$tag.log 2>&1
      rc=$?; tail -12 gpurun_out/pytest_$tag.log; fault gpurun_out/pytest_$tag.log; if [ $rc -ne 0 ]; then exit $rc; fi ;;
    smoke)
      timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_$tag.log 2>&1
      rc=$?; tail -6 gpurun_out/smoke_$tag.log; fault gpurun_out/smoke_$tag.log; if [ $rc -ne 0 ]; then exit $rc; fi ;;
    bench)
      nb=$((nb+1)); out=gpurun_out/bench_$tag; [ $nb -gt 1 ] && out=${out}_$nb
      timeout -k 10 700 python bench.py $arg > $out.json 2> $out.err
      rc=$?; if [ $rc -ne 0 ]; then tail -8 $out.err; exit $rc; fi
      python scripts/bench_brief.py $out.json ;;
    profile) bash scripts/gpu_profile.sh $tag || exit 1 ;;
    trace)      # kernel trace of a short headline run + the launch-by-launch table of one call   -> gpurun_out/timeline_<tag>.md
      ( cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_$tag -- python bench.py --steps 14 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/trace_$tag.log 2>&1 ) || { tail -5 gpurun_out/trace_$tag.log; exit 1; }
      csv=$(find gpurun_out/trace_$tag -name '*kernel_trace.csv' | head -1)
      python scripts/call_timeline.py $csv 10 > gpurun_out/timeline_$tag.md; tail -75 gpurun_out/timeline_$tag.md; rm -rf gpurun_out/trace_$tag ;;
    timeline) timeout -k 10 300 python scripts/call_timeline.py $tag || exit 1 ;;
    ab)
      IFS=, read -r a b reps <<< "$arg"; reps=${reps:-8}
      timeout -k 10 300 python scripts/ab_env.py "$a" "$b" $reps >> gpurun_out/ab_$tag.log 2>&1 || { tail -5 gpurun_out/ab_$tag.log; exit 1; }
      timeout -k 10 300 python scripts/ab_env.py "$b" "$a" $reps >> gpurun_out/ab_$tag.log 2>&1 || { tail -5 gpurun_out/ab_$tag.log; exit 1; }
      tail -12 gpurun_out/ab_$tag.log ;;
    kbench)
      IFS=, read -r kb kargs <<< "$arg"
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $KBENCH_FLAGS -o /tmp/$kb scripts/kbench/$kb.hip 2> gpurun_out/kbench_${kb}_build.err || { tail -5 gpurun_out/kbench_${kb}_build.err; exit 1; }
      timeout -k 10 300 /tmp/$kb $kargs >> gpurun_out/kbench_${kb}_$tag.log 2>&1
      rc=$?; tail -40 gpurun_out/kbench_${kb}_$tag.log; if [ $rc -ne 0 ]; then exit $rc; fi ;;
    sh)         # a shell script of the repo (a harness that builds and runs several variants)  -> gpurun_out/sh_<tag>.log
      IFS=, read -r sc sargs <<< "$arg"
      timeout -k 10 600 bash $sc $sargs >> gpurun_out/sh_$tag.log 2>&1
      rc=$?; tail -40 gpurun_out/sh_$tag.log; fault gpurun_out/sh_$tag.log; if [ $rc -ne 0 ]; then exit $rc; fi ;;
    py)
      IFS=, read -r sc pargs <<< "$arg"
      timeout -k 10 900 python $sc $pargs >> gpurun_out/py_$tag.log 2>&1
      rc=$?; tail -30 gpurun_out/py_$tag.log; if [ $rc -ne 0 ]; then exit $rc; fi ;;
    *) echo "unknown step $step"; exit 2 ;;
  esac
done
