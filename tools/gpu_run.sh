#!/bin/bash
# One parameterised script for every gpurun call (replaces the per-experiment scripts of earlier rounds).  Everything lands in gpurun_out/.
#   tools/gpu_run.sh <step> [<step> ...]     steps run in the order given; a failing step does not stop the later ones
# steps:
#   tests[:<pytest -k expression>]   pytest -m gpu (whole suite, or the subset that matches)
#   smoke                            __graft_entry__.smoke()
#   bench[:<tag>[:<extra args>]]     python bench.py with the library of build tag <tag> ("" = the product's); extra args verbatim
#   quick:<tag>:<ENV=..,ENV=..>[:<args>]   bench.py --quick (default --steps 20 --warmup 5) with environment variables (ablation switches), prints the kernel table
#   trace[:<tag>]                    tools/trace_iteration.py on the kernel trace of the preceding stats step
#   stats[:<tag>[:<bench args>]]     rocprofv3 --kernel-trace --stats of the bench command -> gpurun_out/prof_<tag>/
#   pmc:<counters>[:<tag>[:<bench args>]]   one rocprofv3 --pmc pass (counters space-separated with '+'), kernel-trace only
#   keep:<name>                      copy the kernel-stats csv of the preceding stats step to gpurun_out/<name>_kernel_stats.csv
#   py:<script and args>             python <script> (tools/*.py probes)
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
lib_of() { [ -n "$1" ] && echo "$PWD/loner_amd/_lib/libloner_hip_$1.so"; }
n_step=0
for step in "$@"; do
  n_step=$((n_step + 1))
  kind=${step%%:*}; rest=${step#*:}; [ "$rest" == "$step" ] && rest=""
  echo "===== $step"
  case $kind in
    tests)
      if [ -n "$rest" ]; then timeout 1500 python -m pytest tests -m gpu -q -x --no-header -p no:cacheprovider -k "$rest" > gpurun_out/pytest_gpu_subset.log 2>&1; tail -15 gpurun_out/pytest_gpu_subset.log
      else timeout 1800 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -rf > gpurun_out/pytest_gpu.log 2>&1; tail -25 gpurun_out/pytest_gpu.log; fi ;;
    smoke) timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; tail -3 gpurun_out/smoke.log ;;
    bench)
      tag=${rest%%:*}; args=${rest#*:}; [ "$args" == "$rest" ] && args=""
      out=gpurun_out/bench_${tag:-product}_step${n_step}       # (one log per step: two bench steps of a call do not overwrite each other)
      LNR_LIB_PATH=$(lib_of "$tag") timeout 900 python bench.py $args > $out.log 2> $out.err
      echo "exit $? -> $out.log"; tail -1 $out.log | python tools/bench_kernels.py --all ;;
    quick)
      tag=${rest%%:*}; r2=${rest#*:}; [ "$r2" == "$rest" ] && r2=""
      envs=${r2%%:*}; args=${r2#*:}; [ "$args" == "$r2" ] && args="--steps 20 --warmup 5"
      env $(echo $envs | tr ',' ' ') LNR_LIB_PATH=$(lib_of "$tag") timeout 300 python bench.py --quick $args 2>/dev/null | python tools/bench_kernels.py --all | grep -E "ms_per_step|kernel " ;;
    trace)   # timeline of one iteration from the kernel trace of a stats step: trace:<tag>
      f=$(find gpurun_out/prof_${rest:-product} -name "*kernel_trace.csv" | head -1); python tools/trace_iteration.py $f | tee gpurun_out/trace_${rest:-product}.txt | tail -40 ;;
    stats)
      tag=${rest%%:*}; args=${rest#*:}; [ "$args" == "$rest" ] && args="--steps 30 --warmup 10 --no-cpu-baseline"
      rm -rf gpurun_out/prof_${tag:-product}
      LNR_LIB_PATH=$(lib_of "$tag") timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag:-product} -o run -- python bench.py $args > gpurun_out/prof_${tag:-product}.log 2>&1
      echo "exit $?"; find gpurun_out/prof_${tag:-product} -name "*kernel_stats*" | head -3 ;;
    pmc)
      ctr=${rest%%:*}; r2=${rest#*:}; [ "$r2" == "$rest" ] && r2=""
      tag=${r2%%:*}; args=${r2#*:}; [ "$args" == "$r2" ] && args="--quick --steps 8 --warmup 4"
      out=gpurun_out/pmc_$(echo $ctr | tr '+' '_' | cut -c1-60)_${tag:-product}; rm -rf $out
      LNR_LIB_PATH=$(lib_of "$tag") timeout 600 rocprofv3 --pmc $(echo $ctr | tr '+' ' ') --kernel-trace --output-format csv -d $out -o run -- python bench.py $args > $out.log 2>&1
      echo "exit $?"; python tools/pmc_mean.py $out encode_ table_grad mlp_ | head -60 ;;
    keep)    # keep:<name>: copy the kernel-stats summary of the preceding stats step to gpurun_out/<name>_kernel_stats.csv
      cp gpurun_out/prof_product/run_kernel_stats.csv gpurun_out/${rest}_kernel_stats.csv && head -12 gpurun_out/${rest}_kernel_stats.csv ;;
    py) timeout 900 python $rest 2>&1 | tail -40 ;;
    *) echo "unknown step $step" ;;
  esac
done
