#!/bin/bash
# Runs ON THE GPU BOX (gpurun): one parameterised session instead of one script per session (rounds 2-4 kept ~45 of those; they
# are in git history).  usage: tools/gpu_session.sh <tag> <step> [<step> ...]   — output under gpurun_out/<tag>/
#   tests[:k-expr]   pytest -m gpu (optionally -k <expr>)        smoke            __graft_entry__.smoke()
#   bench[:args]     python bench.py <args> (default line + CPU baseline)         bench_nocpu[:args]  the same with --no-cpu-baseline
#   profile[:dtype]  tools/profile_round.sh (rocprofv3 kernel stats + HBM PMC passes of the bench, engine <dtype>)
#   pmcsq:<engine>   tools/pmc_mx.sh (SQ counters of one trunk engine: fp16 | x3 | mx)
#   rules            tools/rules_bench.py (stand-alone K1 / K2 / K3 / hash kernels, 1 M positions) + its rocprofv3 kernel stats
#   mx               tools/mx_check.py --time (the mx engine against its CPU emulation; launch times of the three fp16 engines)
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
for step in "$@"; do
  name=${step%%:*}; arg=""; [ "$step" != "$name" ] && arg=${step#*:}
  case $name in
    tests) ( timeout 1700 python -m pytest tests -m gpu -q -x -p no:cacheprovider ${arg:+-k "$arg"} > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log ); tail -5 $O/pytest_gpu.log;;
    smoke) ( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log ); tail -3 $O/smoke.log;;
    bench) n=bench_$(echo "${arg:-default}" | tr -c 'A-Za-z0-9\n' '_'); timeout 900 python bench.py $arg > $O/$n.json 2> $O/$n.err; python tools/jline.py $O/$n.json | tee $O/$n.txt; cp gpurun_out/bench_detail_*.json $O/ 2>/dev/null;;
    bench_nocpu) n=bench_$(echo "${arg:-default}" | tr -c 'A-Za-z0-9\n' '_'); timeout 900 python bench.py --no-cpu-baseline $arg > $O/$n.json 2> $O/$n.err; python tools/jline.py $O/$n.json | tee $O/$n.txt; cp gpurun_out/bench_detail_*.json $O/ 2>/dev/null;;
    profile) d=${arg:-fp16}; tools/profile_round.sh $TAG/prof_$d ${arg:+--dtype $arg} > $O/profile_$d.log 2>&1
             # summarised HERE (the raw traces are tens of MB and gpurun merges at most 64 MiB back): profiles_out/ holds what goes to profiles/
             case $d in mx6) k=k_trunk_mx; sfx=_mx;; fp16x2) k=k_trunk_split; sfx=_strict;; *) k=k_tower8; sfx="";; esac
             PROFILES_OUT=$O/profiles_out python tools/summarize_profile.py gpurun_out/$TAG/prof_$d ${ROUND:-r06}_$d $k $sfx > $O/summary_$d.log 2>&1; tail -4 $O/summary_$d.log
             rm -rf gpurun_out/$TAG/prof_$d/stats gpurun_out/$TAG/prof_$d/pmc_f gpurun_out/$TAG/prof_$d/pmc_w;;
    pmcsq) tools/pmc_mx.sh $O/pmcsq_$arg $arg 2>&1 | tail -8 | tee $O/pmcsq_$arg.txt;;
    rules) timeout 600 python tools/rules_bench.py > $O/rules_bench.log 2>&1; tail -12 $O/rules_bench.log; tools/rules_profile.sh $O/rules_prof; rm -rf $O/rules_prof/stats;;
    mx) timeout 600 python tools/mx_check.py --time > $O/mx_check.txt 2>&1; tail -20 $O/mx_check.txt | cut -c1-260;;
    *) echo "unknown step $step";;
  esac
done
