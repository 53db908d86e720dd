#!/bin/bash
# Runs ON THE GPU BOX (gpurun): one parameterised session instead of one script per session (rounds 2-4 kept ~45 of those; they
# are in git history).  usage: tools/gpu_session.sh <tag> <step> [<step> ...]   — output under gpurun_out/<tag>/
#   tests[:k-expr]   pytest -m gpu (optionally -k <expr>)        smoke            __graft_entry__.smoke()
#   bench[:args]     python bench.py <args> (default line + CPU baseline)         bench_nocpu[:args]  the same with --no-cpu-baseline
#   profile[:dtype]  tools/profile_round.sh (rocprofv3 kernel stats + HBM PMC passes of the bench, engine <dtype>)
#   pmcsq:<engine>   tools/pmc_mx.sh (SQ counters of one trunk engine: fp16 | x3 | mx)
#   rules            tools/rules_bench.py (stand-alone K1 / K2 / K3 / hash kernels, 1 M positions) + its rocprofv3 kernel stats
#   mx               tools/mx_check.py --time (the mx engine against its CPU emulation; launch times of the three fp16 engines)
#   mxab[:engine]    the working tree's trunk kernel (engine mx | x3 | fp16, default mx) against every tools/ab/lib_mx_*.so (built here by
#                    tools/experiments/mx_ablate.sh: old=define:X for HEAD's, mx2=mx2 / mx12=mx12 for the round-6 experiments — those
#                    run under CCHESS_MX_KERNEL=2 / 12 —, timing1=define:MX_TIMING for the cycle split), interleaved, 3 rounds
#   latency          the --mode play shape: bench.py --games 1 --search-threads 16 at playout 400 and 1600 (config.ms_per_move)
#   xcache[:steps]   whole-game self-play from the start position, fast engine: no cache / cross-tree table 2^22 / 2^24 (default 160000 lock-steps)
#   loop[:steps]     tools/default_loop.sh: the product's default loop (strict engine + both cache levels), then one whole main.py --mode train batch
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
    mxab) e=${arg:-mx}; T() { ( timeout 200 python tools/mx_check.py --blocks "" --time --engines $e --launches 40 ) 2>&1 | grep "per launch" | tail -1; }
          for r in 1 2 3; do echo "tree: $(T)" | tee -a $O/mxab_$e.txt
            for l in tools/ab/lib_mx_*.so; do k=1; case $l in *timing*) continue;; *mx12*) k=12;; *mx2*) k=2;; esac
              echo "$(basename $l): $(CCHESS_HIP_LIB=$(realpath $l) CCHESS_MX_KERNEL=$k T)" | tee -a $O/mxab_$e.txt; done; done
          for l in tools/ab/lib_mx_timing*.so; do [ -f $l ] || continue; k=1; case $l in *timing12*) k=12;; *timing2*) k=2;; esac
            echo "$(basename $l):" | tee -a $O/mxab_$e.txt; CCHESS_HIP_LIB=$(realpath $l) CCHESS_MX_KERNEL=$k timeout 200 python tools/mx_timing.py 2>&1 | tail -3 | tee -a $O/mxab_$e.txt; done;;
    latency) for po in 400 1600; do n=latency_1x16_p$po
               timeout 600 python bench.py --games 1 --search-threads 16 --playout $po --steps 400 --warmup 16 --age-steps 64 --steady-steps 1000 --alt-steps 0 --no-cpu-baseline > $O/$n.json 2> $O/$n.err
               python -c "import json,sys; d=json.loads([x for x in open('$O/$n.json') if x.startswith('{')][-1]); c=d['config']; print('playout $po: %s %.0f sims/s %.3f ms per lock-step %.2f ms per move (%s launches, %.1f lock-steps per move)' % (d['dtype'], d['value'], d['ms_per_step'], c['ms_per_move'], c['launches_per_lock_step'], c['lock_steps_per_move']))"; done;;
    xcache) st=${arg:-160000}
            for v in "nocache" "xcache22 --eval-cache --xcache 22" "xcache24 --eval-cache --xcache 24"; do set -- $v; n=$1; shift
              timeout 1500 python bench.py --selfplay --start-position --dtype fp16 --steps $st --warmup 16 --age-steps 0 --steady-steps 0 --alt-steps 0 --no-cpu-baseline "$@" > $O/sp_p1600_${st}_$n.json 2> $O/sp_p1600_${st}_$n.err
              python tools/jline.py $O/sp_p1600_${st}_$n.json | head -1; cp gpurun_out/bench_detail_fp16_n1.json $O/detail_$n.json 2>/dev/null; done;;
    loop) tools/default_loop.sh ${arg:-48000} $TAG;;
    *) echo "unknown step $step";;
  esac
done
