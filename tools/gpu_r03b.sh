#!/bin/bash
# round 3, call b: full GPU test suite, bench A/B of the advance driver, multi-rank plumbing on one GPU
set -u
O=gpurun_out/r03b; mkdir -p $O
python -m pytest tests -m gpu -x -q -s > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
B="python bench.py --no-cpu-baseline"
$B > $O/bench_dev_advance.json 2>$O/err1; python tools/jline.py $O/bench_dev_advance.json | head -3
$B --torch-advance > $O/bench_torch_advance.json 2>/dev/null; python tools/jline.py $O/bench_torch_advance.json | head -2
$B > $O/bench_dev_advance2.json 2>/dev/null; python tools/jline.py $O/bench_dev_advance2.json | head -2
$B --eval-cache > $O/bench_cache.json 2>/dev/null; python tools/jline.py $O/bench_cache.json | head -2; python tools/jline.py $O/bench_cache.json | grep "sims per"
# two ranks on one GPU over gloo (plumbing): search loop, and self-play with the fixed-capacity exchange in the timed region
timeout 600 python bench.py --gpus 2 --all-on-device0 --dist-backend gloo --games 1024 --playout 40 --steps 400 --age-steps 80 --steady-steps 0 --cpu-seconds 3 > $O/bench_2rank_gloo.json 2>$O/err2; echo "2rank rc $?"; python tools/jline.py $O/bench_2rank_gloo.json | tail -4
timeout 600 python bench.py --gpus 2 --all-on-device0 --dist-backend gloo --games 1024 --playout 40 --steps 1600 --age-steps 80 --steady-steps 0 --selfplay --timed-gather --no-cpu-baseline > $O/bench_2rank_gloo_selfplay.json 2>$O/err3; echo "2rank selfplay rc $?"; python tools/jline.py $O/bench_2rank_gloo_selfplay.json | tail -4
# RCCL with a world of one (API check of the collectives on device tensors)
timeout 600 python bench.py --force-dist --games 1024 --playout 40 --steps 1600 --age-steps 80 --steady-steps 0 --selfplay --timed-gather --no-cpu-baseline > $O/bench_rccl_world1.json 2>$O/err4; echo "rccl rc $?"; python tools/jline.py $O/bench_rccl_world1.json | tail -4
tail -3 $O/err2 $O/err3 $O/err4
