#!/bin/bash
# Runs ON THE GPU BOX: wall-clock of the product's DEFAULT loop (reference: main.py:1224-1248).
#   1. bench.py --selfplay --start-position with the defaults of main.py --mode train (precision strict, both cache levels) over
#      $1 lock-steps (default 48000)
#   2. main.py --mode train --games 8192 --train_playout 1600 --max_batches 1: one whole self-play batch + its policy updates
O=gpurun_out/${2:-r06b}; mkdir -p $O
STEPS=${1:-48000}
timeout 1500 python bench.py --selfplay --start-position --dtype strict --eval-cache --xcache 24 --steps $STEPS --warmup 16 --age-steps 0 \
    --steady-steps 0 --alt-steps 0 --no-cpu-baseline > $O/sp_strict_xcache24_${STEPS}.json 2> $O/sp_strict_xcache24_${STEPS}.err
python tools/jline.py $O/sp_strict_xcache24_${STEPS}.json | head -12
cp gpurun_out/bench_detail_*.json $O/ 2>/dev/null
if [ "$3" != "nomain" ]; then
  rm -rf /tmp/cz_models /tmp/cz_log; mkdir -p /tmp/cz_run; cd /tmp/cz_run
  ( time timeout 2400 python $GRAFT_REPO_ROOT/main.py --mode train --games 8192 --train_playout 1600 --max_batches 1 ) > $GRAFT_REPO_ROOT/$O/main_train_8192x1600.log 2>&1
  cd $GRAFT_REPO_ROOT
  grep -E "batch_timing|batch i:|real|kl:" $O/main_train_8192x1600.log | tail -8 | cut -c1-900
fi
