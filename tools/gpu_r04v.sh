#!/bin/bash
# round 4, GPU session v: the other lines of the results table on the final kernels — device self-play from the start position at
# playout 1600 (no cache / per-tree cache / + cross-tree table), configs[1] (4096 games, playout 400), bf16
O=gpurun_out/r04v; mkdir -p $O
B="timeout 900 python bench.py --no-cpu-baseline --strict-steps 0 --selfplay --start-position --age-steps 0 --steady-steps 0 --warmup 16 --playout 1600 --steps 12800"
$B --eval-cache --xcache 22 > $O/sp_p1600_xcache22.json 2> $O/e2
$B --eval-cache > $O/sp_p1600_cache.json 2> $O/e3
$B > $O/sp_p1600_nocache.json 2> $O/e4
for f in $O/sp_*.json; do python -c "
import json
d=json.loads([l for l in open('$f') if l.startswith('{')][-1]); t=d['roofline_tree']; c=d['config']; print('$f'.split('/')[-1], '%.0f sims/s' % d['value'], 'select %.1f us expand %.1f us' % (t['us_select'], t['us_expand_backup']), 'ms/step %.3f' % d['ms_per_step'], 'sims/row %.3f' % c['simulations_per_net_row'], 'xcache', c.get('xcache'), 'records', c['selfplay']['records'])"; done | tee $O/selfplay.txt
timeout 600 python bench.py --no-cpu-baseline --games 4096 --playout 400 --steps 800 --steady-steps 1200 > $O/bench_configs1.json 2> $O/e5; python tools/jline.py $O/bench_configs1.json | head -2 | tee $O/configs1.txt
timeout 600 python bench.py --no-cpu-baseline --dtype bf16 --strict-steps 0 > $O/bench_bf16.json 2> $O/e6; python tools/jline.py $O/bench_bf16.json | head -2 | tee $O/bf16.txt
timeout 600 python bench.py --no-cpu-baseline --eval-cache --strict-steps 0 > $O/bench_evalcache.json 2> $O/e7; python tools/jline.py $O/bench_evalcache.json | head -2 | tee $O/evalcache.txt
