#!/bin/bash
# cross-tree cache statistics per tree (no same-address atomics in the probe): cache tests + the self-play line with both levels
O=gpurun_out/r04y; mkdir -p $O
( timeout 900 python -m pytest tests/test_hip_search.py tests/test_selfplay_device.py -m gpu -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log ); tail -3 $O/pytest_gpu.log
B="timeout 900 python bench.py --no-cpu-baseline --strict-steps 0 --selfplay --start-position --age-steps 0 --steady-steps 0 --warmup 16 --playout 1600 --steps 12800"
$B --eval-cache --xcache 22 > $O/sp_p1600_xcache22.json 2> $O/e2
$B --eval-cache > $O/sp_p1600_cache.json 2> $O/e3
$B > $O/sp_p1600_nocache.json 2> $O/e4
for f in $O/sp_*.json; do python -c "
import json
d=json.loads([l for l in open('$f') if l.startswith('{')][-1]); t=d['roofline_tree']; c=d['config']; print('$f'.split('/')[-1], '%.0f sims/s' % d['value'], 'select %.1f us expand %.1f us' % (t['us_select'], t['us_expand_backup']), 'ms/step %.3f' % d['ms_per_step'], 'sims/row %.3f' % c['simulations_per_net_row'], 'xcache', c.get('xcache'), 'records', c['selfplay']['records'])"; done | tee $O/selfplay.txt
