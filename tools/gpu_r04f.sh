#!/bin/bash
# round 4, GPU session f: the cross-tree cache — parity tests, then self-play from the start position at playout 1600 / 400 with
# no cache, the per-tree cache, and both levels (same box, back to back)
O=gpurun_out/r04f; mkdir -p $O
( timeout 1200 python -m pytest tests/test_hip_search.py tests/test_selfplay_device.py tests/test_width.py -m gpu -q -s -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log )
grep -n "cross-tree\|passed\|failed\|rc=\|Error" $O/pytest_gpu.log | cut -c1-330 | tail -12
B="timeout 900 python bench.py --no-cpu-baseline --strict-steps 0 --selfplay --start-position --age-steps 0 --steady-steps 0 --warmup 16"
for P in 1600 400; do
  S=$((P * 8))
  $B --playout $P --steps $S > $O/sp_p${P}_nocache.json 2> $O/sp_p${P}_nocache.err
  $B --playout $P --steps $S --eval-cache > $O/sp_p${P}_cache.json 2> $O/sp_p${P}_cache.err
  $B --playout $P --steps $S --eval-cache --xcache 20 > $O/sp_p${P}_xcache.json 2> $O/sp_p${P}_xcache.err
done
for f in $O/sp_*.json; do echo "== $f"; python tools/jline.py $f 2>&1 | sed -n '1p;/sims per net row/p'; python -c "
import json,sys
d=json.loads([l for l in open('$f') if l.startswith('{')][-1]); c=d['config']; print('   eval_cache', c.get('eval_cache'), 'xcache', c.get('xcache'), 'selfplay games', c['selfplay']['games_finished'], 'records', c['selfplay']['records'])"; done
