#!/bin/bash
# experiment: evaluation-cache hits per tree and select launch (CZ_EC_BUDGET = 2 / 3 / 4 (product) / 6) on the self-play line with
# both cache levels — the select launch's tail against the simulations per net row
O=gpurun_out/r04w; mkdir -p $O
B="timeout 900 python bench.py --no-cpu-baseline --strict-steps 0 --selfplay --start-position --age-steps 0 --steady-steps 0 --warmup 16 --playout 1600 --steps 9600 --eval-cache --xcache 22"
for lib in tools/bin/libcchess_hip_ec2.so tools/bin/libcchess_hip_ec3.so cchess_zero_amd/libcchess_hip.so tools/bin/libcchess_hip_ec6.so; do
  n=$(basename $lib .so)
  CCHESS_HIP_LIB=$(pwd)/$lib $B > $O/sp_$n.json 2> $O/sp_$n.err
  python -c "
import json
d=json.loads([l for l in open('$O/sp_$n.json') if l.startswith('{')][-1]); t=d['roofline_tree']; c=d['config']; print('$n', '%.0f sims/s' % d['value'], 'select %.1f us expand %.1f us' % (t['us_select'], t['us_expand_backup']), 'ms/step %.3f' % d['ms_per_step'], 'sims/row %.3f' % c['simulations_per_net_row'])"
done | tee $O/ec_budget.txt
