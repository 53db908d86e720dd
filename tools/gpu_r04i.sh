#!/bin/bash
# round 4, GPU session i: the branch-free mask kernel (tests + rules bench), the zero-work elasticity experiment on the trunk
O=gpurun_out/r04i; mkdir -p $O
( timeout 900 python -m pytest tests/test_hip_rules.py -m gpu -q -p no:cacheprovider > $O/pytest_rules.log 2>&1; echo "pytest rc=$?" >> $O/pytest_rules.log ); tail -3 $O/pytest_rules.log
( timeout 300 python tools/rules_bench.py > $O/rules_bench.log 2>&1 ); cat $O/rules_bench.log
for rep in 1 2; do for sk in 0 3 9 0; do ./tools/experiments/bin/tower_skip$sk 8192 7 20 | tail -1; done; done > $O/tower_skip.log 2>&1; cat $O/tower_skip.log
./tools/experiments/bin/variants_check 8192 7 1 10 1 | tail -4 > $O/variants_check_sk.log 2>&1; cat $O/variants_check_sk.log
