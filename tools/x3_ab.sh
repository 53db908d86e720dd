#!/bin/bash
# GPU box: k_trunk_split_c128 of tools/ab/lib_mx_old.so against the working tree's, interleaved
T() { ( timeout 200 python tools/mx_check.py --blocks "" --time --engines x3 --launches 30 ) 2>&1 | grep "per launch" | tail -1; }
for r in 1 2 3; do echo "old: $(CCHESS_HIP_LIB=$(realpath tools/ab/lib_mx_old.so) T)"; echo "new: $(T)"; done
timeout 600 python -m pytest tests/test_net.py -m gpu -q -p no:cacheprovider -k "strict_engine or rows_are_independent" 2>&1 | tail -2
