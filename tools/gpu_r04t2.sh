#!/bin/bash
for n in 983040 1179648 1048576; do echo "N=$n"; timeout 300 python tools/rules_bench.py $n 2>&1 | grep "mask only"; done
