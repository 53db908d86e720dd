#!/bin/bash
# Which MIOpen kernels run in the 7-block device training-step test, and does the set differ between runs?  (round 3: the
# per-tensor update error of that test is bimodal across fresh processes.)  Writes gpurun_out/train_algo/run_N.txt =
# the test's PROBE lines + the kernel names with call counts.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/train_algo
for i in $(seq 1 ${1:-10}); do
  rm -rf /tmp/tap; mkdir -p /tmp/tap
  (cd $R && CZ_TRAIN_PROBE=1 rocprofv3 --kernel-trace --stats -d /tmp/tap -o t --output-format csv -- python -m pytest tests/test_train.py -m gpu -q -s -k "on_device_matches and 7-" 2>&1 | grep PROBE | cut -c1-160) > $R/gpurun_out/train_algo/run_$i.txt
  f=$(find /tmp/tap -name "*kernel_stats.csv" | head -1)
  python3 - "$f" >> $R/gpurun_out/train_algo/run_$i.txt <<'P'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(r["Calls"], r["Name"][:150])
P
done
rm -rf /tmp/tap
