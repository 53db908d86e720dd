#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): kernel-trace stats + the two HBM PMC passes of the default bench,
# written under gpurun_out/$1 (default: prof).  tools/summarize_profile.py turns them into profiles/.
# PMC passes are separate runs with --kernel-trace only, as MI355X_MICROARCH.md prescribes.
# usage: tools/profile_round.sh [tag] [extra bench.py arguments, e.g. --dtype strict]
set -u
TAG=${1:-prof}
shift || true
EXTRA="$*"
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="python $ROOT/bench.py --no-cpu-baseline --strict-steps 0 $EXTRA"
(timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o s -- $B --steps 100 --warmup 8 --steady-steps 0 > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.err") < /dev/null
(timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_f" -o f -- $B --steps 40 --warmup 4 --age-steps 200 --steady-steps 0 > "$OUT/pmc_f.out" 2> "$OUT/pmc_f.err") < /dev/null
(timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_w" -o w -- $B --steps 40 --warmup 4 --age-steps 200 --steady-steps 0 > "$OUT/pmc_w.out" 2> "$OUT/pmc_w.err") < /dev/null
# keep only the small CSVs (the merge-back limit is 64 MiB)
find "$OUT" -name '*_kernel_trace.csv' -size +20M -delete
ls -R "$OUT" | head -40
