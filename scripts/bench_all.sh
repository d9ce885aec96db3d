#!/bin/bash
# The single-workload bench lines kept under profiles/bench_rNN_*.json, one after another (through gpurun):
#   bash scripts/bench_all.sh [TAG]      -> gpurun_out/bench_TAG_<name>.json (the compact line) + _full.json (the complete record), stderr of every run in gpurun_out/bench_TAG.log
set -e
TAG=${1:-r05}
LOG=gpurun_out/bench_$TAG.log
: > $LOG
run() { local name=$1; shift; echo "== $name: $*" >> $LOG; python bench.py --no-secondary --full-out gpurun_out/bench_${TAG}_${name}_full.json "$@" > gpurun_out/bench_${TAG}_$name.json 2>> $LOG || { echo "bench $name failed: see $LOG" >&2; tail -5 $LOG >&2; exit 1; }; }
run c1 --workload c1 --spp 16 --steps 5 --warmup 2
run c2 --workload c2
run c2_philox10 --workload c2 --rng philox
run c2_samplerh --workload c2 --rng samplerh
run c3 --workload c3
run c3ref --workload c3ref
run c3ref_samplerh --workload c3ref --rng samplerh
run c4s --workload c4s
run c4f --workload c4f
echo "== default" >> $LOG
python bench.py --full-out gpurun_out/bench_${TAG}_default_full.json > gpurun_out/bench_${TAG}_default.json 2>> $LOG
