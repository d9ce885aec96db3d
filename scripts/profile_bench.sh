#!/bin/bash
# Profiles the default bench command on the GPU box; summaries land in gpurun_out/prof_<tag>/ and are
# copied (by hand) into profiles/.  Usage (through gpurun): scripts/profile_bench.sh r01 [bench args]
set -e
TAG=${1:-r01}; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 bench.py --no-cpu-baseline --no-secondary "$@" > $OUT/bench_kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py --no-cpu-baseline --no-secondary --steps 1 --warmup 0 "$@" > $OUT/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py --no-cpu-baseline --no-secondary --steps 1 --warmup 0 "$@" > $OUT/bench_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/sq -- python3 bench.py --no-cpu-baseline --no-secondary --steps 1 --warmup 0 "$@" > $OUT/bench_sq.log 2>&1
# round 4: what the waves do besides vector arithmetic -- instruction mix, scalar / any-instruction activity, vector-memory level
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM --output-format csv -d $OUT/sq2 -- python3 bench.py --no-cpu-baseline --no-secondary --steps 1 --warmup 0 "$@" > $OUT/bench_sq2.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $OUT/tcc -- python3 bench.py --no-cpu-baseline --no-secondary --steps 1 --warmup 0 "$@" > $OUT/bench_tcc.log 2>&1
python3 scripts/profile_digest.py $OUT > $OUT/digest.json
cat $OUT/digest.json
