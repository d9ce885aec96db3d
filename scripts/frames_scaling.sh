for f in 16 32 64 128 256 512; do VP_PERF_RNG=2 timeout -k 10 120 python3 scripts/perf_workloads.py c2 $f 2 || exit 1; done
