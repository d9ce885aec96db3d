set -e
B="python bench.py --no-secondary"
$B --workload c1 --spp 16 --steps 5 --warmup 2 > gpurun_out/bench_r03_c1.json 2>/dev/null
$B --workload c2 > gpurun_out/bench_r03_c2.json 2>/dev/null
$B --workload c2 --rng philox > gpurun_out/bench_r03_c2_philox10.json 2>/dev/null
$B --workload c2 --rng samplerh > gpurun_out/bench_r03_c2_samplerh.json 2>/dev/null
$B --workload c3 > gpurun_out/bench_r03_c3.json 2>/dev/null
$B --workload c3ref > gpurun_out/bench_r03_c3ref.json 2>/dev/null
$B --workload c4s > gpurun_out/bench_r03_c4s.json 2>/dev/null
$B --workload c4f > gpurun_out/bench_r03_c4f.json 2>/dev/null
$B --workload c3ref --rng samplerh > gpurun_out/bench_r03_c3ref_samplerh.json 2>/dev/null
python bench.py > gpurun_out/bench_r03_final.json 2>/dev/null
