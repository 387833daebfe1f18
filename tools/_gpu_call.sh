set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3a
timeout 1500 python -m pytest tests/test_gpu_configs.py -q -s -x 2>&1 | tail -40 > gpurun_out/r3a/pytest_configs.log
timeout 600 python tools/kscale.py > gpurun_out/r3a/kscale_plane.txt 2>&1
timeout 600 python tools/kscale.py rough 4096 8192 32768 > gpurun_out/r3a/kscale_rough.txt 2>&1
timeout 300 python tools/termination_check.py "r2 model (2 slots per leg)" > gpurun_out/r3a/termination_before.txt 2>&1
timeout 600 python bench.py --steps 40 --warmup 10 --no-cpu-baseline > gpurun_out/r3a/bench_flat.json 2> gpurun_out/r3a/bench_flat.err
tail -5 gpurun_out/r3a/pytest_configs.log; cat gpurun_out/r3a/kscale_plane.txt gpurun_out/r3a/kscale_rough.txt; tail -4 gpurun_out/r3a/termination_before.txt; tail -c 600 gpurun_out/r3a/bench_flat.json
