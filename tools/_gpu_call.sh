cd $GRAFT_REPO_ROOT
O=gpurun_out/r3l; mkdir -p $O
timeout 100 python tools/kscale.py 4096 > $O/kscale.txt 2>&1; timeout 100 python tools/kscale.py rough 4096 >> $O/kscale.txt 2>&1
timeout 300 python -m pytest tests/test_gpu_parity.py -q -x --timeout=100 -k "one_step_parity or golden_sequence or fallen or trimesh_walls or full_size or strict_ops or pretrained or rough" > $O/pytest_parity.log 2>&1
timeout 150 python tools/kbench.py 4096 rough > $O/kbench_rough.txt 2>&1
timeout 150 python bench.py --task go2 --steps 30 --warmup 20 --no-cpu-baseline > $O/bench_go2.json 2> $O/bench_go2.err
grep -v amdgpu.ids $O/kscale.txt; tail -4 $O/pytest_parity.log; grep -o '"value": [0-9.]*\|"kernel_ms": [0-9.]*\|"collection_only": [0-9.]*' $O/bench_go2.json | tr '\n' ' '; echo; tail -14 $O/kbench_rough.txt
