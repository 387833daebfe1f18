cd $GRAFT_REPO_ROOT
O=gpurun_out/r3fb; mkdir -p $O
bash tools/pmc_pass.sh 4096 100 go2_flat > $O/pmc_flat.log 2>&1
bash tools/pmc_pass.sh 4096 100 go2 > $O/pmc_go2.log 2>&1
bash tools/sq_pass.sh 4096 60 go2_flat > $O/sq_flat.log 2>&1
bash tools/sq_pass.sh 4096 60 go2 > $O/sq_go2.log 2>&1
cp gpurun_out/pmc/*.json gpurun_out/pmc/*.csv $O/ 2>/dev/null
timeout 150 python tools/kbench.py 4096 > $O/kbench.txt 2>&1
timeout 150 python tools/kbench.py 4096 rough > $O/kbench_rough.txt 2>&1
timeout 120 python tools/kscale.py 1024 4096 8192 32768 > $O/kscale.txt 2>&1
timeout 100 python tools/policy_bench.py 4096 > $O/policy_bench.txt 2>&1
timeout 100 python tools/gemm_bench.py > $O/gemm_bench.txt 2>&1
timeout 100 python tools/gemm_stamps.py f 512 256 > $O/gemm_stamps.txt 2>&1
timeout 100 python tools/gemm_stamps.py i 512 256 >> $O/gemm_stamps.txt 2>&1
bash tools/gemm_pmc.sh L2i i 512 256 > $O/gemm_pmc_L2i.json 2>/dev/null
timeout 200 python bench.py --steps 40 --warmup 20 --no-cpu-baseline | grep -o '"traffic": [0-9.a-z]*\|"value": [0-9.]*' | head -3
ls $O | tr '\n' ' '
