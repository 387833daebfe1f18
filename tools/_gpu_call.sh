cd $GRAFT_REPO_ROOT
O=gpurun_out/r3f; mkdir -p $O
for n in 4096 16384 32768; do KV_N=$n timeout 90 python tools/kvariants.py build/variants/r2model_1wave.so build/variants/r2model_2waves.so go2_rl_gym_amd/libgo2sim_hip.so 2>&1 | grep -v amdgpu.ids >> $O/kvariants_occupancy.txt; done
timeout 900 python -m pytest tests -m gpu -q --timeout=200 -p no:cacheprovider > $O/pytest_gpu.log 2>&1
cat $O/kvariants_occupancy.txt; tail -15 $O/pytest_gpu.log
