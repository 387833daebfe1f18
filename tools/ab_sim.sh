# tools/ab_sim.sh v0 v1 ... — bench.py with build/variants/libgo2sim_<v>.so in place of the product library, each variant twice, interleaved (one box, one call)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
cp go2_rl_gym_amd/libgo2sim_hip.so /tmp/libgo2sim_keep.so
for rep in 1 2; do for v in "$@"; do
  cp build/variants/libgo2sim_$v.so go2_rl_gym_amd/libgo2sim_hip.so
  echo "$v: $(python bench.py --steps ${STEPS:-30} --warmup 10 --no-cpu-baseline ${BENCH_ARGS} 2>/dev/null | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"collection_only": [0-9.]*\|"achieved": [0-9.]*' | tr '\n' ' ')"
done; done
cp /tmp/libgo2sim_keep.so go2_rl_gym_amd/libgo2sim_hip.so
