cd $GRAFT_REPO_ROOT
bash tools/gemm_mem_pmc.sh L2f f 512 256 2>&1 | tail -8
