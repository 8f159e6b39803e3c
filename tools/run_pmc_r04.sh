cd $GRAFT_REPO_ROOT
bash tools/run_pmc_gemm.sh r04 > gpurun_out/pmc_gemm_r04.log 2>&1; tail -3 gpurun_out/pmc_gemm_r04.log | cut -c1-400
ADT_ATTN_BWD=fused bash tools/run_pmc_attn.sh r04fused > gpurun_out/pmc_attn_r04fused.log 2>&1; grep -A1 "== attn_bwd_fused" gpurun_out/pmc_attn_r04fused.log | cut -c1-900
ADT_ATTN_BWD=fused ADT_ATTN_DROP=0 bash tools/run_pmc_attn.sh r04fused_nodrop > gpurun_out/pmc_attn_r04fused_nodrop.log 2>&1; grep -A1 "== attn_bwd_fused" gpurun_out/pmc_attn_r04fused_nodrop.log | cut -c1-900
bash tools/run_pmc_attn.sh r04 > gpurun_out/pmc_attn_r04.log 2>&1; grep -A1 "== attn" gpurun_out/pmc_attn_r04.log | cut -c1-700
