# usage: bash tools/run_pmc_attn.sh <tag>   (separate --pmc passes over tools/pmc_attn.py, on the GPU box)
TAG=${1:-rXX}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $R/gpurun_out/pmca1_$TAG -- python3 $R/tools/pmc_attn.py > $R/gpurun_out/pmca1_$TAG.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA --output-format csv -d $R/gpurun_out/pmca2_$TAG -- python3 $R/tools/pmc_attn.py > $R/gpurun_out/pmca2_$TAG.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmca3_$TAG -- python3 $R/tools/pmc_attn.py > $R/gpurun_out/pmca3_$TAG.log 2>&1
for k in attn_fwd attn_bwd_dq attn_bwd_dkv attn_bwd_fused; do echo "== $k"; python3 $R/tools/pmc_summary.py $k $R/gpurun_out/pmca1_$TAG $R/gpurun_out/pmca2_$TAG $R/gpurun_out/pmca3_$TAG | tr -d '\n {}' | sed 's/"launches":4,//g; s/"mean"://g'; echo; done | tee $R/gpurun_out/attn_pmc_$TAG.txt
tail -2 $R/gpurun_out/pmca2_$TAG.log
