# usage: bash tools/run_pmc_gemm.sh <tag>  (three separate --pmc passes over tools/pmc_gemm.py, on the GPU box)
TAG=${1:-rXX}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmcg1_$TAG -- python3 $R/tools/pmc_gemm.py > $R/gpurun_out/pmcg1_$TAG.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmcg2_$TAG -- python3 $R/tools/pmc_gemm.py > $R/gpurun_out/pmcg2_$TAG.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/pmcg3_$TAG -- python3 $R/tools/pmc_gemm.py > $R/gpurun_out/pmcg3_$TAG.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmcg4_$TAG -- python3 $R/tools/pmc_gemm.py > $R/gpurun_out/pmcg4_$TAG.log 2>&1
python3 $R/tools/pmc_summary.py gemm_nt_256 $R/gpurun_out/pmcg1_$TAG $R/gpurun_out/pmcg2_$TAG $R/gpurun_out/pmcg3_$TAG $R/gpurun_out/pmcg4_$TAG | tee $R/gpurun_out/gemm_pmc_summary_$TAG.json
tail -2 $R/gpurun_out/pmcg1_$TAG.log
