# usage: bash tools/run_pmc_logmel.sh <tag>  (separate --pmc passes over the log-mel workload of bench.py, on the GPU box)
TAG=${1:-rXX}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/bench.py --workload logmel --steps 3 --warmup 1 --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $R/gpurun_out/pmcl1_$TAG -- $CMD > $R/gpurun_out/pmcl1_$TAG.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $R/gpurun_out/pmcl2_$TAG -- $CMD > $R/gpurun_out/pmcl2_$TAG.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmcl3_$TAG -- $CMD > $R/gpurun_out/pmcl3_$TAG.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmcl4_$TAG -- $CMD > $R/gpurun_out/pmcl4_$TAG.log 2>&1
python3 $R/tools/pmc_summary.py logmel_kernel $R/gpurun_out/pmcl1_$TAG $R/gpurun_out/pmcl2_$TAG $R/gpurun_out/pmcl3_$TAG $R/gpurun_out/pmcl4_$TAG | tee $R/gpurun_out/logmel_pmc_summary_$TAG.json
rm -rf $R/gpurun_out/pmcl?_$TAG
