# usage: bash tools/run_roofline_profile.sh <tag>  -- rocprofv3 --kernel-trace --stats of the roofline kernel alone (tools/pmc_gemm.py:
# 300 launches of the FFN-1 GEMM + GELU epilogue at the training-step shape), so its average duration can be read off directly
TAG=${1:-rXX}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
export ADT_PMC_LAUNCHES=300
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/roofprof_$TAG -- python3 $R/tools/pmc_gemm.py > $R/gpurun_out/roofprof_$TAG.log 2>&1
cat $R/gpurun_out/roofprof_$TAG/*/*kernel_stats.csv | head -5 | cut -c1-200
