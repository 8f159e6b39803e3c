# usage: bash tools/run_clap.sh <tag>   (CLAP workload bench + rocprof kernel stats, on the GPU box through gpurun)
TAG=${1:-rXX}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout 900 python bench.py --workload clap --steps 5 --warmup 2 > gpurun_out/bench_clap_$TAG.log 2>&1; tail -3 gpurun_out/bench_clap_$TAG.log | cut -c1-2500
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_clap_$TAG -- python3 $R/bench.py --workload clap --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_clap_$TAG.log 2>&1
tail -2 $R/gpurun_out/prof_clap_$TAG.log | cut -c1-300
cat $R/gpurun_out/prof_clap_$TAG/*/*kernel_stats.csv | head -30
