# usage: bash tools/run_gpu.sh <tag> [pytest]   (runs on the GPU box through gpurun)
TAG=${1:-rXX}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
if [ "$2" = "pytest" ]; then
  timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_$TAG.log 2>&1; tail -5 gpurun_out/pytest_gpu_$TAG.log
fi
timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/bench_$TAG.log 2>&1; tail -3 gpurun_out/bench_$TAG.log | cut -c1-2500
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-clap > $R/gpurun_out/prof_$TAG.log 2>&1
tail -2 $R/gpurun_out/prof_$TAG.log | cut -c1-300
cat $R/gpurun_out/prof_$TAG/*/*kernel_stats.csv | head -30
