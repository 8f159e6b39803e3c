# usage: bash tools/run_prof_train.sh <tag>  -- rocprofv3 kernel stats of the default training bench (5 steps after 2 warm-up)
TAG=${1:-rXX}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-e2e --no-clap --no-fp32-arm --no-parity-arm > $O/train.log 2>&1
find $O/train -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/train_step_kernel_stats.csv
rm -rf $O/train
head -40 $O/train_step_kernel_stats.csv | cut -c1-200
