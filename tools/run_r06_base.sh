# usage: bash tools/run_r06_base.sh <tag>  -- GPU suite + default bench line + train-step kernel stats (baseline of a build)
TAG=${1:-base}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_$TAG
mkdir -p $O
cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $O/bench_train.json 2> $O/bench_train.err && cut -c1-400 $O/bench_train.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/prof_train -o train -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-clap --no-fp32-arm --no-parity-arm --no-e2e > $O/prof_train.log 2>&1
find $O/prof_train -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/train_step_kernel_stats.csv
find $O/prof_train -name "*.db" -delete; find $O/prof_train -name "*trace.csv" -size +8M -delete
ls $O
