# usage: bash tools/run_round.sh <tag>   -- full GPU suite, smoke, the three bench workloads and their rocprofv3 kernel stats
TAG=${1:-rXX}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/round_$TAG
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_train.log 2>&1; tail -1 $O/bench_train.log | cut -c1-300
timeout 900 python bench.py --steps 20 --warmup 5 --dropout 0.0 --no-cpu-baseline > $O/bench_train_nodrop.log 2>&1; tail -1 $O/bench_train_nodrop.log | cut -c1-200
timeout 900 python bench.py --workload logmel --steps 30 --warmup 5 > $O/bench_logmel.log 2>&1; tail -1 $O/bench_logmel.log | cut -c1-300
timeout 900 python bench.py --workload clap --steps 5 --warmup 2 > $O/bench_clap.log 2>&1; tail -1 $O/bench_clap.log | cut -c1-300
timeout 600 python tools/bench_kernels.py > $O/bench_kernels.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/prof_train.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_logmel -- python3 $R/bench.py --workload logmel --steps 5 --warmup 2 --no-cpu-baseline > $O/prof_logmel.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_clap -- python3 $R/bench.py --workload clap --steps 3 --warmup 1 --no-cpu-baseline > $O/prof_clap.log 2>&1
rm -f $O/prof_*/*/*kernel_trace.csv          # keep the merged output small: the stats tables are what is committed
ls $O $O/prof_train/* | head -40
