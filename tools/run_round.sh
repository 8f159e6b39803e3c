# usage: bash tools/run_round.sh <tag>   -- full GPU suite, smoke, the three bench workloads, kernel micro-benchmarks and their rocprofv3 kernel stats
TAG=${1:-rXX}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/round_$TAG
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log | cut -c1-200
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 900 python bench.py --steps 20 --warmup 5 --roofline-loop > $O/bench_train.json 2> $O/bench_train.err; cut -c1-300 $O/bench_train.json
timeout 900 python bench.py --steps 20 --warmup 5 --dropout 0.0 --no-cpu-baseline > $O/bench_train_nodropout.json 2> /dev/null; cut -c1-200 $O/bench_train_nodropout.json
timeout 900 python bench.py --workload logmel --steps 30 --warmup 5 > $O/bench_logmel.json 2> /dev/null; cut -c1-300 $O/bench_logmel.json
timeout 900 python bench.py --workload clap --steps 5 --warmup 2 > $O/bench_clap.json 2> /dev/null; cut -c1-300 $O/bench_clap.json
# the N > 1 code path on real RCCL at world size 1 (torchrun, one rank): the line carries the `comm` object
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-clap > $O/bench_train_torchrun1.json 2> /dev/null; cut -c1-200 $O/bench_train_torchrun1.json
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-clap --grad-compress bf16 > $O/bench_train_torchrun1_bf16comm.json 2> /dev/null; cut -c1-200 $O/bench_train_torchrun1_bf16comm.json
timeout 600 python tools/bench_hf_trainer.py > $O/bench_hf_trainer.txt 2>&1; tail -3 $O/bench_hf_trainer.txt
timeout 600 python tools/bench_kernels.py > $O/bench_kernels.txt 2>&1
timeout 600 python tools/exp_attn_bwd.py > $O/attn_bwd_paths.txt 2>&1; tail -6 $O/attn_bwd_paths.txt
timeout 600 python tools/exp_attn_fwd.py > $O/attn_fwd.txt 2>&1; tail -6 $O/attn_fwd.txt
timeout 600 python bench.py --steps 5 --warmup 2 --precision fp32 --no-e2e --no-clap --no-cpu-baseline > $O/bench_train_fp32.json 2> /dev/null; cut -c1-200 $O/bench_train_fp32.json
ADT_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-clap > $O/bench_train_2ranks_shared_gpu_debug.json 2> $O/bench_2ranks.err; cut -c1-200 $O/bench_train_2ranks_shared_gpu_debug.json
timeout 600 python tools/e2e.py --shots 4000 --chunks 2048 --check-resume > $O/e2e_config4_scaled.json 2> $O/e2e.err; cut -c1-400 $O/e2e_config4_scaled.json
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-e2e --no-clap > $O/prof_train.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_logmel -- python3 $R/bench.py --workload logmel --steps 5 --warmup 2 --no-cpu-baseline > $O/prof_logmel.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_clap -- python3 $R/bench.py --workload clap --steps 3 --warmup 1 --no-cpu-baseline > $O/prof_clap.log 2>&1
export ADT_PMC_LAUNCHES=300
export ADT_PMC_ALSO_NO_DROPOUT=1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_roofline -- python3 $R/tools/pmc_gemm.py > $O/prof_roofline.log 2>&1
unset ADT_PMC_LAUNCHES ADT_PMC_ALSO_NO_DROPOUT
rm -f $O/prof_*/*/*kernel_trace.csv          # keep the merged output small: the stats tables are what is committed
ls $O
