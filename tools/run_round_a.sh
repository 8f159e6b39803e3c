# usage: bash tools/run_round_a.sh <tag>   -- first half of a round's evidence: smoke, the bench workloads (incl. the reference's own operating point), RCCL at world size 1,
# the HF-Trainer path, kernel micro-benchmarks, attention A/B tools, the fp32 arm, the two-rank debug run
TAG=${1:-rXX}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/round_$TAG
mkdir -p $O
cd $R
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout -k 10 900 python bench.py --steps 20 --warmup 5 --roofline-loop > $O/bench_train.json 2> $O/bench_train.err; cut -c1-300 $O/bench_train.json
timeout -k 10 900 python bench.py --steps 20 --warmup 5 --dropout 0.0 --no-cpu-baseline --no-clap --no-fp32-arm --no-parity-arm > $O/bench_train_nodropout.json 2> /dev/null; cut -c1-200 $O/bench_train_nodropout.json
ADT_ATTN_NO_BITS=1 timeout -k 10 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-clap --no-fp32-arm --no-parity-arm --no-e2e > $O/bench_train_nobits.json 2> /dev/null; cut -c1-200 $O/bench_train_nobits.json
timeout -k 10 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-clap --no-fp32-arm --no-parity-arm --no-e2e > $O/bench_train_bits.json 2> /dev/null; cut -c1-200 $O/bench_train_bits.json
timeout -k 10 900 python bench.py --input-sec 2.56 --sample-rate 24000 --fx-prob 0.3 --steps 40 --warmup 10 --no-clap --no-cpu-baseline > $O/bench_train_native.json 2> /dev/null; cut -c1-200 $O/bench_train_native.json
timeout -k 10 900 python bench.py --workload logmel --steps 30 --warmup 5 > $O/bench_logmel.json 2> /dev/null; cut -c1-300 $O/bench_logmel.json
timeout -k 10 900 python bench.py --workload clap --steps 5 --warmup 2 > $O/bench_clap.json 2> /dev/null; cut -c1-300 $O/bench_clap.json
timeout -k 10 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-clap --no-fp32-arm --no-parity-arm 2> /dev/null | grep "^{" > $O/bench_train_torchrun1.json; cut -c1-200 $O/bench_train_torchrun1.json
timeout -k 10 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-clap --no-fp32-arm --no-parity-arm --grad-compress bf16 2> /dev/null | grep "^{" > $O/bench_train_torchrun1_bf16comm.json; cut -c1-200 $O/bench_train_torchrun1_bf16comm.json
timeout -k 10 600 python tools/bench_hf_trainer.py > $O/bench_hf_trainer.txt 2>&1; tail -3 $O/bench_hf_trainer.txt
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29513 tools/bench_hf_trainer.py > $O/bench_hf_trainer_torchrun1.txt 2>&1; tail -2 $O/bench_hf_trainer_torchrun1.txt
timeout -k 10 600 python tools/bench_kernels.py > $O/bench_kernels.txt 2>&1
timeout -k 10 600 python tools/exp_attn_bwd.py > $O/attn_bwd_paths.txt 2>&1; tail -6 $O/attn_bwd_paths.txt | cut -c1-300
timeout -k 10 600 python tools/exp_attn_fwd.py > $O/attn_fwd.txt 2>&1; tail -6 $O/attn_fwd.txt
ADT_BENCH_SHARE_GPU=1 timeout -k 10 600 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-clap > $O/bench_train_2ranks_shared_gpu_debug.json 2> $O/bench_2ranks.err; cut -c1-200 $O/bench_train_2ranks_shared_gpu_debug.json
ls $O
