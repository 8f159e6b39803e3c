# usage: bash tools/run_rccl_one_rank.sh   -- one box, alternating: plain bench.py vs torchrun --nproc-per-node 1 (RCCL path, f32 and bf16 wire), three rounds;
# then the HF-Trainer path under torchrun with the engine-driven reduction on and off
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/rccl_one_rank
mkdir -p $O
cd $R
COMMON="--steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-clap --no-fp32-arm --no-parity-arm"
for i in 1 2 3; do
  timeout -k 10 300 python bench.py $COMMON > $O/plain_$i.json 2> /dev/null; python -c "import json;d=json.load(open('$O/plain_$i.json'));print('plain $i', round(d['ms_per_step'],3))"
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2951$i bench.py --gpus 1 $COMMON > $O/torchrun_f32_$i.json 2> /dev/null; python -c "import json;d=json.load(open('$O/torchrun_f32_$i.json'));print('torchrun f32 $i', round(d['ms_per_step'],3), d.get('comm'))"
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2952$i bench.py --gpus 1 $COMMON --grad-compress bf16 > $O/torchrun_bf16_$i.json 2> /dev/null; python -c "import json;d=json.load(open('$O/torchrun_bf16_$i.json'));print('torchrun bf16 $i', round(d['ms_per_step'],3), d.get('comm'))"
done
cd /tmp && export TMPDIR=/tmp
# The profiled program itself goes after `--`: no launcher (torch.distributed.run forks and execs its worker under the profiler's preload, which this
# pool forbids).  bench.py builds the nccl process group whenever RANK / WORLD_SIZE are in the environment, so the one-rank RCCL path still runs
# inside the profiled process.
export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29531
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_torchrun -- python3 $R/bench.py --gpus 1 --steps 6 --warmup 3 --no-cpu-baseline --no-e2e --no-clap --no-fp32-arm --no-parity-arm --no-clock > $O/prof_torchrun.log 2>&1
unset RANK LOCAL_RANK WORLD_SIZE MASTER_ADDR MASTER_PORT
cd $R
ls $O/prof_torchrun/*/ 2>/dev/null | head
