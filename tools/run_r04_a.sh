# round 4, first GPU call: the new evidence tests + baseline numbers for this round's performance work
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04a
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_clap_encoder_gpu.py tests/test_bench_contract_gpu.py tests/test_e2e_config4_gpu.py "tests/test_training_loop_gpu.py" tests/test_end_to_end_gpu.py -m gpu -x -q -s > $O/pytest_new.log 2>&1; tail -5 $O/pytest_new.log | cut -c1-300
grep "config\[4\] curation" $O/pytest_new.log | cut -c1-400
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_train.json 2> $O/bench_train.err && cut -c1-250 $O/bench_train.json
timeout 600 python bench.py --steps 3 --warmup 1 --precision fp32 --no-e2e --no-clap --no-cpu-baseline > $O/bench_train_fp32.json 2> $O/bench_train_fp32.err && cut -c1-250 $O/bench_train_fp32.json
timeout 600 python tools/bench_kernels.py attn > $O/bench_kernels_attn.txt 2>&1; tail -12 $O/bench_kernels_attn.txt
