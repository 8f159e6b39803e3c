# usage: bash tools/run_native_point.sh   -- the reference's own operating point (2.56 s @ 24 kHz, use_fx_prob 0.3): bench lines, kernel trace, idle gaps
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/native_point
mkdir -p $O
cd $R
timeout -k 10 600 python bench.py --input-sec 2.56 --sample-rate 24000 --fx-prob 0.3 --steps 40 --warmup 10 --no-clap --no-cpu-baseline > $O/bench_train_native.json 2> $O/bench_train_native.err; cut -c1-300 $O/bench_train_native.json
timeout -k 10 600 python bench.py --input-sec 2.56 --sample-rate 24000 --fx-prob 0.0 --steps 40 --warmup 10 --no-clap --no-cpu-baseline --no-fp32-arm --no-parity-arm > $O/bench_train_native_nofx.json 2> /dev/null; cut -c1-300 $O/bench_train_native_nofx.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --input-sec 2.56 --sample-rate 24000 --fx-prob 0.0 --steps 6 --warmup 3 --no-clap --no-cpu-baseline --no-e2e --no-fp32-arm --no-parity-arm --no-clock > $O/prof.log 2>&1
cd $R
python tools/trace_gaps.py $O/prof > $O/trace_gaps.txt 2>&1; tail -12 $O/trace_gaps.txt
python tools/step_timeline.py $O/prof > $O/step_timeline.txt 2>&1; tail -40 $O/step_timeline.txt
rm -f $O/prof/*/*kernel_trace.csv
