R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_x3
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${1:-fp32} -- python3 $R/bench.py --precision ${1:-fp32} --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-clap --no-clock > $O/prof_${1:-fp32}.log 2>&1
f=$(ls $O/prof_${1:-fp32}/*/*kernel_stats.csv | head -1)
head -25 $f | cut -c1-200
rm -f $O/prof_${1:-fp32}/*/*kernel_trace.csv
