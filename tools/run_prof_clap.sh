# usage: bash tools/run_prof_clap.sh <tag>  -- rocprofv3 kernel stats of the CLAP workload (3 passes of 512 clips after 1 warm-up)
TAG=${1:-rXX}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/clap -- python3 $R/bench.py --workload clap --steps 3 --warmup 1 --no-cpu-baseline > $O/clap.log 2>&1
find $O/clap -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/clap_kernel_stats.csv
rm -rf $O/clap
head -30 $O/clap_kernel_stats.csv | cut -c1-220
