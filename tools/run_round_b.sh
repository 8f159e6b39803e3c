# usage: bash tools/run_round_b.sh <tag>   -- second half: rocprofv3 kernel stats of the three workloads and of the roofline kernel alone, the PMC passes
TAG=${1:-rXX}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/round_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-e2e --no-clap --no-fp32-arm --no-parity-arm > $O/prof_train.log 2>&1
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_logmel -- python3 $R/bench.py --workload logmel --steps 5 --warmup 2 --no-cpu-baseline > $O/prof_logmel.log 2>&1
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_clap -- python3 $R/bench.py --workload clap --steps 3 --warmup 1 --no-cpu-baseline > $O/prof_clap.log 2>&1
export ADT_PMC_LAUNCHES=300
export ADT_PMC_ALSO_NO_DROPOUT=1
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_roofline -- python3 $R/tools/pmc_gemm.py > $O/prof_roofline.log 2>&1
unset ADT_PMC_LAUNCHES ADT_PMC_ALSO_NO_DROPOUT
cd $R
python tools/step_timeline.py $O/prof_train > $O/step_timeline.txt 2>&1
python tools/trace_gaps.py $O/prof_train > $O/trace_gaps.txt 2>&1; tail -3 $O/trace_gaps.txt
rm -f $O/prof_*/*/*kernel_trace.csv          # keep the merged output small: the stats tables are what is committed
bash tools/run_pmc_gemm.sh $TAG > $O/pmc_gemm.log 2>&1; tail -3 $O/pmc_gemm.log
bash tools/run_pmc_attn.sh $TAG > $O/pmc_attn.log 2>&1; tail -3 $O/pmc_attn.log
bash tools/run_pmc_clap.sh $TAG > $O/pmc_clap.log 2>&1; tail -3 $O/pmc_clap.log
ls $O
