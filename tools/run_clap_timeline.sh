# usage (GPU box): bash tools/run_clap_timeline.sh  -- every launch of one CLAP step (512 clips: features + tower + cosine arg-max) in order, with idle gaps
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/clap_tl
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python3 $R/bench.py --workload clap --steps 4 --warmup 2 --no-cpu-baseline > $O/clap.log 2>&1
python3 $R/tools/step_timeline.py $O/tr 2 cosine_argmax > $O/clap_step_timeline.txt
rm -rf $O/tr
tail -40 $O/clap_step_timeline.txt
