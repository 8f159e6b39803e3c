# usage (GPU box): bash tools/probe/gemm_forms.sh  -- which epilogue forms of the 256^2 NT kernel does one training step launch (generic = not instantiated)
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
ADT_GEMM_LOG_FORMS=1 timeout 600 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-e2e --no-clap > $O/forms.json 2> $O/forms.err
grep "nt256 form" $O/forms.err | sort | uniq -c | sort -rn > $O/gemm_forms.txt
cat $O/gemm_forms.txt
