# usage: bash tools/run_pmc_gemm_forms.sh <tag> [M,N,K]   -- the three persistent NT kernels on one bare product, separate --pmc passes each (GPU box)
TAG=${1:-rXX}
export ADT_PMC_SHAPE=${2:-63104,768,3072}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_forms_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for F in 256 2wg ring; do
  export ADT_GEMM_NT=$F
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $O/a_$F -- python3 $R/tools/pmc_gemm_forms.py > $O/a_$F.log 2>&1
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES --output-format csv -d $O/b_$F -- python3 $R/tools/pmc_gemm_forms.py > $O/b_$F.log 2>&1
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/c_$F -- python3 $R/tools/pmc_gemm_forms.py > $O/c_$F.log 2>&1
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/d_$F -- python3 $R/tools/pmc_gemm_forms.py > $O/d_$F.log 2>&1
  python3 $R/tools/pmc_summary.py gemm_nt $O/a_$F $O/b_$F $O/c_$F $O/d_$F > $O/summary_$F.json 2>&1
  echo "== $F"; python3 -c "
import json,sys
d=json.load(open('$O/summary_$F.json'))
print({k: round(v['mean']) for k,v in d.items()})"
done
tail -3 $O/d_ring.log
