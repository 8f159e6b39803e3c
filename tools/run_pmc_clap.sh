# usage: bash tools/run_pmc_clap.sh <tag>   (separate --pmc passes over tools/prof_clap.py, on the GPU box)
TAG=${1:-rXX}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $R/gpurun_out/pmcc1_$TAG -- python3 $R/tools/prof_clap.py 2 > $R/gpurun_out/pmcc1_$TAG.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA --output-format csv -d $R/gpurun_out/pmcc2_$TAG -- python3 $R/tools/prof_clap.py 2 > $R/gpurun_out/pmcc2_$TAG.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmcc3_$TAG -- python3 $R/tools/prof_clap.py 2 > $R/gpurun_out/pmcc3_$TAG.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmcc4_$TAG -- python3 $R/tools/prof_clap.py 2 > $R/gpurun_out/pmcc4_$TAG.log 2>&1
for k in clap_logmel_kernel "htsat_attn_kernel<96" "htsat_attn_big_kernel<192" "htsat_attn_big_kernel<384" "htsat_rowblock_kernel<384, 5" window_attn4_kernel htsat_patch_embed_tok; do echo "== $k"; python3 $R/tools/pmc_summary.py "$k" $R/gpurun_out/pmcc1_$TAG $R/gpurun_out/pmcc2_$TAG $R/gpurun_out/pmcc3_$TAG $R/gpurun_out/pmcc4_$TAG | tr -d '\n {}' | sed 's/"mean"://g'; echo; done | tee $R/gpurun_out/clap_pmc_$TAG.txt
tail -2 $R/gpurun_out/pmcc2_$TAG.log
