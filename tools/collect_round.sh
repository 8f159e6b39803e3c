# usage: bash tools/collect_round.sh <tag> <dest>   -- copy what tools/run_round_a.sh / run_round_b.sh / run_pmc_*.sh left under gpurun_out/ into profiles/<dest>/
TAG=$1; P=profiles/$2; O=gpurun_out/round_$TAG
mkdir -p $P
for f in bench_train.json bench_train_nodropout.json bench_logmel.json bench_clap.json e2e_config4_scaled.json bench_train_torchrun1.json bench_train_torchrun1_bf16comm.json; do [ -f $O/$f ] && cp $O/$f $P/$f; done
[ -f $O/bench_hf_trainer.txt ] && grep -v amdgpu.ids $O/bench_hf_trainer.txt > $P/bench_hf_trainer.txt
grep -v amdgpu.ids $O/bench_kernels.txt > $P/bench_kernels.txt
for f in attn_bwd_paths.txt attn_fwd.txt; do [ -f $O/$f ] && grep -v amdgpu.ids $O/$f > $P/$f; done
[ -f $O/bench_train_fp32.json ] && cp $O/bench_train_fp32.json $P/bench_train_fp32.json
[ -f $O/bench_train_2ranks_shared_gpu_debug.json ] && cp $O/bench_train_2ranks_shared_gpu_debug.json $P/
cp $O/smoke.log $P/smoke.txt
[ -f $O/pytest_gpu.log ] && tail -3 $O/pytest_gpu.log > $P/pytest_gpu.txt
cp "$(ls -t $O/prof_train/*/*kernel_stats.csv | head -1)" $P/train_step_kernel_stats.csv          # (newest: an earlier call with the same tag leaves its files behind)
cp "$(ls -t $O/prof_logmel/*/*kernel_stats.csv | head -1)" $P/logmel_kernel_stats.csv          # (newest: an earlier call with the same tag leaves its files behind)
cp "$(ls -t $O/prof_clap/*/*kernel_stats.csv | head -1)" $P/clap_kernel_stats.csv          # (newest: an earlier call with the same tag leaves its files behind)
cp "$(ls -t $O/prof_roofline/*/*kernel_stats.csv | head -1)" $P/roofline_gemm_kernel_stats.csv          # (newest: an earlier call with the same tag leaves its files behind)
[ -f gpurun_out/gemm_pmc_summary_$TAG.json ] && cp gpurun_out/gemm_pmc_summary_$TAG.json $P/gemm_pmc_summary.json
[ -f gpurun_out/logmel_pmc_summary_$TAG.json ] && cp gpurun_out/logmel_pmc_summary_$TAG.json $P/logmel_pmc_summary.json
[ -f gpurun_out/attn_pmc_$TAG.txt ] && cp gpurun_out/attn_pmc_$TAG.txt $P/attn_pmc.txt
[ -f gpurun_out/clap_pmc_$TAG.txt ] && cp gpurun_out/clap_pmc_$TAG.txt $P/clap_pmc.txt
ls $P
