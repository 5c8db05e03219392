# Build container: copy the round's measurements from gpurun_out/r06 (scratch) into profiles/r06 (tracked).
cd "$(dirname "$0")/.."
S=gpurun_out/r06; D=profiles/r06
mkdir -p $D
for f in bench_default.json bench_profiled_line.json bench_cpo_profiled_line.json feature_split_profiled_line.json \
         kernel_stats_bench.csv kernel_stats_bench_cpo.csv kernel_stats_feature_split.csv \
         gae_dispatch_durations.json gae_dispatch_durations_cpo.json gae_pmc.json hbm_copy_ceiling.json \
         collect_kernels.txt update_ab_rs.txt kl_ab.txt update_phase_cycles_rs.txt update_phase_cycles_h.txt feature_split_bench.txt p2p_loopback.txt \
         bench_dp2_one_gpu.json update_kernel_pmc.json kl_kernel_pmc.json fvp_kernel_pmc.json ma_train_kernels_pmc.json \
         pytest_gpu_final.log bench_dp2_one_gpu.json wide_step.txt wide_rows_phase_cycles.txt dp_feature_split_step.txt wide_rows_kernel_pmc.json bench_dp4_one_gpu.json fullbatch_ab.txt; do
  [ -f $S/$f ] && cp $S/$f $D/$f || echo "missing: $f"
done
ls -la $D
