#!/bin/bash
# What profiles/r6_* holds, in one GPU call (run through gpurun, then copy gpurun_out/round_r6/* into profiles/ with the
# r6_ prefix). Every rocprofv3 pass runs under its own timeout; counter passes carry no trace domain but --kernel-trace.
set -u
TAG=${1:-r6}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/round_$TAG
mkdir -p $OUT
cd $R
# 1. the default bench line (what the driver runs), on its own; bench_also.json beside it
/usr/bin/time -f "bench.py wall seconds: %e" python bench.py > $OUT/bench_c2_default.json 2> $OUT/bench_c2_default.err
cp gpurun_out/bench_also.json $OUT/bench_also.json 2>/dev/null || cp bench_also.json $OUT/bench_also.json 2>/dev/null
# 2. the default bench under rocprofv3 (kernel trace + stats), at the shipped sources
tools/profile.sh $TAG > $OUT/bench_c2_rocprofv3_summary.txt 2>&1
# 3. SQ counters of the headline's trajectory kernel (-> profiles/headline_pmc_latest.json)
tools/profile_trajectory.sh $TAG > $OUT/trajectory_kernel_pmc.txt 2>&1
cp gpurun_out/prof_traj_$TAG/headline_pmc.json $OUT/headline_pmc_latest.json 2>/dev/null
# 4. HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the stepwise kernels, tsde_rheun_* and tsde_levy_area included
tools/profile_traffic.sh $TAG > $OUT/traffic_summary.txt 2>&1
cp gpurun_out/traffic_$TAG/traffic.json $OUT/traffic_latest.json 2>/dev/null
# 5. the matrix-core kernels: MFMA busy / LDS counters
tools/profile_kernel_pmc.sh $TAG c3_euler_general_default_route_b16384_d32_m16 neural_trajectory_kernel > $OUT/pmc_c3_neural_kernel.txt 2>&1
tools/profile_kernel_pmc.sh $TAG c3_rheun_general_default_route_b16384_d32_m16 "neural_rheun_kernel<32, 64, 16, false>" > $OUT/pmc_c3_rheun_forward_kernel.txt 2>&1
tools/profile_kernel_pmc.sh $TAG c3_rheun_adjoint_general_default_route_b16384_d32_m16 "neural_rheun_kernel<32, 64, 16, true>" > $OUT/pmc_c3_rheun_backward_kernel.txt 2>&1
# 6. kernel statistics of the new routes' workloads
: > $OUT/rheun_and_rows_rocprofv3.txt
for W in c3_rheun_adjoint_general_default_route_b16384_d32_m16 sdegan_rheun_adjoint_default_route_b1024_d16_m3_s63 c5_rheun_adjoint_latent_b32768_d128_s500 c3_log_ode_general_b16384_d32_m16 c5_logqp_adjoint_latent_b32768_d128_s500 lorenz_euler_default_route_b262144_d3_s1000 c2_srk_exscalar_default_route_b65536_d64_s1000; do
  tools/profile_workload.sh $W ${TAG}_$W --no-also >> $OUT/rheun_and_rows_rocprofv3.txt 2>&1
done
ls -la $OUT
