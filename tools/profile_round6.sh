#!/bin/bash
# What profiles/r6_* holds, in a few GPU calls (each through gpurun, then copy gpurun_out/round_r6/* into profiles/ with the
# r6_ prefix). Usage: tools/profile_round6.sh <tag> <stage ...>, stages: bench also trace headline kernels workloads traffic.
# Every rocprofv3 pass runs under its own timeout; counter passes carry no trace domain but --kernel-trace; raw traces are
# deleted before the call returns (gpurun copies back at most 64 MiB, and nothing at all beyond that).
# (One call with every stage took more than 50 minutes and came back empty: run the stages in separate calls.)
set -u
TAG=${1:-r6}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/round_$TAG
mkdir -p $OUT
cd $R
stamp() { echo "[$(date +%H:%M:%S)] $*" | tee -a $OUT/stages.log; }
for STAGE in "$@"; do
  stamp "start $STAGE"
  case $STAGE in
    bench)      # the default bench line (what the driver runs), on its own; bench_also.json beside it
      T0=$SECONDS
      python bench.py > $OUT/bench_c2_default.json 2> $OUT/bench_c2_default.err
      echo "bench.py wall seconds: $((SECONDS - T0))" | tee -a $OUT/bench_c2_default.err
      cp bench_also.json $OUT/bench_also_default_budget.json 2>/dev/null ;;
    also)       # every side measurement, without the default run's time budget
      python bench.py --also-budget 0 --no-cpu-baseline > $OUT/bench_c2_all_side_measurements.json 2> $OUT/bench_also.err
      cp bench_also.json $OUT/bench_also.json 2>/dev/null ;;
    trace)      # the default bench under rocprofv3 (kernel trace + stats), at the shipped sources
      tools/profile.sh $TAG > $OUT/bench_c2_rocprofv3_summary.txt 2>&1
      rm -rf gpurun_out/prof_$TAG/trace ;;
    headline)   # SQ counters of the headline's trajectory kernel (-> profiles/headline_pmc_latest.json)
      tools/profile_trajectory.sh $TAG > $OUT/trajectory_kernel_pmc.txt 2>&1
      cp gpurun_out/prof_traj_$TAG/headline_pmc.json $OUT/headline_pmc_latest.json 2>/dev/null
      rm -rf gpurun_out/prof_traj_$TAG/trace gpurun_out/prof_traj_$TAG/pmc_sq gpurun_out/prof_traj_$TAG/pmc_sq2 ;;
    kernels)    # the matrix-core kernels: MFMA busy / LDS counters
      tools/profile_kernel_pmc.sh $TAG c3_euler_general_default_route_b16384_d32_m16 neural_trajectory_kernel > $OUT/pmc_c3_neural_kernel.txt 2>&1
      tools/profile_kernel_pmc.sh $TAG c3_rheun_general_default_route_b16384_d32_m16 "neural_rheun_kernel<32, 64, 16, false>" > $OUT/pmc_c3_rheun_forward_kernel.txt 2>&1
      tools/profile_kernel_pmc.sh $TAG c3_rheun_adjoint_general_default_route_b16384_d32_m16 "neural_rheun_kernel<32, 64, 16, true>" > $OUT/pmc_c3_rheun_backward_kernel.txt 2>&1
      tools/profile_kernel_pmc.sh $TAG c3_rheun_adjoint_general_default_route_b16384_d32_m16 "rheun_last_layer_kernel" > $OUT/pmc_c3_rheun_last_layer_kernel.txt 2>&1
      tools/profile_kernel_pmc.sh $TAG neuraladditive_srk_default_route_b65536_d64_m8 "neural_trajectory_kernel<64, 64, 2" > $OUT/pmc_neuraladditive_srk_kernel.txt 2>&1
      tools/profile_kernel_pmc.sh $TAG c2_srk_netdiag_default_route_b65536_d64_s1000 "neural_trajectory_kernel<64, 64, 0" > $OUT/pmc_netdiag_srk_kernel.txt 2>&1
      rm -rf gpurun_out/pmc_${TAG}_*/trace gpurun_out/pmc_${TAG}_*/pmc1 gpurun_out/pmc_${TAG}_*/pmc2 ;;
    workloads)  # kernel statistics of the new routes' workloads
      : > $OUT/rheun_and_rows_rocprofv3.txt
      for W in c3_rheun_adjoint_general_default_route_b16384_d32_m16 sdegan_rheun_adjoint_default_route_b1024_d16_m3_s63 c5_rheun_adjoint_latent_b32768_d128_s500 c5_logqp_adjoint_latent_b32768_d128_s500 lorenz_euler_default_route_b262144_d3_s1000 c2_srk_exscalar_default_route_b65536_d64_s1000; do
        stamp "  workload $W"
        timeout 300 tools/profile_workload.sh $W ${TAG}_$W --no-also >> $OUT/rheun_and_rows_rocprofv3.txt 2>&1
        rm -rf gpurun_out/prof_${TAG}_$W
      done ;;
    traffic)    # HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the stepwise kernels, tsde_rheun_* included
      tools/profile_traffic.sh $TAG > $OUT/traffic_summary.txt 2>&1
      cp gpurun_out/traffic_$TAG/traffic.json $OUT/traffic_latest.json 2>/dev/null
      rm -rf gpurun_out/traffic_$TAG ;;
    *) stamp "unknown stage $STAGE" ;;
  esac
  stamp "end $STAGE"
done
du -sh $R/gpurun_out | tee -a $OUT/stages.log
ls -la $OUT
