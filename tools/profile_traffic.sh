#!/bin/bash
# HBM traffic (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes) and kernel durations (--kernel-trace --stats) of
# the dominant kernel of EVERY BASELINE configuration, each from `bench.py --workload W` itself -> one
# traffic.json keyed by workload, stamped with the digest of the kernel sources (bench.py attaches it to the bench
# line only when that digest matches). Usage (on the GPU box, through gpurun): tools/profile_traffic.sh <tag>
set -u
TAG=${1:-r4}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/traffic_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
WORKLOADS=${TRAFFIC_WORKLOADS:-"c2_euler_diag_default_route_b65536_d64_s1000 c2_euler_diag_b65536_d64_s1000 c2_milstein_diag c2_srk_diag c3_euler_general_b16384_d32_m16 c3_milstein_general_gradfree_b16384_d32_m16 c3_euler_additive_shared_b16384_d32_m16 c3_euler_additive_shared_b262144_d64_m32 c4_midpoint_diag_b32768_d64 c5_adjoint_latent_b32768_d128_s500 c5_rheun_adjoint_latent_b32768_d128_s500 c3_log_ode_general_b16384_d32_m16"}
# (TRAFFIC_WORKLOADS="w1 w2" measures a subset: twelve workloads x three passes take ~45 minutes of box time; merge the result
#  into profiles/traffic_latest.json by hand -- same csrc digest or unchanged files of the measured kernels)
for W in $WORKLOADS; do
  ARGS="--workload $W --steps 1 --warmup 1 --profile-steps 100"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$W/trace -o b -- python $R/bench.py $ARGS > $OUT/$W.trace.json 2> $OUT/$W.trace.log
  # (counter passes serialise every kernel: eager launches there -- the bytes a kernel moves do not depend on how it was
  #  launched -- so that the recording-time checks of the graphs do not run under the counters)
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/$W/pmc_fetch -o b -- python $R/bench.py $ARGS --eager > /dev/null 2> $OUT/$W.fetch.log
  timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/$W/pmc_write -o b -- python $R/bench.py $ARGS --eager > /dev/null 2> $OUT/$W.write.log
  find $OUT/$W -name "*kernel_trace.csv" -delete
done
python $R/tools/traffic_summary.py $OUT $WORKLOADS > $OUT/summary.txt 2>&1
find $OUT -name "*counter_collection.csv" -delete
for W in $WORKLOADS; do rm -rf $OUT/$W; done          # (traces and counter files: tens of MiB; the summary and traffic.json stay)
cat $OUT/summary.txt
