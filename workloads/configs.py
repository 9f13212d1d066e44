"""The benchmark workloads: BASELINE.json's configurations at their single-GPU sizes (and the variants DESIGN.md
measures), and the synthetic SDEs they run on. Shared by bench.py, tools/ and the full-size parity tests."""
import torch

WORKLOADS = {
    # BASELINE.json configs[1] as a drop-in user gets it -- THE HEADLINE: the untouched GBM module (f = mu * y, g = sigma * y
    # as user torch code) handed to sdeint with no options. torchsde_amd/recognise.py interprets f and g at every solve,
    # finds them per-channel affine and the whole solve is ONE launch of the trajectory kernel (state in registers,
    # increments from the counter RNG). `bytes_per_traj_step` is SURVEY 8d's figure for this configuration (what a
    # one-kernel-per-step design streams); the kernel's real HBM traffic is y0 in + final state out.
    "c2_euler_diag_default_route_b65536_d64_s1000": dict(
        problem="gbm_ito", method="euler", levy="none", B=65536, d=64, m=64, nsteps=1000, dt=2.0 ** -10,
        bytes_per_traj_step=16 * 64, kid=8, trajectory=True, recognised=True, stepwise="c2_euler_diag_b65536_d64_s1000",
        kernel_match=["trajectory_kernel<float"], launches_per_step=1,
        kernel="tsde_trajectory_affine_diag<float, euler> (trajectory_kernel; user module recognised by recognise.py)"),
    "c2_milstein_diag_default_route": dict(
        problem="gbm_ito", method="milstein", levy="none", B=65536, d=64, m=64, nsteps=1000, dt=2.0 ** -10,
        bytes_per_traj_step=20 * 64, kid=8, trajectory=True, recognised=True,
        kernel="tsde_trajectory_affine_diag<float, milstein> (user module recognised)"),
    "c2_srk_diag_default_route": dict(
        problem="gbm_ito", method="srk", levy="space-time", B=65536, d=64, m=64, nsteps=1000, dt=2.0 ** -10,
        bytes_per_traj_step=64 * 64, kid=8, trajectory=True, recognised=True,
        kernel="tsde_trajectory_affine_diag<float, srk> (user module recognised)"),
    "c4_midpoint_diag_default_route_b32768_d64": dict(
        problem="gbm_strat", method="midpoint", levy="none", B=32768, d=64, m=64, nsteps=1000, dt=2.0 ** -10,
        bytes_per_traj_step=32 * 64, kid=8, trajectory=True, recognised=True, stepwise="c4_midpoint_diag_b32768_d64",
        kernel="tsde_trajectory_affine_diag<float, midpoint> (user module recognised)"),
    # training through sdeint (autograd on, loss.backward()) on the same untouched module: the sensitivity kernel
    # (forward-mode tangents in registers, backward() = a few reductions); stepwise beside it: c2_euler_training_stepwise
    "c2_euler_training_default_route_b65536_d64_s1000": dict(
        problem="gbm_ito", method="euler", levy="none", B=65536, d=64, m=64, nsteps=1000, dt=2.0 ** -10,
        bytes_per_traj_step=16 * 64, kid=8, trajectory=True, recognised=True, train=True,
        kernel="tsde_trajectory_affine_diag_sens<float, euler> (user module recognised; sdeint + loss.backward())"),
    # the SDE the reference's own benchmark solves (benchmarks/brownian.py:131-139: f = y, g = exp(-y)), no options
    "c2_euler_expdiff_default_route_b65536_d64_s1000": dict(
        problem="exp_diffusion", method="euler", levy="none", B=65536, d=64, m=64, nsteps=1000, dt=2.0 ** -16,
        bytes_per_traj_step=16 * 64, kid=8, trajectory=True, recognised=True,
        kernel="tsde_trajectory_expr_diag<float, euler> (user module recognised: f = y, g = exp(-y) in the kernel)"),
    # coefficients that depend on t (a schedule: ScheduledDiag): the `_timed` kernels read one coefficient row per stage
    # time; stepwise counterpart below
    "c2_euler_scheduled_default_route_b65536_d64_s1000": dict(
        problem="scheduled_diag", method="euler", levy="none", B=65536, d=64, m=64, nsteps=1000, dt=2.0 ** -10,
        bytes_per_traj_step=16 * 64, kid=8, trajectory=True, recognised=True,
        kernel="tsde_trajectory_affine_diag_timed<float, euler> (user module with time-dependent coefficients recognised)"),
    "c2_srk_scheduled_default_route_b65536_d64_s1000": dict(
        problem="scheduled_diag", method="srk", levy="space-time", B=65536, d=64, m=64, nsteps=1000, dt=2.0 ** -10,
        bytes_per_traj_step=64 * 64, kid=8, trajectory=True, recognised=True,
        kernel="tsde_trajectory_affine_diag_timed<float, srk> (4 coefficient rows per step)"),
    "c2_euler_scheduled_b65536_d64_s1000": dict(
        problem="scheduled_diag", method="euler", levy="none", B=65536, d=64, m=64, nsteps=1000, dt=2.0 ** -10,
        bytes_per_traj_step=16 * 64, kid=1, launches_per_step=1,
        kernel="tsde_step_diag<float> (elementwise_kernel<StepDiagOp<float>>)"),
    # a polynomial SDE (double well, quadratic diffusion): TSDE_FN_POLY3 in the expression kernel; stepwise counterpart below
    "c2_euler_doublewell_default_route_b65536_d64_s1000": dict(
        problem="double_well", method="euler", levy="none", B=65536, d=64, m=64, nsteps=1000, dt=2.0 ** -10,
        bytes_per_traj_step=16 * 64, kid=8, trajectory=True, recognised=True,
        kernel="tsde_trajectory_expr_diag<float, euler> (user module recognised: cubic drift, quadratic diffusion)"),
    "c2_euler_doublewell_b65536_d64_s1000": dict(
        problem="double_well", method="euler", levy="none", B=65536, d=64, m=64, nsteps=1000, dt=2.0 ** -10,
        bytes_per_traj_step=16 * 64, kid=1, launches_per_step=1,
        kernel="tsde_step_diag<float> (elementwise_kernel<StepDiagOp<float>>)"),
    # The reference's scalar-noise problem (ExScalar, tests/problems.py:75-103: f = -p^2 sin(y) cos(y)^3, g = p cos(y)^2 of shape
    # (B, d, 1), one Brownian channel per row) with the method `sdeint` picks for it by default (SRK): drift and diffusion are
    # products of several functions of the state, so they travel as expression programs (recognise.RecognisedProgram) and the
    # solve is ONE launch of tsde_trajectory_prog_diag; stepwise counterpart below
    "c2_srk_exscalar_default_route_b65536_d64_s1000": dict(
        problem="scalar_ito", method="srk", levy="space-time", B=65536, d=64, m=1, nsteps=1000, dt=2.0 ** -10,
        bytes_per_traj_step=64 * 64, kid=8, trajectory=True, recognised=True,
        kernel="tsde_trajectory_prog_diag<float, srk, scalar noise> (user module recognised as expression programs)"),
    "c2_euler_exscalar_default_route_b65536_d64_s1000": dict(
        problem="scalar_ito", method="euler", levy="none", B=65536, d=64, m=1, nsteps=1000, dt=2.0 ** -10,
        bytes_per_traj_step=16 * 64, kid=8, trajectory=True, recognised=True,
        kernel="tsde_trajectory_prog_diag<float, euler, scalar noise> (user module recognised as expression programs)"),
    # ... and TRAINING through sdeint on the same module (autograd on, loss.backward()): the programs on dual numbers
    # (tsde_trajectory_prog_diag_sens), gradients to p through the user's own `-p ** 2`, `p * cos(y) ** 2`. (Stepwise, autograd
    # would have to keep ~12 (B, d) tensors for each of the 1000 steps: ~200 GB at this size.)
    "c2_euler_exscalar_training_default_route_b65536_d64_s1000": dict(
        problem="scalar_ito", method="euler", levy="none", B=65536, d=64, m=1, nsteps=1000, dt=2.0 ** -10,
        bytes_per_traj_step=16 * 64, kid=8, trajectory=True, recognised=True, train=True,
        kernel="tsde_trajectory_prog_diag_sens<float, euler, scalar noise> (user module recognised; sdeint + loss.backward())"),
    "c2_euler_exscalar_b65536_d64_s1000": dict(
        problem="scalar_ito", method="euler", levy="none", B=65536, d=64, m=1, nsteps=1000, dt=2.0 ** -10,
        bytes_per_traj_step=16 * 64, kid=1, launches_per_step=1,
        kernel="tsde_step_diag<float> (user f, g: ~11 torch kernels per step)"),
    "c2_srk_exscalar_b65536_d64_s1000": dict(
        problem="scalar_ito", method="srk", levy="space-time", B=65536, d=64, m=1, nsteps=1000, dt=2.0 ** -10,
        bytes_per_traj_step=64 * 64, bytes_moved_per_traj_step=92 * 64, kid=4, launches_per_step=4,
        kernel="tsde_srk_diag_stage<float> (4 stage kernels; user f, g: ~12 torch kernels per evaluation)"),
    # Heun (heun.py:35-48), the Stratonovich predictor-corrector, on the untouched GBM module: one launch of the affine
    # kernel (two evaluations of f and g per step); stepwise counterpart below (tsde_step_diag + tsde_heun_final)
    "c2_heun_diag_default_route_b65536_d64_s1000": dict(
        problem="gbm_strat", method="heun", levy="none", B=65536, d=64, m=64, nsteps=1000, dt=2.0 ** -10,
        bytes_per_traj_step=(16 + 28) * 64, kid=8, trajectory=True, recognised=True,
        kernel="tsde_trajectory_affine_diag<float, heun> (user module recognised)"),
    "c2_heun_diag_b65536_d64_s1000": dict(
        problem="gbm_strat", method="heun", levy="none", B=65536, d=64, m=64, nsteps=1000, dt=2.0 ** -10,
        bytes_per_traj_step=(16 + 28) * 64, kid=1, launches_per_step=2, bench_steps=200,
        step_kernels={"tsde_step_diag": 1, "tsde_heun_final": 1},
        kernel="tsde_step_diag<float> + tsde_heun_final<float> (user f, g evaluated twice per step)"),
    # The reference's additive-noise problem (ExAdditive, tests/problems.py:106-132: f and g use t; g repeated over m columns)
    # with sdeint's default method for additive noise, SRK = SRA1 (sdeint.py:151, srk.py:90-111), as a drop-in call: the
    # drift travels as an expression program, the diffusion as a table over the stage times (one batched call of g), the
    # solve is ONE launch of tsde_trajectory_prog_additive; stepwise counterpart below (3 contractions + 2 f + 2 g per step)
    "exadditive_srk_default_route_b65536_d64_m8": dict(
        problem="additive_ito", method="srk", levy="space-time", B=65536, d=64, m=8, nsteps=1000, dt=2.0 ** -10,
        bytes_per_traj_step=3 * 4 * (64 * 8 + 3 * 64), kid=8, trajectory=True, recognised=True,
        kernel="tsde_trajectory_prog_additive<float, srk (SRA1), m = 8> (user module recognised: drift program + g(t) table)"),
    "exadditive_euler_default_route_b65536_d64_m8": dict(
        problem="additive_ito", method="euler", levy="none", B=65536, d=64, m=8, nsteps=1000, dt=2.0 ** -10,
        bytes_per_traj_step=4 * (64 * 8 + 3 * 64), kid=8, trajectory=True, recognised=True,
        kernel="tsde_trajectory_prog_additive<float, euler, m = 8> (user module recognised: drift program + g(t) table)"),
    # ... and the reference's NeuralAdditive (tests/problems.py:195-224; hidden 64): f_net of cat([t, y]) on the matrix cores,
    # g_net of t alone tabulated, the default method (SRK = SRA1)
    "neuraladditive_srk_default_route_b65536_d64_m8": dict(
        problem="netadditive_big", method="srk", levy="space-time", B=65536, d=64, m=8, nsteps=1000, dt=2.0 ** -10,
        kid=8, trajectory=True, recognised=True, mfma_flops_per_traj_step=2 * 2 * (64 * 64 + 64 * 64) + 3 * 2 * 64 * 8,
        kernel="tsde_trajectory_mlp_additive<64, 64, srk (SRA1), m = 8> (neural_trajectory_kernel; user module recognised)"),
    "neuraladditive_srk_b65536_d64_m8": dict(
        problem="netadditive_big", method="srk", levy="space-time", B=65536, d=64, m=8, nsteps=1000, dt=2.0 ** -10,
        bytes_per_traj_step=3 * 4 * (64 * 8 + 3 * 64), kid=2, launches_per_step=3, bench_steps=200,
        kernel="tsde_step_general_w<float> (3 weighted contractions per step; user f_net x2, g_net x2)"),
    "exadditive_srk_b65536_d64_m8": dict(
        problem="additive_ito", method="srk", levy="space-time", B=65536, d=64, m=8, nsteps=1000, dt=2.0 ** -10,
        bytes_per_traj_step=3 * 4 * (64 * 8 + 3 * 64), kid=2, launches_per_step=3, bench_steps=200,
        kernel="tsde_step_general_w<float> (3 weighted contractions per step; user f x2, g x2: ~20 torch kernels per step)"),
    "exadditive_euler_b65536_d64_m8": dict(
        problem="additive_ito", method="euler", levy="none", B=65536, d=64, m=8, nsteps=1000, dt=2.0 ** -10,
        bytes_per_traj_step=4 * (64 * 8 + 3 * 64), kid=2, launches_per_step=1, bench_steps=200,
        kernel="tsde_step_general<float> (user f, g: ~12 torch kernels per step)"),
    # BASELINE.json configs[1], STEPWISE (options={"trajectory_kernel": False}): the user's f and g run as torch kernels
    # between the per-step kernels -- the route of every SDE that is not a per-channel expression
    "c2_euler_diag_b65536_d64_s1000": dict(
        problem="gbm_ito", method="euler", levy="none", B=65536, d=64, m=64, nsteps=1000, dt=2.0 ** -10,
        bytes_per_traj_step=16 * 64, kid=1, launches_per_step=1, kernel_match=["StepDiagOp<float>"],
        kernel="tsde_step_diag<float> (elementwise_kernel<StepDiagOp<float>>)"),
    # the other BASELINE configs at their single-GPU size (parity-test cases; measured for DESIGN.md, not the headline)
    "c2_milstein_diag": dict(
        problem="gbm_ito", method="milstein", levy="none", B=65536, d=64, m=64, nsteps=1000, dt=2.0 ** -10,
        bytes_per_traj_step=20 * 64, kid=3, launches_per_step=1, kernel_match=["MilsteinDiagOp<float>"],
        kernel="tsde_milstein_diag<float>"),
    "c2_srk_diag": dict(
        problem="gbm_ito", method="srk", levy="space-time", B=65536, d=64, m=64, nsteps=1000, dt=2.0 ** -10,
        # SURVEY 8d's figure for SRID2 is 64*d (every operand touched once); with user code between the stages the four
        # kernels move 23 streams = 92*d (6 + 8 + 6 + 3, include/torchsde_amd.h) -- both are reported
        bytes_per_traj_step=64 * 64, bytes_moved_per_traj_step=92 * 64, kid=4, launches_per_step=4,
        kernel_match=["SrkDiagOp<float, 1>", "SrkDiagOp<float, 2>", "SrkDiagOp<float, 3>", "SrkDiagOp<float, 4>"],
        kernel="tsde_srk_diag_stage<float> (4 stage kernels)"),
    "c3_euler_general_b16384_d32_m16": dict(
        problem="general_big", method="euler", levy="none", B=16384, d=32, m=16, nsteps=1000, dt=2.0 ** -10,
        bytes_per_traj_step=4 * (32 * 16 + 3 * 32), kid=2, launches_per_step=1, kernel_match=["general_rows_kernel<float"],
        kernel="tsde_step_general<float> (general_rows_kernel)"),
    # BASELINE configs[2] as a drop-in user gets it: the untouched NeuralGeneral-style module (f_net, g_net of cat([t, y]),
    # g reshaped to (B, d, m); tests/problems.py:226-252 at hidden 64) through sdeint with no options. recognise.py follows
    # both nets and the whole solve is ONE launch of tsde_trajectory_mlp_general: four layers and the contraction with the
    # increments on the f32 matrix cores, weights in LDS. MFMA-bound: 2 * (d*h + h*d + d*h + h*d*m) flop per trajectory-step.
    "c3_euler_general_default_route_b16384_d32_m16": dict(
        problem="general_big", method="euler", levy="none", B=16384, d=32, m=16, nsteps=1000, dt=2.0 ** -10,
        kid=8, trajectory=True, recognised=True, stepwise="c3_euler_general_b16384_d32_m16",
        mfma_flops_per_traj_step=2 * (32 * 64 + 64 * 32 + 32 * 64 + 64 * 32 * 16),
        kernel="tsde_trajectory_mlp_general<32, 64, general m = 16, euler> (neural_trajectory_kernel, v_mfma_f32_16x16x4_f32; "
               "user module recognised)"),
    # ... with the OPT-IN split-bf16 products for the diffusion net's second layer (options={"matrix_precision": "bf16x3"};
    # an experiment: NOT the reference's arithmetic, never the headline; its error against float64 is in
    # tests/test_gpu_neural.py). `tflops_f32` below is the exact-f32 flop count per second, for comparison only.
    "c3_euler_general_bf16x3_default_route_b16384_d32_m16": dict(
        problem="general_big", method="euler", levy="none", B=16384, d=32, m=16, nsteps=1000, dt=2.0 ** -10,
        kid=8, trajectory=True, recognised=True, options={"matrix_precision": "bf16x3"},
        mfma_flops_per_traj_step=2 * (32 * 64 + 64 * 32 + 32 * 64 + 64 * 32 * 16),
        kernel="tsde_trajectory_mlp_general<32, 64, general m = 16, euler, split bf16 x3> (opt-in experiment)"),
    # ... and the Stratonovich default for general noise, midpoint (sdeint.py:155): two evaluations of both nets per step
    "c3_midpoint_general_default_route_b16384_d32_m16": dict(
        problem="general_big_strat", method="midpoint", levy="none", B=16384, d=32, m=16, nsteps=1000, dt=2.0 ** -10,
        kid=8, trajectory=True, recognised=True,
        mfma_flops_per_traj_step=4 * (32 * 64 + 64 * 32 + 32 * 64 + 64 * 32 * 16),
        kernel="tsde_trajectory_mlp_general<32, 64, general m = 16, midpoint> (user module recognised)"),
    # The reference's NeuralDiagonal (tests/problems.py:135-162: f_net, g_net of cat([t, y]), g = 0.1 * sigmoid-closed net) at
    # d = hidden = 64 with EVERY default of sdeint -- SRK (SRID2) for diagonal Ito noise: seven net evaluations per step, all on
    # the matrix cores in one launch; stepwise counterpart below
    "c2_srk_netdiag_default_route_b65536_d64_s1000": dict(
        problem="netdiag_big", method="srk", levy="space-time", B=65536, d=64, m=64, nsteps=1000, dt=2.0 ** -10,
        kid=8, trajectory=True, recognised=True, mfma_flops_per_traj_step=7 * 2 * (64 * 64 + 64 * 64),
        kernel="tsde_trajectory_mlp_general<64, 64, diagonal, srk> (neural_trajectory_kernel; user module recognised)"),
    "c2_srk_netdiag_b65536_d64_s1000": dict(
        problem="netdiag_big", method="srk", levy="space-time", B=65536, d=64, m=64, nsteps=1000, dt=2.0 ** -10,
        bytes_per_traj_step=64 * 64, bytes_moved_per_traj_step=92 * 64, kid=4, launches_per_step=4, bench_steps=200,
        kernel="tsde_srk_diag_stage<float> (4 stage kernels; user f_net, g_net: 7 evaluations per step)"),
    # The batch-broadcast diffusion of north_star's "MFMA ... for the dense g.dW batched matmul": additive noise returned
    # as sigma.expand(B, d, m) at the configs[2] shape and at a larger one. One launch of the matrix-core kernel per
    # step: reads y0, f, writes y1 (12*d bytes per trajectory-step), increments generated in registers, S in LDS.
    "c3_euler_additive_shared_b16384_d32_m16": dict(
        problem="additive_shared_ito", method="euler", levy="none", B=16384, d=32, m=16, nsteps=1000, dt=2.0 ** -10,
        bytes_per_traj_step=12 * 32, kid=12, launches_per_step=1, kernel_match=["shared_mfma_kernel<float"],
        kernel="tsde_step_shared<float> (shared_mfma_kernel, v_mfma_f32_16x16x4_f32)"),
    "c3_euler_additive_shared_b262144_d64_m32": dict(
        problem="additive_shared_ito", method="euler", levy="none", B=262144, d=64, m=32, nsteps=200, dt=2.0 ** -10,
        bytes_per_traj_step=12 * 64, kid=12, launches_per_step=1, kernel_match=["shared_mfma_kernel<float"],
        kernel="tsde_step_shared<float> (shared_mfma_kernel, v_mfma_f32_16x16x4_f32)"),
    # BASELINE configs[2] as literally worded: Milstein for GENERAL noise does not exist in the reference (it raises
    # ValueError, milstein.py:25); this is the opt-in extension pinned by reduction tests (tests/test_gpu_milstein_general.py)
    "c3_milstein_general_b16384_d32_m16": dict(
        problem="general_big", method="milstein", levy="foster", B=16384, d=32, m=16, nsteps=1000, dt=2.0 ** -10,
        bytes_per_traj_step=4 * (32 * 16 + 3 * 32), kid=2, launches_per_step=1, options={"general_noise": True},
        eager=True,          # (the JVPs run double backward through autograd inside every step: issued eagerly)
        bench_steps=50,      # 16 double-backward JVPs per step: 50 of the 1000 steps are timed and the rest extrapolated
        kernel="tsde_step_general<float> (+ tsde_levy_area, tsde_iterated_integrals, 16 user JVPs per step)"),
    # ... and its derivative-free form (the reference's derivative-free idea, milstein.py:58-67, per Brownian channel): all m
    # supporting states through ONE call of the user's g on m*B rows, the correction in one kernel that streams that
    # call's (m, B, d, m) result once: per trajectory-step g (d*m) + g at the supporting states (m*d*m) + I (m*m) + y, f, corr
    "c3_milstein_general_gradfree_b16384_d32_m16": dict(
        problem="general_big", method="milstein", levy="foster", B=16384, d=32, m=16, nsteps=1000, dt=2.0 ** -10,
        bytes_per_traj_step=4 * (32 * 16 * 17 + 16 * 16 + 32), kid=11, launches_per_step=1,
        options={"general_noise": True, "grad_free": True}, kernel_match=["gf_correction_rows_kernel<float"],
        kernel="tsde_milstein_gf_general_correction<float> (gf_correction_rows_kernel) + tsde_step_general"),
    "c4_midpoint_diag_b32768_d64": dict(
        problem="gbm_strat", method="midpoint", levy="none", B=32768, d=64, m=64, nsteps=1000, dt=2.0 ** -10,
        bytes_per_traj_step=32 * 64, kid=1, launches_per_step=2, kernel_match=["StepDiagOp<float>"],
        kernel="tsde_step_diag<float> (two stages per step)"),
    # SURVEY 8d's nonlinear second workload: the SDE the reference's own benchmark integrates (benchmarks/brownian.py:
    # 131-139), f = y, g = exp(-y), at the headline's shape -- stepwise, f and g are user torch ops. Horizon 1000 * 2^-16:
    # the noise amplitude exp(-y) grows as a path moves down, and by 1000 * 2^-14 the lowest of the 4M paths has
    # run away to -inf under explicit Euler (the reference's benchmark takes 100 steps of at most 512 x 256 paths and
    # only times them)
    "c2_euler_expdiff_b65536_d64_s1000": dict(
        problem="exp_diffusion", method="euler", levy="none", B=65536, d=64, m=64, nsteps=1000, dt=2.0 ** -16,
        bytes_per_traj_step=16 * 64, kid=1, launches_per_step=1,
        kernel="tsde_step_diag<float> (elementwise_kernel<StepDiagOp<float>>)"),
    # The headline dynamics (same mu, sigma, seed addressing: bit-identical final states) handed over as a closed-form
    # SDE (torchsde_amd.AffineDiagonalSDE): the whole solve is ONE launch of the trajectory kernel, state in
    # registers. VALU-bound (Philox + Box-Muller), so the HBM roofline fraction is ~0 by design.
    "c2_euler_closed_form_b65536_d64_s1000": dict(
        problem="gbm_closed_form", method="euler", levy="none", B=65536, d=64, m=64, nsteps=1000, dt=2.0 ** -10,
        kid=8, trajectory=True, kernel="tsde_trajectory_affine_diag<float, euler> (trajectory_kernel)"),
    # ... and the reference's benchmark SDE (c2_euler_expdiff above) stated as an elementwise-expression module
    # (torchsde_amd.ElementwiseDiagonalSDE): the whole solve is one launch of tsde_trajectory_expr_diag
    "c2_euler_expdiff_closed_form_b65536_d64_s1000": dict(
        problem="exp_diffusion_closed_form", method="euler", levy="none", B=65536, d=64, m=64, nsteps=1000,
        dt=2.0 ** -16, kid=8, trajectory=True,
        kernel="tsde_trajectory_expr_diag<float, euler> (trajectory_expr_kernel: f = y, g = exp(-y) in the kernel)"),
    "c2_milstein_closed_form": dict(
        problem="gbm_closed_form", method="milstein", levy="none", B=65536, d=64, m=64, nsteps=1000, dt=2.0 ** -10,
        kid=8, trajectory=True, kernel="tsde_trajectory_affine_diag<float, milstein> (trajectory_kernel)"),
    "c2_srk_closed_form": dict(
        problem="gbm_closed_form", method="srk", levy="space-time", B=65536, d=64, m=64, nsteps=1000, dt=2.0 ** -10,
        kid=8, trajectory=True, kernel="tsde_trajectory_affine_diag<float, srk> (trajectory_kernel)"),
    "c4_midpoint_closed_form_b32768_d64": dict(
        problem="gbm_closed_form_strat", method="midpoint", levy="none", B=32768, d=64, m=64, nsteps=1000,
        dt=2.0 ** -10, kid=8, trajectory=True,
        kernel="tsde_trajectory_affine_diag<float, midpoint> (trajectory_kernel)"),
    # Neural-SDE SAMPLING at the configs[4] shape (forward only): drift = Linear(128,128)-Softplus-Linear(128,128) like the
    # latent-SDE workload below, affine diagonal diffusion, handed over as torchsde_amd.MLPDriftDiagonalSDE: one launch
    # of the perceptron-drift kernel, both layers on the f32 matrix cores. MFMA-bound: 4*d*hidden flop per trajectory-step.
    "c5_sampling_mlp_b32768_d128_s500": dict(
        problem="mlp_drift", method="euler", levy="none", B=32768, d=128, m=128, nsteps=500, dt=2.0 ** -9,
        kid=8, trajectory=True, mfma_flops_per_traj_step=4 * 128 * 128,
        kernel="tsde_trajectory_mlp_diag<128, 128, softplus> (mlp_trajectory_kernel, v_mfma_f32_16x16x4_f32)"),
    # ... and with the reference's DEFAULT method for a diagonal Ito SDE, SRK (SRID2): three drift evaluations per step
    "c5_sampling_mlp_srk_b32768_d128_s500": dict(
        problem="mlp_drift", method="srk", levy="space-time", B=32768, d=128, m=128, nsteps=500, dt=2.0 ** -9,
        kid=8, trajectory=True, mfma_flops_per_traj_step=12 * 128 * 128,
        kernel="tsde_trajectory_mlp_diag<128, 128, softplus, srk> (mlp_trajectory_kernel, v_mfma_f32_16x16x4_f32)"),
    # The TRAINING step of the SDE of c5_adjoint_latent below (same parameter values, stated as the closed-form module):
    # forward + loss.backward() through the solver, Euler: sampling kernel writing every step, reverse sweep (three
    # products per step on the matrix cores), tall-K weight-gradient products.
    # Roofline: the reverse sweep, 3 * 2*d*hidden flop per trajectory-step.
    "c5_training_mlp_b32768_d128_s500": dict(
        problem="latent_diag_closed_form", method="euler", levy="none", B=32768, d=128, m=128, nsteps=500, dt=2.0 ** -9,
        bytes_per_traj_step=0, kid=9, launches_per_step=1, trajectory=True, train=True,
        mfma_flops_per_traj_step=6 * 128 * 128,
        kernel="tsde_trajectory_mlp_diag_backward<128, 128, softplus> (mlp_backward_kernel, v_mfma_f32_16x16x4_f32)"),
    # BASELINE configs[4] through the API it names -- sdeint_adjoint, method = adjoint_method = "euler" -- with the latent
    # SDE stated as the closed-form module: forward = the sampling kernel (outputs only), backward = the stochastic
    # adjoint on the matrix cores (tsde_adjoint_mlp_diag: y reconstructed, four products per step) + the weight-gradient
    # products. Roofline: the adjoint kernel, 4 * 2*d*hidden flop per trajectory-step.
    "c5_adjoint_mlp_b32768_d128_s500": dict(
        problem="latent_diag_closed_form", method="euler", adjoint_method="euler", levy="none", B=32768, d=128, m=128,
        nsteps=500, dt=2.0 ** -9, bytes_per_traj_step=0, kid=10, launches_per_step=1, trajectory=True, adjoint=True,
        mfma_flops_per_traj_step=8 * 128 * 128,
        kernel="tsde_adjoint_mlp_diag<128, 128, softplus> (mlp_adjoint_kernel, v_mfma_f32_16x16x4_f32)"),
    # ... and with Milstein steps both ways (adjoint_method="milstein" is what `sdeint_adjoint` picks by default for a
    # diagonal Ito SDE, adjoint.py:281-296): the same four products per step, more elementwise terms
    "c5_adjoint_mlp_milstein_b32768_d128_s500": dict(
        problem="latent_diag_closed_form", method="milstein", adjoint_method="milstein", levy="none", B=32768, d=128,
        m=128, nsteps=500, dt=2.0 ** -9, bytes_per_traj_step=0, kid=10, launches_per_step=1, trajectory=True,
        adjoint=True, mfma_flops_per_traj_step=8 * 128 * 128,
        kernel="tsde_adjoint_mlp_diag<128, 128, softplus, milstein> (mlp_adjoint_kernel, v_mfma_f32_16x16x4_f32)"),
    # ... and with EVERY default of `sdeint_adjoint(sde, y0, ts)` for a diagonal Ito SDE: forward SRK (adjoint.py /
    # sdeint.py:246-253), backward Milstein
    "c5_adjoint_mlp_defaults_b32768_d128_s500": dict(
        problem="latent_diag_closed_form", method="srk", adjoint_method="milstein", levy="space-time", B=32768, d=128,
        m=128, nsteps=500, dt=2.0 ** -9, bytes_per_traj_step=0, kid=10, launches_per_step=1, trajectory=True,
        adjoint=True, mfma_flops_per_traj_step=8 * 128 * 128,
        kernel="tsde_adjoint_mlp_diag<128, 128, softplus, milstein> (mlp_adjoint_kernel, v_mfma_f32_16x16x4_f32)"),
    # BASELINE configs[4] as a drop-in user gets it: the latent-SDE user module (nn.Sequential drift, 0.1 * sigmoid(w*y + b)
    # diffusion; nothing of this package in it) through sdeint_adjoint with no options. recognise.py finds the perceptron
    # form; forward = tsde_trajectory_mlp_diag, backward = tsde_adjoint_mlp_diag + tsde_gram_partials, gradients on the
    # module's own parameters. Stepwise beside it: c5_adjoint_latent_b32768_d128_s500.
    "c5_adjoint_latent_default_route_b32768_d128_s500": dict(
        problem="latent_diag", method="euler", adjoint_method="euler", levy="none", B=32768, d=128, m=128,
        nsteps=500, dt=2.0 ** -9, bytes_per_traj_step=0, kid=10, launches_per_step=1, trajectory=True, recognised=True,
        adjoint=True, mfma_flops_per_traj_step=8 * 128 * 128,
        kernel="tsde_adjoint_mlp_diag<128, 128, softplus> (mlp_adjoint_kernel, v_mfma_f32_16x16x4_f32; user module recognised)"),
    "c5_adjoint_latent_defaults_default_route_b32768_d128_s500": dict(
        problem="latent_diag", method="srk", adjoint_method="milstein", levy="space-time", B=32768, d=128, m=128,
        nsteps=500, dt=2.0 ** -9, bytes_per_traj_step=0, kid=10, launches_per_step=1, trajectory=True, recognised=True,
        adjoint=True, mfma_flops_per_traj_step=8 * 128 * 128,
        kernel="tsde_adjoint_mlp_diag<128, 128, softplus, milstein> (user module recognised; every default of sdeint_adjoint)"),
    "c5_adjoint_latent_b32768_d128_s500": dict(
        problem="latent_diag", method="euler", adjoint_method="euler", levy="none", B=32768, d=128, m=128,
        nsteps=500, dt=2.0 ** -9, bytes_per_traj_step=32 * 128, kid=5, launches_per_step=1, adjoint=True,
        kernel_match=["aug_multi_kernel<float>"],
        kernel="tsde_aug_update<float> (aug_multi_kernel, backward sweep)"),
    # ---- SURVEY 8(f) rows on the routes they have today (stepwise): measured so that they are rows with numbers -------------
    # 8f rank 1: the reference's recommended training pair (DOCUMENTATION.md:97,118; reversible_heun.py:48-144) at the
    # configs[4] shape. Forward per step: tsde_rheun_z (y, z, f, g -> z': 5 streams) + tsde_rheun_y (y, f, f', g, g' -> y': 6
    # streams); backward per step: tsde_rheun_adj_a (3 in, 2 out) + _z (5) + _y (6) + tsde_rheun_adj_b (3 in, 4 out):
    # (11 + 23) * 4 * d bytes per trajectory-step forward + backward.
    "c5_rheun_adjoint_latent_b32768_d128_s500": dict(
        problem="latent_diag_strat", method="reversible_heun", adjoint_method="adjoint_reversible_heun", levy="none",
        B=32768, d=128, m=128, nsteps=500, dt=2.0 ** -9, bytes_per_traj_step=(11 + 23) * 4 * 128, kid=7, adjoint=True,
        launches_per_step=6,
        step_kernels={"tsde_rheun_z": 2, "tsde_rheun_y": 2, "tsde_rheun_adj_a": 1, "tsde_rheun_adj_b": 1},
        kernel_match=["RheunZOp<float>", "RheunYOp<float>", "RheunAdjAOp<float>", "RheunAdjBOp<float>"],
        kernel="tsde_rheun_z / _y / _adj_a / _adj_b <float> (6 launches per forward + backward step; user f, g and their "
               "vector-Jacobian products between them)"),
    # 8f rank 4: `sdeint_adjoint(..., logqp=True)` (base_sde.py:240-306) at the configs[4] shape: the state carries one more
    # column, u = (f - h) / g, through the same stepwise kernels as c5_adjoint_latent (d + 1 = 129 channels: rows are no longer
    # 16-byte groups)
    "c5_logqp_adjoint_latent_b32768_d128_s500": dict(
        problem="latent_diag_logqp", method="euler", adjoint_method="euler", levy="none", B=32768, d=128, m=129,
        nsteps=500, dt=2.0 ** -9, bytes_per_traj_step=32 * 129, kid=5, launches_per_step=1, adjoint=True, logqp=True,
        kernel_match=["aug_multi_kernel<float>"],
        kernel="tsde_aug_update<float> on the (B, d + 1) logqp state (aug_multi_kernel, backward sweep; user f, g, h and "
               "their VJPs between the launches)"),
    # 8f rank 3: log-ODE (log_ode.py:39-56) with Foster's Levy area (brownian_interval.py:78-99) at the configs[2] shape:
    # per step tsde_levy_area (W, H -> A (B, m, m)), two general contractions, and the user's m-column JVP
    "c3_log_ode_general_b16384_d32_m16": dict(
        problem="general_big_strat", method="log_ode", levy="foster", B=16384, d=32, m=16, nsteps=1000, dt=2.0 ** -10,
        bytes_per_traj_step=2 * 4 * (32 * 16 + 3 * 32) + 4 * (16 * 16 + 2 * 16), kid=2, launches_per_step=2, eager=True,
        bench_steps=50, kernel_match=["general_rows_kernel<float", "levy_area"],
        step_kernels={"tsde_step_general": 2},
        kernel="tsde_step_general x2 + tsde_levy_area <float> (+ the user's g and its Jacobian-vector products per step)"),
    # The stochastic Lorenz system of the reference's examples/latent_sde_lorenz.py:56-86, written as that example writes it
    # (split / cat; workloads/problems.py): channels that read each other. One
    # lane owns a row (recognise_rows.py), the model is generated from the user's code and compiled at run time; stepwise twin
    # below (split / cat and ~20 small torch kernels per step). The example's own sizes: 1024 rows; here also 262144.
    "lorenz_euler_default_route_b262144_d3_s1000": dict(
        problem="stochastic_lorenz", method="euler", levy="none", B=262144, d=3, m=3, nsteps=1000, dt=2.0 ** -12,
        bytes_per_traj_step=16 * 3, kid=8, trajectory=True, recognised=True,
        kernel="trajectory_prog_kernel<float, euler, 3, RowModel> (one lane per row; model generated by specialise.source_rows)"),
    "lorenz_euler_b262144_d3_s1000": dict(
        problem="stochastic_lorenz", method="euler", levy="none", B=262144, d=3, m=3, nsteps=1000, dt=2.0 ** -12,
        bytes_per_traj_step=16 * 3, kid=1, launches_per_step=1, bench_steps=200,
        kernel="tsde_step_diag<float> (user f, g: split, cat and ~20 torch kernels per step)"),
    "lorenz_srk_default_route_b1024_d3_s1000": dict(
        problem="stochastic_lorenz", method="srk", levy="space-time", B=1024, d=3, m=3, nsteps=1000, dt=2.0 ** -12,
        bytes_per_traj_step=64 * 3, kid=8, trajectory=True, recognised=True,
        kernel="trajectory_prog_kernel<float, srk, 3, RowModel> (the example's batch of 1024, sdeint's default method)"),
    "lorenz_srk_b1024_d3_s1000": dict(
        problem="stochastic_lorenz", method="srk", levy="space-time", B=1024, d=3, m=3, nsteps=1000, dt=2.0 ** -12,
        bytes_per_traj_step=64 * 3, kid=0, launches_per_step=4, bench_steps=200,
        kernel="tsde_srk_diag_stage<float> (4 stage kernels; user f, g: 7 evaluations per step)"),
    # ---- the reversible pair on the matrix cores (csrc/tsde_neural_rheun.h): the reference's recommended training method ------
    # The generator of the reference's examples/sde_gan.py at the example's own sizes (hidden 16, noise 3, mlp 16, batch 1024,
    # 64 output times a unit step apart), `sdeint_adjoint(method="reversible_heun", adjoint_method="adjoint_reversible_heun")`
    # as :129-130 calls it: forward ONE launch, backward ONE launch + the weight-gradient products; stepwise twin below
    "sdegan_rheun_adjoint_default_route_b1024_d16_m3_s63": dict(
        problem="sdegan_generator", method="reversible_heun", adjoint_method="adjoint_reversible_heun", levy="none", B=1024,
        d=16, m=3, nsteps=63, dt=1.0, output_every_step=True, kid=13, trajectory=True, recognised=True, adjoint=True,
        mfma_flops_per_traj_step=3 * 2 * (16 * 32 + 32 * 16 + 16 * 32 + 32 * 16 * 4),
        kernel="tsde_rheun_mlp_forward + _backward<16, 32, general m <= 4> (neural_rheun_kernel; user module recognised; "
               "flops of the padded tiles)"),
    "sdegan_rheun_adjoint_b1024_d16_m3_s63": dict(
        problem="sdegan_generator", method="reversible_heun", adjoint_method="adjoint_reversible_heun", levy="none", B=1024,
        d=16, m=3, nsteps=63, dt=1.0, output_every_step=True, kid=0, launches_per_step=2, adjoint=True,
        bytes_per_traj_step=0, kernel="stepwise pair: tsde_step_general_w + torch ops + autograd VJPs of the user's nets"),
    # ... sampling from the same generator with the Stratonovich default of `sdeint`, midpoint (sdeint.py:155), 16384 paths:
    # tsde_deep_mlp_forward (the reversible-Heun kernel's evaluation under a stateless scheme); stepwise twin below
    "sdegan_midpoint_default_route_b16384_d16_m3_s1000": dict(
        problem="sdegan_generator", method="midpoint", levy="none", B=16384, d=16, m=3, nsteps=1000, dt=2.0 ** -10,
        kid=13, trajectory=True, recognised=True, mfma_flops_per_traj_step=2 * 2 * (16 * 32 + 32 * 16 + 16 * 32 + 32 * 16 * 4),
        kernel="tsde_deep_mlp_forward<16, 32, general m <= 4, midpoint> (LipSwish nets closed by tanh; flops of the padded tiles)"),
    "sdegan_midpoint_b16384_d16_m3_s1000": dict(
        problem="sdegan_generator", method="midpoint", levy="none", B=16384, d=16, m=3, nsteps=1000, dt=2.0 ** -10,
        bytes_per_traj_step=2 * 4 * (16 * 3 + 3 * 16), kid=0, launches_per_step=2, bench_steps=200,
        kernel="tsde_step_general<float> x2 per step (user nets: ~30 torch kernels per step)"),
    # ... and at the BASELINE configs[2] shape (NeuralGeneral-style nets, hidden 64): sampling, then forward + backward
    "c3_rheun_general_default_route_b16384_d32_m16": dict(
        problem="general_big_strat", method="reversible_heun", levy="none", B=16384, d=32, m=16, nsteps=1000, dt=2.0 ** -10,
        kid=13, trajectory=True, recognised=True, mfma_flops_per_traj_step=2 * (32 * 64 + 64 * 32 + 32 * 64 + 64 * 32 * 16),
        kernel="tsde_rheun_mlp_forward<32, 64, general m = 16> (neural_rheun_kernel: one pass of both nets per step, two "
               "contractions)"),
    "c3_rheun_general_b16384_d32_m16": dict(
        problem="general_big_strat", method="reversible_heun", levy="none", B=16384, d=32, m=16, nsteps=1000, dt=2.0 ** -10,
        bytes_per_traj_step=2 * 4 * (32 * 16 + 3 * 32), kid=2, launches_per_step=2, bench_steps=200,
        kernel="tsde_step_general_w<float> x2 per step (+ tsde_lincomb2 x3; user f_net, g_net once)"),
    "c3_rheun_adjoint_general_default_route_b16384_d32_m16": dict(
        problem="general_big_strat", method="reversible_heun", adjoint_method="adjoint_reversible_heun", levy="none", B=16384,
        d=32, m=16, nsteps=1000, dt=2.0 ** -10, kid=13, trajectory=True, recognised=True, adjoint=True,
        mfma_flops_per_traj_step=3 * 2 * (32 * 64 + 64 * 32 + 32 * 64 + 64 * 32 * 16),
        kernel="tsde_rheun_mlp_forward + _backward<32, 64, general m = 16> (forward pass + reconstruction pass with the "
               "transposed products: 3 passes' worth of flops per trajectory-step)"),
    "c3_rheun_adjoint_general_b16384_d32_m16": dict(
        problem="general_big_strat", method="reversible_heun", adjoint_method="adjoint_reversible_heun", levy="none", B=16384,
        d=32, m=16, nsteps=1000, dt=2.0 ** -10, bytes_per_traj_step=0, kid=2, launches_per_step=2, adjoint=True, bench_steps=50,
        eager=True, kernel="stepwise pair: tsde_step_general_w + torch outer products + autograd VJPs of the user's nets"),
}


def make_problem(name, d, m, dev):
    from . import problems
    if name == "stochastic_lorenz":
        return problems.StochasticLorenz()
    if name == "sdegan_generator":
        return problems.SdeGanGenerator(noise_size=m, hidden_size=d).to(dev)
    if name == "latent_diag_strat":
        return problems.LatentDiagStrat(d).to(dev)
    if name == "latent_diag_logqp":
        return problems.LatentDiagLogqp(d).to(dev)
    if name == "general_big":      # NeuralGeneral-style: hidden 64
        return problems.MLPGeneral(d, m, "ito", hidden=64).to(dev)
    if name == "netadditive_big":
        return problems.MLPNetAdditive(d, m, "ito", hidden=64).to(dev)
    if name == "netdiag_big":
        return problems.MLPNetDiag(d, "ito", hidden=64).to(dev)
    if name == "general_big_strat":
        return problems.MLPGeneral(d, m, "stratonovich", hidden=64).to(dev)
    if name == "latent_diag":      # latent-SDE-style diagonal SDE
        return problems.LatentDiag(d).to(dev)
    if name == "double_well":
        return problems.DoubleWell(d).to(dev)
    if name == "scheduled_diag":
        return problems.ScheduledDiag(d).to(dev)
    if name == "exp_diffusion":    # the reference's own benchmark SDE (benchmarks/brownian.py:131-139): f = y, g = exp(-y)
        return problems.ExpDiffusion().to(dev)
    if name == "exp_diffusion_closed_form":
        import torchsde_amd
        return torchsde_amd.ElementwiseDiagonalSDE("identity", "exp", (1.0, 1.0, 0.0, 0.0), (1.0, -1.0, 0.0, 0.0),
                                                   dtype=torch.float32).to(dev)
    if name == "latent_diag_closed_form":
        # the SAME SDE as "latent_diag" (same parameter values), stated as the closed-form module the trajectory kernels
        # take: drift Linear-Softplus-Linear, diffusion 0.1 * sigmoid(w * y + b)
        import torchsde_amd
        latent = make_problem("latent_diag", d, m, "cpu")
        sde = torchsde_amd.MLPDriftDiagonalSDE(d, d, activation="softplus", diffusion="sigmoid", diff_scale=0.1,
                                               diff_rate=latent.w.detach(), diff_shift=latent.b.detach())
        with torch.no_grad():
            for dst, src in ((sde.lin1, latent.net[0]), (sde.lin2, latent.net[2])):
                dst.weight.copy_(src.weight)
                dst.bias.copy_(src.bias)
        return sde.to(dev)
    if name == "mlp_drift":
        import torchsde_amd
        torch.manual_seed(0)
        return torchsde_amd.MLPDriftDiagonalSDE(d, 128, activation="softplus", diff_rate=0.0, diff_shift=0.1).to(dev)
    if name.startswith("gbm_closed_form"):
        import torchsde_amd
        strat = name.endswith("_strat")
        gbm = problems.make("gbm_strat" if strat else "gbm_ito", d=d, m=m)
        mu, sigma = gbm.mu.detach(), gbm.sigma.detach()
        rate = mu - 0.5 * sigma ** 2 if strat else mu
        return torchsde_amd.AffineDiagonalSDE(rate, 0.0, sigma, 0.0, sde_type="stratonovich" if strat else "ito",
                                              dtype=torch.float32).to(dev)
    return problems.make(name, d=d, m=m).to(dev)
