/*
 * ndp_nmpc.h -- C-ABI of the MI355X-native batched NMPC + downwash control step.
 *
 * Drop-in boundary for the hot path of Li-Jinjie/ndp_nmpc_qd.  In the reference this
 * boundary is acados_template's ctypes binding of the generated solver
 * (AcadosOcpSolver.set / get / solve_for_x0, used at
 *  ndp_nmpc/scripts/nmpc_ctl/nmpc_body_rate_ctl.py:84-112 and
 *  ndp_nmpc/scripts/ndp_nmpc_ctl/ndp_nmpc_body_rate_ctl.py:82-112)
 * plus torch's CUDA module call in dnwash_nn_est/downwash_nn.py:21-29.
 * Each entry point below names the reference interface it replaces.
 *
 * Conventions
 *  - plain C, no C++ types, no exceptions across the boundary;
 *  - every function returns 0 on success, >0 = solver status of the worst instance
 *    (acados ints: 1 NaN, 4 QP failure), <0 = API misuse or HIP error
 *    (text via ndp_last_error);
 *  - host arrays are row-major, batch-major: x0[B][10], xr[B][N+1][10], ur[B][N][4],
 *    f[B][N+1][3] (fp32), other[B][N+1][10], ego_xy[B][2], u0[B][4];
 *    state order [px,py,pz,vx,vy,vz,qw,qx,qy,qz] (nmpc_body_rate_ctl.py:130),
 *    control order [wx,wy,wz,c] (:144);
 *  - the caller owns every pointer for the duration of the call only; the library owns
 *    device memory and the persistent SQP iterate (the warm start acados keeps);
 *  - a handle is internally serialised (mutex): update / reset / get may be called from
 *    different threads, as rospy does (nmpc_node.py:94,152,237);
 *  - *_device variants take device pointers (HBM-resident inputs) and a hipStream_t
 *    passed as void*; they enqueue work and do not synchronise.
 */
#ifndef NDP_NMPC_H
#define NDP_NMPC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NDP_NX 10
#define NDP_NU 4
#define NDP_MLP_NPARAM 17859 /* 6-128-64-128-3 with biases, nn_net.py:7-18 */

/* Version of this interface: bumped whenever an exported signature or the layout of ndp_cfg changes.  A binding built against
 * another header must refuse to run: ndp_abi_version() is what the loaded library was built with, ndp_cfg_size() its
 * sizeof(ndp_cfg) (ndp_create / ndp_default_cfg read and write that many bytes of the caller's struct). */
#define NDP_ABI_VERSION 7
int ndp_abi_version(void);
size_t ndp_cfg_size(void);

#define NDP_QP_AUTO 0       /* exact early exit when no bound is active, else interior point */
#define NDP_QP_IPM_ALWAYS 1 /* always run the interior-point loop (what HPIPM does) */

#define NDP_PREC_F64 0        /* product path: v_mfma_f64_16x16x4_f64 */
#define NDP_PREC_F32_STUDY 1  /* f64 kernel, operands of the sweeps' matrix instructions rounded to fp32 (numerics only) */
#define NDP_PREC_BF16_STUDY 2 /* ... rounded to bf16 */
#define NDP_PREC_F32_MFMA 3   /* the Riccati sweeps on v_mfma_f32_16x16x4_f32 */
#define NDP_PREC_BF16_MFMA 4  /* the Riccati sweeps on v_mfma_f32_16x16x16_bf16 (bf16 in, fp32 accumulate) */
#define NDP_PREC_COND_F32 5   /* BASELINE configs[4] as worded -- "fp32 vs bf16 MFMA on the CONDENSED QP": the first solve of every QP in condensed
                               * form (prediction matrices, H = R + Gamma' Q Gamma and the gradient as tiled products on v_mfma_f32_16x16x4_f32, fp32
                               * Cholesky in LDS, csrc/cond_qp.hpp); kept when it passes the fp64 inside-the-box test, else the fp64 Riccati path
                               * solves the QP.  A STUDY mode (the reference does not condense: qp_solver_cond_N = N): N a multiple of 4, <= 40 */
#define NDP_PREC_COND_BF16 6  /* ... the products on v_mfma_f32_16x16x16_bf16 (fp32 accumulate, fp32 Cholesky) */

typedef struct ndp_cfg {
    int32_t batch;      /* B: independent OCP instances in this handle            */
    int32_t N;          /* shooting intervals        params/nmpc_params.py:9      */
    int32_t n_rti;      /* SQP-RTI iterations per step (reference: 1)             */
    int32_t use_fd;     /* 1 = NDP model, disturbance force in p (ndp_...ctl.py:146-162) */
    int32_t qp_mode;    /* NDP_QP_*                                               */
    int32_t iter_max;   /* interior-point iteration cap (acados default 50)       */
    int32_t device;     /* HIP device ordinal                                     */
    int32_t qp_precision; /* NDP_PREC_*: 0 = product path (fp64).  BASELINE config 5 ("fp32 vs bf16 MFMA on the QP") only:
                           * 3 / 4 run the Riccati sweeps on the fp32 / bf16-input matrix instructions (everything else stays
                           * fp64); 1 / 2 are the first round's operand-rounding studies on the fp64 kernel */
    int32_t work_queue; /* instances whose QP needs the interior-point loop are re-distributed over all SIMDs through an
                         * work list (producer + consumer launch per step): 0 = automatic, 1 = on, 2 = off.  Automatic: the N = 40 /
                         * 2-iteration shape always takes the list; the reference shape (N = 20, 1 iteration) with qp_mode AUTO and a
                         * batch >= 2 x the device's SIMD count starts in place and switches by what its steps do -- the list on when
                         * >= 4 % of the instances of a window of 8 steps went into the interior-point loop, off at <= 1.5 % (the list
                         * costs a step that lists nothing 8-15 %, and gains 70 % when a fifth of the instances iterate); decided at
                         * launch time, so a captured graph keeps its form */
    int32_t ipm_refine; /* interior point: while a STATE bound's barrier term lambda / t exceeds refine_gamma, the 4x4 block is
                         * factorised L D L' and applied by substitution (an explicit inverse costs the recursion its definiteness
                         * there) and every Newton system's solution is refined this many times with the factorisation at hand
                         * (default 2; 0 = never: round 3's loop).  The loop is in absolute form, so an iteration's answer is as
                         * accurate as its last solve -- cond ~ lambda / t.
                         * The reference's velocity box (+-20 m/s, nmpc_body_rate_ctl.py:59-61) is never active in its envelope:
                         * this only acts on boxes shrunk on purpose.
                         * WHERE IT ACTS: the three-slot kernels only (N <= 27: run-time and compile-time horizons alike), in place and
                         * through the work list.  The five-slot kernels (N >= 28, e.g. config 5's N = 40) and the lean late-force step
                         * (ndp_step_device_prefetched) carry no stiff sweep and no second solve -- they sit at the register limit.
                         * ndp_create REFUSES ipm_refine > 0 for a shape whose kernels have no such path (N >= 28, or qp_precision != 0:
                         * error -2; set ipm_refine = 0 -- a strongly active state bound at a tight tolerance then ends in status 4,
                         * never in a silent answer).  The lean late-force step of an N <= 27 handle runs without it as well
                         * (ndp_refine_active() = what the handle's ordinary steps do). */
    double dt;          /* T_horizon / N_node        params/nmpc_params.py:10,12  */
    double mass;        /* params/fhnp_params.py:9   */
    double gravity;     /* params/fhnp_params.py:12  */
    double r_horiz;     /* params/downwash_params.py:10 */
    double Qd[10];      /* diag(Q)   nmpc_body_rate_ctl.py:48 */
    double Rd[4];       /* diag(R)   nmpc_body_rate_ctl.py:49 */
    double lbu[4], ubu[4]; /* nmpc_body_rate_ctl.py:56-58 */
    double lbv[3], ubv[3]; /* nmpc_body_rate_ctl.py:59-61 (stages 1..N-1) */
    double mu0, thr0, tol, tau; /* interior-point constants */
    double auto_margin;         /* NDP_QP_AUTO accepts the equality-constrained minimiser only if it is this far inside every
                                 * bound (default 0.1); closer to a bound the interior-point loop runs, as in the reference */
    double ts_nmpc;             /* control period, params/nmpc_params.py:11 (0.02): spacing of the reference list entries */
    double mu_floor;            /* interior point: the centring target sigma*mu never goes below mu_floor * tol (default 0.1) --
                                 * slacks are differences, so driving mu far below tol only loses digits */
    double refine_gamma;        /* see ipm_refine (default 1e4) */
    /* NDP_QP_AUTO, active set on the INPUT bounds (nmpc_body_rate_ctl.py:56-58 -- the only bounds ever active in the reference's
     * envelope): when the equality-constrained minimiser leaves the box, inputs beyond a bound are pinned there, pins whose
     * multiplier has the wrong sign are released, and the QP is solved again -- ONE Riccati sweep per iteration -- until the set
     * reproduces itself; the KKT conditions of the box-constrained QP then hold: it IS the solution HPIPM iterates towards
     * (any method converging the same strictly convex QP is parity-equivalent).  The set is kept per instance between control steps
     * (the warm start the reference leaves off, nmpc_body_rate_ctl.py:73-74): a step whose set still holds costs one sweep.
     * as_iter_max: sweeps with pins allowed behind the first one (default 8; 0 = off: rounds 1-5's behaviour, early exit only
     * auto_margin inside every bound, else the interior-point loop).  A violated VELOCITY bound, a set that does not settle, a failed
     * factorisation: the interior-point loop takes the QP, as before.  ndp_reset / ndp_set_iterate empty the kept sets.
     * as_gamma: the weight that holds a pinned input on its bound (default 1e12: the input is then within multiplier / 1e12 of
     * the bound and set onto it exactly). */
    int32_t as_iter_max;
    int32_t reserved0;
    double as_gamma;
} ndp_cfg;

typedef struct ndp_handle ndp_handle;

/* Fills cfg with the reference constants (the params package), batch = 1, N = 20. */
int ndp_default_cfg(ndp_cfg *cfg);

/* Replaces AcadosOcpSolver(ocp, json_file, build=...) (nmpc_body_rate_ctl.py:84):
 * allocates device state for cfg->batch instances.  Fails (<0) when no HIP device is usable. */
int ndp_create(const ndp_cfg *cfg, ndp_handle **out);
int ndp_destroy(ndp_handle *h);
const char *ndp_last_error(const ndp_handle *h); /* h may be NULL: last create error */

/* Replaces nn_model.load_state_dict(torch.load(...)) (downwash_nn.py:14-16).
 * blob: W1[128][6] b1 W2[64][128] b2 W3[128][64] b3 W4[3][128] b4, fp32, n = NDP_MLP_NPARAM. */
int ndp_set_mlp_weights(ndp_handle *h, const float *blob, size_t n);

/* Replaces NMPCBodyRateController.reset (nmpc_body_rate_ctl.py:86-91):
 * iterate x_k := xr[k] (k=0..N), u_k := ur[k] (k=0..N-1), for every instance. */
int ndp_reset(ndp_handle *h, const double *xr, const double *ur);
int ndp_reset_device(ndp_handle *h, const void *d_xr, const void *d_ur, void *stream);

/* Replaces update(x0, xr, ur[, f]) = 2N+2 solver.set calls + solver.solve_for_x0(x0)
 * (nmpc_body_rate_ctl.py:93-112; ndp_nmpc_body_rate_ctl.py:91-112), batched.
 *   f      : [B][N+1][3] fp32 disturbance force, or NULL            (NDP: p_k[4:7])
 *   other  : [B][N+1][10] neighbour reference window, or NULL.  When given (and f is NULL,
 *            use_fd = 1) the force is predicted on the device exactly as
 *            NDPLeaderNode.sub_xf_pred_callback does (ndp_nmpc_leader_node.py:60-76):
 *            f = MLP((other - xr)[:, 0:6]) if |other[0].xy - ego_xy|^2 < r_horiz^2 else 0
 *   ego_xy : [B][2] ego odometry xy for that gate, or NULL = gate always open
 *   u0     : [B][4] out, u_0 after the full step
 * Returns the worst per-instance status (0 = all converged). */
int ndp_step(ndp_handle *h, const double *x0, const double *xr, const double *ur, const float *f,
             const double *other, const double *ego_xy, double *u0);
int ndp_step_device(ndp_handle *h, const void *d_x0, const void *d_xr, const void *d_ur, const void *d_f,
                    const void *d_other, const void *d_ego_xy, void *d_u0, void *stream);
/* ndp_step plus, from the same call (one synchronisation, no further round trips), what the reference's callers read
 * after solve_for_x0: the new iterate (solver.get(i,"x"/"u"), nmpc_node.py:237), solver.status
 * (nmpc_body_rate_ctl.py:109) and the interior-point iterations.  Any of the four output pointers may be NULL.
 *
 * How a host-array step moves its data (both forms; the reference's call shape is host arrays in, host array out,
 * nmpc_body_rate_ctl.py:93-112).  The handle owns two slots of page-locked host mirrors (allocated by the first host step).
 * The caller's arrays are packed into a slot's input mirror by the handle's pack threads (NDP_PACK_THREADS, default: half
 * the hardware threads, at most 8, beside the caller); the kernel reads that mirror over PCIe and writes u0 / status /
 * iterations -- and a copy of the new iterate when X_out / U_out are given -- into the slot's output mirror itself: one
 * launch and one wait per call, no DMA operation, at every batch size.
 * The persistent iterate itself always lives in HBM (ndp_device_iterate_x / _u), whatever the batch size. */
int ndp_step_ex(ndp_handle *h, const double *x0, const double *xr, const double *ur, const float *f,
                const double *other, const double *ego_xy, double *u0, double *X_out, double *U_out,
                int32_t *status_out, int32_t *ipm_iters_out);
/* ndp_step_ex with the disturbance force in DOUBLE precision, f[B][N+1][3] fp64 (or NULL).  The reference concatenates the force
 * into a float64 parameter vector (p_k = [q_r, f_k], ndp_nmpc_body_rate_ctl.py:97-99) and hands that to acados: its own caller
 * only ever supplies DownwashNN's fp32 values (downwash_nn.py:28), for which this call and ndp_step_ex agree bit
 * for bit, but any other caller of the same Python API gets its float64 force to the last digit only through this entry (the
 * fp32 form would round it: 6e-8 relative).  The drop-in class's .solver facade always takes this one. */
int ndp_step_ex_f64(ndp_handle *h, const double *x0, const double *xr, const double *ur, const double *f, double *u0,
                    double *X_out, double *U_out, int32_t *status_out, int32_t *ipm_iters_out);
/* The same step in two halves, so that a caller can keep two control ticks in flight: ndp_step_begin packs the inputs into a
 * slot's page-locked mirror, enqueues ONE launch -- whose waves read that mirror over PCIe and write u0 / status / iterations into
 * the slot's page-locked output block themselves (zero-copy: no H2D / D2H copy operation) -- and returns without waiting;
 * ndp_step_end waits for the OLDEST begun step and hands its results over.  With a second ndp_step_begin issued before the first
 * ndp_step_end, the packing of tick i+1 runs while tick i's kernel does -- the reference's control loop has that freedom too: x0
 * comes from odometry, not from the previous u0 (nmpc_node.py:202-226).  At most two steps may be in flight (a third
 * ndp_step_begin returns -14); steps complete in order; the warm start of tick i+1 is tick i's iterate, as always (stream order).
 *   flags bit 0: ndp_step_end will be asked for the iterate (X_out / U_out) of this step.
 * Every begun step must be drained with ndp_step_end BEFORE ndp_reset / ndp_set_iterate (they wait for the device, but the
 * results of a step still in flight would be lost and its slot stays busy) and before ndp_destroy (which frees the mirrors a
 * running kernel reads and writes: it synchronises the handle's stream first, but steps begun on it are the caller's to finish).
 * ndp_step / ndp_step_ex = begin + end under one lock. */
int ndp_step_begin(ndp_handle *h, const double *x0, const double *xr, const double *ur, const float *f,
                   const double *other, const double *ego_xy, int flags);
int ndp_step_end(ndp_handle *h, double *u0, double *X_out, double *U_out, int32_t *status_out, int32_t *ipm_iters_out);
/* ndp_step_device with the neighbour windows addressed the way a multi-GPU exchange leaves them (the reference's PredXU
 * traffic, nmpc_node.py:116-133 -> ndp_nmpc_leader_node.py:60-76):
 *   other_stride   : doubles per node of d_other, 10 (full windows) or 6 (positions + velocities, all the gate and the MLP
 *                    read, downwash_nn.py:22) -- d_other is [rows][N+1][other_stride]
 *   d_other_index  : int32[B], row of d_other holding instance i's neighbour, < 0 = no neighbour (force 0: the plain NMPC
 *                    followers of a formation, three_qd_ndp_nmpc.launch:8,12); NULL = row i */
int ndp_step_device_ex(ndp_handle *h, const void *d_x0, const void *d_xr, const void *d_ur, const void *d_f,
                       const void *d_other, int other_stride, const void *d_other_index, const void *d_ego_xy,
                       void *d_u0, void *stream);

/* The downwash one tick ahead, beside the control step.  In the reference the network does not run inside the control loop: a
 * subscriber callback predicts the force when a neighbour message arrives (ndp_nmpc_leader_node.py:60-76) and the next control tick
 * uses the latest one (nmpc_node.py:202-209).  The same split on the device, as two launches that overlap:
 *   ndp_downwash_prefetch_device : gate + MLP for the NEXT control step on the handle's second stream (LDS-free kernel, resident
 *                                  beside the control-step kernel); d_other / other_stride / d_other_index / d_ego_xy as in
 *                                  ndp_step_device_ex, d_ego_ref = that tick's xr.  after_stream: a stream whose work so far (the
 *                                  producer of those windows) it must follow, or NULL; inside a stream capture pass the
 *                                  capturing stream on the first call (fork), NULL afterwards, and ndp_prefetch_join at the end.
 *                                  on_stream: NULL = the handle's second stream, or a stream of the caller's to launch on
 *                                  instead (it must not share a hardware queue with the control steps' stream: create it at
 *                                  another priority).  The two chains of launches order themselves through device-side tick
 *                                  words, so they may also be captured into two SEPARATE graphs and replayed side by side --
 *                                  measured on ROCm 7.2: parallel branches inside ONE hipGraph run one after the other.
 *   ndp_step_device_prefetched   : the control step that consumes the oldest unconsumed prediction: linearises without the
 *                                  force, takes it when it is there (normally long before) and corrects the dynamics defects --
 *                                  the force enters them additively.  One prediction per step, in order; at most two
 *                                  predictions may be ahead (two force slots).  A prediction that does not arrive within 100 ms
 *                                  (no matching prefetch call) = zero force, per-instance status 5, counted.
 *   ndp_prefetch_join            : makes `stream` wait for the second stream's work so far (end of a capture, or before reading
 *                                  the slots).
 *   ndp_prefetch_stats           : synchronises; out5 = {predictions completed, steps that took theirs, control-step waves whose
 *                                  wait timed out, downwash launches whose wait for a free slot timed out, control-step waves
 *                                  that started before their prediction was complete (they wait for it and read it past the L2)}.
 *   ndp_device_force_slot        : the two force slots, [B][N+1][3] fp32 (prediction m is in slot m & 1).
 * Not combined with the interior-point work list (cfg.work_queue = 2) and fp64 only. */
int ndp_downwash_prefetch_device(ndp_handle *h, const void *d_other, int other_stride, const void *d_other_index,
                                 const void *d_ego_ref, const void *d_ego_xy, void *after_stream, void *on_stream);
int ndp_step_device_prefetched(ndp_handle *h, const void *d_x0, const void *d_xr, const void *d_ur, void *d_u0, void *stream);
int ndp_prefetch_join(ndp_handle *h, void *stream);
int ndp_prefetch_stats(ndp_handle *h, unsigned long long *out5);
void *ndp_device_force_slot(ndp_handle *h, int slot);

/* Replaces DownwashNN.update(other_pred_x, ego_pred_x) (downwash_nn.py:21-29), batched, with the
 * optional r_horiz gate.  f_out: [B][N+1][3] fp32. */
int ndp_downwash(ndp_handle *h, const double *other, const double *ego_ref, const double *ego_xy, float *f_out);
int ndp_downwash_device(ndp_handle *h, const void *d_other, const void *d_ego_ref, const void *d_ego_xy,
                        void *d_f_out, void *stream);

/* Replaces solver.get(i,"x"/"u") / solver.set(i,"x"/"u") (nmpc_node.py:237; nmpc_body_rate_ctl.py:88-91):
 * whole iterate of every instance, X[B][N+1][10], U[B][N][4]. */
int ndp_get_iterate(ndp_handle *h, double *X, double *U);
int ndp_set_iterate(ndp_handle *h, const double *X, const double *U);

/* Replaces solver.status (nmpc_body_rate_ctl.py:109): per-instance status of the last step and the
 * interior-point iterations it took (0 = early exit).  Either pointer may be NULL. */
int ndp_get_status(ndp_handle *h, int32_t *status, int32_t *ipm_iters);

/* What NDP_QP_AUTO's active-set iterations did in the last step (cfg.as_iter_max; no counterpart in the reference, whose HPIPM is
 * cold-started every call, nmpc_body_rate_ctl.py:71-74): sweeps[B] = Riccati sweeps the step's QPs took before the interior-point
 * loop, if that ran at all (1 = the first solve's set held); act[B][4N] = the set kept for the next step, element 4k + i = input i of
 * stage k: +1 on its upper bound, -1 on its lower, 0 free.  Either pointer may be NULL. */
int ndp_get_active_set(ndp_handle *h, int32_t *sweeps, int8_t *act);
/* The kept set handed in (same layout; entries -1, 0, +1): the warm start of the next step's QPs -- an instance that moves to another
 * handle / rank takes its set along with its iterate (ndp_set_iterate empties the set: call this after it).  A set that does not fit the
 * next QP costs sweeps, never the answer: the iterations end in a set that reproduces itself or in the interior-point loop. */
int ndp_set_active_set(ndp_handle *h, const int8_t *act);

/* Device views for callers that keep everything in HBM (bench, multi-GPU driver).  The getters above and
 * ndp_synchronize also wait for the last stream a *_device call was given. */
void *ndp_device_iterate_x(ndp_handle *h);
void *ndp_device_iterate_u(ndp_handle *h);
void *ndp_device_force(ndp_handle *h);   /* [B][N+1][3] fp32 written by the fused downwash */
int ndp_synchronize(ndp_handle *h);
int ndp_refine_active(ndp_handle *h);        /* 1: cfg.ipm_refine > 0 and this handle's control steps carry the refinement path (see ndp_cfg.ipm_refine); 0: ignored */
int ndp_work_queue_enabled(ndp_handle *h);   /* 1 if the handle's NEXT step runs the interior-point work list (cfg.work_queue; the automatic rule's current state) */

/* Per-kernel timing with HIP events recorded on the stream each kernel is launched on:
 * on = n > 0 brackets every n-th launch of each kernel (n = 1: every launch), on = 0 stops and clears.
 * ndp_timing_read: name is "rti" or "mlp"; total over the bracketed launches.  Returns <0 if nothing was timed. */
int ndp_timing_enable(ndp_handle *h, int on);
int ndp_timing_read(ndp_handle *h, const char *name, double *total_ms, int64_t *launches);

/* ---- "next" row f3: the step after the path (hover-throttle estimator + actuator command), batched.
 * Replaces, per vehicle, HoverThrottleEstimator(ts).update(vz, throttle) -> k_throttle
 * (hv_throttle_est/hover_throttle_estimator.py:15-53, differentiator.py:10-23, params/estimator_params.py:13-18;
 *  called from nmpc_node.py:251-253) and nmpc_u_2_att_tgt's thrust = c * mass / k_throttle (nmpc_node.py:273-283).
 * One estimator per instance of the handle; state lives on the device.
 *   ndp_throttle_reset : x = [0, k_init], P = I, differentiator cleared
 *   ndp_throttle_update: vz[B], throttle[B] (the thrust command sent last tick) -> k_throttle[B]
 *   ndp_actuator_cmd   : u0[B][4] = [wx,wy,wz,c] -> cmd[B][4] = [wx,wy,wz, c*mass/k_throttle (0 if k == 0)] */
int ndp_throttle_reset(ndp_handle *h);
int ndp_throttle_update(ndp_handle *h, const double *vz, const double *throttle, double *k_throttle);
int ndp_throttle_update_device(ndp_handle *h, const void *d_vz, const void *d_throttle, void *d_k_throttle, void *stream);
int ndp_actuator_cmd(ndp_handle *h, const double *u0, const double *k_throttle, double *cmd);
int ndp_actuator_cmd_device(ndp_handle *h, const void *d_u0, const void *d_k_throttle, void *d_cmd, void *stream);
int ndp_throttle_get_state(ndp_handle *h, double *state /* [B][8]: x0 x1 P00 P01 P10 P11 vz_prev az_prev */);

/* ---- "next" row f2: the step beside the path -- the follower's reference relay (nmpc_follower_node.py:44-77).
 *   ndp_relay_formation: one formation_ref message per instance, form[B][3]; three AlphaFilters (alpha 0.8,
 *                        y0 = first message; hv_throttle_est/alpha_filter.py:11-20) -> filtered offset, kept on the device
 *   ndp_relay_reference: leader windows xr_lead[B][N+1][10] -> follower reference xr_out (x[:,0:3] += offset);
 *                        the leader's ur is used unchanged (:72-74), so it needs no copy
 *   ndp_relay_reset    : filters back to "no message seen" */
int ndp_relay_reset(ndp_handle *h);
int ndp_relay_formation(ndp_handle *h, const double *form, double *offset_out /* [B][3] or NULL */);
int ndp_relay_reference(ndp_handle *h, const double *xr_lead, double *xr_out);
int ndp_relay_reference_device(ndp_handle *h, const void *d_xr_lead, void *d_xr_out, void *stream);

/* ---- "next" row f1: the step before the path -- reference window generation, batched.
 * Replaces, per vehicle, NMPCRefPublisher.reset(traj_coeff, ros_t) + get_nmpc_pts(ros_t)
 * (pt_pub/pt_publisher.py:57-103): trajectory point from the piecewise polynomials (pt_pub/base_pt_publisher.py:81-133,
 * polym_optimizer.py:104-139), differential flatness (pt_publisher.py:188-248), x/u packing (:115-146), and the
 * 0.1 s node spacing of params/nmpc_params.py:40-43.  Past the end of the trajectory every node hovers at final_pt.
 *   ndp_ref_set_trajectory: the arrays of TrajCoefficients.msg for every instance, all with n_seg segments:
 *       coeff_x/y/z[B][n_seg*8] (minimum snap), coeff_yaw[B][n_seg*4] (minimum acceleration),
 *       time_cum[B][n_seg+1], time_seg[B][n_seg], final_pt[B][3]
 *   ndp_ref_window        : t[B] = trajectory time of node 0 ((ros_t - start_ros_t).to_sec()) ->
 *       xr[B][N+1][10], ur[B][N][4], node k at t + k*dt.  The device form writes buffers that ndp_step_device reads.
 *   Accuracy / range contract of the flatness map on the device: reciprocals and the thrust norm by hardware seeds + two Newton steps,
 *   sin / cos of the yaw by Cody-Waite reduction -- within a few 1e-16 of the Python publisher (held to 1e-12 by
 *   tests/test_ref_window_row.py), for FINITE yaw with |yaw| < 1e5 rad; a trajectory whose yaw polynomial leaves that range (or is
 *   not finite) is outside what these calls serve. */
int ndp_ref_set_trajectory(ndp_handle *h, int n_seg, const double *coeff_x, const double *coeff_y, const double *coeff_z,
                           const double *coeff_yaw, const double *time_cum, const double *time_seg, const double *final_pt);
int ndp_ref_window(ndp_handle *h, const double *t, double *xr, double *ur);
int ndp_ref_window_device(ndp_handle *h, const void *d_t, void *d_xr, void *d_ur, void *stream);

/* The reference's own bookkeeping of that window (NMPCRefPublisher, pt_pub/pt_publisher.py:36-103): a list of 5N+1 reference
 * points per vehicle, ts_nmpc apart; every control tick drops the oldest entry and appends the point at ros_t + T_horizon;
 * the controller's window is every 5th entry (params/nmpc_params.py:40-43).  Kept on the device as a ring per vehicle.
 *   ndp_ref_list_reset  : _gen_long_list_w_traj (:62-76) from the trajectory of ndp_ref_set_trajectory: points at
 *                         i*ts_nmpc, i = 0..5N-1, the first one duplicated in front
 *   ndp_ref_list_fix_pt : gen_fix_pt_ref (:40-55): every entry = x_odom[B][10], u = [0, 0, 0, mass*g] (quirk_b1 = 1: the
 *                         reference's value although u[3] is an acceleration) or [0, 0, 0, g] (quirk_b1 = 0)
 *   ndp_ref_list_window : t[B] = (ros_t - start_ros_t).to_sec(): get_nmpc_pts (:79-97) = pop, append the point at
 *                         t + T_horizon, return the window; t = NULL: get_nmpc_ref_from_long_list only (:99-103)
 *   *_device            : the two halves separately, on device buffers.  The list position is host state baked into each
 *                         launch's arguments: these calls must NOT be captured into a hipGraph (a replay would reuse the
 *                         position of the captured tick); ndp_ref_window_device has no such state and is capturable.
 * Device layout: phase-major, every entry stored twice, so that every window is contiguous (csrc/ndp_hip.hip: RingGeom) -- the
 * stand-alone window call is a dense copy and ndp_tick's control step reads its windows in place. */
int ndp_ref_list_reset(ndp_handle *h);
int ndp_ref_list_fix_pt(ndp_handle *h, const double *x_odom, int quirk_b1);
int ndp_ref_list_window(ndp_handle *h, const double *t, double *xr, double *ur);
int ndp_ref_list_advance_device(ndp_handle *h, const void *d_t, void *stream);
int ndp_ref_list_window_device(ndp_handle *h, void *d_xr, void *d_ur, void *stream);

/* ---- The node's control tick, end to end on the device: odometry in, actuator command out, references resident.
 * Replaces, per vehicle and control period, ControllerNode.nmpc_callback + hover_throttle_callback (nmpc_node.py:211-231,251-253):
 *     nmpc_x_ref, nmpc_u_ref = ref_pub.get_nmpc_pts(now)        (:162; pt_pub/pt_publisher.py:78-103)   -- f1, the list on the device
 *     k_throttle = hv_th_estimator.update(vz, body_rate_cmd.thrust)   (:251-253)                         -- f3
 *     u0 = nmpc_ctl.update(x0, nmpc_x_ref, nmpc_u_ref[, disturb_force])   (:202-209,223-224)            -- a5 / a6, a7, a8
 *     body_rate_cmd = nmpc_u_2_att_tgt(*u0)                    (:225,273-283)                            -- f3
 * with the downwash force predicted from the NEIGHBOUR's window of the same tick (ndp_nmpc_leader_node.py:60-76; the neighbour
 * is another instance of this handle: its window is read out of its list, nothing is copied) and gated on the ego ODOMETRY xy
 * (:65-74).  Only what is new information crosses PCIe: 80 bytes of odometry per vehicle and tick (+ 8 each for t,
 * vz, throttle when given) in, 32 bytes of command (+ 4 status) out -- against 3.4 KB per vehicle for ndp_step's x0 + xr + ur +
 * neighbour columns.  ONE launch per tick at the reference configuration (N = 20, 1 RTI iteration): the control step's wave first
 * makes its vehicle's newest list entry -- node N of its window -- and the neighbour's, runs the estimator, reads the rest of both
 * windows straight out of the list (the phase-major layout makes every window contiguous), and writes the actuator command itself,
 * beside u0.  Other shapes: a small launch (tick_pre_kernel: list advance + estimator) in front of the control step.
 *   ndp_tick_config : other_index[B] = the instance whose window is vehicle i's neighbour (< 0: none, plain NMPC vehicle), or NULL
 *                     = no vehicle has one; gate_on_odometry = 1: the r_horiz gate of ndp_nmpc_leader_node.py:65-74, 0: always open.
 *                     Neighbours need use_fd = 1 and ndp_set_mlp_weights.
 *   ndp_tick_reset  : nmpc_ctl.reset(*ref_pub.get_nmpc_ref_from_long_list()) (nmpc_node.py:92,151-152): iterate := the list's
 *                     current window.  Call after ndp_ref_list_fix_pt (start-up) / ndp_ref_list_reset (a new trajectory).
 *   ndp_tick_begin  : x_odom[B][10] = odom_2_nmpc_x(px4_odom) of every vehicle; t[B] = (ros_t - start_ros_t).to_sec() -> the
 *                     list is advanced (get_nmpc_pts), or NULL -> the list stays (hover at the fixed point; a vehicle whose
 *                     trajectory has ended keeps being advanced: its points are final_pt); vz[B] or NULL = x_odom[:,5];
 *                     throttle[B] or NULL = the thrust this handle commanded on the previous tick (0 before the first).
 *                     flags bit 0: run the estimator this tick (the reference stops its timer while a trajectory is tracked,
 *                     nmpc_node.py:146,196); bit 1: ndp_tick_end will be asked for u0 as well; bit 2: NDP_TICK_T_UNIFORM.  Returns without waiting; at most two
 *                     ticks (or steps) in flight, drained in order, as with ndp_step_begin.
 *   ndp_tick_end    : waits for the oldest tick: cmd[B][4] = [wx, wy, wz, thrust]; u0[B][4], status[B], ipm_iters[B] or NULL.
 *                     Returns the worst status.
 *   ndp_tick        : begin + end under one lock.
 *   ndp_tick_device : the same launches on device pointers and a caller's stream; d_u0 may be NULL; status: ndp_get_status.
 *                     d_t: device-accessible [B] doubles -- EXCEPT with NDP_TICK_T_UNIFORM, where d_t is a HOST pointer to one double
 *                     that is read inside the call (before it returns), not by the device.
 *                     The list position is host state baked into the launches: not capturable into a hipGraph. */
/* The tick with neighbours on OTHER ranks / GPUs: their windows arrive through the caller's exchange (ndp_xchg_*, peer-mapped windows)
 * in d_windows[rows][N+1][stride] (stride 6: the position / velocity columns the gate and the network read; or 10); other_index[B] = the
 * row of every vehicle's neighbour (< 0: none).  A tick is then three enqueues on one stream, the exchange between them:
 *   ndp_tick_advance_device   : list advance (d_t as in ndp_tick_device; NULL: none) + estimator (flags bit 0) -- this rank's window
 *                               of the tick is complete (node N included);
 *   ndp_tick_window_pv_device : that window's columns [B][N+1][6] into the exchange's send buffer;      ... the exchange ...
 *   ndp_tick_step_device      : the control step against the exchanged windows + the actuator command.
 * Bit-equal with ndp_tick_device on the same vehicles in ONE handle.  ndp_tick_config switches back to neighbours of the same handle. */
int ndp_tick_config_remote(ndp_handle *h, const void *d_windows, int stride, int64_t rows, const int32_t *other_index, int gate_on_odometry);
int ndp_tick_advance_device(ndp_handle *h, const void *d_x_odom, const void *d_t, const void *d_vz, const void *d_throttle, int flags, void *stream);
int ndp_tick_window_pv_device(ndp_handle *h, void *d_pv, void *stream);
int ndp_tick_step_device(ndp_handle *h, const void *d_x_odom, void *d_cmd, void *d_u0, void *stream);
#define NDP_TICK_ESTIMATE 1
#define NDP_TICK_WANT_U0 2
#define NDP_TICK_T_UNIFORM 4   /* t points at ONE double (host memory, also for ndp_tick_device): the trajectory time of every vehicle
                                * (one host, one clock, the trajectories started together).  It travels in the kernel arguments: no
                                * vehicle's first load of the tick has to cross PCIe. */
int ndp_tick_config(ndp_handle *h, const int32_t *other_index, int gate_on_odometry);
int ndp_tick_reset(ndp_handle *h);
int ndp_tick_begin(ndp_handle *h, const double *x_odom, const double *t, const double *vz, const double *throttle, int flags);
int ndp_tick_end(ndp_handle *h, double *cmd, double *u0, int32_t *status_out, int32_t *ipm_iters_out);
int ndp_tick(ndp_handle *h, const double *x_odom, const double *t, const double *vz, const double *throttle, int flags,
             double *cmd, double *u0, int32_t *status_out, int32_t *ipm_iters_out);
int ndp_tick_device(ndp_handle *h, const void *d_x_odom, const void *d_t, const void *d_vz, const void *d_throttle, int flags,
                    void *d_cmd, void *d_u0, void *stream);

/* ---- "next" row f4: plant step for closed-loop rollouts on the device (dop_sim is absent from the reference).
 * x[B][10] in/out, u[B][4], f[B][3] force or NULL; RK4 with `substeps` over dt, quaternion renormalised. */
int ndp_plant_step(ndp_handle *h, double *x, const double *u, const double *f, double dt, int substeps);
int ndp_plant_step_device(ndp_handle *h, void *d_x, const void *d_u, const void *d_f, double dt, int substeps, void *stream);
/* Closed-loop rollout of every instance on the device (what nmpc_node.py's control timer + a simulator would do tick by
 * tick): reset at the trajectory's window at t0, then for k = 0..ticks-1: reference window at t0 + k*dt_tick
 * (ndp_ref_set_trajectory must have been called) -> control step from the plant state -> plant step with u0.
 * d_x[B][10] plant state in/out; d_log[ticks][B][10] receives the state after every tick, or NULL.  3 launches per tick
 * enqueued back to back on the stream; nothing returns to the host in between, nothing is synchronised. */
int ndp_rollout_device(ndp_handle *h, int ticks, double t0, double dt_tick, int substeps, void *d_x, void *d_log, void *stream);

/* ---- Peer windows: the neighbour exchange across GPUs as publish / subscribe over xGMI.
 * Replaces the PredXU topic between processes: the publisher is nmpc_node.py:116-133,229-230 (`nmpc_x_ref`, 21x10 float64, a NEW
 * message every control tick), the subscriber ndp_nmpc_leader_node.py:40,60-76 (consumes the latest one).  One process per GPU:
 * the publisher allocates its window buffer with ndp_peer_alloc and sends the 64-byte handle to the subscriber's process once
 * (any channel: torch.distributed object collectives, a socket); the subscriber maps it with ndp_peer_open.  Every control tick
 * each rank then calls ndp_peer_publish_device in front of its control-step launch (a copy launch + a one-wave launch): it writes
 * this tick's windows into one of the two slots of its own buffer, publishes the tick number (an epoch word, system scope) and
 * waits for the neighbour's epoch of the same tick; the control-step kernel launched next reads the neighbour's slot straight out of the
 * publisher's HBM (peer access over xGMI) -- no collective, no host round trip.  Writer -> reader ordering, slot reuse (the
 * reader's acknowledgement), bounded waits and their counters: csrc/peer_epoch.hpp.
 *   ndp_peer_layout : bytes of a buffer whose slots hold n_doubles each ([B][N+1][10] windows: n = B*(N+1)*10), the offset
 *                     of slot 0 and the slot stride (bytes).  The first 512 bytes are protocol words.
 *   ndp_peer_alloc  : zeroed device memory on `device` -> *ptr and its IPC handle (64 bytes).  Fine-grained (coherent between
 *                     agents while kernels run); ordinary device memory if the runtime refuses or NDP_PEER_COARSE=1 is set.
 *   ndp_peer_open   : maps another process's buffer into this process for use on `device` -> *ptr.  The mapped range is
 *                     remembered: neighbour windows (`other`) that lie inside it are read with system-scope loads by every
 *                     kernel of this library, so that they are fetched from the owner's memory, not from a stale cache line.
 *   ndp_peer_publish_device : the per-tick launch described above.  d_src: this rank's windows of the tick (n_doubles, device
 *                     memory); own_buf / nb_buf: this rank's buffer and the mapped neighbour buffer (the same pointer with one
 *                     rank); slot = parity of the tick (first tick = 1 -> slot 1, then alternating): the slot whose address
 *                     the caller hands to the control step as `other` (a mismatch with the device-side tick is counted);
 *                     timeout_us bounds each wait.  Capturable into a hipGraph (an even number of ticks per graph).
 *   ndp_peer_stats  : synchronises `device`; out4 = {ticks published, reader-acknowledgement timeouts, neighbour-epoch
 *                     timeouts (the slot was read as it was: stale), slot-parity mismatches}.
 *   ndp_peer_close / ndp_peer_free: undo open / alloc (the mapping keeps the memory alive: a reader whose publisher has
 *                     exited goes on reading the last published windows and counts epoch timeouts).
 * All return 0 or a negative error code. */
int ndp_peer_layout(size_t n_doubles, size_t *buffer_bytes, size_t *slot0_offset, size_t *slot_stride);
int ndp_peer_alloc(int device, size_t bytes, void **ptr, unsigned char *handle64);
int ndp_peer_open(int device, const unsigned char *handle64, void **ptr);
int ndp_peer_publish_device(int device, const void *d_src, size_t n_doubles, void *own_buf, void *nb_buf, int slot,
                            unsigned timeout_us, void *stream);
int ndp_peer_stats(int device, const void *own_buf, unsigned long long *out4);
int ndp_peer_close(int device, void *ptr);
int ndp_peer_free(int device, void *ptr);

/* ---- The neighbour exchange as ONE RCCL all-gather per control tick, issued by the library on a HIP stream of its own.
 * Replaces the same PredXU traffic (nmpc_node.py:116-133,229-230 -> ndp_nmpc_leader_node.py:40,60-76) with the collective the
 * build contract names: every rank contributes the position / velocity columns of its windows ([B_local][N+1][6] fp64, all the
 * gate and the network read, downwash_nn.py:22), every rank receives all of them ([world * B_local][N+1][6]; rank r's leaders
 * read the rows of rank (r+1) % world, or the rows other_index names).  RCCL is bound at run time (rccl_path: the librccl the
 * process already holds, e.g. torch's -- NULL or "" = the loader's search path); one process per GPU, one ndp_xchg per process.
 *   ndp_xchg_unique_id : rank 0 draws the communicator's id (128 bytes) and hands it to the others (any channel)
 *   ndp_xchg_create    : collective -- every rank calls it with the same id; creates the communicator, the exchange stream
 *                        (its own hardware queue) and two events
 *   ndp_xchg_begin     : packs this rank's d_xr ([rows][10] doubles, rows = B_local * (N+1)) and starts the all-gather into
 *                        d_gathered on the exchange stream, ordered behind everything `after_stream` holds so far (NULL: no
 *                        such ordering) and behind `after_event` (NULL: none; e.g. ndp_last_step_event: the last reader of
 *                        d_gathered); returns at once.  The windows are trajectory-generator output (functions of time): the gather of tick i+1 is
 *                        started before tick i's control step is launched and runs beside it (two gathered buffers).
 *   ndp_xchg_end       : `stream` waits on the device for the gather started last; the control step launched next on `stream`
 *                        may read d_gathered.  No host synchronisation anywhere; begin / end / step are capturable together
 *                        (the FIRST ndp_xchg_begin of a given size allocates the send buffer: call it once outside a capture).
 * All return 0 or a negative error code (-20 / -21: RCCL not found / symbols missing, -22: an RCCL call failed, see
 * ndp_xchg_last_error). */
typedef struct ndp_xchg ndp_xchg;
int ndp_xchg_unique_id(const char *rccl_path, unsigned char *id128);
int ndp_xchg_create(int device, int rank, int world, const unsigned char *id128, const char *rccl_path, ndp_xchg **out);
int ndp_xchg_begin(ndp_xchg *x, const void *d_xr, size_t rows, void *d_gathered, void *after_stream, void *after_event);
/* Ordering another stream behind a control step without a packet on the step's own stream: with ndp_track_steps(h, 1) the
 * completion of every control step launched for h marks an event through the dispatch packet's own completion signal
 * (hipExtLaunchKernel); ndp_last_step_event returns the event of the step launched last (valid until four more have been
 * launched).  Typical use: the all-gather of tick i+2 overwrites the gathered buffer tick i read -- it is begun behind tick i's
 * step with ndp_xchg_begin(..., NULL, that event).  (An event recorded on the compute stream instead costs two command-processor
 * packets between consecutive control steps: 38 against 29 us per tick at batch 1024.)  Tracked steps are not graph-capturable. */
int ndp_track_steps(ndp_handle *h, int on);
int ndp_last_step_event(ndp_handle *h, void **event);
int ndp_xchg_end(ndp_xchg *x, void *stream);
/* The two calls of one tick of the pipelined form in one: ndp_xchg_end(x, stream) for the gather begun last (this tick's windows), then
 * ndp_xchg_begin of the NEXT tick's windows behind the last reader of d_gathered_next -- the completion event of the control step
 * launched last for h when its steps are tracked, else (h NULL, or not tracked) everything `stream` holds so far. */
int ndp_xchg_tick(ndp_xchg *x, ndp_handle *h, void *stream, const void *d_xr_next, size_t rows, void *d_gathered_next);
/* ndp_tick with neighbours on other ranks (ndp_tick_config_remote): stage 2 and the exchange in one call -- this tick's window columns
 * of h's reference list packed into the exchange's send buffer and all-gathered into d_gathered ([world * B][N+1][6] doubles, the
 * buffer ndp_tick_config_remote was given), both on `stream` (NULL: the handle's), i.e. behind ndp_tick_advance_device and in front of
 * ndp_tick_step_device by stream order alone.  Replaces, on the reference's side, the publish of PredXU per control period
 * (nmpc_node.py:229-230) and its subscription in the neighbour's node (ndp_nmpc_leader_node.py:40,60-76). */
int ndp_xchg_tick_windows(ndp_xchg *x, ndp_handle *h, void *d_gathered, void *stream);
/* The same tick with the exchange AHEAD of the control steps: a window is a function of time alone, so the list advance of a later tick,
 * its window columns and their all-gather run on the exchange's own stream BESIDE the control steps, into one of TWO or THREE gather
 * buffers the caller cycles through (each [world * B][N+1][6]; ndp_tick_config_remote names any: the step is told which one it reads).
 * Begins and steps pair up in order (step k reads what begin k gathered); at most two begins may be ahead.  Per period, two buffers:
 *     ndp_xchg_tick_step (tick i: estimator if NDP_TICK_ESTIMATE, device-side wait for gather i, control step + command on `stream`)
 *     ndp_xchg_tick_begin(tick i+1: d_t / flags as ndp_tick_advance_device takes them -- THAT period's trajectory time; NULL: no advance)
 * with one begin in front of the first step; three buffers: step(i) then begin(i+2), two begins in front -- the gather then fills a
 * buffer whose last reader is long over and never waits for a control step (38.3 against 34.9 M solves/s on one rank).  Ordering is the library's (events on the device, nothing waits on the host);
 * with ndp_track_steps the gather waits for exactly the control step that read its buffer last.  Needs a list with >= 2 entries per
 * node spacing (the reference: 5), else -17: the serial form above serves.  Same results as the serial form and the one-handle tick. */
int ndp_xchg_tick_begin(ndp_xchg *x, ndp_handle *h, const void *d_t, int flags, void *d_gathered_next);
/* on = 1: a begin's launches on the exchange's stream (event wait, advance + columns, ncclAllGather, event record: ~15 us of host time)
 * are made by a thread the exchange owns; ndp_xchg_tick_begin then only describes them (~2 us) and ndp_xchg_tick_step waits until its
 * tick's have been made.  One host thread's launches are what bounds the remote tick one period ahead; with two the device does.  Errors
 * of that thread surface at the next begin / step.  Switch only with no gather ahead (-14 otherwise); off (the default) = the caller's thread. */
int ndp_xchg_tick_async(ndp_xchg *x, int on);
int ndp_xchg_tick_step(ndp_xchg *x, ndp_handle *h, const void *d_x_odom, const void *d_vz, const void *d_throttle, int flags,
                       void *d_cmd, void *d_u0, const void *d_gathered, void *stream);
const char *ndp_xchg_last_error(const ndp_xchg *x);
int ndp_xchg_destroy(ndp_xchg *x);

/* Test hook: number of doubles of the LDS image dump, and a step that also dumps it (B = 1 use). */
int ndp_debug_lds_doubles(int N);
/* Test hook: where things sit in that dump: out8 = {XI, MB, CB, MB stride, CB stride, image size, first stamp, 0} (doubles). */
int ndp_debug_lds_layout(int N, int *out8);
/* Test hook: one v_mfma_f64_16x16x4_f64 on caller-chosen per-lane operands a[64], b[64], c[4][64]; d has 832 doubles:
 * d[0..255] = result registers [4][64], d[256..319] = a cross-lane checksum (readlane + wave reductions),
 * d[320..383] = one v_mfma_f64_4x4x4_4b_f64 (four blocks) on a, b with accumulator c[0], d[384..639] = a after the row
 * broadcasts of lanes 0, 4, 8, 12 of every 16-lane row (DPP row_newbcast), d[640..831] = a after the row rotations by 4, 8, 12
 * lanes (DPP row_ror). */
int ndp_debug_mfma_probe(const double *a, const double *b, const double *c, double *d);
/* Test hook: the config-5 instructions through their backends.  mode 0: one v_mfma_f32_16x16x4_f32 on a[0][64], b[0][64];
 * mode 1: one v_mfma_f32_16x16x16_bf16 on four packed contraction steps a[4][64], b[4][64]; c[4][64] -> d[0..255];
 * d[256..319] = the four-lane row sum of a[0] (lanes 4 apart inside each 16-lane row). */
int ndp_debug_mfma_probe_f32(const float *a, const float *b, const float *c, float *d, int mode);
/* Profiling hook: enable = 1 makes every instance of the following steps write its 24 phase stamps (shader clock; NDP_NSTAMP in
 * rti_wave.hpp); out (or NULL) receives the [B][24] stamps of the last step before the switch is applied (0-8: the wave program's
 * phases, 9-15: the kernel's -- downwash tile, entry / exit clocks --, 16-23: the one-launch tick's prologue). */
int ndp_debug_stamps(ndp_handle *h, int enable, double *out);
/* Measurement hook: ndp_downwash_device through the LDS-free form of the downwash kernel (weights streamed from L2, at most 192
 * registers per lane: it can be resident on a CU beside the control-step kernel). */
int ndp_debug_downwash_stream_device(ndp_handle *h, const void *d_other, const void *d_ego_ref, const void *d_ego_xy,
                                     void *d_f_out, void *stream);
/* Profiling hook: where the last host-array step spent its time on the CPU, microseconds: out4 = {packing the inputs into the
 * page-locked mirror, enqueueing (launches / copies), waiting for the results, copying them out}. */
int ndp_debug_host_timing(ndp_handle *h, double *out4);
/* ... and what the host path runs on: out3 = {hardware threads of the machine, cores this process may really use (affinity mask and
 * cgroup CPU quota), pack threads the handle started (-1: no host-array step yet)}. */
int ndp_debug_host_info(ndp_handle *h, int32_t *out3);
int ndp_step_debug(ndp_handle *h, const double *x0, const double *xr, const double *ur, const float *f,
                   const double *other, const double *ego_xy, double *u0, double *lds_dump);

#ifdef __cplusplus
}
#endif
#endif
