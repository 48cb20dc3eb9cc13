/*
 * ndp_oracle.h -- CPU fp64 restatement of the ndp_nmpc_qd control step.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / the timed CPU baseline.
 *
 * PARITY UNPINNED (NMPC half): the reference's NMPC arithmetic lives in
 * acados / HPIPM / BLASFEO / CasADi-generated C, none of which is vendored in
 * /root/reference, pinned to a version (README.md:24) or installed in the
 * build image, and the reference has no tests or golden vectors.  This file
 * restates the *published algorithm* those options select
 * (nmpc_body_rate_ctl.py:71-80: SQP_RTI, ERK, GAUSS_NEWTON, HPIPM) and is
 * pinned instead by independent cross-checks in tests/ (finite differences,
 * dense KKT solves, an active-set QP solver, analytic hover answers).
 * The MLP half IS pinned: tests/golden/mlp_golden.npz was produced by
 * importing the reference's nn_net.py + shipped SN=4 weights.
 */
#ifndef NDP_ORACLE_H
#define NDP_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_NX 10
#define ORC_NU 4
#define ORC_NMAX 64

typedef struct {
    int N;          /* shooting intervals          nmpc_params.py:9  */
    int n_rti;      /* SQP-RTI iterations per call (reference: 1)    */
    double dt;      /* T_horizon / N_node          nmpc_params.py:12 */
    double mass;    /* fhnp_params.py:9  */
    double g;       /* fhnp_params.py:12 */
    double Qd[10];  /* diag of Q   nmpc_body_rate_ctl.py:48 */
    double Rd[4];   /* diag of R   nmpc_body_rate_ctl.py:49 */
    double lbu[4], ubu[4]; /* nmpc_body_rate_ctl.py:56-58 */
    double lbv[3], ubv[3]; /* nmpc_body_rate_ctl.py:59-61, stages 1..N-1 */
    int use_fd;     /* 1 = NDP variant (ndp_nmpc_body_rate_ctl.py:155-157) */
    /* interior point (HPIPM-like Mehrotra predictor-corrector) */
    double mu0, thr0, tol, tau;
    int iter_max;
    int qp_mode;        /* 1 (default) = interior point always, what HPIPM does; 0 = the device's QP_AUTO early exit */
    double auto_margin; /* qp_mode 0: accept the equality-constrained minimiser only this far inside every bound */
    double mu_floor;    /* the centring target sigma*mu never goes below mu_floor * tol (default 0.1), see orc_qp_solve */
    /* Iterative refinement of a Newton system's solution through further solves with the SAME factorisation (default 2), applied
     * only when a STATE bound's barrier term Gamma = lambda / t exceeds refine_gamma (default 1e4): the solve is then accurate to
     * cond * eps ~ Gamma * 1e-16 only, and the interior-point loop is written in absolute form (every iteration's answer IS the
     * last solve's), so that error is the answer's.  The refinement gradient is g + H z (block-wise), the correction the
     * minimiser of the same QP with that gradient, zero defects and zero initial state: see qp_solve_ws.  HPIPM's BALANCE mode
     * (nmpc_body_rate_ctl.py:72) refines its KKT solves likewise [acados-knowledge]. */
    int refine;
    double refine_gamma;
    /* qp_mode 0 only -- NOT part of the reference's algorithm (HPIPM iterates on every QP, cold-started): the device's active-set
     * iterations on the input bounds (ndp_cfg.as_iter_max, rti_wave.hpp: as_check), restated so that the device's default mode has a
     * CPU twin doing the same arithmetic.  The parity claim against the reference's QP solver is made against qp_mode 1. */
    int as_iter_max;    /* sweeps with pinned inputs allowed behind the first one (default 8; 0 = the early exit only) */
    double as_gamma;    /* weight of a pin (default 1e12) */
} orc_cfg;

typedef struct {
    int status;       /* 0 ok, 1 NaN, 4 QP failure (acados status ints) */
    int ipm_iters;    /* summed over rti iterations */
    int n_active;     /* bounds active at the last QP solution */
    double mu;        /* final complementarity */
    int as_sweeps;    /* qp_mode 0: Riccati sweeps of the active-set iterations, summed over rti iterations */
} orc_stats;

void orc_default_cfg(orc_cfg *c);

/* Sensitivity study of SURVEY A.4's [acados-knowledge] list (scripts/acados_sensitivity.py): a process-wide switch, 0 = off. */
#define ORC_VAR_TERMINAL_TIMES_DT 1 /* terminal cost scaled by the interval like the stage costs              */
#define ORC_VAR_BOUNDS_STAGE_N 2    /* the velocity box also on the terminal state                            */
#define ORC_VAR_ERK_2_STEPS 4       /* sim_method_num_steps = 2: two RK4 steps per shooting interval          */
#define ORC_VAR_NO_DT_SCALING 8     /* stage costs NOT scaled by the interval (terminal by 1 either way)      */
void orc_set_variant(int bits);
int orc_get_variant(void);

/* a1: continuous dynamics, nmpc_body_rate_ctl.py:147-158 */
void orc_dynamics(const orc_cfg *c, const double *x, const double *u, const double *fd, double *xdot);
/* analytic Jacobians of a1, row-major A[10][10], B[10][4] */
void orc_jacobians(const double *x, const double *u, double *A, double *B);
/* ERK4, one step of h, with forward sensitivities (acados sim_erk) */
void orc_rk4_sens(const orc_cfg *c, const double *x, const double *u, const double *fd,
                  double *xn, double *A, double *B);
/* a2: residual y - yref and J_x^T W J_x, J_x^T W r; nmpc_body_rate_ctl.py:164-180 */
void orc_cost_stage(const orc_cfg *c, double scale, const double *x, const double *u,
                    const double *xr, const double *ur,
                    double *Q /*10x10*/, double *q /*10*/, double *Rdiag /*4*/, double *r /*4*/);

/* box-constrained OCP-QP in the step variables; dense per-stage row-major blocks.
 * A[N][10][10] B[N][10][4] b[N][10] Q[N+1][10][10] q[N+1][10] Rd[N][4] r[N][4]
 * lu/uu[N][4] bounds on du_k (k=0..N-1); lv/uv[N+1][3] bounds on dv_k (used k=1..N-1)
 * out: dx[N+1][10], du[N][4]                                                   */
int orc_qp_solve(const orc_cfg *c, int N, const double *A, const double *B, const double *b,
                 const double *Q, const double *q, const double *Rd, const double *r,
                 const double *dx0, const double *lu, const double *uu,
                 const double *lv, const double *uv,
                 double *dx, double *du, orc_stats *st);

/* unconstrained (equality-only) Riccati solve of the same QP, for cross-checks */
void orc_qp_riccati(int N, const double *A, const double *B, const double *b,
                    const double *Q, const double *q, const double *Rd, const double *r,
                    const double *dx0, double *dx, double *du);

/* linearise at (X,U): fills the QP blocks above (a5 steps 2-3) */
void orc_linearize(const orc_cfg *c, const double *x0, const double *xr, const double *ur,
                   const double *f, const double *X, const double *U,
                   double *A, double *B, double *b, double *Q, double *q, double *Rd, double *r,
                   double *dx0, double *lu, double *uu, double *lv, double *uv);

/* a4: reset -- X[k]=xr[k], U[k]=ur[k]   nmpc_body_rate_ctl.py:86-91 */
void orc_reset(const orc_cfg *c, const double *xr, const double *ur, double *X, double *U);

/* a5/a6: update() -- one (or n_rti) SQP-RTI iterations, iterate updated in place.
 * x0[10] xr[(N+1)*10] ur[N*4] f[(N+1)*3] or NULL; X[(N+1)*10] U[N*4]; u0[4]   */
int orc_step(const orc_cfg *c, const double *x0, const double *xr, const double *ur,
             const double *f, double *X, double *U, double *u0, orc_stats *st);

/* batch of independent instances, OpenMP over the batch (nthreads<=0: all cores) */
int orc_step_batch(const orc_cfg *c, int B, const double *x0, const double *xr, const double *ur,
                   const double *f, double *X, double *U, double *u0, int *status, int *ipm_iters,
                   int nthreads);
/* the same with the instances' kept active sets (qp_mode 0, as_iter_max > 0): act[B][4N] signed bytes, in = the previous step's
 * sets (the warm start), out = this step's; sweeps[B] out.  Either may be NULL (no warm start / not reported). */
int orc_step_batch_as(const orc_cfg *c, int B, const double *x0, const double *xr, const double *ur,
                      const double *f, double *X, double *U, double *u0, int *status, int *ipm_iters,
                      int nthreads, signed char *act, int *sweeps);
/* orc_qp_solve with an active set handed in and out (act[4N], or NULL) */
int orc_qp_solve_as(const orc_cfg *c, int N, const double *A, const double *B, const double *b,
                    const double *Q, const double *q, const double *Rd, const double *r,
                    const double *dx0, const double *lu, const double *uu,
                    const double *lv, const double *uv,
                    double *dx, double *du, orc_stats *st, signed char *act);
int orc_num_threads(void);

/* a7: MLP 6-128-64-128-3, fp32 (nn_net.py:7-18). blob = W1(128x6) b1 W2(64x128) b2
 * W3(128x64) b3 W4(3x128) b4, row-major [out][in], 17859 floats.                */
#define ORC_MLP_NPARAM 17859
void orc_mlp_forward(const float *blob, int rows, const float *in /*rows x 6*/, float *out /*rows x 3*/);

/* a7+a8: DownwashNN.update + the gate at ndp_nmpc_leader_node.py:65-76.
 * other[(N+1)*10], ego_ref[(N+1)*10] fp64; ego_xy[2] = ego odometry xy, or NULL = no gate.
 * f_out[(N+1)*3] fp32.                                                          */
void orc_downwash(const float *blob, int N, double r_horiz, const double *other,
                  const double *ego_ref, const double *ego_xy, float *f_out);
void orc_downwash_batch(const float *blob, int B, int N, double r_horiz, const double *other,
                        const double *ego_ref, const double *ego_xy, float *f_out, int nthreads);

/* f3: HoverThrottleEstimator.update (hv_throttle_est/hover_throttle_estimator.py:37-53) with its
 * Tustin differentiator (differentiator.py:10-23), one estimator per vehicle, state = [x0,x1,P00,P01,P10,P11,
 * vz_prev, az_prev]; and nmpc_u_2_att_tgt's thrust conversion (nmpc_node.py:273-283). */
typedef struct { double ts, tau, mass, g, R, Q0, Q1, k_init; } orc_thr_cfg;
void orc_thr_default_cfg(orc_thr_cfg *c);
void orc_thr_reset(const orc_thr_cfg *c, int V, double *state /*V x 8*/);
void orc_thr_update(const orc_thr_cfg *c, int V, double *state, const double *vz, const double *throttle, double *k_out);
void orc_att_thrust(const orc_thr_cfg *c, int V, const double *cacc, const double *k, double *thrust);

/* f2: follower reference relay.  AlphaFilter y = a*y + (1-a)*u with y0 = first input
 * (hv_throttle_est/alpha_filter.py:11-20, nmpc_follower_node.py:44-56); reference x[:,0:3] += offset, u copied
 * (nmpc_follower_node.py:58-74).  state[V][4] = [ox, oy, oz, initialised]. */
void orc_relay_formation(double alpha, int V, double *state, const double *form /*V x 3*/);
void orc_relay_reference(int V, int N, const double *state, const double *xr_lead, double *xr_out);

/* f4: plant = a1 dynamics (+ f/mass), RK4 with `sub` substeps over dt, quaternion renormalised
 * (no reference implementation exists: dop_sim is an empty submodule). x[V][10] in/out, u[V][4], f[V][3] or NULL. */
void orc_plant_step(const orc_cfg *c, int V, double *x, const double *u, const double *f, double dt, int sub);

/* f1: reference window generation -- the step before the path.
 * Per vehicle a piecewise polynomial trajectory (TrajCoefficients.msg: coeff_x/y/z 8 per segment (minimum snap),
 * coeff_yaw 4 per segment (minimum acceleration), traj_time_cum[n_seg+1], traj_time_seg[n_seg], final_pt).
 *   orc_traj_point   : BasePtPublisher.get_traj_pt (pt_pub/base_pt_publisher.py:81-133) at trajectory time t:
 *                      pva j[12] = pos, vel, acc, jerk; yaw[2] = yaw, yaw_dot; past the end: final_pt, all else 0
 *   orc_diff_flatness: diff_flatness (pt_pub/pt_publisher.py:188-248) + traj_full_pt_2_x_u (:115-146):
 *                      -> x[10] = [p, v, qw, qx, qy, qz], u[4] = [wx, wy, wz, collective_force / mass];
 *                      the quaternion follows tf.transformations.quaternion_from_matrix (ROS; not vendored, restated)
 *   orc_ref_window   : what NMPCRefPublisher.get_nmpc_pts hands the controller (pt_publisher.py:79-103 with the
 *                      index rule of params/nmpc_params.py:40-43): node k at trajectory time t + k*dt.
 * Layouts: coeff[V][n_seg][28] = x(8) y(8) z(8) yaw(4) per segment. */
void orc_traj_point(int n_seg, const double *coeff, const double *t_cum, const double *t_seg, const double *final_pt,
                    double t, double *pvaj /*12*/, double *yaw /*2*/);
void orc_diff_flatness(double mass, double g, const double *pvaj, const double *yaw, double *x /*10*/, double *u /*4*/);
void orc_ref_window(int V, int N, double dt, double mass, double g, int n_seg, const double *coeff, const double *t_cum,
                    const double *t_seg, const double *final_pt, const double *t /*V*/, double *xr, double *ur);

#ifdef __cplusplus
}
#endif
#endif
