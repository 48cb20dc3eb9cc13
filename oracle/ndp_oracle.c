/*
 * ndp_oracle.c -- CPU fp64 restatement of the ndp_nmpc_qd control step.
 * TEST INFRASTRUCTURE ONLY -- see ndp_oracle.h for the scope and the
 * "parity unpinned" statement.  Reference citations are relative to
 * /root/reference/ndp_nmpc/scripts/.
 */
#include "ndp_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define NX ORC_NX
#define NU ORC_NU

/* ------------------------------------------------------------------ config */

void orc_default_cfg(orc_cfg *c)
{
    memset(c, 0, sizeof(*c));
    c->N = 20;                 /* params/nmpc_params.py:9  */
    c->n_rti = 1;              /* SQP_RTI: one iteration per update() call */
    c->dt = 2.0 / 20.0;        /* params/nmpc_params.py:10,12 */
    c->mass = 1.4844;          /* params/fhnp_params.py:9  */
    c->g = 9.81;               /* params/fhnp_params.py:12 */
    /* nmpc_body_rate_ctl.py:48 with params/nmpc_params.py:28-35 */
    const double Qd[10] = {300, 300, 400, 10, 10, 10, 0, 10, 10, 100};
    const double Rd[4] = {10, 10, 10, 5};
    memcpy(c->Qd, Qd, sizeof(Qd));
    memcpy(c->Rd, Rd, sizeof(Rd));
    for (int i = 0; i < 3; ++i) {
        c->lbu[i] = -6.0;      /* w_min  nmpc_params.py:19-20 */
        c->ubu[i] = 6.0;
        c->lbv[i] = -20.0;     /* v_min  nmpc_params.py:24-25 */
        c->ubv[i] = 20.0;
    }
    c->lbu[3] = 0.0;           /* c_min  nmpc_params.py:22 */
    c->ubu[3] = 9.81 / 0.36;   /* c_max  fhnp_params.py:19 */
    c->use_fd = 0;
    c->mu0 = 10.0;
    c->thr0 = 0.1;
    c->tol = 1e-8;     /* HPIPM's default [acados-knowledge] */
    c->mu_floor = 0.1; /* centring target >= mu_floor * tol */
    c->tau = 0.995;
    c->iter_max = 50;
    c->qp_mode = 1;
    c->auto_margin = 0.1;
    c->refine = 2;
    c->refine_gamma = 1e4;
    c->as_iter_max = 8;
    c->as_gamma = 1e12;
}

/* ---------------------------------------------------------------- dynamics */

/* nmpc_body_rate_ctl.py:147-158; NDP: ndp_nmpc_body_rate_ctl.py:151-162 */
void orc_dynamics(const orc_cfg *c, const double *x, const double *u, const double *fd, double *xd)
{
    const double vx = x[3], vy = x[4], vz = x[5];
    const double qw = x[6], qx = x[7], qy = x[8], qz = x[9];
    const double wx = u[0], wy = u[1], wz = u[2], cc = u[3];
    xd[0] = vx;
    xd[1] = vy;
    xd[2] = vz;
    xd[3] = 2.0 * (qx * qz + qw * qy) * cc;
    xd[4] = 2.0 * (qy * qz - qw * qx) * cc;
    xd[5] = (1.0 - 2.0 * qx * qx - 2.0 * qy * qy) * cc - c->g;
    if (c->use_fd && fd) {
        xd[3] += fd[0] / c->mass;
        xd[4] += fd[1] / c->mass;
        xd[5] += fd[2] / c->mass;
    }
    xd[6] = (-wx * qx - wy * qy - wz * qz) * 0.5;
    xd[7] = (wx * qw + wz * qy - wy * qz) * 0.5;
    xd[8] = (wy * qw - wz * qx + wx * qz) * 0.5;
    xd[9] = (wz * qw + wy * qx - wx * qy) * 0.5;
}

/* SURVEY Appendix A.2 (derivatives of the expressions above) */
void orc_jacobians(const double *x, const double *u, double *A, double *B)
{
    const double qw = x[6], qx = x[7], qy = x[8], qz = x[9];
    const double wx = u[0], wy = u[1], wz = u[2], cc = u[3];
    memset(A, 0, sizeof(double) * NX * NX);
    memset(B, 0, sizeof(double) * NX * NU);
#define A_(i, j) A[(i)*NX + (j)]
#define B_(i, j) B[(i)*NU + (j)]
    A_(0, 3) = 1.0; A_(1, 4) = 1.0; A_(2, 5) = 1.0;
    A_(3, 6) = 2 * cc * qy;  A_(3, 7) = 2 * cc * qz;  A_(3, 8) = 2 * cc * qw; A_(3, 9) = 2 * cc * qx;
    A_(4, 6) = -2 * cc * qx; A_(4, 7) = -2 * cc * qw; A_(4, 8) = 2 * cc * qz; A_(4, 9) = 2 * cc * qy;
    A_(5, 7) = -4 * cc * qx; A_(5, 8) = -4 * cc * qy;
    A_(6, 7) = -0.5 * wx; A_(6, 8) = -0.5 * wy; A_(6, 9) = -0.5 * wz;
    A_(7, 6) = 0.5 * wx;  A_(7, 8) = 0.5 * wz;  A_(7, 9) = -0.5 * wy;
    A_(8, 6) = 0.5 * wy;  A_(8, 7) = -0.5 * wz; A_(8, 9) = 0.5 * wx;
    A_(9, 6) = 0.5 * wz;  A_(9, 7) = 0.5 * wy;  A_(9, 8) = -0.5 * wx;
    B_(3, 3) = 2.0 * (qx * qz + qw * qy);
    B_(4, 3) = 2.0 * (qy * qz - qw * qx);
    B_(5, 3) = 1.0 - 2.0 * qx * qx - 2.0 * qy * qy;
    B_(6, 0) = -0.5 * qx; B_(6, 1) = -0.5 * qy; B_(6, 2) = -0.5 * qz;
    B_(7, 0) = 0.5 * qw;  B_(7, 1) = -0.5 * qz; B_(7, 2) = 0.5 * qy;
    B_(8, 0) = 0.5 * qz;  B_(8, 1) = 0.5 * qw;  B_(8, 2) = -0.5 * qx;
    B_(9, 0) = -0.5 * qy; B_(9, 1) = 0.5 * qx;  B_(9, 2) = 0.5 * qw;
#undef A_
#undef B_
}

/* one stage of the variational ODE: K = A(xs) S + [0 | B(xs)], S is 10 x 14 */
static void vde_stage(const orc_cfg *c, const double *xs, const double *u, const double *fd,
                      const double *S, double *k, double *K)
{
    double A[NX * NX], B[NX * NU];
    orc_dynamics(c, xs, u, fd, k);
    orc_jacobians(xs, u, A, B);
    for (int i = 0; i < NX; ++i)
        for (int j = 0; j < NX + NU; ++j) {
            double s = 0.0;
            for (int l = 0; l < NX; ++l) s += A[i * NX + l] * S[l * (NX + NU) + j];
            if (j >= NX) s += B[i * NU + (j - NX)];
            K[i * (NX + NU) + j] = s;
        }
}

/* acados sim_erk: classical RK4 tableau, num_stages=4, num_steps=1 (options
 * not set by nmpc_body_rate_ctl.py:71-80, so acados defaults), forward
 * sensitivities from the variational equation with the same tableau.       */
/* ---- [acados-knowledge] assumptions as switches, for the sensitivity study only (scripts/acados_sensitivity.py): what u0 would be if
 * acados did one of the things SURVEY A.4 assumes it does not.  0 = the restatement as the tests pin it.  A process-wide test switch,
 * not part of orc_cfg (whose layout the Python binding mirrors); not thread-safe against concurrent changes. */
static int g_variant = 0;
void orc_set_variant(int bits) { g_variant = bits; }
int orc_get_variant(void) { return g_variant; }

static void rk4_sens_h(const orc_cfg *c, double h, const double *x, const double *u, const double *fd,
                       double *xn, double *A, double *B);

void orc_rk4_sens(const orc_cfg *c, const double *x, const double *u, const double *fd,
                  double *xn, double *A, double *B)
{
    if (!(g_variant & ORC_VAR_ERK_2_STEPS)) { rk4_sens_h(c, c->dt, x, u, fd, xn, A, B); return; }
    /* sim_method_num_steps = 2: two RK4 steps of h / 2; sensitivities chained: A = A2 A1, B = A2 B1 + B2 */
    double xm[NX], A1[NX * NX], B1[NX * NU], A2[NX * NX], B2[NX * NU];
    rk4_sens_h(c, 0.5 * c->dt, x, u, fd, xm, A1, B1);
    rk4_sens_h(c, 0.5 * c->dt, xm, u, fd, xn, A2, B2);
    for (int i = 0; i < NX; ++i) {
        for (int j = 0; j < NX; ++j) {
            double s = 0.0;
            for (int l = 0; l < NX; ++l) s += A2[i * NX + l] * A1[l * NX + j];
            A[i * NX + j] = s;
        }
        for (int j = 0; j < NU; ++j) {
            double s = B2[i * NU + j];
            for (int l = 0; l < NX; ++l) s += A2[i * NX + l] * B1[l * NU + j];
            B[i * NU + j] = s;
        }
    }
}

static void rk4_sens_h(const orc_cfg *c, double h, const double *x, const double *u, const double *fd,
                       double *xn, double *A, double *B)
{
    enum { NS = NX + NU };
    double S0[NX * NS], St[NX * NS], xs[NX];
    double k1[NX], k2[NX], k3[NX], k4[NX];
    double K1[NX * NS], K2[NX * NS], K3[NX * NS], K4[NX * NS];
    memset(S0, 0, sizeof(S0));
    for (int i = 0; i < NX; ++i) S0[i * NS + i] = 1.0;

    vde_stage(c, x, u, fd, S0, k1, K1);
    for (int i = 0; i < NX; ++i) xs[i] = x[i] + 0.5 * h * k1[i];
    for (int i = 0; i < NX * NS; ++i) St[i] = S0[i] + 0.5 * h * K1[i];
    vde_stage(c, xs, u, fd, St, k2, K2);
    for (int i = 0; i < NX; ++i) xs[i] = x[i] + 0.5 * h * k2[i];
    for (int i = 0; i < NX * NS; ++i) St[i] = S0[i] + 0.5 * h * K2[i];
    vde_stage(c, xs, u, fd, St, k3, K3);
    for (int i = 0; i < NX; ++i) xs[i] = x[i] + h * k3[i];
    for (int i = 0; i < NX * NS; ++i) St[i] = S0[i] + h * K3[i];
    vde_stage(c, xs, u, fd, St, k4, K4);

    for (int i = 0; i < NX; ++i)
        xn[i] = x[i] + h / 6.0 * (k1[i] + 2.0 * k2[i] + 2.0 * k3[i] + k4[i]);
    for (int i = 0; i < NX; ++i)
        for (int j = 0; j < NS; ++j) {
            const int ij = i * NS + j;
            const double s = S0[ij] + h / 6.0 * (K1[ij] + 2.0 * K2[ij] + 2.0 * K3[ij] + K4[ij]);
            if (j < NX) A[i * NX + j] = s;
            else B[i * NU + (j - NX)] = s;
        }
}

/* -------------------------------------------------------------------- cost */

/* NONLINEAR_LS residual and Gauss-Newton blocks.
 * y = [p, v, qwr, qe + q_r,xyz, u]  (nmpc_body_rate_ctl.py:164-180,194-195)
 * yref = [xr, ur], p = xr[6:10]    (nmpc_body_rate_ctl.py:96-100)
 * => residual = [p-pr, v-vr, 0, qe, u-ur]; d(qe)/dq = E(q_r)
 * scale = dt for stages 0..N-1, 1 for the terminal stage (acados scales the
 * stage cost by the shooting interval; the terminal cost is unscaled).      */
void orc_cost_stage(const orc_cfg *c, double scale, const double *x, const double *u,
                    const double *xr, const double *ur,
                    double *Q, double *q, double *Rdiag, double *r)
{
    const double qwr = xr[6], qxr = xr[7], qyr = xr[8], qzr = xr[9];
    const double qw = x[6], qx = x[7], qy = x[8], qz = x[9];
    double E[3][4] = {{-qxr, qwr, -qzr, qyr}, {-qyr, qzr, qwr, -qxr}, {-qzr, -qyr, qxr, qwr}};
    double qe[3];
    /* nmpc_body_rate_ctl.py:164-166 */
    qe[0] = qwr * qx - qw * qxr + qyr * qz - qy * qzr;
    qe[1] = qwr * qy - qw * qyr - qxr * qz + qx * qzr;
    qe[2] = qxr * qy - qx * qyr + qwr * qz - qw * qzr;
    memset(Q, 0, sizeof(double) * NX * NX);
    memset(q, 0, sizeof(double) * NX);
    for (int i = 0; i < 6; ++i) {
        Q[i * NX + i] = scale * c->Qd[i];
        q[i] = scale * c->Qd[i] * (x[i] - xr[i]);
    }
    /* row 6 (qwr - yref[6] = 0) has zero Jacobian and zero weight */
    for (int a = 0; a < 4; ++a) {
        double g = 0.0;
        for (int m = 0; m < 3; ++m) g += E[m][a] * c->Qd[7 + m] * qe[m];
        q[6 + a] = scale * g;
        for (int b = 0; b < 4; ++b) {
            double s = 0.0;
            for (int m = 0; m < 3; ++m) s += E[m][a] * c->Qd[7 + m] * E[m][b];
            Q[(6 + a) * NX + 6 + b] = scale * s;
        }
    }
    if (u) {
        for (int i = 0; i < NU; ++i) {
            Rdiag[i] = scale * c->Rd[i];
            r[i] = scale * c->Rd[i] * (u[i] - ur[i]);
        }
    }
}

/* ----------------------------------------------------------------- Riccati */

typedef struct {
    double K[ORC_NMAX][NU * NX];
    double kff[ORC_NMAX][NU];
} ric_gain;

static int chol4(double *L) /* in: SPD 4x4 row-major (lower used); out: lower factor */
{
    for (int j = 0; j < NU; ++j) {
        double d = L[j * NU + j];
        for (int l = 0; l < j; ++l) d -= L[j * NU + l] * L[j * NU + l];
        if (!(d > 0.0)) return 1;
        d = sqrt(d);
        L[j * NU + j] = d;
        for (int i = j + 1; i < NU; ++i) {
            double s = L[i * NU + j];
            for (int l = 0; l < j; ++l) s -= L[i * NU + l] * L[j * NU + l];
            L[i * NU + j] = s / d;
        }
    }
    return 0;
}

static void chol4_solve(const double *L, double *v) /* v <- (L L^T)^-1 v */
{
    for (int i = 0; i < NU; ++i) {
        double s = v[i];
        for (int l = 0; l < i; ++l) s -= L[i * NU + l] * v[l];
        v[i] = s / L[i * NU + i];
    }
    for (int i = NU - 1; i >= 0; --i) {
        double s = v[i];
        for (int l = i + 1; l < NU; ++l) s -= L[l * NU + i] * v[l];
        v[i] = s / L[i * NU + i];
    }
}

/* Backward Riccati recursion + forward rollout of
 *   min sum 1/2 dx'Q dx + q'dx + 1/2 du'R du + r'du  s.t. dx+ = A dx + B du + b
 * Qe/qe/Re/re are the effective (barrier-augmented) blocks.                 */
static int riccati_solve(int N, const double *A, const double *B, const double *b,
                         const double *Qe, const double *qe, const double *Re, const double *re,
                         const double *dx0, double *dx, double *du, ric_gain *G)
{
    double P[NX * NX], p[NX];
    memcpy(P, Qe + (size_t)N * NX * NX, sizeof(P));
    memcpy(p, qe + (size_t)N * NX, sizeof(p));
    for (int k = N - 1; k >= 0; --k) {
        const double *Ak = A + (size_t)k * NX * NX, *Bk = B + (size_t)k * NX * NU;
        const double *bk = b + (size_t)k * NX;
        double Pb[NX], PA[NX * NX], PB[NX * NU];
        for (int i = 0; i < NX; ++i) {
            double s = p[i];
            for (int l = 0; l < NX; ++l) s += P[i * NX + l] * bk[l];
            Pb[i] = s;
        }
        for (int i = 0; i < NX; ++i) {
            for (int j = 0; j < NX; ++j) {
                double s = 0.0;
                for (int l = 0; l < NX; ++l) s += P[i * NX + l] * Ak[l * NX + j];
                PA[i * NX + j] = s;
            }
            for (int j = 0; j < NU; ++j) {
                double s = 0.0;
                for (int l = 0; l < NX; ++l) s += P[i * NX + l] * Bk[l * NU + j];
                PB[i * NU + j] = s;
            }
        }
        double Lam[NU * NU], Hux[NU * NX], hu[NU], Hxx[NX * NX], hx[NX];
        for (int i = 0; i < NU; ++i) {
            for (int j = 0; j < NU; ++j) {
                double s = (i == j) ? Re[k * NU + i] : 0.0;
                for (int l = 0; l < NX; ++l) s += Bk[l * NU + i] * PB[l * NU + j];
                Lam[i * NU + j] = s;
            }
            for (int j = 0; j < NX; ++j) {
                double s = 0.0;
                for (int l = 0; l < NX; ++l) s += Bk[l * NU + i] * PA[l * NX + j];
                Hux[i * NX + j] = s;
            }
            double s = re[k * NU + i];
            for (int l = 0; l < NX; ++l) s += Bk[l * NU + i] * Pb[l];
            hu[i] = s;
        }
        for (int i = 0; i < NX; ++i) {
            for (int j = 0; j < NX; ++j) {
                double s = Qe[(size_t)k * NX * NX + i * NX + j];
                for (int l = 0; l < NX; ++l) s += Ak[l * NX + i] * PA[l * NX + j];
                Hxx[i * NX + j] = s;
            }
            double s = qe[k * NX + i];
            for (int l = 0; l < NX; ++l) s += Ak[l * NX + i] * Pb[l];
            hx[i] = s;
        }
        if (chol4(Lam)) return 4;
        /* K = -Lam^-1 Hux, kff = -Lam^-1 hu */
        for (int j = 0; j < NX; ++j) {
            double col[NU];
            for (int i = 0; i < NU; ++i) col[i] = Hux[i * NX + j];
            chol4_solve(Lam, col);
            for (int i = 0; i < NU; ++i) G->K[k][i * NX + j] = -col[i];
        }
        double kf[NU];
        memcpy(kf, hu, sizeof(kf));
        chol4_solve(Lam, kf);
        for (int i = 0; i < NU; ++i) G->kff[k][i] = -kf[i];
        /* P = Hxx + Hux' K (symmetrised), p = hx + Hux' kff */
        for (int i = 0; i < NX; ++i) {
            for (int j = 0; j < NX; ++j) {
                double s = Hxx[i * NX + j];
                for (int l = 0; l < NU; ++l) s += Hux[l * NX + i] * G->K[k][l * NX + j];
                P[i * NX + j] = s;
            }
            double s = hx[i];
            for (int l = 0; l < NU; ++l) s += Hux[l * NX + i] * G->kff[k][l];
            p[i] = s;
        }
        for (int i = 0; i < NX; ++i)
            for (int j = i + 1; j < NX; ++j) {
                const double s = 0.5 * (P[i * NX + j] + P[j * NX + i]);
                P[i * NX + j] = s;
                P[j * NX + i] = s;
            }
    }
    memcpy(dx, dx0, sizeof(double) * NX);
    for (int k = 0; k < N; ++k) {
        const double *Ak = A + (size_t)k * NX * NX, *Bk = B + (size_t)k * NX * NU;
        const double *xk = dx + (size_t)k * NX;
        double *uk = du + (size_t)k * NU, *xn = dx + (size_t)(k + 1) * NX;
        for (int i = 0; i < NU; ++i) {
            double s = G->kff[k][i];
            for (int l = 0; l < NX; ++l) s += G->K[k][i * NX + l] * xk[l];
            uk[i] = s;
        }
        for (int i = 0; i < NX; ++i) {
            double s = b[(size_t)k * NX + i];
            for (int l = 0; l < NX; ++l) s += Ak[i * NX + l] * xk[l];
            for (int l = 0; l < NU; ++l) s += Bk[i * NU + l] * uk[l];
            xn[i] = s;
        }
    }
    return 0;
}

void orc_qp_riccati(int N, const double *A, const double *B, const double *b,
                    const double *Q, const double *q, const double *Rd, const double *r,
                    const double *dx0, double *dx, double *du)
{
    ric_gain *G = (ric_gain *)malloc(sizeof(ric_gain));
    riccati_solve(N, A, B, b, Q, q, Rd, r, dx0, dx, du, G);
    free(G);
}

/* --------------------------------------------------- interior-point method */

/* Bounded step variables, in the order the QP uses them:
 *   du_k[0..3], k=0..N-1              (idxbu = 0..3, nmpc_body_rate_ctl.py:56-58)
 *   dv_k[0..2] = dx_k[3..5], k=1..N-1 (idxbx = 3,4,5, nmpc_body_rate_ctl.py:59-61;
 *                                      lbx/ubx apply to intermediate stages only,
 *                                      stage 0 is the x0 equality)             */
typedef struct {
    double lo, hi;      /* bounds on the step variable */
    double tl, tu, ll, lu; /* slacks and multipliers */
    double dtl, dtu, dll, dlu;
    int stage, idx;     /* idx<4: u index, else 4+(v index) */
} ipm_con;

static double con_value(const ipm_con *cn, const double *dx, const double *du)
{
    return cn->idx < NU ? du[cn->stage * NU + cn->idx] : dx[cn->stage * NX + 3 + (cn->idx - NU)];
}

/* Primal-dual Mehrotra predictor-corrector in "absolute" form: every Newton
 * system is the equality-constrained QP with Hessian diag += Gamma and
 * gradient += gamma, solved by the Riccati recursion (what HPIPM does with
 * d_ocp_qp_fact_solve_kkt_step).  All linear residuals (stationarity,
 * dynamics, slack definitions) contract by (1-alpha) per iteration, so they
 * are tracked by the scalar rho.                                             */
/* scratch of one QP solve; allocated once per thread by the batch driver (malloc per instance does not scale
 * under OpenMP) */
typedef struct {
    ipm_con cn[7 * ORC_NMAX];
    double Qe[(ORC_NMAX + 1) * NX * NX], qe[(ORC_NMAX + 1) * NX], Re[ORC_NMAX * NU], re[ORC_NMAX * NU];
    double zx[(ORC_NMAX + 1) * NX], zu[ORC_NMAX * NU], nx_[(ORC_NMAX + 1) * NX], nu_[ORC_NMAX * NU];
    double gx[(ORC_NMAX + 1) * NX], gu[ORC_NMAX * NU], cx[(ORC_NMAX + 1) * NX], cu[ORC_NMAX * NU], zb[ORC_NMAX * NX];
    ric_gain G;
} qp_ws;

/* Iterative refinement of the minimiser (dx, du) of the equality-constrained QP with blocks Qe, qe, Re, re: the gradient of its
 * objective at (dx, du) is g = qe + Qe dx | re + Re du; the minimiser of the SAME quadratic with gradient g, zero dynamics defects
 * and zero initial state is exactly (z* - z) -- a correction of the size of the first solve's error, computed to the same RELATIVE
 * accuracy.  (The gradient of a stiff coordinate, Gamma (dx_v - bound) + ..., is itself only accurate to Gamma * eps -- which moves
 * that coordinate by eps.)  Why this and not a factored (square-root) recursion: measured on the shrunk-velocity-box problems of
 * tests/test_oracle_pins.py, propagating a Cholesky / QR factor of P leaves the error where it is (1e-5 .. 1e-2 at tol 1e-10, the
 * same as the explicit recursion): what is lost is lost in the SOLUTION of an ill-conditioned system (cond ~ Gamma), not in forming
 * P; two refinement solves bring the same problems to 1e-9 .. 1e-11. */
#define ORC_REFINE_FAIL 1e-5
static int refine_solution(int N, const double *A, const double *B, const double *Qe, const double *qe, const double *Re,
                           const double *re, double *dx, double *du, qp_ws *w, double *corr_max)
{
    const double dx0z[NX] = {0};
    memset(w->zb, 0, sizeof(double) * (size_t)N * NX);
    for (int k = 0; k <= N; ++k)
        for (int i = 0; i < NX; ++i) {
            double s = qe[k * NX + i];
            for (int j = 0; j < NX; ++j) s += Qe[(size_t)k * NX * NX + i * NX + j] * dx[k * NX + j];
            w->gx[k * NX + i] = s;
        }
    for (int i = 0; i < N * NU; ++i) w->gu[i] = re[i] + Re[i] * du[i];
    if (riccati_solve(N, A, B, w->zb, Qe, w->gx, Re, w->gu, dx0z, w->cx, w->cu, &w->G)) return 4;
    double cm = 0.0;
    for (int i = 0; i < (N + 1) * NX; ++i) { dx[i] += w->cx[i]; cm = fmax(cm, fabs(w->cx[i])); }
    for (int i = 0; i < N * NU; ++i) { du[i] += w->cu[i]; cm = fmax(cm, fabs(w->cu[i])); }
    if (corr_max) *corr_max = cm;
    return 0;
}

/* act: the kept active set of the input bounds (qp_mode 0, as_iter_max > 0), [4N] signed bytes in / out, or NULL */
static int qp_solve_ws(const orc_cfg *c, int N, const double *A, const double *B, const double *b,
                       const double *Q, const double *q, const double *Rd, const double *r,
                       const double *dx0, const double *lu, const double *uu,
                       const double *lv, const double *uv,
                       double *dx, double *du, orc_stats *st, qp_ws *w, signed char *act)
{
    const int m = NU * N + 3 * (N - 1);
    ipm_con *cn = w->cn;
    double *Qe = w->Qe, *qe = w->qe, *Re = w->Re, *re = w->re, *zx = w->zx, *zu = w->zu, *nx_ = w->nx_, *nu_ = w->nu_;
    ric_gain *G = &w->G;
    int status = 0, iters = 0, n = 0, failed = 0;
    double norm0 = 1.0, mu = 0.0, rho = 1.0;
    memset(cn, 0, sizeof(ipm_con) * (size_t)m);
    memset(zx, 0, sizeof(double) * (size_t)(N + 1) * NX);
    memset(zu, 0, sizeof(double) * (size_t)N * NU);

    for (int k = 0; k < N; ++k)
        for (int i = 0; i < NU; ++i, ++n) {
            cn[n].stage = k; cn[n].idx = i;
            cn[n].lo = lu[k * NU + i]; cn[n].hi = uu[k * NU + i];
        }
    for (int k = 1; k < N + ((g_variant & ORC_VAR_BOUNDS_STAGE_N) ? 1 : 0); ++k)
        for (int i = 0; i < 3; ++i, ++n) {
            cn[n].stage = k; cn[n].idx = NU + i;
            cn[n].lo = lv[k * 3 + i]; cn[n].hi = uv[k * 3 + i];
        }

    /* qp_mode 0 (the device's QP_AUTO; qp_mode 1, the default, is HPIPM-like: always iterate).  as_iter_max = 0: the
     * equality-constrained minimiser, if it lies auto_margin inside every bound, IS the QP solution (all multipliers zero).
     * as_iter_max > 0: primal-dual active-set iterations on the INPUT bounds, the device's rule (rti_wave.hpp: as_check) --
     * inputs beyond a bound are pinned there by the weight as_gamma, pins whose multiplier as_gamma (du - d) has the wrong sign are
     * released, all at once, one Riccati solve per iteration, until the set reproduces itself: the KKT conditions of the
     * box-constrained QP then hold.  A velocity bound violated (or closer than auto_margin), a set that does not settle, a failed
     * factorisation with pins: the interior-point loop below takes the QP. */
    int sweeps = 0;
    if (c->qp_mode == 0) {
        const int as_on = c->as_iter_max > 0;
        const double um = as_on ? 0.0 : c->auto_margin;
        const int nu = NU * N;
        signed char a[NU * ORC_NMAX], na[NU * ORC_NMAX];
        for (int i = 0; i < nu; ++i) a[i] = (as_on && act) ? act[i] : 0;
        for (;;) {
            int any = 0;
            double *be = w->zb;                 /* the defects with the pinned inputs' step bounds moved into them (see below) */
            memcpy(Re, Rd, sizeof(double) * (size_t)nu);
            memcpy(re, r, sizeof(double) * (size_t)nu);
            memcpy(be, b, sizeof(double) * (size_t)N * NX);
            /* a pinned input is solved for in re-centred form, du = d + delta with the weight on delta alone: gradient r + R d,
             * defect b + B d, diagonal R + as_gamma -- the solve returns delta = lambda / as_gamma itself, to full relative accuracy
             * (read off du - d, two numbers that agree to 12 digits, the multiplier is right to ~1e-3 only: rti_wave.hpp, as_apply) */
            for (int i = 0; i < nu; ++i)
                if (a[i]) {
                    const double d = a[i] > 0 ? cn[i].hi : cn[i].lo;
                    const int k = i / NU, j = i % NU;
                    Re[i] += c->as_gamma;
                    re[i] += Rd[i] * d;
                    for (int row = 0; row < NX; ++row) be[k * NX + row] += B[(size_t)k * NX * NU + row * NU + j] * d;
                    any = 1;
                }
            const int rc = riccati_solve(N, A, B, any ? be : b, Q, q, Re, re, dx0, nx_, nu_, G);
            ++sweeps;
            if (rc) {
                if (!any) { status = 4; failed = 1; goto done; }
                break;
            }
            int vok = 1, same = 1;
            for (int i = nu; i < m; ++i) {
                const double zn = con_value(&cn[i], nx_, nu_);
                vok = vok && zn > cn[i].lo + c->auto_margin && zn < cn[i].hi - c->auto_margin;
            }
            if (!vok) break;
            for (int i = 0; i < nu; ++i) {
                const double zn = nu_[i];
                /* pinned: kept while its multiplier as_gamma delta has the right sign, else released (not re-pinned in this pass);
                 * free: beyond a bound -> pinned there */
                if (a[i]) na[i] = ((a[i] > 0 ? zn : -zn) >= 0.0) ? a[i] : 0;            /* (zn = delta) */
                else na[i] = zn > cn[i].hi - um ? 1 : (zn < cn[i].lo + um ? -1 : 0);
                same = same && na[i] == a[i];
            }
            if (same) {
                for (int i = 0; i < nu; ++i)
                    if (a[i]) nu_[i] = a[i] > 0 ? cn[i].hi : cn[i].lo;       /* onto the bound exactly */
                memcpy(zx, nx_, sizeof(double) * (size_t)(N + 1) * NX);
                memcpy(zu, nu_, sizeof(double) * (size_t)N * NU);
                if (as_on && act) memcpy(act, a, (size_t)nu);
                goto done;
            }
            memcpy(a, na, (size_t)nu);
            if (!(as_on && sweeps <= c->as_iter_max)) break;
        }
        if (as_on && act) memset(act, 0, (size_t)nu);          /* the interior-point loop's answer carries no set */
    }
    /* cold start (qp_solver_warm_start left at 0, nmpc_body_rate_ctl.py:73-74) */
    for (int i = 0; i < m; ++i) {
        cn[i].tl = fmax(-cn[i].lo, c->thr0);
        cn[i].tu = fmax(cn[i].hi, c->thr0);
        cn[i].ll = c->mu0 / cn[i].tl;
        cn[i].lu = c->mu0 / cn[i].tu;
        mu += cn[i].ll * cn[i].tl + cn[i].lu * cn[i].tu;
        norm0 = fmax(norm0, fmax(cn[i].ll, cn[i].lu));
        norm0 = fmax(norm0, fmax(fabs(-cn[i].lo - cn[i].tl), fabs(cn[i].hi - cn[i].tu)));
    }
    mu /= (2.0 * m);
    for (int i = 0; i < (N + 1) * NX; ++i) norm0 = fmax(norm0, fabs(q[i]));
    for (int i = 0; i < N * NU; ++i) norm0 = fmax(norm0, fabs(r[i]));
    for (int i = 0; i < N * NX; ++i) norm0 = fmax(norm0, fabs(b[i]));
    for (int i = 0; i < NX; ++i) norm0 = fmax(norm0, fabs(dx0[i]));

    for (;;) {
        if (mu <= c->tol && rho * norm0 <= c->tol) break;
        if (iters >= c->iter_max) { status = 4; break; }
        ++iters;
        double sigma_mu = 0.0;
        for (int pass = 0; pass < 2; ++pass) {
            /* barrier-augmented blocks */
            memcpy(Qe, Q, sizeof(double) * (size_t)(N + 1) * NX * NX);
            memcpy(qe, q, sizeof(double) * (size_t)(N + 1) * NX);
            memcpy(Re, Rd, sizeof(double) * (size_t)N * NU);
            memcpy(re, r, sizeof(double) * (size_t)N * NU);
            for (int i = 0; i < m; ++i) {
                ipm_con *cc = &cn[i];
                const double sl = pass ? sigma_mu - cc->dll * cc->dtl : 0.0;
                const double su = pass ? sigma_mu - cc->dlu * cc->dtu : 0.0;
                const double gl = cc->ll / cc->tl, gu = cc->lu / cc->tu;
                const double Gam = gl + gu;
                const double gam = -sl / cc->tl - cc->ll - gl * cc->lo + su / cc->tu + cc->lu - gu * cc->hi;
                if (cc->idx < NU) {
                    Re[cc->stage * NU + cc->idx] += Gam;
                    re[cc->stage * NU + cc->idx] += gam;
                } else {
                    const int j = 3 + cc->idx - NU;
                    Qe[(size_t)cc->stage * NX * NX + j * NX + j] += Gam;
                    qe[cc->stage * NX + j] += gam;
                }
            }
            if (riccati_solve(N, A, B, b, Qe, qe, Re, re, dx0, nx_, nu_, G)) { status = 4; failed = 1; goto done; }
            if (c->refine > 0) {
                double gmax = 0.0;           /* largest barrier term on a STATE bound (input bounds sit on R's diagonal: benign) */
                for (int i = 0; i < m; ++i)
                    if (cn[i].idx >= NU) gmax = fmax(gmax, cn[i].ll / cn[i].tl + cn[i].lu / cn[i].tu);
                if (gmax > c->refine_gamma) {
                    double corr = 0.0;
                    for (int rf = 0; rf < c->refine; ++rf)
                        if (refine_solution(N, A, B, Qe, qe, Re, re, nx_, nu_, w, &corr)) { status = 4; failed = 1; goto done; }
                    /* a LAST correction that is still visible at the parity bar means the solves of this system cannot be trusted
                     * (barrier terms beyond what fp64 carries through the recursion): a QP failure, said so -- not a silent answer */
                    /* (the corrector's solve -- the one the iterate is updated from; the predictor only steers the centring) */
                    if (pass && !(corr <= ORC_REFINE_FAIL)) { status = 4; failed = 1; goto done; }
                }
            }
            /* slack / multiplier steps and the largest feasible step length */
            double alpha = 1.0;
            for (int i = 0; i < m; ++i) {
                ipm_con *cc = &cn[i];
                const double zn = con_value(cc, nx_, nu_);
                const double sl = pass ? sigma_mu - cc->dll * cc->dtl : 0.0;
                const double su = pass ? sigma_mu - cc->dlu * cc->dtu : 0.0;
                const double dtl = zn - cc->lo - cc->tl, dtu = cc->hi - zn - cc->tu;
                const double dll = sl / cc->tl - cc->ll - cc->ll / cc->tl * dtl;
                const double dlu = su / cc->tu - cc->lu - cc->lu / cc->tu * dtu;
                cc->dtl = dtl; cc->dtu = dtu; cc->dll = dll; cc->dlu = dlu;
                if (dtl < 0.0) alpha = fmin(alpha, -cc->tl / dtl);
                if (dtu < 0.0) alpha = fmin(alpha, -cc->tu / dtu);
                if (dll < 0.0) alpha = fmin(alpha, -cc->ll / dll);
                if (dlu < 0.0) alpha = fmin(alpha, -cc->lu / dlu);
            }
            if (pass == 0) {
                double mu_aff = 0.0;
                for (int i = 0; i < m; ++i) {
                    const ipm_con *cc = &cn[i];
                    mu_aff += (cc->ll + alpha * cc->dll) * (cc->tl + alpha * cc->dtl)
                            + (cc->lu + alpha * cc->dlu) * (cc->tu + alpha * cc->dtu);
                }
                mu_aff /= (2.0 * m);
                const double s = mu_aff / mu;
                /* Mehrotra's centring target, kept from undershooting the tolerance: slacks are formed by subtraction
                 * (t = z - lb), so their relative accuracy -- and with it the Newton systems' -- is eps / t; a target
                 * of 1e-12 would cost six digits of the answer for nothing.  HPIPM guards the same way with lower
                 * thresholds on t and lambda [acados-knowledge]. */
                sigma_mu = fmax(s * s * s * mu, c->mu_floor * c->tol);
            } else {
                if (alpha < 1.0) alpha *= c->tau;
                for (int i = 0; i < (N + 1) * NX; ++i) zx[i] += alpha * (nx_[i] - zx[i]);
                for (int i = 0; i < N * NU; ++i) zu[i] += alpha * (nu_[i] - zu[i]);
                mu = 0.0;
                for (int i = 0; i < m; ++i) {
                    ipm_con *cc = &cn[i];
                    cc->tl += alpha * cc->dtl; cc->tu += alpha * cc->dtu;
                    cc->ll += alpha * cc->dll; cc->lu += alpha * cc->dlu;
                    mu += cc->ll * cc->tl + cc->lu * cc->tu;
                }
                mu /= (2.0 * m);
                rho *= (1.0 - alpha);
            }
        }
        if (!(mu == mu)) { status = 1; failed = 1; break; }
    }
done:
    /* No usable step (factorisation failure, NaN): hand back a zero step, i.e. the caller's iterate stays as it is --
     * acados' SQP_RTI returns ACADOS_QP_FAILURE before it updates the variables [acados-knowledge].  An exhausted
     * iteration budget (status 4 as well) still hands over the last interior-point iterate. */
    if (failed) {
        memset(zx, 0, sizeof(double) * (size_t)(N + 1) * NX);
        memset(zu, 0, sizeof(double) * (size_t)N * NU);
    }
    memcpy(dx, zx, sizeof(double) * (size_t)(N + 1) * NX);
    memcpy(du, zu, sizeof(double) * (size_t)N * NU);
    if (st) {
        st->status = status;
        st->ipm_iters = iters;
        st->mu = mu;
        st->n_active = 0;
        for (int i = 0; i < m; ++i)
            st->n_active += (cn[i].ll > 1e-6) + (cn[i].lu > 1e-6);
        st->as_sweeps = sweeps;
    }
    return status;
}

int orc_qp_solve(const orc_cfg *c, int N, const double *A, const double *B, const double *b,
                 const double *Q, const double *q, const double *Rd, const double *r,
                 const double *dx0, const double *lu, const double *uu,
                 const double *lv, const double *uv,
                 double *dx, double *du, orc_stats *st)
{
    return orc_qp_solve_as(c, N, A, B, b, Q, q, Rd, r, dx0, lu, uu, lv, uv, dx, du, st, NULL);
}

int orc_qp_solve_as(const orc_cfg *c, int N, const double *A, const double *B, const double *b,
                    const double *Q, const double *q, const double *Rd, const double *r,
                    const double *dx0, const double *lu, const double *uu,
                    const double *lv, const double *uv,
                    double *dx, double *du, orc_stats *st, signed char *act)
{
    qp_ws *w = (qp_ws *)malloc(sizeof(qp_ws));
    const int rc = qp_solve_ws(c, N, A, B, b, Q, q, Rd, r, dx0, lu, uu, lv, uv, dx, du, st, w, act);
    free(w);
    return rc;
}

/* ------------------------------------------------------------- SQP-RTI step */

void orc_reset(const orc_cfg *c, const double *xr, const double *ur, double *X, double *U)
{
    /* nmpc_body_rate_ctl.py:86-91 */
    memcpy(X, xr, sizeof(double) * (size_t)(c->N + 1) * NX);
    memcpy(U, ur, sizeof(double) * (size_t)c->N * NU);
}

void orc_linearize(const orc_cfg *c, const double *x0, const double *xr, const double *ur,
                   const double *f, const double *X, const double *U,
                   double *A, double *B, double *b, double *Q, double *q, double *Rd, double *r,
                   double *dx0, double *lu, double *uu, double *lv, double *uv)
{
    const int N = c->N;
    for (int k = 0; k < N; ++k) {
        const double *xk = X + (size_t)k * NX, *uk = U + (size_t)k * NU;
        double xn[NX], fdk[3] = {0, 0, 0};
        /* p_k = [xr_k[6:10], f[k,:]]  ndp_nmpc_body_rate_ctl.py:97-99 */
        if (c->use_fd && f) { fdk[0] = f[k * 3]; fdk[1] = f[k * 3 + 1]; fdk[2] = f[k * 3 + 2]; }
        orc_rk4_sens(c, xk, uk, fdk, xn, A + (size_t)k * NX * NX, B + (size_t)k * NX * NU);
        for (int i = 0; i < NX; ++i) b[k * NX + i] = xn[i] - X[(size_t)(k + 1) * NX + i];
        orc_cost_stage(c, (g_variant & ORC_VAR_NO_DT_SCALING) ? 1.0 : c->dt, xk, uk, xr + (size_t)k * NX, ur + (size_t)k * NU,
                       Q + (size_t)k * NX * NX, q + (size_t)k * NX, Rd + (size_t)k * NU, r + (size_t)k * NU);
        for (int i = 0; i < NU; ++i) {
            lu[k * NU + i] = c->lbu[i] - uk[i];
            uu[k * NU + i] = c->ubu[i] - uk[i];
        }
        for (int i = 0; i < 3; ++i) {
            lv[k * 3 + i] = c->lbv[i] - xk[3 + i];
            uv[k * 3 + i] = c->ubv[i] - xk[3 + i];
        }
    }
    /* terminal: yref_N = xr_N, W_e = Q unscaled (nmpc_body_rate_ctl.py:53,101-104) */
    orc_cost_stage(c, (g_variant & ORC_VAR_TERMINAL_TIMES_DT) ? c->dt : 1.0, X + (size_t)N * NX, NULL, xr + (size_t)N * NX, NULL,
                   Q + (size_t)N * NX * NX, q + (size_t)N * NX, NULL, NULL);
    for (int i = 0; i < 3; ++i) {
        const int on = (g_variant & ORC_VAR_BOUNDS_STAGE_N) != 0;
        lv[N * 3 + i] = on ? c->lbv[i] - X[(size_t)N * NX + 3 + i] : -1e30;
        uv[N * 3 + i] = on ? c->ubv[i] - X[(size_t)N * NX + 3 + i] : 1e30;
    }
    /* solve_for_x0: lbx_0 = ubx_0 = x0 (nmpc_body_rate_ctl.py:107) */
    for (int i = 0; i < NX; ++i) dx0[i] = x0[i] - X[i];
}

typedef struct {
    double lin[ORC_NMAX * (NX * NX + NX * NU + NX + NU + NU + NU + NU) + (ORC_NMAX + 1) * (NX * NX + NX + 3 + 3 + NX) + NX + ORC_NMAX * NU];
    qp_ws qp;
} step_ws;

static int step_ws_run(const orc_cfg *c, const double *x0, const double *xr, const double *ur,
                       const double *f, double *X, double *U, double *u0, orc_stats *st, step_ws *ws, signed char *act)
{
    const int N = c->N;
    double *w = ws->lin;
    double *A = w, *B = A + (size_t)N * NX * NX, *b = B + (size_t)N * NX * NU;
    double *Rd = b + (size_t)N * NX, *r = Rd + (size_t)N * NU, *lu = r + (size_t)N * NU, *uu = lu + (size_t)N * NU;
    double *Q = uu + (size_t)N * NU, *q = Q + (size_t)(N + 1) * NX * NX;
    double *lv = q + (size_t)(N + 1) * NX, *uv = lv + (size_t)(N + 1) * 3;
    double *dx = uv + (size_t)(N + 1) * 3, *dx0 = dx + (size_t)(N + 1) * NX, *du = dx0 + NX;
    orc_stats acc = {0, 0, 0, 0.0, 0};
    for (int it = 0; it < c->n_rti; ++it) {
        orc_stats s1;
        orc_linearize(c, x0, xr, ur, f, X, U, A, B, b, Q, q, Rd, r, dx0, lu, uu, lv, uv);
        qp_solve_ws(c, N, A, B, b, Q, q, Rd, r, dx0, lu, uu, lv, uv, dx, du, &s1, &ws->qp, act);
        /* full step, no line search */
        for (int i = 0; i < (N + 1) * NX; ++i) X[i] += dx[i];
        for (int i = 0; i < N * NU; ++i) U[i] += du[i];
        acc.ipm_iters += s1.ipm_iters;
        acc.as_sweeps += s1.as_sweeps;
        acc.n_active = s1.n_active;
        acc.mu = s1.mu;
        if (s1.status && !acc.status) acc.status = s1.status;
    }
    for (int i = 0; i < NU; ++i) {
        u0[i] = U[i];
        if (!(u0[i] == u0[i])) acc.status = 1;
    }
    if (st) *st = acc;
    return acc.status;
}

int orc_step(const orc_cfg *c, const double *x0, const double *xr, const double *ur,
             const double *f, double *X, double *U, double *u0, orc_stats *st)
{
    if (c->N > ORC_NMAX) return -1;
    step_ws *ws = (step_ws *)malloc(sizeof(step_ws));
    const int rc = step_ws_run(c, x0, xr, ur, f, X, U, u0, st, ws, NULL);
    free(ws);
    return rc;
}

int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

int orc_step_batch(const orc_cfg *c, int B, const double *x0, const double *xr, const double *ur,
                   const double *f, double *X, double *U, double *u0, int *status, int *ipm_iters,
                   int nthreads)
{
    return orc_step_batch_as(c, B, x0, xr, ur, f, X, U, u0, status, ipm_iters, nthreads, NULL, NULL);
}

int orc_step_batch_as(const orc_cfg *c, int B, const double *x0, const double *xr, const double *ur,
                      const double *f, double *X, double *U, double *u0, int *status, int *ipm_iters,
                      int nthreads, signed char *act, int *sweeps)
{
    const int N = c->N;
    int worst = 0;
    if (N > ORC_NMAX) return -1;
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
    if (nthreads > B) nthreads = B;
#pragma omp parallel num_threads(nthreads) reduction(max : worst)
#endif
    {
        step_ws *ws = (step_ws *)malloc(sizeof(step_ws));   /* one scratch per thread */
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
        for (int i = 0; i < B; ++i) {
            orc_stats s;
            step_ws_run(c, x0 + (size_t)i * NX, xr + (size_t)i * (N + 1) * NX, ur + (size_t)i * N * NU,
                        f ? f + (size_t)i * (N + 1) * 3 : NULL, X + (size_t)i * (N + 1) * NX,
                        U + (size_t)i * N * NU, u0 + (size_t)i * NU, &s, ws, act ? act + (size_t)i * N * NU : NULL);
            if (sweeps) sweeps[i] = s.as_sweeps;
            if (status) status[i] = s.status;
            if (ipm_iters) ipm_iters[i] = s.ipm_iters;
            if (s.status > worst) worst = s.status;
        }
        free(ws);
    }
    return worst;
}

/* --------------------------------------------------------------------- MLP */

/* One dense layer for a BLOCK of rows, activations held [feature][row]: the inner loop runs over the rows of the block -- independent
 * accumulators, so the compiler vectorises it (8 rows per AVX2 instruction) without being allowed to re-associate anything: every
 * row's sum is accumulated in exactly the order of the scalar loop  s = bias; for i: s += W[o][i] * in[i]  (bit-identical to it).
 * A row-at-a-time dot product, by contrast, is a serial chain of dependent adds that gcc may not vectorise without -ffast-math:
 * as the CPU baseline's network it ran ~4x slower than this. */
#define MLP_RB 32
static void dense_block(const float *W, const float *bias, int nout, int nin, const float *in, float *out, int relu)
{
    for (int o = 0; o < nout; ++o) {
        float s[MLP_RB];
        for (int r = 0; r < MLP_RB; ++r) s[r] = bias[o];
        for (int i = 0; i < nin; ++i) {
            const float w = W[o * nin + i];
            const float *x = in + (size_t)i * MLP_RB;
            for (int r = 0; r < MLP_RB; ++r) s[r] += w * x[r];
        }
        float *y = out + (size_t)o * MLP_RB;
        if (relu)
            for (int r = 0; r < MLP_RB; ++r) y[r] = s[r] < 0.0f ? 0.0f : s[r];
        else
            for (int r = 0; r < MLP_RB; ++r) y[r] = s[r];
    }
}

/* nn_net.py:7-18: Linear(6,128) ReLU Linear(128,64) ReLU Linear(64,128) ReLU Linear(128,3) */
void orc_mlp_forward(const float *blob, int rows, const float *in, float *out)
{
    const float *W1 = blob, *b1 = W1 + 128 * 6;
    const float *W2 = b1 + 128, *b2 = W2 + 64 * 128;
    const float *W3 = b2 + 64, *b3 = W3 + 128 * 64;
    const float *W4 = b3 + 128, *b4 = W4 + 3 * 128;
    for (int r0 = 0; r0 < rows; r0 += MLP_RB) {
        const int nr = rows - r0 < MLP_RB ? rows - r0 : MLP_RB;
        float x[6 * MLP_RB], h1[128 * MLP_RB], h2[64 * MLP_RB], h3[128 * MLP_RB], y[3 * MLP_RB];
        for (int i = 0; i < 6; ++i)
            for (int r = 0; r < MLP_RB; ++r) x[i * MLP_RB + r] = r < nr ? in[(size_t)(r0 + r) * 6 + i] : 0.0f;
        dense_block(W1, b1, 128, 6, x, h1, 1);
        dense_block(W2, b2, 64, 128, h1, h2, 1);
        dense_block(W3, b3, 128, 64, h2, h3, 1);
        dense_block(W4, b4, 3, 128, h3, y, 0);
        for (int r = 0; r < nr; ++r)
            for (int c = 0; c < 3; ++c) out[(size_t)(r0 + r) * 3 + c] = y[c * MLP_RB + r];
    }
}

void orc_downwash(const float *blob, int N, double r_horiz, const double *other,
                  const double *ego_ref, const double *ego_xy, float *f_out)
{
    const int rows = N + 1;
    /* gate: ndp_nmpc_leader_node.py:65-68 (other.x[0] xy vs ego ODOMETRY xy) */
    if (ego_xy) {
        const double dx = other[0] - ego_xy[0], dy = other[1] - ego_xy[1];
        if (!(dx * dx + dy * dy < r_horiz * r_horiz)) {
            memset(f_out, 0, sizeof(float) * (size_t)rows * 3); /* :75-76 */
            return;
        }
    }
    float in[(ORC_NMAX + 1) * 6];
    /* downwash_nn.py:22-23: (other - ego)[:, 0:6] in fp64, then cast to fp32 */
    for (int k = 0; k < rows; ++k)
        for (int j = 0; j < 6; ++j)
            in[k * 6 + j] = (float)(other[k * NX + j] - ego_ref[k * NX + j]);
    orc_mlp_forward(blob, rows, in, f_out);
}

void orc_downwash_batch(const float *blob, int B, int N, double r_horiz, const double *other,
                        const double *ego_ref, const double *ego_xy, float *f_out, int nthreads)
{
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#pragma omp parallel for schedule(static) num_threads(nthreads)
#endif
    for (int i = 0; i < B; ++i)
        orc_downwash(blob, N, r_horiz, other + (size_t)i * (N + 1) * NX, ego_ref + (size_t)i * (N + 1) * NX,
                     ego_xy ? ego_xy + (size_t)i * 2 : NULL, f_out + (size_t)i * (N + 1) * 3);
}

/* ------------------------------------------------- f3: hover-throttle estimator */

void orc_thr_default_cfg(orc_thr_cfg *c)
{
    c->ts = 0.02;        /* params/estimator_params.py:15 */
    c->tau = 0.05;       /* differentiator.py:17 */
    c->mass = 1.4844;    /* params/fhnp_params.py:9 */
    c->g = 9.81;
    c->R = 1.225;        /* params/estimator_params.py:17 */
    c->Q0 = 0.1;         /* params/estimator_params.py:18 */
    c->Q1 = 0.1;
    c->k_init = 50.0;    /* params/estimator_params.py:13 */
}

void orc_thr_reset(const orc_thr_cfg *c, int V, double *st)
{
    for (int v = 0; v < V; ++v) {
        double *s = st + (size_t)v * 8;
        s[0] = 0.0; s[1] = c->k_init;                 /* hover_throttle_estimator.py:21 */
        s[2] = 1.0; s[3] = 0.0; s[4] = 0.0; s[5] = 1.0; /* P = I  :22 */
        s[6] = 0.0; s[7] = 0.0;                       /* differentiator.py:12-13 */
    }
}

void orc_thr_update(const orc_thr_cfg *c, int V, double *st, const double *vz, const double *throttle, double *k_out)
{
    const double a1 = (2.0 * c->tau - c->ts) / (2.0 * c->tau + c->ts);   /* differentiator.py:18-19 */
    const double a2 = 2.0 / (2.0 * c->tau + c->ts);
    const double hm = 1.0 / c->mass;                                      /* H = [1/mass, 0]  :31 */
    for (int v = 0; v < V; ++v) {
        double *s = st + (size_t)v * 8;
        const double az = a1 * s[7] + a2 * (vz[v] - s[6]);                /* differentiator.py:21-23 */
        s[6] = vz[v];
        s[7] = az;
        const double th = throttle[v];
        if (0.1 < th && th < 1.0) {                                       /* :40 */
            const double z = az + c->g;
            /* P = Phi P Phi' + Q with Phi = [[0, th], [0, 1]]   :45 */
            const double P11 = s[5];
            const double p00 = (th * P11) * th + c->Q0, p01 = th * P11, p10 = P11 * th, p11 = P11 + c->Q1;
            /* K = P H' inv(H P H' + R)   :46 */
            const double inv = 1.0 / ((hm * p00) * hm + c->R);
            const double K0 = (p00 * hm) * inv, K1 = (p10 * hm) * inv;
            /* x = Phi x; x = x + K (z - H x)   :47-48 */
            const double x0 = th * s[1], x1 = s[1];
            const double innov = z - hm * x0;
            s[0] = x0 + K0 * innov;
            s[1] = x1 + K1 * innov;
            /* P = (I - K H) P   :49 */
            const double i00 = 1.0 - K0 * hm, i10 = -(K1 * hm);
            s[2] = i00 * p00; s[3] = i00 * p01;
            s[4] = i10 * p00 + p10; s[5] = i10 * p01 + p11;
        }
        k_out[v] = s[1];                                                   /* :51 */
    }
}

void orc_att_thrust(const orc_thr_cfg *c, int V, const double *cacc, const double *k, double *thrust)
{
    for (int v = 0; v < V; ++v) thrust[v] = k[v] != 0.0 ? cacc[v] * c->mass / k[v] : 0.0;   /* nmpc_node.py:281 */
}

/* ------------------------------------------------- f2: follower reference relay */

void orc_relay_formation(double alpha, int V, double *st, const double *form)
{
    for (int v = 0; v < V; ++v) {
        double *s = st + (size_t)v * 4;
        for (int a = 0; a < 3; ++a) {
            const double u = form[v * 3 + a];
            if (s[3] == 0.0) s[a] = u;                          /* AlphaFilter(alpha, y0=msg)  nmpc_follower_node.py:45-52 */
            s[a] = alpha * s[a] + (1.0 - alpha) * u;            /* alpha_filter.py:19 */
        }
        s[3] = 1.0;
    }
}

void orc_relay_reference(int V, int N, const double *st, const double *xr_lead, double *xr_out)
{
    for (int v = 0; v < V; ++v)
        for (int k = 0; k <= N; ++k)
            for (int i = 0; i < NX; ++i) {
                const size_t idx = ((size_t)v * (N + 1) + k) * NX + i;
                xr_out[idx] = i < 3 ? xr_lead[idx] + st[v * 4 + i] : xr_lead[idx];   /* nmpc_follower_node.py:66-70 */
            }
}

/* ------------------------------------------------- f4: plant step */

void orc_plant_step(const orc_cfg *c, int V, double *x, const double *u, const double *f, double dt, int sub)
{
    orc_cfg cc = *c;
    cc.use_fd = f != NULL;
    const double h = dt / sub;
    for (int v = 0; v < V; ++v) {
        double *xv = x + (size_t)v * NX;
        const double *uv = u + (size_t)v * NU, *fv = f ? f + (size_t)v * 3 : NULL;
        for (int s = 0; s < sub; ++s) {
            double k1[NX], k2[NX], k3[NX], k4[NX], xs[NX];
            orc_dynamics(&cc, xv, uv, fv, k1);
            for (int i = 0; i < NX; ++i) xs[i] = xv[i] + 0.5 * h * k1[i];
            orc_dynamics(&cc, xs, uv, fv, k2);
            for (int i = 0; i < NX; ++i) xs[i] = xv[i] + 0.5 * h * k2[i];
            orc_dynamics(&cc, xs, uv, fv, k3);
            for (int i = 0; i < NX; ++i) xs[i] = xv[i] + h * k3[i];
            orc_dynamics(&cc, xs, uv, fv, k4);
            for (int i = 0; i < NX; ++i) xv[i] += h / 6.0 * (k1[i] + 2.0 * k2[i] + 2.0 * k3[i] + k4[i]);
        }
        const double n = sqrt(xv[6] * xv[6] + xv[7] * xv[7] + xv[8] * xv[8] + xv[9] * xv[9]);
        for (int i = 6; i < 10; ++i) xv[i] /= n;
    }
}

/* ------------------------------------------------- f1: reference window generation */

/* PolymOptimizer.get_poly_params(deriv, t) (pt_pub/polym_optimizer.py:104-139) for a polynomial with n coefficients:
 * params[i] = i (i-1) ... (i-deriv+1) * t^(i-deriv) for i >= deriv, 0 below. */
static void poly_params(int n, int deriv, double t, double *params)
{
    for (int i = 0; i < n; ++i) {
        double p = 1.0;
        int ord = i;
        for (int j = 0; j < deriv; ++j) {
            p *= (double)ord;
            if (ord > 0) --ord;
        }
        params[i] = p * pow(t, (double)ord);
    }
}

/* _get_output_value (base_pt_publisher.py:136-143): (params / t_segment^deriv) @ coeff */
static double poly_value(int n, int deriv, double ts, double tseg, const double *c)
{
    double params[8], acc = 0.0;
    poly_params(n, deriv, ts, params);
    const double sc = pow(tseg, (double)deriv);
    for (int i = 0; i < n; ++i) acc += params[i] / sc * c[i];
    return acc;
}

void orc_traj_point(int n_seg, const double *coeff, const double *t_cum, const double *t_seg, const double *final_pt,
                    double t, double *pvaj, double *yaw)
{
    for (int i = 0; i < 12; ++i) pvaj[i] = 0.0;
    yaw[0] = yaw[1] = 0.0;
    if (t >= t_cum[n_seg]) {                   /* base_pt_publisher.py:93-94: hover at final_pt after the end */
        for (int i = 0; i < 3; ++i) pvaj[i] = final_pt[i];
        return;
    }
    int idx = 0;                               /* :100: first i with time_cum[i] > t, minus one */
    while (idx < n_seg && !(t_cum[idx] > t)) ++idx;
    idx -= 1;
    if (idx < 0) idx = 0;                      /* t < time_cum[0]: callers never ask; clamp instead of numpy's wrap-around */
    const double tseg = t_seg[idx];
    const double ts = (t - t_cum[idx]) / tseg; /* :102-103 */
    const double *c = coeff + (size_t)idx * 28;
    for (int d = 0; d < 4; ++d)                /* :110-124: position, velocity, acceleration, jerk */
        for (int a = 0; a < 3; ++a) pvaj[3 * d + a] = poly_value(8, d, ts, tseg, c + 8 * a);
    yaw[0] = poly_value(4, 0, ts, tseg, c + 24);   /* :127-129 */
    yaw[1] = poly_value(4, 1, ts, tseg, c + 24);
}

static void cross3(const double *a, const double *b, double *o)
{
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

/* tf.transformations.quaternion_from_matrix (ROS geometry, tf/src/tf/transformations.py) on a homogeneous matrix
 * with M[3][3] = 1; R columns = x_b, y_b, z_b.  Returns (x, y, z, w). */
static void quat_from_matrix(const double R[3][3], double *q)
{
    const double m33 = 1.0;
    double t = R[0][0] + R[1][1] + R[2][2] + m33;
    if (t > m33) {
        q[3] = t;
        q[2] = R[1][0] - R[0][1];
        q[1] = R[0][2] - R[2][0];
        q[0] = R[2][1] - R[1][2];
    } else {
        int i = 0, j = 1, k = 2;
        if (R[1][1] > R[0][0]) { i = 1; j = 2; k = 0; }
        if (R[2][2] > R[i][i]) { i = 2; j = 0; k = 1; }
        t = R[i][i] - (R[j][j] + R[k][k]) + m33;
        q[i] = t;
        q[j] = R[i][j] + R[j][i];
        q[k] = R[k][i] + R[i][k];
        q[3] = R[k][j] - R[j][k];
    }
    const double s = 0.5 / sqrt(t * m33);
    for (int a = 0; a < 4; ++a) q[a] *= s;
}

void orc_diff_flatness(double mass, double g, const double *pvaj, const double *yaw, double *x, double *u)
{
    const double *acc = pvaj + 6, *jerk = pvaj + 9;
    double t_des[3] = {acc[0] + 0.0, acc[1] + 0.0, acc[2] + g};              /* pt_publisher.py:198 */
    const double tn = sqrt(t_des[0] * t_des[0] + t_des[1] * t_des[1] + t_des[2] * t_des[2]);
    double zb[3] = {t_des[0] / tn, t_des[1] / tn, t_des[2] / tn};            /* :204 */
    const double u1 = tn * mass;                                             /* :206 */
    const double xc[3] = {cos(yaw[0]), sin(yaw[0]), 0.0};                    /* :208 */
    double zx[3], yb[3], xb[3];
    cross3(zb, xc, zx);                                                      /* :209 */
    const double nzx = sqrt(zx[0] * zx[0] + zx[1] * zx[1] + zx[2] * zx[2]);
    for (int i = 0; i < 3; ++i) yb[i] = zx[i] / nzx;                         /* :215 */
    cross3(yb, zb, xb);                                                      /* :217 */
    const double zj = zb[0] * jerk[0] + zb[1] * jerk[1] + zb[2] * jerk[2];
    double ho[3];
    for (int i = 0; i < 3; ++i) ho[i] = (mass / u1) * (jerk[i] - zj * zb[i]); /* :222 */
    const double p = -(ho[0] * yb[0] + ho[1] * yb[1] + ho[2] * yb[2]);       /* :223 */
    const double q = ho[0] * xb[0] + ho[1] * xb[1] + ho[2] * xb[2];          /* :224 */
    const double r = yaw[1] * zb[2];                                         /* :225 */
    double R[3][3], qq[4];
    for (int i = 0; i < 3; ++i) { R[i][0] = xb[i]; R[i][1] = yb[i]; R[i][2] = zb[i]; }   /* :218 */
    quat_from_matrix(R, qq);                                                 /* :234 */
    for (int i = 0; i < 6; ++i) x[i] = pvaj[i];
    x[6] = qq[3]; x[7] = qq[0]; x[8] = qq[1]; x[9] = qq[2];                  /* :237-240, :115-128 */
    u[0] = p; u[1] = q; u[2] = r; u[3] = u1 / mass;                          /* :138-145 */
}

void orc_ref_window(int V, int N, double dt, double mass, double g, int n_seg, const double *coeff, const double *t_cum,
                    const double *t_seg, const double *final_pt, const double *t, double *xr, double *ur)
{
    for (int v = 0; v < V; ++v)
        for (int k = 0; k <= N; ++k) {
            double pvaj[12], yaw[2], x[10], u[4];
            orc_traj_point(n_seg, coeff + (size_t)v * n_seg * 28, t_cum + (size_t)v * (n_seg + 1), t_seg + (size_t)v * n_seg,
                           final_pt + (size_t)v * 3, t[v] + k * dt, pvaj, yaw);
            orc_diff_flatness(mass, g, pvaj, yaw, x, u);
            for (int i = 0; i < 10; ++i) xr[((size_t)v * (N + 1) + k) * 10 + i] = x[i];
            if (k < N)
                for (int i = 0; i < 4; ++i) ur[((size_t)v * N + k) * 4 + i] = u[i];
        }
}
