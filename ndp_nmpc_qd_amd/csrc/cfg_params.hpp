// cfg_params.hpp -- ndp_cfg (C-ABI) -> RtiParams (kernel argument block) and the reference's constants.
#pragma once
#include <string.h>

#include "../../include/ndp_nmpc.h"
#include "rti_wave.hpp"

namespace ndp {

// Reference constants: params/nmpc_params.py:9-35, params/fhnp_params.py:9,12,19, params/downwash_params.py:10
inline void fill_default_cfg(ndp_cfg *c)
{
    memset(c, 0, sizeof(*c));
    c->batch = 1;
    c->N = 20;
    c->n_rti = 1;
    c->use_fd = 0;
    c->qp_mode = NDP_QP_AUTO;
    c->iter_max = 50;
    c->device = 0;
    c->dt = 2.0 / 20.0;
    c->mass = 1.4844;
    c->gravity = 9.81;
    c->r_horiz = 1.0;
    const double Qd[10] = {300, 300, 400, 10, 10, 10, 0, 10, 10, 100};
    const double Rd[4] = {10, 10, 10, 5};
    memcpy(c->Qd, Qd, sizeof(Qd));
    memcpy(c->Rd, Rd, sizeof(Rd));
    for (int i = 0; i < 3; ++i) {
        c->lbu[i] = -6.0; c->ubu[i] = 6.0;
        c->lbv[i] = -20.0; c->ubv[i] = 20.0;
    }
    c->lbu[3] = 0.0;
    c->ubu[3] = 9.81 / 0.36;
    c->mu0 = 10.0;
    c->thr0 = 0.1;
    c->tol = 1e-8;     // HPIPM's default [acados-knowledge]
    c->mu_floor = 0.1; // the centring target never goes below mu_floor * tol (see RtiWave::ipm)
    c->tau = 0.995;
    c->auto_margin = 0.1;
    c->ts_nmpc = 0.02;
    c->ipm_refine = 2;
    c->refine_gamma = 1e4;
    c->as_iter_max = 8;
    c->as_gamma = 1e12;
}

inline RtiParams to_params(const ndp_cfg &c)
{
    RtiParams p;
    p.N = c.N; p.n_rti = c.n_rti; p.use_fd = c.use_fd; p.qp_mode = c.qp_mode; p.iter_max = c.iter_max;
    p.dt = c.dt; p.inv_mass = 1.0 / c.mass; p.g = c.gravity;
    memcpy(p.Qd, c.Qd, sizeof(p.Qd)); memcpy(p.Rd, c.Rd, sizeof(p.Rd));
    memcpy(p.lbu, c.lbu, sizeof(p.lbu)); memcpy(p.ubu, c.ubu, sizeof(p.ubu));
    memcpy(p.lbv, c.lbv, sizeof(p.lbv)); memcpy(p.ubv, c.ubv, sizeof(p.ubv));
    p.mu0 = c.mu0; p.thr0 = c.thr0; p.tol = c.tol; p.tau = c.tau; p.auto_margin = c.auto_margin; p.mu_floor = c.mu_floor;
    p.refine = c.ipm_refine; p.refine_gamma = c.refine_gamma;
    p.as_iter_max = c.as_iter_max; p.as_gamma = c.as_gamma;
    fill_quotients(p);
    return p;
}

// constraint slots needed for horizon N (64 box constraints per slot, 7N-3 constraints)
inline int slots_for(int N) { return (7 * N - 3 + 63) / 64; }

}  // namespace ndp
