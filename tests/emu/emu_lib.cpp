// emu_lib.cpp -- runs the product's wave program (rti_wave.hpp) on the host wave emulator.
// TEST INFRASTRUCTURE ONLY: built by tests/emu/Makefile into tests/emu/libndp_emu.so, loaded by tests/.
#include <vector>

#include "wave_emu.hpp"
#include "../../ndp_nmpc_qd_amd/csrc/cfg_params.hpp"

extern "C" {

int emu_default_cfg(ndp_cfg *c) { ndp::fill_default_cfg(c); return 0; }

int emu_lds_doubles(int N) { return ndp::lds_doubles(N) + ndp::DBG_EXTRA; }

int emu_lds_layout(int N, int *out8) { ndp::lds_layout(N, out8); return 0; }

// One instance, one emulated wave.  counters: [mfma (16x16x4), lds_ld, lds_st, readlane, mfma4 (4x4x4, four blocks)]
// act: the instance's kept active set (RtiIo::act: ndp::act_pitch(N) bytes), or null; *iters = the step's raw iteration word
int emu_rti_step_act(const ndp_cfg *cfg, const double *x0, const double *xr, const double *ur, const float *f,
                     double *X, double *U, double *u0, int *status, int *iters, double *lds_dump, long *counters, signed char *act);

int emu_act_pitch(int N) { return ndp::act_pitch(N); }

int emu_rti_step(const ndp_cfg *cfg, const double *x0, const double *xr, const double *ur, const float *f,
                 double *X, double *U, double *u0, int *status, int *iters, double *lds_dump, long *counters)
{
    return emu_rti_step_act(cfg, x0, xr, ur, f, X, U, u0, status, iters, lds_dump, counters, nullptr);
}

int emu_rti_step_act(const ndp_cfg *cfg, const double *x0, const double *xr, const double *ur, const float *f,
                     double *X, double *U, double *u0, int *status, int *iters, double *lds_dump, long *counters, signed char *act)
{
    ndp::RtiParams P = ndp::to_params(*cfg);
    const int n = ndp::lds_doubles(P.N);
    std::vector<double> lds((size_t)n, 0.0 / 0.0);   // NaN-poisoned: any read of unwritten LDS shows up
    emu::Wave::lds_limit() = n;
    emu::stats() = emu::Stats();
    double kc[ndp::KC_HOST];
    ndp::fill_kc(P, kc);
    ndp::RtiIo io{x0, xr, ur, f, X, U, u0, status, iters, lds_dump, 0, kc};
    std::vector<int> tb(ndp::TB_WORDS);
    ndp::fill_tables(P.N, tb.data(), cfg->qp_precision >= 3 ? 1 : 0);
    io.tables = tb.data();
    io.act = act;
    const int ns = ndp::slots_for(P.N);
    // horizon 20 and the 5-slot form run as the product does (host-built tables; N = 20 also compile-time horizon);
    // the other horizons build the tables in the wave program
    if (cfg->qp_precision == 3) ndp::RtiWave<emu::Wave32, 5, 0, true>::run(P, io, lds.data());          // the real fp32 / bf16 backends' layout
    else if (cfg->qp_precision == 4) ndp::RtiWave<emu::WaveBF16, 5, 0, true>::run(P, io, lds.data());
    else if (cfg->qp_precision == 1) ndp::RtiWave<emu::Wave, 5, 0, true, 0, 1>::run(P, io, lds.data());
    else if (cfg->qp_precision == 2) ndp::RtiWave<emu::Wave, 5, 0, true, 0, 2>::run(P, io, lds.data());
    else if (P.N == 20 && P.n_rti == 1) ndp::RtiWave<emu::Wave, 3, 20, true, 1>::run(P, io, lds.data());
    else if (ns <= 3) ndp::RtiWave<emu::Wave, 3>::run(P, io, lds.data());
    else if (ns <= 5) ndp::RtiWave<emu::Wave, 5, 0, true>::run(P, io, lds.data());
    else return -1;
    if (counters) {
        counters[0] = emu::stats().mfma; counters[1] = emu::stats().lds_ld;
        counters[2] = emu::stats().lds_st; counters[3] = emu::stats().readlane;
        counters[4] = emu::stats().mfma4;
    }
    return 0;
}

// The late-force path (RtiIo::f_late: the downwash force of a launch that ran one tick ahead on a second stream): the same step
// with f handed over as f_late behind a flag that is already set (ready = 1) or not (ready = 0: the wait "times out" at once on the
// emulator -> zero force, status 5).
int emu_rti_step_late(const ndp_cfg *cfg, const double *x0, const double *xr, const double *ur, const float *f, int ready,
                      double *X, double *U, double *u0, int *status, int *iters, int *missed)
{
    ndp::RtiParams P = ndp::to_params(*cfg);
    if (P.N != 20 || P.n_rti != 1 || cfg->qp_precision != 0) return -1;
    const int n = ndp::lds_doubles(P.N);
    std::vector<double> lds((size_t)n, 0.0 / 0.0);
    emu::Wave::lds_limit() = n;
    double kc[ndp::KC_HOST];
    ndp::fill_kc(P, kc);
    ndp::RtiIo io{x0, xr, ur, nullptr, X, U, u0, status, iters, nullptr, 0, kc};
    std::vector<int> tb(ndp::TB_WORDS);
    ndp::fill_tables(P.N, tb.data(), 0);
    io.tables = tb.data();
    const unsigned long long flag = ready ? 7 : 6;
    io.f_late = f; io.late_flag = &flag; io.late_flag2 = &flag; io.late_want = 7; io.late_timeout_us = 1; io.late_missed = missed;
    ndp::RtiWave<emu::Wave, 3, 20, true, 1>::run(P, io, lds.data());
    return 0;
}

// The producer launch's view of one instance (work list, QMODE 1): RtiWave::run<DEFER = true> on the reference configuration's
// instantiation.  Returns 1 if the instance was deferred (needs the interior-point loop: nothing may have been written),
// 0 if it was solved by the early exit, < 0 on misuse.
int emu_rti_step_defer_act(const ndp_cfg *cfg, const double *x0, const double *xr, const double *ur, const float *f,
                           double *X, double *U, double *u0, int *status, int *iters, signed char *act);
int emu_rti_step_defer(const ndp_cfg *cfg, const double *x0, const double *xr, const double *ur, const float *f,
                       double *X, double *U, double *u0, int *status, int *iters)
{
    return emu_rti_step_defer_act(cfg, x0, xr, ur, f, X, U, u0, status, iters, nullptr);
}
int emu_rti_step_defer_act(const ndp_cfg *cfg, const double *x0, const double *xr, const double *ur, const float *f,
                           double *X, double *U, double *u0, int *status, int *iters, signed char *act)
{
    ndp::RtiParams P = ndp::to_params(*cfg);
    if (P.N != 20 || P.n_rti != 1 || cfg->qp_precision != 0) return -1;
    const int n = ndp::lds_doubles(P.N);
    std::vector<double> lds((size_t)n, 0.0 / 0.0);
    emu::Wave::lds_limit() = n;
    double kc[ndp::KC_HOST];
    ndp::fill_kc(P, kc);
    ndp::RtiIo io{x0, xr, ur, f, X, U, u0, status, iters, nullptr, 0, kc};
    std::vector<int> tb(ndp::TB_WORDS);
    ndp::fill_tables(P.N, tb.data(), 0);
    io.tables = tb.data();
    io.act = act;
    using Prog = ndp::RtiWave<emu::Wave, 3, 20, true, 1>;
    Prog::InBuf inb;
    emu::vd x0v;
    Prog::issue_first(P, io, inb, x0v);
    return Prog::run<true>(P, io, lds.data(), inb, x0v) ? 1 : 0;
}
}
