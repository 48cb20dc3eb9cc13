// rti_wave.hpp -- one SQP-RTI control step of one quadrotor OCP, executed by ONE 64-lane wavefront.
//
// Replaces, for a batch of independent instances, what the reference reaches through
// NMPCBodyRateController.update / NDPNMPCBodyRateController.update
// (ndp_nmpc/scripts/nmpc_ctl/nmpc_body_rate_ctl.py:93-112, ndp_nmpc_ctl/ndp_nmpc_body_rate_ctl.py:91-112):
// the acados-generated SQP_RTI / ERK / GAUSS_NEWTON / HPIPM solver configured at nmpc_body_rate_ctl.py:36-80.
//
// The program is written SPMD-style against a small "wave backend" W (per-lane value types vd/vi/vb,
// LDS access, cross-lane reads, the 16x16x4 f64 matrix instruction).  The product instantiates it with
// the gfx950 backend (wave_gfx950.hpp: v_mfma_f64_16x16x4_f64, v_readlane, LDS); tests instantiate the
// same text with a host lock-step emulator so the algorithm can be checked without a GPU.
// Control flow is wave-uniform everywhere; per-lane decisions are selects / predicated stores.
//
// Design:
//  * everything of one instance lives in that wave's LDS slice (40 KB at N=20) for the whole step;
//  * linearisation (RK4 + forward sensitivities) is lane-parallel over (stage, sensitivity column);
//  * the Riccati recursion runs in homogeneous coordinates z~ = [x(10), 1, 0, u(4)] so that one
//    16x16 f64 MFMA tile carries the Hessian AND the gradient: per stage
//        W = P~ M~ (3 mfma), H~ = M~' W + C~ (3), G = Lam^-1 H~ux (1), P~ = H~ - H~xu G (1), K~' (1)
//    with every operand already in the register layout the previous instruction produced it in;
//  * bounds are handled by a Mehrotra predictor-corrector interior-point loop around that recursion
//    (same algorithm as oracle/ndp_oracle.c), with an exact early exit when the equality-constrained
//    minimiser is strictly inside the box.
#pragma once
// A branch the wave almost never takes.  More than a layout hint: the register allocator weighs every use by its block's estimated
// frequency, and a LOOP in a cold branch (x32 per nesting level by default) outweighs the straight-line hot path in front of it --
// the hot sweep then pays accumulation-register reads for values the cold loops were given registers for.
#define NDP_RARELY(x) (__builtin_expect_with_probability(!!(x), 0, 0.9999))

#ifndef NDP_D        // wave-program functions: __device__ in the gfx950 build, plain inline under the emulator
#define NDP_D inline
#endif
// the stage loops of the sweeps are fully unrolled when the horizon is a compile-time constant (clang only: the
// host emulator build ignores it)
#if defined(__clang__)
#define NDP_UNROLL_STAGES _Pragma("unroll UNROLL_STAGES")
#define NDP_UNROLL_SWEEP _Pragma("unroll UNROLL_SWEEP")
#define NDP_KEEP_LOOP _Pragma("unroll 1")            // rare-path loops: a few registers, whatever the trip count
#else
#define NDP_UNROLL_STAGES
#define NDP_UNROLL_SWEEP
#define NDP_KEEP_LOOP
#endif
#ifdef NDP_FINE_STAMPS
#define NDP_FINE(x) x
#else
#define NDP_FINE(x)
#endif
#ifndef NDP_HD       // layout helpers used by host and device
#define NDP_HD inline
#endif

namespace ndp {

enum { NX = 10, NU = 4 };
enum { QP_AUTO = 0, QP_IPM_ALWAYS = 1 };
// Every per-stage operand of the sweeps lives in a stage block of ONE stride, constants included, so that the address
// of element e of stage k is (per-lane base of e) + k * stride: with a compile-time horizon the stage term is the
// immediate offset of the DS instruction and the sweeps carry no address arithmetic at all.
enum { MB_STRIDE = 138, CB_STRIDE = 48 };
// stage block MB_k: 6x8 [d(p,v)+/d(q,u)] | 4x7 [dq+/d(q,w)] | b(10) | constants 0, 1, h | K~'(12x4, written by the backward sweep)
// | one dump slot (where the lanes that hold no K~' entry store, so that the sweep's stores need no predicate)
enum { MB_PV = 0, MB_Q = 48, MB_B = 76, MB_ZERO = 86, MB_ONE = 87, MB_H = 88, MB_KT = 89, MB_DUMP = 137 };
// cost block CB_k: Qq(4x4) | qe(10) | re(4) | dex(6) | constant 0 | qbv(3) | deu(4) | rb(4)
enum { CB_QQ = 0, CB_QE = 16, CB_RE = 26, CB_DEX = 30, CB_ZERO = 36, CB_QBV = 37, CB_DEU = 40, CB_RB = 44 };
// constants area
enum { KC_ZERO = 0, KC_ONE = 1, KC_H = 2, KC_QD = 3, KC_RD = 13, KC_LBV = 17, KC_UBV = 21, KC_LBU = 24, KC_UBU = 28, KC_SC = 32, KC_DUMP = 48,
       // the box as QP_AUTO's acceptance test looks at it: every bound moved inwards by its margin (velocities: auto_margin; inputs: auto_margin
       // with the active set off, 0 with it on) -- made by the host (fill_kc), so that the test behind the sweep needs no scalar parameter
       // (a scalar load issued in front of the sweep makes the sweep's first LDS wait a wait for it as well: one counter for both)
       KC_LBVM = 50, KC_UBVM = 53, KC_LBUM = 56, KC_UBUM = 60, KC_SIZE = 64, KC_HOST = 64 };
// A box constraint (interior-point loop) touches eight LDS places: the step variable, the iterate value, the diagonal / gradient /
// base-gradient entries of its cost block, its two bounds and its weight.  The two layouts above are arranged so that FIVE of them
// sit at the same distance from another one for input bounds and velocity bounds alike -- three offsets per constraint slot live in
// registers instead of eight (25 registers per lane at N = 40), the rest are immediates of the DS instructions:
static_assert(CB_RE - CB_DEU == (CB_QE + 3) - (CB_DEX + 3), "gradient entry: the same distance from the diagonal entry for u and v bounds");
static_assert(CB_RB - CB_DEU == CB_QBV - (CB_DEX + 3), "base gradient: the same distance from the diagonal entry for u and v bounds");
static_assert(KC_UBU - KC_LBU == KC_UBV - KC_LBV, "upper bound: the same distance from the lower bound for u and v bounds");
static_assert(KC_RD - KC_LBU == (KC_QD + 3) - KC_LBV, "weight: the same distance from the lower bound for u and v bounds");
enum { SL_GE = CB_RE - CB_DEU, SL_GB = CB_RB - CB_DEU, SL_UB = KC_UBU - KC_LBU, SL_DW = KC_RD - KC_LBU };

struct RtiParams {
    int N, n_rti, use_fd, qp_mode, iter_max;
    double dt, inv_mass, g;
    double Qd[10], Rd[4], lbu[4], ubu[4], lbv[3], ubv[3];
    double mu0, thr0, tol, tau, auto_margin, mu_floor;
    int refine;             // interior point: refinement solves per Newton system while a state bound's barrier term exceeds refine_gamma
    double refine_gamma;
    // QP_AUTO: active-set iterations on the INPUT bounds (see RtiWave::as_check): at most as_iter_max sweeps with pinned inputs behind the
    // first one (0 = none: the equality-constrained minimiser or the interior-point loop, as in rounds 1-5); as_gamma = weight of a pin
    int as_iter_max;
    double as_gamma;
    // host-evaluated quotients (an f64 divide is a ~30-instruction VALU sequence on the device, even for uniforms)
    double h_6, h2_6, h4_24, h3_6, h4_12, two_over_h2, inv2m;
};

NDP_HD void fill_quotients(RtiParams &p)
{
    const double h = p.dt, h2 = h * h;
    p.h_6 = h / 6.0; p.h2_6 = h2 / 6.0; p.h4_24 = h2 * h2 / 24.0; p.h3_6 = h * h2 / 6.0; p.h4_12 = h2 * h2 / 12.0;
    p.two_over_h2 = 2.0 / h2;
    p.inv2m = 1.0 / (2.0 * (7 * p.N - 3));
}

enum { NDP_NSTAMP = 24 };  // doubles per instance of the whole-batch phase stamps (ndp_debug_stamps): 0-8 the program's phases, 9-15 the kernel's, 16-23 the tick prologue's

struct RtiIo {            // global-memory views of ONE instance
    const double *x0;     // [10]
    const double *xr;     // [(N+1)*10]
    const double *ur;     // [N*4]
    const float *f;       // [(N+1)*3] or null
    double *X, *U;        // persistent iterate, updated in place
    double *u0;           // [4]
    int *status, *iters;  // per-instance
    double *dbg;          // optional dump area (tests), or null
    int f_in_lds;         // 1: the caller already left f (as doubles) in the LDS staging slot TF (fused downwash)
    const double *kc;     // [KC_HOST] lane-indexable constants block prepared by the host (fill_kc)
    double *stamps = nullptr;   // optional [16] per-instance phase stamps (whole-batch profiling), or null
    const int *tables = nullptr;   // [TB_WORDS] host-built index tables (fill_tables), read when RtiWave<..., HT = true>
    double *Xm = nullptr, *Um = nullptr;   // optional second destination of the new iterate (a page-locked host block the caller reads), or null
    // Late force (downwash predicted by ANOTHER launch, one tick ahead, on a second stream): this instance's [(N+1)*3] floats, valid
    // once *late_flag and *late_flag2 are >= late_want.  The force enters the dynamics additively (dv = h f/m, dp = h^2/2 f/m: it changes the defects
    // b_k, not A_k, B_k or the cost), so it is not needed before the backward sweep: the flag is tested after the cost phase, the
    // loads fly under the linearisation, the defects are corrected after it.  f and f_in_lds must be unset then.
    const float *f_late = nullptr;
    const unsigned long long *late_flag = nullptr, *late_flag2 = nullptr;   // both words must have reached late_want (the one or two
    unsigned long long late_want = 0;                                        // 32-row tiles of the prediction that hold this instance's rows)
    int late_ready = 0;            // 1: the prediction was complete before this launch started -- no wait, ordinary (cached) loads
    unsigned late_timeout_us = 0;
    int *late_missed = nullptr;    // counts waves whose wait timed out (the step then runs with zero force and status 5)
    unsigned long long *ipm_ctr = nullptr;   // counts (monotonic) the instances whose QP went into the interior-point loop IN PLACE: what the
                                             // handle's automatic work-list rule looks at (ndp_hip.hip: queue_policy)
    int *late_slow = nullptr;      // counts waves that started before their prediction was complete (epoch path)
    // "done reading" accounting of the force slot (the launch that refills it two ticks later waits for it), see W::late_count
    unsigned *late_cnt = nullptr;          // this workgroup's group counter (monotonic)
    unsigned long long *late_done_word = nullptr;   // groups completed (monotonic)
    unsigned late_gsize = 1;               // workgroups per group and launch
    void *late_group = nullptr;            // backend-specific counter shared by the waves of a workgroup, or null
    unsigned late_group_size = 1;
    // ndp_tick: the actuator command nmpc_u_2_att_tgt makes of u_0 (nmpc_node.py:273-283), written by the control step itself beside
    // u_0: cmd[4] = [wx, wy, wz, c mass / k_throttle]; the thrust is also kept for the next estimator update.  Null = not a tick.
    double *cmd = nullptr, *thrust_keep = nullptr;
    double kthr = 0.0, cmd_mass = 0.0;
    // ndp_tick in one launch: node N of the reference window as the caller's wave made it (the list's newest entry; its slot in the
    // list is being written during this launch) -- staged over what was loaded for that row
    int have_xrN = 0;
    double xrN[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    int f_is_f64 = 0;     // 1: f points at doubles (the reference hands acados a float64 p, ndp_nmpc_body_rate_ctl.py:97-99); 0: fp32, what DownwashNN returns
    // QP_AUTO's active set of this instance, kept between control steps (the warm start of the next step's QP): one signed byte per
    // input bound, element 4k + i = input i of stage k: +1 pinned at its upper bound, -1 at its lower, 0 free.  Null = no warm start,
    // nothing kept.  (The sweeps a step's QPs took ride in the high half of the iteration word: ITERS_SWEEP_SHIFT.)
    signed char *act = nullptr;
};
NDP_HD int act_pitch(int N) { return 4 * N; }      // bytes per instance of the active-set record
// *RtiIo::iters = interior-point iterations of the step (low half) + Riccati sweeps its QP_AUTO solves took before them (high half);
// COND_ACCEPTED in the sweep count (config 5's study): a condensed solve's result was kept
enum { ITERS_SWEEP_SHIFT = 16, ITERS_IPM_MASK = 0xffff, COND_ACCEPTED = 0x1000 };

struct LdsMap {
    int KC, SC, XI, UI, ZX, ZU, CX, CU, ZD, MB, CB, KT, TXR, TUR, TF, AS, total;
};

NDP_HD LdsMap make_map(int N)
{
    LdsMap m;
    int o = 0;
    m.KC = o; o += KC_SIZE;
    m.SC = m.KC + KC_SC;   // 16-double scratch for the 4x4 inverse
    m.XI = o; o += (N + 1) * NX;
    m.UI = o; o += N * NU;
    m.ZX = o; o += (N + 1) * NX;
    m.ZU = o; o += N * NU;
    m.CX = o; o += (N + 1) * NX;
    m.CU = o; o += N * NU;
    m.ZD = o; o += (N + 1) * NX + N * NU;   // shadow of ZX|ZU: where the lanes that hold no forward-sweep result store
    m.MB = o; o += N * MB_STRIDE;
    m.CB = o; o += (N + 1) * CB_STRIDE;
    // the active set of the input bounds (RtiWave::ActSet) where the five-slot kernels keep it (they sit at the register limit and park
    // the set here instead of holding it across the sweeps; the three-slot kernels hold it in registers and leave the area alone)
    m.AS = o; o += N * NU;
    m.total = o;
    m.KT = m.MB + MB_KT;   // K~' of stage 0; stage k at + k * MB_STRIDE
    // staged inputs (17N + 13 doubles) alias ZX|ZU|CX|CU|ZD (42N + 30), which are dead until the first sweep / step
    m.TXR = m.ZX;
    m.TUR = m.TXR + (N + 1) * NX;
    m.TF = m.TUR + N * NU;
    return m;
}

NDP_HD int lds_doubles(int N) { return make_map(N).total; }
enum { DBG_EXTRA = 32 };   // the debug dump is the LDS image followed by 16 phase stamps and 16 fine stamps
// layout facts for tests and scripts: {XI, MB, CB, MB_STRIDE, CB_STRIDE, total, stamps, 0}
NDP_HD void lds_layout(int N, int *out)
{
    const LdsMap m = make_map(N);
    out[0] = m.XI; out[1] = m.MB; out[2] = m.CB; out[3] = MB_STRIDE; out[4] = CB_STRIDE; out[5] = m.total; out[6] = m.total; out[7] = 0;
}

// the constants area of the LDS image, written once by the host and copied by the kernel (one coalesced load
// instead of a 30-way select chain per launch)
NDP_HD void fill_kc(const RtiParams &P, double *kc)
{
    for (int i = 0; i < KC_HOST; ++i) kc[i] = 0.0;
    kc[KC_ONE] = 1.0;
    kc[KC_H] = P.dt;
    for (int i = 0; i < 10; ++i) kc[KC_QD + i] = P.Qd[i];
    for (int i = 0; i < 4; ++i) { kc[KC_RD + i] = P.Rd[i]; kc[KC_LBU + i] = P.lbu[i]; kc[KC_UBU + i] = P.ubu[i]; }
    for (int i = 0; i < 3; ++i) { kc[KC_LBV + i] = P.lbv[i]; kc[KC_UBV + i] = P.ubv[i]; }
    const double um = (P.qp_mode == QP_AUTO && P.as_iter_max > 0) ? 0.0 : P.auto_margin;
    for (int i = 0; i < 4; ++i) { kc[KC_LBUM + i] = P.lbu[i] + um; kc[KC_UBUM + i] = P.ubu[i] - um; }
    for (int i = 0; i < 3; ++i) { kc[KC_LBVM + i] = P.lbv[i] + P.auto_margin; kc[KC_UBVM + i] = P.ubv[i] - P.auto_margin; }
}

// NC > 0: the horizon is a compile-time constant (every LDS offset folds into the DS instruction's immediate, every
// task-range predicate into a constant lane mask); NC = 0: horizon read from RtiParams at run time.
// HT: the per-lane index tables come from a host-built block (RtiIo::tables, fill_tables) instead of ~900 integer
// VALU instructions per launch.
// NR > 0: compile-time number of RTI iterations per step (NR = 1 removes the iteration loop, which otherwise makes the
// compiler hoist every address computation of the loop body in front of it and spill them).
// PREC: precision study of BASELINE config 5 (0 = the product path).  1 / 2 round every operand of every matrix
// instruction of the Riccati sweeps to fp32 / bf16 and every accumulator to fp32 after each instruction -- the numbers
// a sweep on v_mfma_f32_16x16x4_f32 / a bf16-input MFMA with fp32 accumulation would see; everything else stays f64.
// LEAN: no stiff path in the interior-point loop (see ipm(): STIFF).  For the control step that runs BESIDE the downwash launch of the next
// tick (late force): the two kernels' waves share a SIMD's 512 registers, 184 of them the downwash launch's -- the stiff sweeps cost 80.
// COND: BASELINE config 5's study of the CONDENSED QP (cond_qp.hpp; gfx950 backend only): 1 / 2 = the first solve of every QP in condensed
// form on the fp32 / bf16 matrix instructions; its result is kept when it passes the fp64 inside-the-box test, else the fp64 path below
// solves the QP.  0 = the product.
template <class W, int NSLOT, int NC = 0, bool HT = false, int NR = 0, int PREC = 0, bool LEAN = false, int COND = 0>
struct RtiWave {
    static NDP_D int horizon(const RtiParams &P) { return NC ? NC : P.N; }
    static constexpr int UNROLL_STAGES = NC > 0 ? NC : 1;
    static constexpr int RESYM = 10;      // riccati_sweep: stages between two re-symmetrisations of H~
    using vd = typename W::vd;
    using vi = typename W::vi;
    using vb = typename W::vb;
    using vd4 = typename W::vd4;
    // Matrix values of the Riccati sweeps: what the backend's matrix instruction consumes and produces.  The product backend
    // has md = double (v_mfma_f64_16x16x4_f64); the BASELINE-config-5 backends have md = float (v_mfma_f32_16x16x4_f32, and the
    // bf16-input v_mfma_f32_16x16x16_bf16 with fp32 accumulators).  Everything else -- linearisation, the 4x4 inverse, the
    // interior-point bookkeeping, the LDS image -- stays in double; W::to_m / W::to_d convert at the sweeps' boundary.
    using md = typename W::md;
    using md4 = typename W::md4;
    // Products with one four-wide side -- the gains K~' = H~ux' (-Lam^-1), adj(Lam) T, and every matrix-VECTOR product of the
    // forward and second-solve sweeps -- run on the backend's four-block 4x4x4 instruction where it has one (W::mfma4, the f64
    // backend): same operand registers as the 16x16x4 form, a quarter of its issue time.  A result then holds element 4b + i of
    // a vector in lane (j, b, i) = j + 4b + 16i, whereas the next product wants element 4c + k as chunk register c of the lanes
    // with l >> 4 == k: W::rowb<c> (one row broadcast) makes that register from block c.
    static constexpr bool MMA4 = W::has_mma4 && PREC == 0;
#ifndef NDP_DELTA_MAX_N
#define NDP_DELTA_MAX_N 40
#endif
    static constexpr int DELTA_MAX_N = NDP_DELTA_MAX_N;   // longest compile-time horizon whose corrector is a second solve (2 registers per stage)
    static NDP_D md mma4(md a, md b, md c) { return W::mfma4(a, b, c); }
    // one 4-deep contraction step D = A B + C
    static NDP_D md4 mma(md a, md b, const md4 &c)
    {
        if constexpr (W::packed_k) {
            const md aa[1] = {a}, bb[1] = {b};
            return W::mfma_k(aa, bb, 1, c);
        } else if constexpr (PREC == 0) {
            return W::mfma(a, b, c);
        } else {
            md4 d = W::mfma(W::round_op(a, PREC), W::round_op(b, PREC), c);
            for (int r = 0; r < 4; ++r) d.r[r] = W::round_op(d.r[r], 1);
            return d;
        }
    }
    // n <= 4 contraction steps on one accumulator: a backend whose instruction contracts 16 deep (bf16) packs them into one
    template <int n>
    static NDP_D md4 mman(const md *a, const md *b, const md4 &c)
    {
        if constexpr (W::packed_k) return W::mfma_k(a, b, n, c);
        else {
            md4 d = c;
            for (int i = 0; i < n; ++i) d = mma(a[i], b[i], d);
            return d;
        }
    }
    // Lane id of the per-iteration phases (inputs, cost, linearisation).  With several RTI iterations per step (unrolled: NR = 2,
    // config 5) the compiler finds the phases' index arithmetic -- task -> (stage, row) -> LDS address, ~100 values per lane -- common
    // to the copies, computes it ONCE at the kernel's start and keeps it alive across the first iteration's sweeps and its
    // interior-point loop: 93 scratch stores in the set-up block, 51 loads in front of the linearisations (profiles/r03_isa_audit.txt).
    // An opaque lane id per phase makes every copy compute its own (a few hundred integer instructions per iteration).  One-iteration
    // kernels (the headline) keep the plain id: their code does not change.
    static NDP_D vi lane_it() { if constexpr (NR == 1) return W::lane(); else return W::lane_here(); }
    // -Lam_k^-1 of every stage, kept for the second solve (delta_sweep).  On the four-block instruction only the lanes of block 3
    // (columns 12..15: rows 12..15 of the A operand [K~' ; -Lam^-1 - I]) ever use it, while every block holds a copy when it is
    // produced: FOUR stages share one register -- the lanes of block b keep stage 4q + b -- and the second solve brings the block it
    // wants to block 3 with one row rotation (DPP row_ror, no LDS).  N / 4 registers per lane instead of N: at N = 40 the 40
    // registers (80 VGPRs) were the part of the interior-point state that went to scratch memory, one store and one load per stage.
    static constexpr bool PACKL = MMA4;
    static constexpr int linv_regs(int n) { return PACKL ? (n + 3) / 4 : n; }
    // register q of the pack, reached through CONSTANT indices only (a switch that folds once the stage loops are unrolled): an
    // index computed from the loop counter keeps the array in private memory -- the compiler promotes it to registers before it
    // unrolls the loops, and only if every index is a constant by then
    template <class F>
    static NDP_D void linv_reg(md *linv, int q, F &&f)
    {
        switch (q) {
#define NDP_LC(i) case i: f(linv[i]); break;
        NDP_LC(0) NDP_LC(1) NDP_LC(2) NDP_LC(3) NDP_LC(4) NDP_LC(5) NDP_LC(6) NDP_LC(7)
        NDP_LC(8) NDP_LC(9) NDP_LC(10) NDP_LC(11) NDP_LC(12) NDP_LC(13) NDP_LC(14) NDP_LC(15)
#undef NDP_LC
        default: break;
        }
    }
    static_assert(NC <= 64, "linv_reg serves packs of up to 16 registers");
    static NDP_D void linv_put(md *linv, int k, int N, md v)
    {
        if constexpr (PACKL) {
            // the backward sweep comes down from stage N - 1: the first stage that writes a register fills all its lanes (every
            // lane then holds a finite value of SOME stage -- the lanes the second solve does not read are multiplied by zero)
            const bool whole = (k & 3) == 3 || k == N - 1;
            const vb mine = ((W::lane() >> 2) & 3) == (k & 3);
            linv_reg(linv, k >> 2, [&](md &r) { r = whole ? v : W::msel(mine, v, r); });
        } else linv[k] = v;
    }
    static NDP_D md linv_get(const md *linv, int k)      // PACKL: valid in the lanes of block 3 only (all the second solve reads)
    {
        if constexpr (PACKL) {
            md v = W::to_m(vd(0.0));
            linv_reg(const_cast<md *>(linv), k >> 2, [&](md &r) { v = r; });
            switch (k & 3) {
            case 0: return W::template rowror4<3>(v);
            case 1: return W::template rowror4<2>(v);
            case 2: return W::template rowror4<1>(v);
            default: return v;
            }
        } else return linv[k];
    }
    static NDP_D int mb(int k) { return k * int(MB_STRIDE); }   // stage offsets (immediates once the stage loops are unrolled)
    static NDP_D int cb(int k) { return k * int(CB_STRIDE); }
    using lp = typename W::lds_ptr;   // pointer into this wave's LDS slice

    struct Tables {               // per-lane LDS offsets of stage 0 (doubles); stage k adds k * MB_STRIDE / CB_STRIDE / NX / NU
        vi mk_off[3];             // M~ as B operand / M~' as A operand: element (4c+g, j)
        vi fw_off[3];             // forward A operand [M~x ; K~]: rows j<12 element (j, 4c+g) of M~, rows j>=12 K~[j-12][4c+g]
        vi mu_off;                // M~ columns 12..15 (B~) as A operand: element (j, 12+g)
        vi c_off[4];              // C~ in accumulator layout: element (g+4r, j)
        vi kt_off[3];             // where lanes j>=12 keep K~'[4c+g][j-12]
        vi kt_st[3];              // store form of kt_off: lanes j<12 aim at the block's dump slot
        vi zu_st, zx_st[3];       // forward-sweep results du[g], x+[4c+g] of lanes j == 0; the other lanes aim at the shadow ZD
        // 4x4x4 form (MMA4): a result register holds row x = 4b + i in lane j + 4b + 16i (j = l & 3, b = (l >> 2) & 3, i = l >> 4)
        vi kt_st4;                // K~'[x][j] of every lane with x < 12 (the others: dump slot): ONE store per stage
        vi zx_st4;                // x+[x] of the lanes with j == 0 and x < 10 (the others: shadow ZD): ONE store per stage
        vb kt_pred;               // j >= 12
        vb lo4;                   // j < 4
        vb col0;                  // j == 0
        // 4x4 inverse (lam_inverse): LDS scratch SC holds Lam row-major
        vi lam_w_off;             // lanes j >= 12 publish H~[12+g][j] to SC[g*4 + j-12]
        vi minor_one[3], minor_pair[3];   // the 3x3 minor of (g, j&3), row a: one single element + one ALIGNED pair of adjacent ones
                                  // (columns {0..3} \ {j&3} always hold the pair (2,3) or (0,1): one 16-byte read instead of two
                                  // 8-byte ones; the column order (single, pair) is the natural one or a cyclic shift of it --
                                  // the same determinant)
        vi own_off;               // Lam[g][j&3]
        vd cof_sign;              // (-1)^(g + j&3)
        vd adj_a, adj_b;          // cof_sign * [j < 4], -cof_sign * [j >= 12]: minor determinant -> MFMA operand in one multiply
        vb lam_diag;              // g == j&3
    };

    struct Slots {                // box constraints, 64 per slot
        vb valid[NSLOT];
        vi zoff[NSLOT];           // where the bounded step variable lives in ZX/ZU; the iterate value: + io (XI|UI and ZX|ZU are laid out alike)
        vi de_off[NSLOT];         // diagonal entry of the cost block; gradient: + SL_GE, base gradient: + SL_GB
        vi lb_off[NSLOT];         // lower bound in the constants area; upper bound: + SL_UB, weight: + SL_DW
        int io;                   // XI - ZX
        vd lo[NSLOT], hi[NSLOT];  // step bounds lb - cur, ub - cur: filled by load_bounds for the inside-the-box test; the interior-point
                                  // loop re-reads them from LDS where it needs them (bounds()) instead of holding 4 NSLOT registers
        vd tl[NSLOT], tu[NSLOT], ll[NSLOT], lu[NSLOT];
        vd pl[NSLOT], pu[NSLOT];  // predictor's dlam * dt per bound: all that the corrector keeps of the affine step across its sweep
    };

    // ---------------------------------------------------------------- index tables
    // element (r, c) of M~_0 = [[A b 0 B], [0 1 0 0], [0 0 0 0]] -> LDS offset (stage k: + k * MB_STRIDE)
    static NDP_HD vi m_entry(const LdsMap &m, vi r, vi c)
    {
        vb isP = r < 3, isPV = r < 6, isQ = (r >= 6) && (r < 10);
        vb colq = (c >= 6) && (c < 10), colu = c >= 12;
        vi colpos = W::sel(colq, c - 6, c - 8);
        vb var_pv = isPV && (colq || colu);
        vb var_q = isQ && (colq || (colu && (c < 15)));
        vb var_b = (r < 10) && (c == 10);
        vb one = (isPV && (c == r)) || ((r == 10) && (c == 10));
        vb hh = isP && (c == r + 3);
        vi rel = W::sel(var_pv, r * 8 + colpos + int(MB_PV),
                 W::sel(var_q, (r - 6) * 7 + colpos + int(MB_Q),
                 W::sel(var_b, r + int(MB_B),
                 W::sel(one, vi(int(MB_ONE)), W::sel(hh, vi(int(MB_H)), vi(int(MB_ZERO)))))));
        return rel + m.MB;
    }

    // element (row, col) of C~_0 = [[Q q 0 0], [q' 0 0 r'], [0], [0 r 0 R]] -> LDS offset (stage k: + k * CB_STRIDE)
    static NDP_HD vi c_entry(const LdsMap &m, vi row, vi col)
    {
        vb rx = row < 10, cx = col < 10, ru = row >= 12, cu = col >= 12;
        vb dgx = rx && (row == col) && (row < 6);
        vb qq = (row >= 6) && rx && (col >= 6) && cx;
        vb gx = (rx && (col == 10)) || ((row == 10) && cx);
        vi gxi = W::sel(rx, row, col);
        vb dgu = ru && (col == row);
        vb gu = (ru && (col == 10)) || ((row == 10) && cu);
        vi gui = W::sel(ru, row - 12, col - 12);
        vi rel = W::sel(dgx, row + int(CB_DEX),
                 W::sel(qq, (row - 6) * 4 + (col - 6) + int(CB_QQ),
                 W::sel(gx, gxi + int(CB_QE),
                 W::sel(dgu, row - 12 + int(CB_DEU),
                 W::sel(gu, gui + int(CB_RE), vi(int(CB_ZERO)))))));
        return rel + m.CB;
    }

    // lane-derived predicates and constants of the tables (cheap: a handful of compares)
    // Lane (g, j): g = lane >> 4 is the lane's row group, j = W::lcol(lane) the matrix column it holds: lane & 15 for the f64
    // instruction; the f32 / bf16 instructions keep row 4g + r (not g + 4r) in accumulator register r, which becomes the f64
    // picture again once rows AND columns are renumbered by i -> (i >> 2) + 4 (i & 3) -- so those backends only report a
    // different j and everything below (tables, predicates, operand chaining) reads the same.
    static NDP_HD void lane_preds(Tables &T)
    {
        vi lane = W::lane();
        vi g = lane >> 4, j = W::lcol(lane), jc = j & 3;
        T.kt_pred = j >= 12;
        T.lo4 = j < 4;
        T.col0 = j == 0;
        T.cof_sign = W::sel(((g + jc) & 1) == 1, vd(-1.0), vd(1.0));
        T.lam_diag = g == jc;
        // the 4x4x4 products read the 4x4 operand from every lane (block b = its own copy): no lane mask
        T.adj_a = MMA4 ? T.cof_sign : W::sel(T.lo4, T.cof_sign, vd(0.0));
        T.adj_b = MMA4 ? -T.cof_sign : W::sel(T.kt_pred, -T.cof_sign, vd(0.0));
    }

    static NDP_HD void build_tables(const LdsMap &m, Tables &T)
    {
        lane_preds(T);
        vi lane = W::lane();
        vi g = lane >> 4, j = W::lcol(lane);
        for (int c = 0; c < 3; ++c) T.mk_off[c] = m_entry(m, g + 4 * c, j);
        for (int r = 0; r < 4; ++r) T.c_off[r] = c_entry(m, g + 4 * r, j);
        for (int c = 0; c < 3; ++c) {
            T.kt_off[c] = (g + 4 * c) * 4 + (j & 3) + m.KT;
            T.fw_off[c] = W::sel(T.kt_pred, T.kt_off[c], m_entry(m, j, g + 4 * c));
            T.kt_st[c] = W::sel(T.kt_pred, T.kt_off[c], vi(m.MB + int(MB_DUMP)));
            vi xi = W::sel(g + 4 * c < 10, g + 4 * c, vi(0));
            T.zx_st[c] = xi + W::sel(T.col0 && (g + 4 * c < 10), vi(m.ZX), vi(m.ZD));
        }
        T.mu_off = m_entry(m, j, g + 12);
        T.zu_st = g + W::sel(T.col0, vi(m.ZU), vi(m.ZD + (m.ZU - m.ZX)));
        vi jc = j & 3;
        T.lam_w_off = W::sel(T.kt_pred, g * 4 + jc + m.SC, vi(m.KC + KC_DUMP));
        for (int a = 0; a < 3; ++a) {
            vi ra = W::sel(g <= a, vi(a + 1), vi(a));        // rows {0..3} \ {g}
            // cols {0..3} \ {jc}: jc < 2 -> single {1 or 0}, pair (2,3);  jc >= 2 -> pair (0,1), single {3 or 2}
            vi single = W::sel(jc < 2, W::sel(jc == 0, vi(1), vi(0)), W::sel(jc == 2, vi(3), vi(2)));
            vi pair0 = W::sel(jc < 2, vi(2), vi(0));
            T.minor_one[a] = ra * 4 + single + m.SC;
            T.minor_pair[a] = ra * 4 + pair0 + m.SC;
        }
        T.own_off = g * 4 + jc + m.SC;
        vi x4 = ((lane >> 2) & 3) * 4 + g, j4 = lane & 3;
        T.kt_st4 = W::sel(x4 < 12, x4 * 4 + j4 + m.KT, vi(m.MB + int(MB_DUMP)));
        T.zx_st4 = W::sel(x4 < 10, x4, vi(0)) + W::sel((j4 == 0) && (x4 < 10), vi(m.ZX), vi(m.ZD));
    }

    // the integer fields of Tables in a fixed order: f(index, field).  Used by the host to serialise the tables
    // (fill_tables) and by the device to read them back (load_tables).
    enum { TB_FIELDS = 36 };
    template <class F>
    static NDP_HD void for_each_int(Tables &T, F &&f)
    {
        int i = 0;
        for (int c = 0; c < 3; ++c) { f(i++, T.mk_off[c]); f(i++, T.fw_off[c]); f(i++, T.kt_off[c]); f(i++, T.zx_st[c]); }
        for (int c = 0; c < 3; ++c) f(i++, T.kt_st[c]);
        i++;   // pad to a multiple of four fields
        for (int r = 0; r < 4; ++r) f(i++, T.c_off[r]);
        f(i++, T.mu_off); f(i++, T.zu_st); f(i++, T.lam_w_off); f(i++, T.own_off);
        for (int a = 0; a < 3; ++a) f(i++, T.minor_one[a]);
        for (int a = 0; a < 3; ++a) f(i++, T.minor_pair[a]);
        i += 3;   // (three fields fewer than the nine single offsets of rounds 1-3: the positions behind stay where they were)
        f(i++, T.kt_st4); f(i++, T.zx_st4);
    }
    // block layout [field / 4][lane][field % 4]: the four fields of a group are one 16-byte load per lane, a
    // contiguous 1 KB per wave
    static NDP_HD int tb_word(int field, int lane) { return ((field >> 2) * 64 + lane) * 4 + (field & 3); }

    static NDP_D void load_tables(const int *g, Tables &T)
    {
        lane_preds(T);
        vi lane = W::lane();
        for_each_int(T, [&](int i, vi &fld) { fld = W::gldi(g, lane * 4 + ((i >> 2) * 256 + (i & 3))); });
    }

    // ---------------------------------------------------------------- inputs
    // largest horizon this instantiation serves (7N-3 box constraints in 64*NSLOT slots) and the number of
    // 64-lane rounds needed to move its arrays: compile-time so that every global load is in flight at once
    static constexpr int NMAXS = (64 * NSLOT + 3) / 7;
    static constexpr int RX = ((NMAXS + 1) * NX + 63) / 64, RU = (NMAXS * NU + 63) / 64, RF = ((NMAXS + 1) * 3 + 63) / 64;

    struct InBuf { vd xr[RX], xi[RX], ur[RU], ui[RU], f[RF], kc; };

    // issue every global load of this instance (nothing waits here)
    static NDP_D void issue_inputs(const RtiParams &P, const RtiIo &io, InBuf &b, bool first)
    {
        const int N = horizon(P);
        vi lane = lane_it();
        const int nx = (N + 1) * NX, nu = N * NU, nf = (N + 1) * 3;
        const bool have_f = P.use_fd && io.f && !io.f_in_lds;
        // unconditional loads from a clamped index (a predicated load compiles to a branchy block each);
        // lanes past the end fetch the last element and are masked at the LDS store
        for (int t = 0; t < RX; ++t) {
            vi i = W::imin(lane + 64 * t, nx - 1);
            b.xr[t] = W::gldu(io.xr, i);
            b.xi[t] = first ? W::gldu(io.X, i) : vd(0.0);
        }
        for (int t = 0; t < RU; ++t) {
            vi i = W::imin(lane + 64 * t, nu - 1);
            b.ur[t] = W::gldu(io.ur, i);
            b.ui[t] = first ? W::gldu(io.U, i) : vd(0.0);
        }
        for (int t = 0; t < RF; ++t) {
            vi i = W::imin(lane + 64 * t, nf - 1);
            b.f[t] = have_f ? (NDP_RARELY(io.f_is_f64 != 0) ? W::gldu(reinterpret_cast<const double *>(io.f), i) : W::gldfu(io.f, i)) : vd(0.0);
        }
        b.kc = first ? W::gldu(io.kc, W::imin(lane, KC_HOST - 1)) : vd(0.0);
    }

    // land them in LDS (first use of the loaded values: the wait sits here)
    static NDP_D void commit_inputs(const RtiParams &P, const LdsMap &m, const InBuf &b, lp lds, bool first, const RtiIo &io)
    {
        const int N = horizon(P);
        vi lane = lane_it();
        const int nx = (N + 1) * NX, nu = N * NU, nf = (N + 1) * 3;
        // the loads came from clamped indices: lanes past the end hold a copy of the last element and store it to the last
        // slot again -- an identical duplicate, so no store needs a predicate (a predicated LDS store is an exec-mask
        // save / branch / restore around it)
        for (int t = 0; t < RX; ++t) {
            vi i = W::imin(lane + 64 * t, nx - 1);
            W::st(lds, i + m.TXR, b.xr[t]);
            if (first) W::st(lds, i + m.XI, b.xi[t]);
        }
        if (NDP_RARELY(io.have_xrN != 0)) {       // (DS operations of a wave execute in order: this lands on top of row N's loaded values)
            vd v = vd(io.xrN[9]);
            for (int i = 8; i >= 0; --i) v = W::sel(lane == i, vd(io.xrN[i]), v);
            W::stp(lds, lane + (m.TXR + N * int(NX)), v, lane < int(NX));
        }
        for (int t = 0; t < RU; ++t) {
            vi i = W::imin(lane + 64 * t, nu - 1);
            W::st(lds, i + m.TUR, b.ur[t]);
            if (first) W::st(lds, i + m.UI, b.ui[t]);
        }
        for (int t = 0; t < RF; ++t) {
            vi i = W::imin(lane + 64 * t, nf - 1);
            W::st(lds, i + m.TF, b.f[t]);
        }
        if (first) W::st(lds, W::imin(lane, KC_HOST - 1) + m.KC, b.kc);   // constants area (the scratch in its middle is dead here)
        W::sync();
    }

    // ---------------------------------------------------------------- Gauss-Newton cost blocks
    // residual [p-pr, v-vr, 0, E(qr) q, u-ur]; nmpc_body_rate_ctl.py:164-180
    static NDP_D void build_cost(const RtiParams &P, const LdsMap &m, lp lds)
    {
        const int N = horizon(P);
        vi lane = lane_it();
        // Three task families, each at most a few 64-lane rounds; lanes past a family's last task repeat that task and store
        // the same values to the same places (no predicates).  Every LDS read of every round is issued before the
        // first result is needed (one wait per family instead of one per round), then computed, then stored.
        constexpr int RA = (4 * (NMAXS + 1) + 63) / 64, RB = (6 * (NMAXS + 1) + 63) / 64, RC = (4 * NMAXS + 63) / 64;
        // (a) quaternion block, one lane per (stage, row a)
        vd qr[RA][4], qi[RA][4];
        vd xb[RB], rb[RB], wb[RB], uc[RC], rc[RC], wc[RC];
        for (int t = 0; t < RA; ++t) {
            vi task = W::imin(lane + 64 * t, 4 * (N + 1) - 1);
            vi k = task >> 2;
            for (int i = 0; i < 4; ++i) {
                qr[t][i] = W::ld(lds, k * NX + (m.TXR + 6 + i));
                qi[t][i] = W::ld(lds, k * NX + (m.XI + 6 + i));
            }
        }
        // (b) position / velocity rows, one lane per (stage, i < 6)
        for (int t = 0; t < RB; ++t) {
            vi task = W::imin(lane + 64 * t, 6 * (N + 1) - 1);
            vi k = W::div6(task);
            vi i = task - k * 6;
            wb[t] = W::ld(lds, i + (m.KC + KC_QD));
            xb[t] = W::ld(lds, k * NX + i + m.XI);
            rb[t] = W::ld(lds, k * NX + i + m.TXR);
        }
        // (c) control rows, one lane per (stage < N, i < 4)
        for (int t = 0; t < RC; ++t) {
            vi ti = W::imin(lane + 64 * t, 4 * N - 1);
            wc[t] = W::ld(lds, (ti & 3) + (m.KC + KC_RD));
            uc[t] = W::ld(lds, ti + m.UI);
            rc[t] = W::ld(lds, ti + m.TUR);
        }
        for (int t = 0; t < RA; ++t) {
            vi task = W::imin(lane + 64 * t, 4 * (N + 1) - 1);
            vi k = task >> 2;
            vi a = task & 3;
            vd s = W::sel(k < N, vd(P.dt), vd(1.0));
            const vd qwr = qr[t][0], qxr = qr[t][1], qyr = qr[t][2], qzr = qr[t][3];
            // E = [[-qx, qw,-qz, qy], [-qy, qz, qw,-qx], [-qz,-qy, qx, qw]] (of q_r)
            vd E0[4] = {-qxr, qwr, -qzr, qyr};
            vd E1[4] = {-qyr, qzr, qwr, -qxr};
            vd E2[4] = {-qzr, -qyr, qxr, qwr};
            vd ea0 = W::sel(a == 0, E0[0], W::sel(a == 1, E0[1], W::sel(a == 2, E0[2], E0[3])));
            vd ea1 = W::sel(a == 0, E1[0], W::sel(a == 1, E1[1], W::sel(a == 2, E1[2], E1[3])));
            vd ea2 = W::sel(a == 0, E2[0], W::sel(a == 1, E2[1], W::sel(a == 2, E2[2], E2[3])));
            vd w0 = ea0 * (s * P.Qd[7]), w1 = ea1 * (s * P.Qd[8]), w2 = ea2 * (s * P.Qd[9]);
            vd grad = 0.0;
            vi cb = k * int(CB_STRIDE) + m.CB;
            for (int b = 0; b < 4; ++b) {
                vd h = w0 * E0[b] + w1 * E1[b] + w2 * E2[b];
                W::st(lds, cb + a * 4 + (int(CB_QQ) + b), h);
                grad = grad + h * qi[t][b];
            }
            W::st(lds, cb + a + (int(CB_QE) + 6), grad);
            W::st(lds, cb + int(CB_ZERO), vd(0.0));                     // the block's structural zero (all four lanes of the stage)
        }
        for (int t = 0; t < RB; ++t) {
            vi task = W::imin(lane + 64 * t, 6 * (N + 1) - 1);
            vi k = W::div6(task);
            vi i = task - k * 6;
            vd s = W::sel(k < N, vd(P.dt), vd(1.0));
            vd de = s * wb[t];
            vd grad = de * (xb[t] - rb[t]);
            vi cb = k * int(CB_STRIDE) + m.CB;
            W::st(lds, cb + i + int(CB_DEX), de);
            W::st(lds, cb + i + int(CB_QE), grad);
            W::stp(lds, cb + i + (int(CB_QBV) - 3), grad, i >= 3);
        }
        for (int t = 0; t < RC; ++t) {
            vi task = W::imin(lane + 64 * t, 4 * N - 1);
            vi k = task >> 2;
            vi i = task & 3;
            vd de = P.dt * wc[t];
            vd grad = de * (uc[t] - rc[t]);
            vi cb = k * int(CB_STRIDE) + m.CB;
            W::st(lds, cb + i + int(CB_DEU), de);
            W::st(lds, cb + i + int(CB_RE), grad);
            W::st(lds, cb + i + int(CB_RB), grad);
        }
    }

    // ---------------------------------------------------------------- linearisation
    // q-dot = 1/2 Omega(w) q   (nmpc_body_rate_ctl.py:154-157)
    static NDP_D void qdot(const vd q[4], const vd w[3], vd o[4])
    {
        o[0] = (-w[0] * q[1] - w[1] * q[2] - w[2] * q[3]) * 0.5;
        o[1] = (w[0] * q[0] + w[2] * q[2] - w[1] * q[3]) * 0.5;
        o[2] = (w[1] * q[0] - w[2] * q[1] + w[0] * q[3]) * 0.5;
        o[3] = (w[2] * q[0] + w[1] * q[1] - w[0] * q[2]) * 0.5;
    }
    // thrust direction R(q) e3   (nmpc_body_rate_ctl.py:151-153)
    static NDP_D void thrust_dir(const vd q[4], vd o[3])
    {
        o[0] = (q[1] * q[3] + q[0] * q[2]) * 2.0;
        o[1] = (q[2] * q[3] - q[0] * q[1]) * 2.0;
        o[2] = 1.0 - (q[1] * q[1] + q[2] * q[2]) * 2.0;
    }
    // directional derivative of R(q) e3 along s
    static NDP_D void thrust_dir_tan(const vd q[4], const vd s[4], vd o[3])
    {
        o[0] = (s[1] * q[3] + q[1] * s[3] + s[0] * q[2] + q[0] * s[2]) * 2.0;
        o[1] = (s[2] * q[3] + q[2] * s[3] - s[0] * q[1] - q[0] * s[1]) * 2.0;
        o[2] = (q[1] * s[1] + q[2] * s[2]) * -4.0;
    }

    // Scalars of the closed-form ERK4 step.  The attitude ODE q' = Z q, Z = 1/2 Omega(w), is linear with Z^2 = -sigma I
    // (sigma = |w|^2/4), so every RK node is q_i = a_i q + b_i r with r = Z q:
    //   (a,b) = (1,0), (1,h/2), (1-h^2 sigma/4, h/2), (1-h^2 sigma/2, h(1-h^2 sigma/4)),  q+ = A q + B r.
    // The thrust direction is td(x) = 1/2 tdt(x,x) + e3 with tdt the symmetric bilinear form thrust_dir_tan, hence
    //   v+ = v + c Dv + h acc,  p+ = p + h v + c Dp + h^2/2 acc,   D = sum_i beta_i td(q_i),
    // beta_v = h/6 (1,2,2,1), beta_p = h^2/6 (1,1,1,0), and all sensitivities are tdt(q,U) + tdt(r,V) with U, V linear
    // combinations of a few basis vectors weighted by the sums S_xy = sum_i beta_i x_i y_i below.  This is the exact
    // derivative of the same RK4 map the oracle differentiates numerically stage by stage (acados sim_erk defaults).
    struct RkScal {
        vd a3, a4, b4, A, B;
        vd Sv_aa, Sv_ab, Sv_bb, Sp_aa, Sp_ab, Sp_bb;
    };
    static NDP_D void rk_scalars(const RtiParams &P, vd sg, RkScal &K)
    {
        const double h = P.dt, h2 = h * h, hh = 0.5 * h, h6 = P.h_6, hp = P.h2_6;
        K.a3 = 1.0 - sg * (0.25 * h2);
        K.a4 = 1.0 - sg * (0.5 * h2);
        K.b4 = K.a3 * h;
        K.A = 1.0 - sg * (0.5 * h2) + sg * sg * P.h4_24;
        K.B = h - sg * P.h3_6;
        K.Sv_aa = (3.0 + K.a3 * K.a3 * 2.0 + K.a4 * K.a4) * h6;
        K.Sv_ab = (2.0 * hh + K.a3 * (2.0 * hh) + K.a4 * K.b4) * h6;
        K.Sv_bb = (4.0 * hh * hh + K.b4 * K.b4) * h6;
        K.Sp_aa = (2.0 + K.a3 * K.a3) * hp;
        K.Sp_ab = (hh + K.a3 * hh) * hp;
        K.Sp_bb = (2.0 * hh * hh) * hp;
    }

    // ERK4 (one step of dt) + forward sensitivities -> stage blocks MB_k, in closed form (see rk_scalars)
    static NDP_D void linearize(const RtiParams &P, const LdsMap &m, lp lds)
    {
        const int N = horizon(P);
        const double h = P.dt, h2 = h * h, hh = 0.5 * h, h6 = P.h_6, hp = P.h2_6;
        vi lane = lane_it();
        // ---- d/dq columns: one lane per (stage, j), 4N tasks
        for (int t = 0; t < 4 * N; t += 64) {
            vi task = W::imin(lane + t, 4 * N - 1);       // lanes past the last task repeat it: identical duplicate stores, no predicates
            vi k = task >> 2;
            vi j = task & 3;
            vi xi = k * NX + m.XI + 6, ui = k * NU + m.UI;
            vd q[4] = {W::ld(lds, xi), W::ld(lds, xi + 1), W::ld(lds, xi + 2), W::ld(lds, xi + 3)};
            vd w[3] = {W::ld(lds, ui), W::ld(lds, ui + 1), W::ld(lds, ui + 2)};
            vd c = W::ld(lds, ui + 3);
            vd r[4], e[4], z[4];
            qdot(q, w, r);
            RkScal K;
            rk_scalars(P, (w[0] * w[0] + w[1] * w[1] + w[2] * w[2]) * 0.25, K);
            for (int i = 0; i < 4; ++i) e[i] = W::sel(j == i, vd(1.0), vd(0.0));
            qdot(e, w, z);                                   // Z e_j
            vd Uv[4], Vv[4], Up[4], Vp[4], t1[3], t2[3];
            for (int i = 0; i < 4; ++i) {
                Uv[i] = e[i] * K.Sv_aa + z[i] * K.Sv_ab; Vv[i] = e[i] * K.Sv_ab + z[i] * K.Sv_bb;
                Up[i] = e[i] * K.Sp_aa + z[i] * K.Sp_ab; Vp[i] = e[i] * K.Sp_ab + z[i] * K.Sp_bb;
            }
            vi mb = k * int(MB_STRIDE) + m.MB + j;
            thrust_dir_tan(q, Uv, t1); thrust_dir_tan(r, Vv, t2);
            for (int i = 0; i < 3; ++i) W::st(lds, mb + (int(MB_PV) + (3 + i) * 8), (t1[i] + t2[i]) * c);
            thrust_dir_tan(q, Up, t1); thrust_dir_tan(r, Vp, t2);
            for (int i = 0; i < 3; ++i) W::st(lds, mb + (int(MB_PV) + i * 8), (t1[i] + t2[i]) * c);
            for (int i = 0; i < 4; ++i) W::st(lds, mb + (int(MB_Q) + i * 7), e[i] * K.A + z[i] * K.B);
        }
        // ---- d/dw columns: one lane per (stage, mth rate), 3N tasks.  d q_i = eps (a_i' q + b_i' r) + b_i g,
        //      eps = w_m/2 = d sigma/d w_m, g = Z_m q; only a_3' = -h^2/4, a_4' = -h^2/2, b_4' = -h^3/4 are non-zero
        for (int t = 0; t < 3 * N; t += 64) {
            vi task = W::imin(lane + t, 3 * N - 1);
            vi k = W::div3(task);
            vi mm = task - k * 3;
            vi xi = k * NX + m.XI + 6, ui = k * NU + m.UI;
            vd q[4] = {W::ld(lds, xi), W::ld(lds, xi + 1), W::ld(lds, xi + 2), W::ld(lds, xi + 3)};
            vd w[3] = {W::ld(lds, ui), W::ld(lds, ui + 1), W::ld(lds, ui + 2)};
            vd c = W::ld(lds, ui + 3);
            vd r[4], em[3], g[4];
            qdot(q, w, r);
            RkScal K;
            rk_scalars(P, (w[0] * w[0] + w[1] * w[1] + w[2] * w[2]) * 0.25, K);
            for (int i = 0; i < 3; ++i) em[i] = W::sel(mm == i, vd(1.0), vd(0.0));
            qdot(q, em, g);                                  // Z_m q
            vd eps = (em[0] * w[0] + em[1] * w[1] + em[2] * w[2]) * 0.5;
            const double da3 = -0.25 * h2, da4 = -0.5 * h2, db4 = -0.25 * h * h2;
            // sums with one primed factor; beta_v = h/6 (1,2,2,1), beta_p = h^2/6 (1,1,1,0); b_2 = b_3 = h/2
            vd Sv_aad = (K.a3 * (2.0 * da3) + K.a4 * da4) * h6, Sv_abd = (K.a4 * db4) * h6;
            vd Sv_bad = (K.b4 * da4 + 2.0 * hh * da3) * h6, Sv_bbd = (K.b4 * db4) * h6;
            vd Sp_aad = K.a3 * (da3 * hp);
            const double Sp_bad = hh * da3 * hp;            // Sp_abd = Sp_bbd = 0 (node 4 carries no position weight)
            vd dA = sg_lin(P, K), dB = vd(-P.h3_6);
            vd Uv[4], Vv[4], Up[4], Vp[4], t1[3], t2[3];
            for (int i = 0; i < 4; ++i) {
                Uv[i] = (q[i] * Sv_aad + r[i] * Sv_abd) * eps + g[i] * K.Sv_ab;
                Vv[i] = (q[i] * Sv_bad + r[i] * Sv_bbd) * eps + g[i] * K.Sv_bb;
                Up[i] = (q[i] * Sp_aad) * eps + g[i] * K.Sp_ab;
                Vp[i] = (q[i] * Sp_bad) * eps + g[i] * K.Sp_bb;
            }
            vi mb = k * int(MB_STRIDE) + m.MB + 4 + mm;
            thrust_dir_tan(q, Uv, t1); thrust_dir_tan(r, Vv, t2);
            for (int i = 0; i < 3; ++i) W::st(lds, mb + (int(MB_PV) + (3 + i) * 8), (t1[i] + t2[i]) * c);
            thrust_dir_tan(q, Up, t1); thrust_dir_tan(r, Vp, t2);
            for (int i = 0; i < 3; ++i) W::st(lds, mb + (int(MB_PV) + i * 8), (t1[i] + t2[i]) * c);
            for (int i = 0; i < 4; ++i) W::st(lds, mb + (int(MB_Q) + i * 7), (q[i] * dA + r[i] * dB) * eps + g[i] * K.B);
        }
        // ---- d/dc column + nominal step and dynamics defect b_k = phi(x_k,u_k) - x_{k+1}; one lane per stage
        for (int t = 0; t < N; t += 64) {
            vi k = W::imin(lane + t, N - 1);
            {   // the block's structural constants 0, 1, h (entries of M~ that are the same at every stage)
                vi mc = k * int(MB_STRIDE) + m.MB;
                W::st(lds, mc + int(MB_ZERO), vd(0.0));
                W::st(lds, mc + int(MB_ONE), vd(1.0));
                W::st(lds, mc + int(MB_H), vd(P.dt));
            }
            vi xi = k * NX + m.XI, ui = k * NU + m.UI, fi = k * 3 + m.TF;
            vd x[10];
            for (int i = 0; i < 10; ++i) x[i] = W::ld(lds, xi + i);
            vd w[3] = {W::ld(lds, ui), W::ld(lds, ui + 1), W::ld(lds, ui + 2)};
            vd c = W::ld(lds, ui + 3);
            // disturbance acceleration f/m (ndp_nmpc_body_rate_ctl.py:155-157) and gravity
            vd acc[3] = {W::ld(lds, fi) * P.inv_mass, W::ld(lds, fi + 1) * P.inv_mass, W::ld(lds, fi + 2) * P.inv_mass - P.g};
            vd xn1[10];
            for (int i = 0; i < 10; ++i) xn1[i] = W::ld(lds, xi + (NX + i));
            vd q[4] = {x[6], x[7], x[8], x[9]}, r[4];
            qdot(q, w, r);
            RkScal K;
            rk_scalars(P, (w[0] * w[0] + w[1] * w[1] + w[2] * w[2]) * 0.25, K);
            vd tqq[3], tqr[3], trr[3];
            thrust_dir_tan(q, q, tqq); thrust_dir_tan(q, r, tqr); thrust_dir_tan(r, r, trr);
            vi mb = k * int(MB_STRIDE) + m.MB;
            for (int i = 0; i < 3; ++i) {
                const double e3 = i == 2 ? 1.0 : 0.0;
                vd Dv = (tqq[i] * K.Sv_aa + tqr[i] * (K.Sv_ab * 2.0) + trr[i] * K.Sv_bb) * 0.5 + e3 * h;
                vd Dp = (tqq[i] * K.Sp_aa + tqr[i] * (K.Sp_ab * 2.0) + trr[i] * K.Sp_bb) * 0.5 + e3 * (0.5 * h2);
                W::st(lds, mb + (int(MB_PV) + (3 + i) * 8 + 7), Dv);      // column 7 = d/dc
                W::st(lds, mb + (int(MB_PV) + i * 8 + 7), Dp);
                vd vn = x[3 + i] + Dv * c + acc[i] * h;
                vd pn = x[i] + x[3 + i] * h + Dp * c + acc[i] * (0.5 * h2);
                W::st(lds, mb + (int(MB_B) + 3 + i), vn - xn1[3 + i]);
                W::st(lds, mb + (int(MB_B) + i), pn - xn1[i]);
            }
            for (int i = 0; i < 4; ++i) W::st(lds, mb + (int(MB_B) + 6 + i), q[i] * K.A + r[i] * K.B - xn1[6 + i]);
        }
    }

    // dA/dsigma of q+ = A q + B r
    static NDP_D vd sg_lin(const RtiParams &P, const RkScal &K)
    {
        // A = 1 - h^2 sigma/2 + h^4 sigma^2/24 and a4 = 1 - h^2 sigma/2  =>  sigma = (1 - a4) 2/h^2
        vd sg = (1.0 - K.a4) * P.two_over_h2;
        return sg * P.h4_12 - 0.5 * (P.dt * P.dt);
    }

    // ---------------------------------------------------------------- Riccati sweep (MFMA)
    // Lam^-1 for the SPD 4x4 block Lam = H~uu, lane-parallel: lane (g, j<4) computes its own entry
    // adj(Lam)[g][j] / det(Lam) from a 16-double LDS copy of Lam.  No sequential pivots: on gfx950 a
    // dependent f64 VALU op costs ~32 cycles and a divide ~100, so a 4-pivot factorisation computed
    // redundantly on every lane was ~40 % of a backward stage.  Lam = R + B'PB + barrier diagonal: R > 0 and
    // the barrier terms only add to the diagonal, so the cofactor expansion is well conditioned here.
    // The inverse in three phases so that the caller can interleave them with matrix instructions:
    //   lam_gather   : publish Lam (register 3 of H~) to the LDS scratch and read this lane's 3x3 minor + own entry
    //   lam_cofactor : signed cofactor adj(Lam)[g][j&3], in EVERY lane (callers mask the columns they need)
    //   lam_rdet     : 1/det by row expansion (DPP quad-sum) + PD flag
    struct LamRegs { vd mm[9], own; };
    static NDP_D void lam_gather(const Tables &T, lp lds, vd h3, LamRegs &L)
    {
        W::st(lds, T.lam_w_off, h3);                         // H~[12+g][12+b] -> SC[g*4+b]; lanes j<12 -> dump slot
        W::sync();
        for (int a = 0; a < 3; ++a) {
            L.mm[3 * a] = W::ld(lds, T.minor_one[a]);
            W::ld2(lds, T.minor_pair[a], L.mm[3 * a + 1], L.mm[3 * a + 2]);
        }
        L.own = W::ld(lds, T.own_off);
    }
    static NDP_D vd lam_cofactor(const Tables &T, const LamRegs &L)
    {
        const vd *mm = L.mm;
        vd d0 = mm[4] * mm[8] - mm[5] * mm[7];
        vd d1 = mm[3] * mm[8] - mm[5] * mm[6];
        vd d2 = mm[3] * mm[7] - mm[4] * mm[6];
        return (mm[0] * d0 - mm[1] * d1 + mm[2] * d2) * T.cof_sign;
    }
    static NDP_D vd lam_rdet(const Tables &T, const LamRegs &L, vd cof, bool &ok)
    {
        vd det = L.own * cof;                                 // row expansion over the four lanes holding one row of Lam
        det = det + W::csum1(det);
        det = det + W::csum2(det);
        vb pd = (det > 0.0) && (!T.lam_diag || (cof > 0.0));
        ok = W::all(pd) && ok;
        return W::rcp(det);
    }

    // Lam^-1 for a STIFF Lam (ROBUST sweeps).  A barrier term Gamma on a STATE bound enters Lam = R + B'PB as Gamma b b' with b = dv+/du:
    // a rank-one part with large off-diagonal entries.  The cofactor expansion above cancels (Gamma |b|^2)^3 down to Gamma |b|^2 --
    // garbage from Gamma |b|^2 ~ 1e6 on (found with the shrunk-velocity-box problems of tests/test_wave_program_emulated.py: the sweep
    // reported a failed factorisation where the oracle's Cholesky had none).  Here every lane factorises the 4x4 matrix itself,
    // Lam = L D L' without square roots, and forms the entry (g, j & 3) of L^-T D^-1 L^-1 it holds: ~100 dependent f64 operations
    // per stage, run only in the interior-point iterations in which a state bound's barrier term exceeds P.refine_gamma.
    // The factors are the same in every lane (each lane factorises the whole 4x4 block): they are parked in the block's own LDS image
    // (strict lower triangle: L, diagonal: 1 / D) and read back where a substitution uses them -- twenty registers that would otherwise
    // be live across a whole stage of a ROBUST sweep, and through the allocator's live-range splitting cost the (unrelated) hot sweep
    // sixteen more accumulation-register reads (1.2 % of the headline step, measured A/B).
    // X = Lam^-1 B for a 4 x 16 right-hand side in register-3 form (lane (g, j) holds B[g][j]), by forward / back substitution with the
    // factors -- a SOLVE, not a multiplication with the explicit inverse: with cond(Lam) ~ 1e8 the explicit inverse carries
    // cond * eps in its small entries, and Hxx - Hxu (Lam^-1 Hux) formed with it loses the recursion's definiteness within a few stages
    // (numpy, both forms side by side: the explicit inverse fails on exactly the problems the device failed on, the solve on none --
    // whichever order the products are taken in).  Each lane gathers its column through a 64-double LDS scratch (the shadow ZD, idle
    // during a backward sweep), solves it redundantly and keeps row g.
    static NDP_D vd ldl_solve(const LdsMap &m, lp lds, vd rhs)
    {
        vi lane = W::lane_here();
        vi g = lane >> 4, j = W::lcol(lane);
        auto F = [&](int r, int c) { return W::ld(lds, lane * 0 + (m.SC + 4 * r + c)); };
        W::sync();
        W::st(lds, g * 16 + j + m.ZD, rhs);
        W::sync();
        const vd b0 = W::ld(lds, j + m.ZD), b1 = W::ld(lds, j + (m.ZD + 16)), b2 = W::ld(lds, j + (m.ZD + 32)), b3 = W::ld(lds, j + (m.ZD + 48));
        const vd l10 = F(1, 0), l20 = F(2, 0), l21 = F(2, 1), l30 = F(3, 0), l31 = F(3, 1), l32 = F(3, 2);
        const vd y0 = b0, y1 = b1 - l10 * y0, y2 = b2 - l20 * y0 - l21 * y1, y3 = b3 - l30 * y0 - l31 * y1 - l32 * y2;
        const vd x3 = y3 * F(3, 3);
        const vd x2 = y2 * F(2, 2) - l32 * x3;
        const vd x1 = y1 * F(1, 1) - l21 * x2 - l31 * x3;
        const vd x0 = y0 * F(0, 0) - l10 * x1 - l20 * x2 - l30 * x3;
        return W::sel(g == 0, x0, W::sel(g == 1, x1, W::sel(g == 2, x2, x3)));
    }
    static NDP_D void lam_factor_ldl(const LdsMap &m, const Tables &T, lp lds, vd h3, vb &okv)
    {
        W::st(lds, T.lam_w_off, h3);
        W::sync();
        // entries are read where they are used (broadcast reads of the block's LDS image), the factors written back over them: nothing
        // but the running column lives in registers
        auto A = [&](int r, int c) { return W::ld(lds, W::lane_here() * 0 + (m.SC + 4 * r + c)); };
        auto put = [&](int r, int c, vd v) { W::st(lds, W::lane_here() * 0 + (m.SC + 4 * r + c), v); };   // every lane writes the same value
        const vd d0 = A(0, 0), r0 = W::rcp(d0);
        const vd a10 = A(1, 0), a20 = A(2, 0), a30 = A(3, 0);
        const vd l10 = a10 * r0, l20 = a20 * r0, l30 = a30 * r0;
        const vd d1 = A(1, 1) - l10 * a10, r1 = W::rcp(d1);
        const vd l21 = (A(2, 1) - l20 * a10) * r1, l31 = (A(3, 1) - l30 * a10) * r1;
        const vd l21d = l21 * d1;
        const vd d2 = A(2, 2) - l20 * a20 - l21 * l21d, r2 = W::rcp(d2);
        const vd l32 = (A(3, 2) - l30 * a20 - l31 * l21d) * r2;
        const vd d3 = A(3, 3) - l30 * a30 - l31 * (l31 * d1) - l32 * (l32 * d2), r3 = W::rcp(d3);
        okv = okv && (d0 > 0.0) && (d1 > 0.0) && (d2 > 0.0) && (d3 > 0.0);
        W::sync();
        put(0, 0, r0); put(1, 1, r1); put(2, 2, r2); put(3, 3, r3);
        put(1, 0, l10); put(2, 0, l20); put(2, 1, l21); put(3, 0, l30); put(3, 1, l31); put(3, 2, l32);
        W::sync();
    }
    // Lam^-1[g][j & 3] in every lane -- the layout of the explicit inverse the second-solve sweeps keep (delta_sweep's -Lam^-1) -- as the
    // solution of Lam X = [I I I I] with the parked factors
    static NDP_D vd lam_inverse_ldl(const LdsMap &m, lp lds)
    {
        vi lane = W::lane_here();
        return ldl_solve(m, lds, W::sel((W::lcol(lane) & 3) == (lane >> 4), vd(1.0), vd(0.0)));
    }

    // backward: H~_k = M~_k' P~_{k+1} M~_k + C~_k with P~_{k+1} = H~xx - H~xu Lam^-1 H~ux of stage k+1, P~_N = C~_N.
    // P~ is never formed: the next stage needs only W' = P~ M~' = H~xx M~' - H~xu (Lam^-1 (H~ux M~')), and
    // [H~xx ; H~ux] M~' is ONE product (the accumulator registers of H~ as A operand), which does not depend on
    // Lam^-1 -- only two dependent MFMAs follow the inverse.  Stores K~' per stage.
    // forward: z~_0 = [dx0,1,0]; du = K~ z~; z~+ = M~ [z~; du].  Writes ZX[1..N], ZU[0..N-1].
    // The LDS operands of the next stage are requested one stage ahead, so the LDS latency is paid while the MFMA chain runs (LDS, unlike the wave's own VALU work, does proceed under it).
    // linv (compile-time horizons only, else null): per stage, -Lam^-1 in the lanes j >= 12 (the B operand of the K~' product),
    // kept in registers for delta_sweep.
    // KEEP (compile time, not "linv != null": the address of a private array compared with null is a RUN-time test after the
    // address-space cast, a branch around every store, and it keeps the array in scratch memory): store -Lam^-1 of every stage.
    // ROBUST (compile time): the 4x4 inverse by lam_inverse_ldl instead of the cofactor expansion -- the interior-point loop's sweeps
    // while a state bound's barrier term is large.
    template <bool KEEP = false, bool ROBUST = false>
    static NDP_D bool riccati_sweep(const RtiParams &P, const LdsMap &m, const Tables &T, lp lds,
                                    const RtiIo *io = nullptr, md *linv = nullptr)
    {
        const int N = horizon(P);
        // the stage loops of a ROBUST sweep stay loops (a cold path: unrolled, its ten factor registers per stage and two LDS-gathered
        // solves pushed every N = 20 kernel -- the headline included -- over the register file: 372 B of scratch per lane)
        constexpr int UNROLL_SWEEP = ROBUST ? 1 : UNROLL_STAGES;
        (void)UNROLL_SWEEP;
        bool ok = true;
        vi lane = W::lane();
        vi g = lane >> 4, j = W::lcol(lane);
        vb okv = lane >= 0;        // per-lane positive-definiteness flags, reduced once after the sweep
        md4 H;
        {   // stage N-1 from the terminal block (no control part: keep columns 12..15 exactly zero)
            md Pt[3];
            for (int c = 0; c < 3; ++c) Pt[c] = W::to_m(W::sel(j < 12, W::ld(lds, T.c_off[c] + cb(N)), vd(0.0)));
            md mk[3];
            for (int c = 0; c < 3; ++c) mk[c] = W::to_m(W::ld(lds, T.mk_off[c] + mb(N - 1)));
            for (int r = 0; r < 4; ++r) H.r[r] = W::to_m(W::ld(lds, T.c_off[r] + cb(N - 1)));
            md4 Wm = mman<3>(Pt, mk, W::mzero4());
            H = mman<3>(mk, Wm.r, H);
        }
        md mk[3], cc[4];       // operands of the NEXT stage to be formed (k-1), requested one iteration ahead
        {
            const int kn = N > 1 ? N - 2 : 0;
            for (int c = 0; c < 3; ++c) mk[c] = W::to_m(W::ld(lds, T.mk_off[c] + mb(kn)));
            for (int r = 0; r < 4; ++r) cc[r] = W::to_m(W::ld(lds, T.c_off[r] + cb(kn)));
        }
        md4 Ktp = W::mzero4();
        md Ktq = W::to_m(vd(0.0));     // MMA4: K~' of the previous stage in ONE register
        int kprev = -1;
        NDP_UNROLL_SWEEP
        for (int k = N - 1; k >= 1; --k) {
            md nmk[3], ncc[4];
            const int kp = k >= 2 ? k - 2 : 0;      // prefetch stage k-2; at k = 1 there is none: re-read stage 0 (values unused)
            // Lam[a][b] = H~[12+a][12+b] and H~ux both sit in accumulator register 3
            md hux = H.r[3];
            // [H~xx ; H~ux] M~_{k-1}: H~'s registers as A operand mean H~' -- equal up to rounding (see the
            // re-symmetrisation below); rows 12..15 of the result are T = H~ux M~_{k-1}
            //   M~'' H~xu = (H~ux M~')' = T', hence  H~' = [C~' + M~'' (H~xx M~')] - T' Lam^-1 T:
            // the bracket needs no Lam^-1, only two dependent MFMAs (adj T, then the rank-4 correction) follow it.
            // Issue order: a wave issues nothing under its own MFMA, and a DEPENDENT f64 VALU op waits ~32 cycles for its operand.  The
            // inverse's dependency chain (cofactor 5 levels, determinant 3, 1/det 2, scale 1) is written in pieces between the MFMAs that
            // do not need it; what counts is the number of instructions and of chain levels (dropping the second Newton step of 1/det
            // saved 5 %).  The ORDER is the compiler's: rounds 2-3 pinned it with scheduling fences between the pieces (worth ~1 % then);
            // once the hot path had its registers to itself (round 4: cold branches marked) the scheduler's own order beat the pinned one
            // by 3 % (20.65 against 21.30 us per step, interior point always 59.8 against 63.6) and the fences went.
            if constexpr (ROBUST) {
                // a cold path, written for few live registers (no operand prefetch, nothing carried between stages but H~): what is
                // live here at its widest is what the allocator parks for the whole kernel, the hot sweep included
                lam_factor_ldl(m, T, lds, W::to_d(hux), okv);                        // the factors parked in LDS
                if constexpr (KEEP) {
                    const vd inv = lam_inverse_ldl(m, lds);
                    linv_put(linv, k, N, W::to_m(MMA4 ? -inv : W::sel(T.kt_pred, -inv, vd(0.0))));
                }
                {   // K~' = -(Lam^-1 H~ux)': lane (g, j) holds (Lam^-1 H~ux)[g][j] = -K~'[j][g], stored where K~' lives (row-major 12 x 4)
                    const vd G1 = ldl_solve(m, lds, W::to_d(hux));
                    vi ln = W::lane_here();
                    vi gg = ln >> 4, jj = W::lcol(ln);
                    W::st(lds, W::sel(jj < 12, jj * 4 + gg + m.KT, vi(m.MB + int(MB_DUMP))) + mb(k), -G1);
                }
                md rmk[3];
                for (int c = 0; c < 3; ++c) rmk[c] = W::to_m(W::ld(lds, T.mk_off[c] + mb(k - 1)));
                md4 Wf = mman<3>(H.r, rmk, W::mzero4());
                md4 Hb;
                for (int r = 0; r < 4; ++r) Hb.r[r] = W::to_m(W::ld(lds, T.c_off[r] + cb(k - 1)));
                Hb = mman<3>(rmk, Wf.r, Hb);
                md tt = Wf.r[3];
                const md G0 = W::to_m(ldl_solve(m, lds, W::to_d(tt)));            // Lam^-1 T by substitution (see ldl_solve)
                md4 Hn = mma(-tt, G0, Hb);
                {   // re-symmetrised at EVERY stage here (the oracle does): with entries of 1e10 the antisymmetric rounding part, which
                    // the open-loop map doubles per stage, reaches the size of the O(1) eigenvalues within a few stages
                    md ey[4];
                    vi ln = W::lane_here();
                    vi gg = ln >> 4, jj = W::lcol(ln);
                    for (int c = 0; c < 4; ++c) ey[c] = W::to_m(W::sel(jj == gg + 4 * c, vd(1.0), vd(0.0)));
                    md4 Tp = mman<4>(Hn.r, ey, W::mzero4());
                    for (int r = 0; r < 4; ++r) Hn.r[r] = W::mavg(Hn.r[r], Tp.r[r]);
                }
                H = Hn;
                continue;
            }
            // -DNDP_FINE_STAMPS (B = 1, debug path): the timeline of ONE stage (k = N / 2) -- the clock when each result is available
            // (every stamp waits for its operand: the stamped stage runs a little slower than the others).  scripts/sweep_stage_timeline.py
            NDP_FINE(const bool fst = io && io->dbg && k == N / 2; if (fst) fstamp(*io, m, 0, W::to_d(H.r[0]));)
            LamRegs LR;
            lam_gather(T, lds, W::to_d(hux), LR);
            NDP_FINE(if (fst) fstamp(*io, m, 1, LR.own);)                          // Lam's entries gathered
            md4 Wf = mman<3>(H.r, mk, W::mzero4());
            NDP_FINE(if (fst) fstamp(*io, m, 2, W::to_d(Wf.r[3]));)               // W = [H~xx ; H~ux] M~' (3 matrix instructions)
            if (kprev >= 0) {
                if constexpr (MMA4) W::st(lds, T.kt_st4 + mb(kprev), W::to_d(Ktq));
                else for (int c = 0; c < 3; ++c) W::st(lds, T.kt_st[c] + mb(kprev), W::to_d(Ktp.r[c]));
            }
            const vd *mm = LR.mm;
            vd p0 = mm[4] * mm[8], p1 = mm[3] * mm[8], p2 = mm[3] * mm[7];
            vd os = LR.own * T.cof_sign;
            md4 Hb;
            for (int r = 0; r < 4; ++r) Hb.r[r] = cc[r];
            if constexpr (!W::packed_k) Hb = mma(mk[0], Wf.r[0], Hb);
            vd d0 = p0 - mm[5] * mm[7], d1 = p1 - mm[5] * mm[6], d2 = p2 - mm[4] * mm[6];
            if constexpr (!W::packed_k) Hb = mma(mk[1], Wf.r[1], Hb);
            vd cx = mm[0] * d0, cy = mm[1] * d1;
            if constexpr (W::packed_k) Hb = mman<3>(mk, Wf.r, Hb);
            else Hb = mma(mk[2], Wf.r[2], Hb);
            vd cz = cx + mm[2] * d2;
            vd oy = os * cy;
            // unsigned cofactor cz - cy (the 3x3 minor's determinant) and, one level earlier than through it, this lane's
            // term of the row expansion of det: own * sign * (cz - cy)
            vd dq = os * cz - oy;
            vd cofu = cz - cy;
            dq = dq + W::csum1(dq);                           // the four lanes holding one row of Lam (a quad in the f64 layout)
            vd ladj = cofu * T.adj_a;                         // A operand: adj(Lam)[g][j], j < 4 (sign and lane mask in one factor)
            vd nahi;                                          // B operand of K~': -adj[g][j-12] in columns 12..15
            if constexpr (MMA4) nahi = -ladj;                 // (four-block form: every block reads its own copy, adj_b = -adj_a; the sign rides on the consumer)
            else nahi = cofu * T.adj_b;
            md tt = Wf.r[3];                                  // T = H~ux M~' (rows 12..15 of Wf)
            md G0;                                            // adj T, lane l: row l >> 4, column l & 15
            if constexpr (MMA4) G0 = mma4(W::to_m(ladj), tt, W::to_m(vd(0.0)));
            else G0 = mma(W::to_m(ladj), tt, W::mzero4()).r[0];
            NDP_FINE(if (fst) fstamp(*io, m, 3, W::to_d(Hb.r[0]));)               // bracket C~' + M~'' (H~xx M~') (3 matrix instructions)
            NDP_FINE(if (fst) fstamp(*io, m, 4, W::to_d(G0));)                    // adj(Lam) T (one four-block instruction) -- after the cofactors
            vd det = dq + W::csum2(dq);
            vd r0 = W::rcp_seed(det);
            for (int c = 0; c < 3; ++c) nmk[c] = W::to_m(W::ld(lds, T.mk_off[c] + mb(kp)));
            // 1/det = r0 (2 - det r0): v_rcp_f64 seed (4.5e-8) + ONE Newton step = 2.2e-15 (profiles/r01_ubench_mfma_latency.txt),
            // below the cofactors' own cond * eps.  The scale of Lam^-1 T is applied as (G r0) e0 so that G r0 runs beside e0.
            vd e0 = W::fma(-det, r0, vd(2.0));
            vd g0 = W::to_d(G0) * r0;
            for (int r = 0; r < 2; ++r) ncc[r] = W::to_m(W::ld(lds, T.c_off[r] + cb(kp)));
            vd gs = g0 * e0;                                  // Lam^-1 T
            vd rdet = r0 * e0;
            for (int r = 2; r < 4; ++r) ncc[r] = W::to_m(W::ld(lds, T.c_off[r] + cb(kp)));
            okv = okv && (det > 0.0) && (!T.lam_diag || (cofu > 0.0));   // on the diagonal the cofactor's sign factor is +1
            NDP_FINE(if (fst) fstamp(*io, m, 5, gs);)                             // Lam^-1 T scaled (1 / det ready)
            md4 Hn = mma(-tt, W::to_m(gs), Hb);               // - T' Lam^-1 T on top of the bracket
            // K~' = H~ux' (-Lam^-1): the 1/det rides in the B operand (one multiply instead of one per result register); lands in
            // column 12+b = rows 12..15 of the forward operand; stored behind the next stage's first MFMAs
            const md nli = W::to_m(nahi * rdet);
            if constexpr (KEEP) linv_put(linv, k, N, nli);
            if constexpr (MMA4) Ktq = mma4(hux, nli, W::to_m(vd(0.0)));     // K~'[4b + i][j] in lane j + 4b + 16i
            else Ktp = mma(hux, nli, W::mzero4());
            NDP_FINE(if (fst) { fstamp(*io, m, 13, W::to_d(Hn.r[0])); fstamp(*io, m, 14, W::to_d(MMA4 ? Ktq : Ktp.r[0])); })   // H~_k; K~'
            kprev = k;
            if (k % RESYM == 0) {
                // H~ re-enters the next stage as an A operand, i.e. transposed.  Its antisymmetric rounding part
                // therefore propagates with the OPEN-loop map (x2.2 per stage measured) instead of contracting:
                // re-symmetrise every RESYM-th stage (10: growth x2.6e3 in between, 1e-16 -> 3e-13; one re-symmetrisation per sweep at N = 20,
                // three at N = 40).  H~' = (H~ as A operand) x I costs four MFMAs, no LDS.
                md ey[4];          // identity as B operand: chunk c, lane (g, j) = [j == 4c + g] -- made here, every 8th stage, not kept
                {
                    vi ln = W::lane_here();
                    vi gg = ln >> 4, jj = W::lcol(ln);
                    for (int c = 0; c < 4; ++c) ey[c] = W::to_m(W::sel(jj == gg + 4 * c, vd(1.0), vd(0.0)));
                }
                md4 Tp = mman<4>(Hn.r, ey, W::mzero4());
                for (int r = 0; r < 4; ++r) Hn.r[r] = W::mavg(Hn.r[r], Tp.r[r]);
            }
            H = Hn;
            for (int c = 0; c < 3; ++c) mk[c] = nmk[c];
            for (int r = 0; r < 4; ++r) cc[r] = ncc[r];
        }
        if (kprev >= 0) {
            if constexpr (MMA4) W::st(lds, T.kt_st4 + mb(kprev), W::to_d(Ktq));
            else for (int c = 0; c < 3; ++c) W::st(lds, T.kt_st[c] + mb(kprev), W::to_d(Ktp.r[c]));
        }
        {   // stage 0: only the gain is needed
            md hux = H.r[3];
            vd cof, rdet;
            if constexpr (ROBUST) {
                lam_factor_ldl(m, T, lds, W::to_d(hux), okv);
                cof = KEEP ? lam_inverse_ldl(m, lds) : vd(0.0);
                rdet = vd(1.0);
                const vd G1 = ldl_solve(m, lds, W::to_d(hux));
                vi ln = W::lane_here();
                vi gg = ln >> 4, jj = W::lcol(ln);
                W::st(lds, W::sel(jj < 12, jj * 4 + gg + m.KT, vi(m.MB + int(MB_DUMP))), -G1);
            } else {
                LamRegs LR;
                lam_gather(T, lds, W::to_d(hux), LR);
                cof = lam_cofactor(T, LR);
                rdet = lam_rdet(T, LR, cof, ok);
            }
            vd nahi = MMA4 ? -cof : W::sel(T.kt_pred, -cof, vd(0.0));
            if constexpr (KEEP) linv_put(linv, 0, N, W::to_m(nahi * rdet));
            if constexpr (ROBUST) {
                // (the gain of stage 0 was stored above)
            } else if constexpr (MMA4) {
                W::st(lds, T.kt_st4, W::to_d(mma4(hux, W::to_m(nahi), W::to_m(vd(0.0)))) * rdet);
            } else {
                md4 Kt = mma(hux, W::to_m(nahi), W::mzero4());
                for (int c = 0; c < 3; ++c) W::st(lds, T.kt_st[c], W::to_d(Kt.r[c]) * rdet);
            }
        }
        ok = W::all(okv) && ok;
        W::sync();
        if (io) stamp(*io, m, 6);
        // forward rollout; z~ index 4c+g lives in chunk c of the lanes with j == 0 (MMA4: of every lane -- the four blocks of the
        // 4x4x4 instruction each read their own copy of the vector)
        md zc[3];
        for (int c = 0; c < 3; ++c) {
            vi idx = g + 4 * c;
            vb mine = MMA4 ? (lane >= 0) : T.col0;
            vd v = W::ldp(lds, idx + m.ZX, mine && (idx < 10));
            zc[c] = W::to_m(W::sel(mine && (idx == 10), vd(1.0), v));
        }
        // per stage: Y = [M~x ; K~] z~ (3 MFMAs) holds M~x z~ in rows 0..11 and du = K~ z~ in rows 12..15, i.e. du is
        // accumulator register 3 -- exactly the B operand of the 4th MFMA, which adds B~ du to rows 0..11.
        md fw[3], mu;
        for (int c = 0; c < 3; ++c) fw[c] = W::to_m(W::ld(lds, T.fw_off[c]));
        mu = W::to_m(W::ld(lds, T.mu_off));
        NDP_UNROLL_SWEEP
        for (int k = 0; k < N; ++k) {
            md nfw[3], nmu;
            const int kn = k + 1 < N ? k + 1 : k;
            for (int c = 0; c < 3; ++c) nfw[c] = W::to_m(W::ld(lds, T.fw_off[c] + mb(kn)));
            nmu = W::to_m(W::ld(lds, T.mu_off + mb(kn)));
            if constexpr (MMA4) {
                // four 4x4x4 products (a matrix-VECTOR product needs one column): y[4b + i] in lane j + 4b + 16i, any j
                md y = mma4(fw[0], zc[0], W::to_m(vd(0.0)));
                y = mma4(fw[1], zc[1], y);
                y = mma4(fw[2], zc[2], y);
                md du = W::template rowb<3>(y);                      // rows 12..15 = K~ z~, as the next product's B operand
                md xn = mma4(mu, du, y);
                W::st(lds, T.zu_st + k * int(NU), W::to_d(du));
                W::st(lds, T.zx_st4 + (k + 1) * int(NX), W::to_d(xn));
                zc[0] = W::template rowb<0>(xn); zc[1] = W::template rowb<1>(xn); zc[2] = W::template rowb<2>(xn);
            } else {
                md4 Y = mman<3>(fw, zc, W::mzero4());
                md du = Y.r[3];
                md4 xn = mma(mu, du, Y);
                W::st(lds, T.zu_st + k * int(NU), W::to_d(du));
                for (int c = 0; c < 3; ++c) {
                    zc[c] = xn.r[c];
                    W::st(lds, T.zx_st[c] + (k + 1) * int(NX), W::to_d(xn.r[c]));
                }
            }
            for (int c = 0; c < 3; ++c) fw[c] = nfw[c];
            mu = nmu;
        }
        W::sync();
        return ok;
    }

    // ---------------------------------------------------------------- second solve with the same factorisation
    // Mehrotra's corrector (and any further right-hand side) differs from the predictor only in the gradient of the bounded
    // variables: the Hessians, Lam, the gains are the same, and the Riccati recursion is linear in the gradient.  With
    // dc_k = change of the gradient of stage k (rows 3..5: v bounds, rows 12..15: u bounds), the change of the solution is
    //   backward:  dg_k = dc_k + M~_k' dp_{k+1}     (3 MFMAs, dp as a single column)
    //              [dp_k ; dk_k] = dg_k + [K~_k' ; -Lam_k^-1 - I] dg_k(u)     (1 MFMA; dk_k = change of the feed-forward)
    //   forward:   du_k = K~_k dx_k + dk_k,  dx_{k+1} = A_k dx_k + B_k du_k,  dx_0 = 0      (the 4 MFMAs of the full forward sweep)
    // and ZX|ZU += (dx, du).  No 4x4 inverse, no cost blocks: ~0.55 of a full sweep.  Needs -Lam_k^-1 of every stage (riccati_sweep's
    // linv, in registers: compile-time horizons) and K~' where the sweep left it in LDS.
    struct DeltaTabs { vi dc_off[4], kta_off, dc_off4, dk_off, zd_off, zd_str; md eye12, m12; vb row10, g2; };
    static NDP_D void build_delta_tabs(const LdsMap &m, DeltaTabs &D)
    {
        vi lane = W::lane_here();
        vi g = lane >> 4, j = W::lcol(lane);
        vb c0 = j == 0;
        for (int r = 0; r < 4; ++r) {
            vi row = g + 4 * r;
            vb isv = c0 && (row >= 3) && (row < 6), isu = c0 && (row >= 12);
            D.dc_off[r] = W::sel(isv, row + (m.CB + int(CB_QE)), W::sel(isu, row + (m.CB + int(CB_RE) - 12), vi(m.CB + int(CB_ZERO))));
        }
        D.kta_off = W::sel(j < 10, j * 4 + g + m.KT, vi(m.MB + int(MB_ZERO)));     // K~'[j][g] as A operand, rows 10..15 read a structural 0
        D.eye12 = W::to_m(W::sel((j >= 12) && (j - 12 == g), vd(1.0), vd(0.0)));
        D.row10 = c0 && (g == 2);                                                  // row 10 = g + 4r with g = 2, r = 2
        // 4x4x4 form: dc of row x = 4b + i in lane j + 4b + 16i; riccati_sweep's linv is then unmasked (-Lam^-1 in every lane)
        vi x4 = ((lane >> 2) & 3) * 4 + g;
        D.dc_off4 = W::sel((x4 >= 3) && (x4 < 6), x4 + (m.CB + int(CB_QE)), W::sel(x4 >= 12, x4 + (m.CB + int(CB_RE) - 12), vi(m.CB + int(CB_ZERO))));
        D.m12 = W::to_m(W::sel(j >= 12, vd(1.0), vd(0.0)));
        D.g2 = g == 2;
        // 4x4x4 form: the feed-forward change dk_k (4 values per stage) waits for the forward pass in the U part of the shadow
        // ZD, entry 4k + g -- every lane of row g holds the same value, so all sixteen store it (identical duplicates).  The
        // forward sweeps' own dump stores reach entry 4k + g only at the END of stage k, after dk_k has been read.
        D.dk_off = g + (m.ZD + (m.ZU - m.ZX));
        // refinement solves (refine_gradient / delta_sweep<true>): the full gradient of stage k waits in the shadow ZD, laid out like
        // ZX|ZU -- x row i at ZD + 10 k + i, u row i at ZD + (ZU - ZX) + 4 k + i: a per-lane stride, rows 10 / 11 read a structural zero
        D.zd_off = W::sel(x4 < 10, x4 + m.ZD, W::sel(x4 >= 12, x4 - 12 + (m.ZD + (m.ZU - m.ZX)), vi(m.MB + int(MB_ZERO))));
        D.zd_str = W::sel(x4 < 10, vi(int(NX)), W::sel(x4 >= 12, vi(int(NU)), vi(0)));
    }

    // Refinement of the solution z = ZX|ZU of the Newton system at hand (oracle: refine_solution).  The gradient of its quadratic at
    // z, g = qe + Qe z | re + Re du with the effective blocks as they stand in the cost blocks, goes to the shadow ZD; delta_sweep<true>
    // then solves the same system for that gradient with zero defects and zero initial state -- the factorisation (gains in LDS, -Lam^-1
    // in registers) is the one at hand -- and adds the result to z.
    static NDP_D void refine_gradient(const RtiParams &P, const LdsMap &m, lp lds)
    {
        const int N = horizon(P);
        vi lane = W::lane_here();
        vi e = lane & 15, kq = lane >> 4;
        vb isx = e < 10, isd = e < 6, isq = isx && !isd, isu = e >= 12;
        vi a4 = W::sel(isq, (e - 6) * 4, vi(0));
        for (int t = 0; 4 * t <= N; ++t) {
            vi k = kq + 4 * t;
            vb ok = (isx && (k <= N)) || (isu && (k < N));
            vi kc = W::imin(k, W::sel(isu, vi(N - 1), vi(N)));      // lanes past the end re-read the last stage (their store is masked)
            vi cbk = kc * int(CB_STRIDE) + m.CB;
            vi zo = W::sel(isu, kc * int(NU) + (e - 12) + m.ZU, kc * int(NX) + W::sel(isx, e, vi(0)) + m.ZX);
            vd z = W::ld(lds, zo);
            vd d = W::ld(lds, cbk + W::sel(isd, e + int(CB_DEX), W::sel(isu, e - 12 + int(CB_DEU), vi(int(CB_ZERO)))));
            vd ge = W::ld(lds, cbk + W::sel(isx, e + int(CB_QE), W::sel(isu, e - 12 + int(CB_RE), vi(int(CB_ZERO)))));
            vd acc = d * z + ge;
            vd qacc = ge;
            for (int b = 0; b < 4; ++b)
                qacc = qacc + W::ld(lds, cbk + a4 + (int(CB_QQ) + b)) * W::ld(lds, kc * int(NX) + (m.ZX + 6 + b));
            vd g = W::sel(isq, qacc, acc);
            vi so = W::sel(isu, kc * int(NU) + (e - 12) + (m.ZD + (m.ZU - m.ZX)), kc * int(NX) + W::sel(isx, e, vi(0)) + m.ZD);
            W::stp(lds, so, g, ok);
        }
        W::sync();
    }

    // FROM_ZD: the gradient is the FULL one refine_gradient left in the shadow ZD (all state and input rows, the terminal stage
    // included) instead of the corrector's change of the bounded rows in the cost blocks.
    // Returns (FROM_ZD) the largest correction it added, |dx|, |du| over the horizon -- the caller's check that the solves of a stiff
    // system can still be trusted (REFINE_FAIL) -- else 0.
    static constexpr double REFINE_FAIL = 1e-5;
    template <bool FROM_ZD = false>
    static NDP_D double delta_sweep(const RtiParams &P, const LdsMap &m, const Tables &T, const DeltaTabs &D, lp lds, const md *linv)
    {
        const int N = horizon(P);
        vd cmax = 0.0;
        md vc[3] = {W::to_m(vd(0.0)), W::to_m(vd(0.0)), W::to_m(vd(0.0))};
        if constexpr (MMA4) {
            // every product here is matrix x vector: the 4x4x4 instruction, vectors as chunk registers replicated over the columns.
            // The LDS operands of a stage are requested one stage ahead (as in riccati_sweep): they arrive under the previous
            // stage's dependent chain instead of in front of this one's.
            auto dc_at = [&](int k) { return FROM_ZD ? W::ld(lds, D.zd_off + D.zd_str * k) : W::ld(lds, D.dc_off4 + cb(k)); };
            md Dg = W::to_m(dc_at(N - 1));
            md a[3], kta = W::to_m(W::ld(lds, D.kta_off + mb(N - 1)));
            for (int c = 0; c < 3; ++c) a[c] = W::to_m(vd(0.0));           // stage N-1 has no successor term ...
            if constexpr (FROM_ZD) {                                       // ... but for the terminal stage's own gradient
                vi g4 = W::lane() >> 4;
                for (int c = 0; c < 3; ++c) {
                    a[c] = W::to_m(W::ld(lds, T.mk_off[c] + mb(N - 1)));
                    vi row = g4 + 4 * c;
                    vc[c] = W::to_m(W::ldp(lds, W::sel(row < 10, row, vi(0)) + (m.ZD + N * int(NX)), row < 10));
                }
            }
            NDP_UNROLL_STAGES
            for (int k = N - 1; k >= 0; --k) {
                md nDg = Dg, na[3] = {a[0], a[1], a[2]}, nkta = kta;
                if (k > 0) {
                    nDg = W::to_m(dc_at(k - 1));
                    for (int c = 0; c < 3; ++c) na[c] = W::to_m(W::ld(lds, T.mk_off[c] + mb(k - 1)));
                    nkta = W::to_m(W::ld(lds, D.kta_off + mb(k - 1)));
                }
                md akl = kta + W::fma(linv_get(linv, k), D.m12, -D.eye12);
                if (FROM_ZD || k != N - 1)
                    for (int c = 0; c < 3; ++c) Dg = mma4(a[c], vc[c], Dg);
                md Dp = mma4(akl, W::template rowb<3>(Dg), Dg);
                W::st(lds, D.dk_off + k * int(NU), W::to_d(W::template rowb<3>(Dp)));   // dk_k (kept in LDS: NC registers pairs less)
                vc[0] = W::template rowb<0>(Dp); vc[1] = W::template rowb<1>(Dp);
                vc[2] = W::msel(D.g2, W::to_m(vd(0.0)), W::template rowb<2>(Dp));   // the constant-term row must not feed back
                Dg = nDg; kta = nkta;
                for (int c = 0; c < 3; ++c) a[c] = na[c];
            }
            md zc[3] = {W::to_m(vd(0.0)), W::to_m(vd(0.0)), W::to_m(vd(0.0))};
            md fw[3], mu;
            for (int c = 0; c < 3; ++c) fw[c] = W::to_m(W::ld(lds, T.fw_off[c]));
            mu = W::to_m(W::ld(lds, T.mu_off));
            W::sync();       // the dk stores above are read by other lanes below
            vd zu0 = W::ld(lds, T.zu_st), zx0 = W::ld(lds, T.zx_st4 + int(NX));
            md dkk = W::to_m(W::ld(lds, D.dk_off));
            NDP_UNROLL_STAGES
            for (int k = 0; k < N; ++k) {
                md nfw[3] = {fw[0], fw[1], fw[2]}, nmu = mu, ndk = dkk;
                vd nzu0 = zu0, nzx0 = zx0;
                if (k + 1 < N) {
                    for (int c = 0; c < 3; ++c) nfw[c] = W::to_m(W::ld(lds, T.fw_off[c] + mb(k + 1)));
                    nmu = W::to_m(W::ld(lds, T.mu_off + mb(k + 1)));
                    nzu0 = W::ld(lds, T.zu_st + (k + 1) * int(NU));
                    nzx0 = W::ld(lds, T.zx_st4 + (k + 2) * int(NX));
                    ndk = W::to_m(W::ld(lds, D.dk_off + (k + 1) * int(NU)));
                }
                md y = W::to_m(vd(0.0)), du = dkk;
                if (k != 0) {
                    for (int c = 0; c < 3; ++c) y = mma4(fw[c], zc[c], y);
                    du = W::template rowb<3>(y) + dkk;
                }
                md xn = mma4(mu, du, y);
                if constexpr (FROM_ZD) {      // rows 0..9 of xn are dx_{k+1} (row x = 4b + i in lane j + 4b + 16i), du: all four rows
                    vi x4 = ((W::lane() >> 2) & 3) * 4 + (W::lane() >> 4);
                    cmax = W::vmax(cmax, W::vmax(W::vabs(W::to_d(du)), W::sel(x4 < 10, W::vabs(W::to_d(xn)), vd(0.0))));
                }
                W::st(lds, T.zu_st + k * int(NU), zu0 + W::to_d(du));
                W::st(lds, T.zx_st4 + (k + 1) * int(NX), zx0 + W::to_d(xn));
                zc[0] = W::template rowb<0>(xn); zc[1] = W::template rowb<1>(xn); zc[2] = W::template rowb<2>(xn);
                for (int c = 0; c < 3; ++c) fw[c] = nfw[c];
                mu = nmu; zu0 = nzu0; zx0 = nzx0; dkk = ndk;
            }
            W::sync();
            if constexpr (FROM_ZD) return W::wave_max(cmax);
            return 0.0;
        }
        md dk[NC > 0 ? NC : 1];
        NDP_UNROLL_STAGES
        for (int k = N - 1; k >= 0; --k) {
            md4 C;
            for (int r = 0; r < 4; ++r) C.r[r] = W::to_m(W::ld(lds, D.dc_off[r] + cb(k)));
            md a[3];
            for (int c = 0; c < 3; ++c) a[c] = W::to_m(W::ld(lds, T.mk_off[c] + mb(k)));
            md akl = W::to_m(W::ld(lds, D.kta_off + mb(k))) + (linv_get(linv, k) - D.eye12);
            md4 Dg = k == N - 1 ? C : mman<3>(a, vc, C);
            md4 Dp = mma(akl, Dg.r[3], Dg);
            dk[k] = Dp.r[3];
            vc[0] = Dp.r[0]; vc[1] = Dp.r[1];
            vc[2] = W::msel(D.row10, W::to_m(vd(0.0)), Dp.r[2]);                    // the constant-term row must not feed back
        }
        md zc[3] = {W::to_m(vd(0.0)), W::to_m(vd(0.0)), W::to_m(vd(0.0))};
        NDP_UNROLL_STAGES
        for (int k = 0; k < N; ++k) {
            md fw[3], mu;
            for (int c = 0; c < 3; ++c) fw[c] = W::to_m(W::ld(lds, T.fw_off[c] + mb(k)));
            mu = W::to_m(W::ld(lds, T.mu_off + mb(k)));
            vd zu0 = W::ld(lds, T.zu_st + k * int(NU));
            vd zx0[3];
            for (int c = 0; c < 3; ++c) zx0[c] = W::ld(lds, T.zx_st[c] + (k + 1) * int(NX));
            md4 Y = k == 0 ? W::mzero4() : mman<3>(fw, zc, W::mzero4());
            md du = Y.r[3] + dk[k];
            md4 xn = mma(mu, du, Y);
            W::st(lds, T.zu_st + k * int(NU), zu0 + W::to_d(du));
            for (int c = 0; c < 3; ++c) {
                zc[c] = xn.r[c];
                W::st(lds, T.zx_st[c] + (k + 1) * int(NX), zx0[c] + W::to_d(xn.r[c]));
            }
        }
        W::sync();
        return 0.0;
    }

    // ---------------------------------------------------------------- box constraints / interior point
    // order: du_k[0..3] k=0..N-1 (idxbu), then dv_k[0..2] k=1..N-1 (idxbx = 3,4,5; nmpc_body_rate_ctl.py:56-61)
    static NDP_D void build_slots(const RtiParams &P, const LdsMap &m, Slots &S)
    {
        const int N = horizon(P), nu = 4 * N, mcon = 7 * N - 3;
        vi lane = W::lane_here();      // (computed here, at the interior-point loop's door -- not hoisted to the kernel's start)
        for (int s = 0; s < NSLOT; ++s) {
            vi n = lane + 64 * s;
            S.valid[s] = n < mcon;
            vb isu = n < nu;
            vi nv = W::sel(isu, vi(0), n - nu);
            vi kv = W::div3(nv);
            vi iv = nv - kv * 3;
            vi k = W::sel(isu, n >> 2, kv + 1);
            k = W::sel(S.valid[s], k, vi(0));
            vi i = W::sel(isu, n & 3, iv);
            vi cb = k * int(CB_STRIDE) + m.CB;
            S.zoff[s] = W::sel(isu, k * NU + i + m.ZU, k * NX + i + (m.ZX + 3));
            S.de_off[s] = cb + W::sel(isu, i + int(CB_DEU), i + (int(CB_DEX) + 3));
            S.lb_off[s] = W::sel(isu, i + (m.KC + KC_LBU), i + (m.KC + KC_LBV));
        }
        S.io = m.XI - m.ZX;           // (= UI - ZU: the two pairs of arrays are laid out alike)
    }

    // the step bounds of slot s.  Kernels with up to three constraint slots (N <= 27: the reference configuration) hold them in
    // registers (load_bounds); the five-slot kernels (N = 40) read them from LDS where they are needed -- the iterate does not move
    // during a QP solve -- which is 20 registers per lane less across the sweeps and costs ~60 LDS reads per interior-point iteration
    // (measured on the N = 20 kernels, where it is therefore NOT done: interior point on every instance 14.2 against 14.55 M solves/s)
    static constexpr bool KEEP_BOUNDS = NSLOT <= 3;
    static NDP_D void bounds(const Slots &S, lp lds, int s, vd &lo, vd &hi)
    {
        if constexpr (KEEP_BOUNDS) { lo = S.lo[s]; hi = S.hi[s]; }
        else {
            vd cur = W::ld(lds, S.zoff[s] + S.io);
            lo = W::ld(lds, S.lb_off[s]) - cur;
            hi = W::ld(lds, S.lb_off[s] + int(SL_UB)) - cur;
        }
    }

    static NDP_D void load_bounds(const LdsMap &, Slots &S, lp lds)
    {
        for (int s = 0; s < NSLOT; ++s) {
            vd cur = W::ld(lds, S.zoff[s] + S.io);
            S.lo[s] = W::ld(lds, S.lb_off[s]) - cur;
            S.hi[s] = W::ld(lds, S.lb_off[s] + int(SL_UB)) - cur;
        }
    }

    // ---------------------------------------------------------------- active set on the input bounds (QP_AUTO)
    // The reference's QP solver iterates on every QP (HPIPM, cold start: nmpc_body_rate_ctl.py:71-74); any method that converges the
    // same strictly convex QP is parity-equivalent, and the bounds that are ever active in the reference's envelope are
    // the INPUT bounds (nmpc_body_rate_ctl.py:56-58).  Primal-dual active set (Hintermueller / Ito / Kunisch) on those: with the set
    // A of pinned inputs (du = its step bound d), solve the equality-constrained QP, then
    //     pinned, multiplier of the wrong sign      -> released          inactive, beyond a bound -> pinned there
    // all at once; a set that reproduces itself satisfies the KKT conditions of the box-constrained QP -- it IS the solution.  One
    // Riccati sweep per iteration (the interior-point loop: ~1.55 per iteration, 5-9 iterations); typically two or three sweeps from
    // an empty set, ONE when the previous control step's set still holds (RtiIo::act: the warm start HPIPM is not given).
    // A pin is a weight: the term as_gamma/2 (du - d)^2, solved for delta = du - d (as_apply) -- it rides on the diagonal of Lam exactly
    // where the interior-point loop's barrier terms ride (up to 1e10 in its last iterations), costs the sweep nothing, leaves du within
    // lambda / as_gamma of d (then set to d exactly) and hands back the multiplier lambda = as_gamma delta.
    // Velocity bounds are not pinned (a weight on a STATE makes the recursion stiff: see ROBUST): a violated one, a set that does not
    // settle within as_iter_max sweeps, or a failed factorisation hand the QP to the interior-point loop, which starts from the base
    // cost blocks whatever the pins did to them.
#if defined(NDP_DEV_NO_AS)
    static constexpr bool ASET = false;
#elif defined(NDP_DEV_NO_AS5)
    static constexpr bool ASET = NSLOT <= 3;
#else
    static constexpr bool ASET = true;
#endif
    static constexpr int RUA = NC ? (NC * NU + 63) / 64 : NSLOT;     // 64-lane rounds over the 4N input bounds as they lie in ZU
    // The set: one value per input bound, +1 pinned at its upper bound, -1 at its lower, 0 free.  Three-slot kernels: RUA registers per
    // lane.  Five-slot kernels (N = 40: 500 of 512 registers before this existed): parked in LDS (LdsMap::AS, as doubles) once the kept
    // set has arrived, read where it is looked at.
    static constexpr bool A_LDS = NSLOT > 3;
    struct ActSet { vi a[RUA]; };
    static NDP_D vi a_get(const ActSet &A, const LdsMap &m, lp lds, int t, vi e)
    {
        if constexpr (A_LDS) return W::d2i(W::ld(lds, e + m.AS));
        else return A.a[t];
    }
    static NDP_D void a_put(ActSet &A, const LdsMap &m, lp lds, int t, vi e, vi v)
    {
        if constexpr (A_LDS) W::st(lds, e + m.AS, W::i2d(v));
        else A.a[t] = v;
    }
    // element index of round t (lanes past the end repeat the last element: identical duplicates everywhere)
    static NDP_D vi a_elem(const RtiParams &P, vi lane, int t) { return W::imin(lane + 64 * t, horizon(P) * NU - 1); }
    // A_LDS: the kept set (requested by as_issue) -> its LDS area; called where the step waits for its inputs anyway
    static NDP_D void as_park(const RtiParams &P, const LdsMap &m, lp lds, ActSet &A)
    {
        if constexpr (A_LDS) {
            vi lane = W::lane_here();
            for (int t = 0; t < RUA; ++t) W::st(lds, a_elem(P, lane, t) + m.AS, W::i2d(A.a[t]));
            W::sync();
        }
    }
    static NDP_D void as_clear(const RtiParams &P, const LdsMap &m, lp lds, ActSet &A)
    {
        vi lane = W::lane_here();
        for (int t = 0; t < RUA; ++t) a_put(A, m, lds, t, a_elem(P, lane, t), vi(0));
        if constexpr (A_LDS) W::sync();
    }

    // request the instance's kept set (no wait here)
    static NDP_D void as_issue(const RtiParams &P, const RtiIo &io, ActSet &A)
    {
        const int nzu = horizon(P) * NU;
        vi lane = W::lane();
        for (int t = 0; t < RUA; ++t)
            A.a[t] = io.act ? W::gld_i8(io.act, W::imin(lane + 64 * t, nzu - 1)) : vi(0);
    }
    static NDP_D bool as_any(const RtiParams &P, const LdsMap &m, lp lds, const ActSet &A)
    {
        vi lane = W::lane_here();
        vb nz = lane < 0;
        for (int t = 0; t < RUA; ++t) nz = W::bor(nz, !(a_get(A, m, lds, t, a_elem(P, lane, t)) == 0));
        return W::any(nz);
    }
    // The QP's blocks <- base blocks + the pins of A (every input row is rewritten: released pins disappear).  A pinned input is solved
    // for in RE-CENTRED form: du = d + delta with the pin's weight on delta alone -- cost gradient r + R d, dynamics defect b + B d (d = 0
    // on free inputs), diagonal R + as_gamma, no as_gamma d term anywhere.  The sweep then returns delta = lambda / as_gamma itself
    // (|delta| ~ 1e-12 .. 1e-9) to full RELATIVE accuracy, and with it the multiplier's sign.  (Round 6's first form put as_gamma d into
    // the gradient and read the multiplier off du - d: a difference of two numbers that agree to 12 digits, right to ~1e-3 in lambda --
    // one closed-loop instance in ~20 000 kept a pin whose multiplier was negative, 1e-4 .. 1e-2 off the QP's solution.)
    // The defects as the linearisation left them are parked in CX (dead until the interior-point loop, which gets them back: as_restore_b).
    static NDP_D void as_apply(const RtiParams &P, const LdsMap &m, lp lds, const ActSet &A, bool &pinned)
    {
        const int N = horizon(P), nzu = N * NU, nb = N * NX;
        vi lane = W::lane_here();
        auto rows = [&](int t) {
            vi e = W::imin(lane + 64 * t, nzu - 1);               // lanes past the end repeat the last element (identical duplicate stores)
            vi c = e & 3;
            vi cbk = (e >> 2) * int(CB_STRIDE) + c + m.CB;
            vd cu = W::ld(lds, e + m.UI);
            const vi at = a_get(A, m, lds, t, e);
            vb up = at > 0, on = !(at == 0);
            vd d = W::sel(on, W::ld(lds, c + (m.KC + int(KC_LBU)) + W::sel(up, vi(int(SL_UB)), vi(0))) - cu, vd(0.0));
            vd w = W::sel(on, vd(W::late_params(P)->as_gamma), vd(0.0));
            vd rd = P.dt * W::ld(lds, c + (m.KC + int(KC_RD)));
            W::st(lds, cbk + int(CB_DEU), rd + w);
            W::st(lds, cbk + int(CB_RE), W::ld(lds, cbk + int(CB_RB)) + rd * d);
            W::st(lds, e + m.CU, d);
        };
        if constexpr (A_LDS) {            // (the set is read from LDS: the loop stays a loop -- those kernels sit at the register limit)
            NDP_KEEP_LOOP
            for (int t = 0; t < RUA; ++t) rows(t);
        } else {
            for (int t = 0; t < RUA; ++t) rows(t);
        }
        W::sync();
        NDP_KEEP_LOOP
        for (int t = 0; t * 64 < nb; ++t) {                       // one lane per (stage, state row): b_k[row] = base + B_k[row][:] d_k
            vi e = W::imin(lane + 64 * t, nb - 1);
            vi k = (e * 6554) >> 16, row = e - k * 10;             // e / 10, e < 16 384
            vi mb = k * int(MB_STRIDE) + m.MB;
            vb pv = row < 6;                                       // rows 0..5: 6x8 block, inputs in columns 4..7; rows 6..9: 4x7 block, columns 4..6 (no thrust column)
            vi bo = mb + W::sel(pv, row * 8 + (int(MB_PV) + 4), (row - 6) * 7 + (int(MB_Q) + 4));
            vd base;
            if (!pinned) {                                         // the first pins of this QP: the defects are still the linearisation's
                base = W::ld(lds, mb + row + int(MB_B));
                W::st(lds, e + m.CX, base);
            } else base = W::ld(lds, e + m.CX);
            vd s = base;
            for (int j = 0; j < 4; ++j) {
                vi off = j < 3 ? bo + j : W::sel(pv, bo + 3, mb + int(MB_ZERO));
                s = s + W::ld(lds, off) * W::ld(lds, k * 4 + j + m.CU);
            }
            W::st(lds, mb + row + int(MB_B), s);
        }
        pinned = true;
        W::sync();
    }
    static NDP_D void as_save_b(const RtiParams &P, const LdsMap &m, lp lds)
    {
        const int nb = horizon(P) * NX;
        vi lane = W::lane_here();
        NDP_KEEP_LOOP
        for (int t = 0; t * 64 < nb; ++t) {
            vi e = W::imin(lane + 64 * t, nb - 1);
            vi k = (e * 6554) >> 16, row = e - k * 10;
            W::st(lds, e + m.CX, W::ld(lds, k * int(MB_STRIDE) + row + (m.MB + int(MB_B))));
        }
        W::sync();
    }
    // the defects as the linearisation left them, back from CX (in front of the interior-point loop, which starts from the base blocks)
    static NDP_D void as_restore_b(const RtiParams &P, const LdsMap &m, lp lds)
    {
        const int nb = horizon(P) * NX;
        vi lane = W::lane_here();
        NDP_KEEP_LOOP
        for (int t = 0; t * 64 < nb; ++t) {
            vi e = W::imin(lane + 64 * t, nb - 1);
            vi k = (e * 6554) >> 16, row = e - k * 10;
            W::st(lds, k * int(MB_STRIDE) + row + (m.MB + int(MB_B)), W::ld(lds, e + m.CX));
        }
        W::sync();
    }
    // The sweep's solution (ZX|ZU) against the set it was made with.  Returns 0: the set reproduces itself, every free input and every
    // velocity is inside its box (velocities by auto_margin) -- the QP is solved, pinned inputs are set onto their bounds; 1: the set
    // changed (A updated); 2: a velocity bound is violated or closer than auto_margin -- not this method's case.
    // With as_iter_max = 0 the test is rounds 1-5's: accepted only if inside EVERY bound by auto_margin (the moved bounds of fill_kc) -- what keeps the early
    // exit within ~1e-6 of an interior-point solve stopped at mu <= tol (HPIPM, the oracle's qp_mode 1): at a slack t the barrier leaves a
    // multiplier mu / t on an INACTIVE bound, which moves the solution by ~mu / (t * weight).  With the active set on, the inputs' margin is 0: the
    // answer is the QP's exact solution, and an interior-point answer is compared with it at ITS accuracy (tests: the oracle at a tight
    // tolerance).  The velocity margin stays: a velocity bound that close belongs to the interior-point loop.
    // Evaluated without the constraint slots (which the interior-point loop needs, this does not): the 4N input bounds as they lie in
    // ZU (element e: component e & 3), the 3(N-1) velocity bounds four lanes per stage (component 3 idle) -- no division by three.
    // update = false (as_iter_max = 0): the set stays empty whatever the verdict (it is looked at again by the next RTI iteration).
    // (A NaN makes every comparison false: the QP then goes the interior-point loop's way, which reports it.)
    // pinned = false: the sweep was made with an empty set (the usual case: the caller knows) -- the test is then the handful of
    // comparisons rounds 1-5 made, and everything that deals with pins sits behind a branch the wave almost never takes.
    static NDP_D int as_check(const RtiParams &P, const LdsMap &m, lp lds, ActSet &A, bool update, bool pinned)
    {
        const int N = horizon(P), nzu = N * NU, nv4 = 4 * (N - 1);
        constexpr int RVm = NC ? (4 * (NC - 1) + 63) / 64 : NSLOT;
        // (an opaque lane id: this sits in a loop now, and index arithmetic the compiler can trace to the thread id is loop-invariant --
        // hoisted in front of the sweep and held across it)
        const vi lane = W::lane_here();
        const int io = m.XI - m.ZX;
        // every LDS read of the test is requested before the first comparison (one wait), the verdicts are combined without control flow
        vd zu[RUA], cu[RUA], lu[RUA], hu[RUA], zv[RVm], cv[RVm], lv[RVm], hv[RVm];
        for (int t = 0; t < RUA; ++t) {
            vi e = W::imin(lane + 64 * t, nzu - 1);
            vi c = e & 3;
            zu[t] = W::ld(lds, e + m.ZU); cu[t] = W::ld(lds, e + (m.ZU + io));
            lu[t] = W::ld(lds, c + (m.KC + int(KC_LBUM))); hu[t] = W::ld(lds, c + (m.KC + int(KC_UBUM)));    // (moved inwards by the margin: fill_kc)
        }
        for (int t = 0; t < RVm; ++t) {
            vi q = W::imin(lane + 64 * t, nv4 - 1);
            vi k = (q >> 2) + 1, c = W::imin(q & 3, 2);           // component 3 repeats component 2
            vi zo = k * int(NX) + c + (m.ZX + 3);
            zv[t] = W::ld(lds, zo); cv[t] = W::ld(lds, zo + io);
            lv[t] = W::ld(lds, c + (m.KC + int(KC_LBVM))); hv[t] = W::ld(lds, c + (m.KC + int(KC_UBVM)));
        }
        vb vok = lane >= 0, uok = lane >= 0;
        for (int t = 0; t < RUA; ++t)
            uok = W::band(uok, W::band(zu[t] > lu[t] - cu[t], zu[t] < hu[t] - cu[t]));
        for (int t = 0; t < RVm; ++t)
            vok = W::band(vok, W::band(zv[t] > lv[t] - cv[t], zv[t] < hv[t] - cv[t]));
        if (!pinned && W::all(W::band(uok, vok))) return 0;         // empty set, everything inside: the QP's solution
        if (!W::all(vok)) return 2;
        // ---- the rare part: pins in play, or a free input beyond a bound
        vb same = lane >= 0;
        vi at[RUA], na[RUA];
        for (int t = 0; t < RUA; ++t) {
            at[t] = a_get(A, m, lds, t, W::imin(lane + 64 * t, nzu - 1));
            const vd lo = lu[t] - cu[t], hi = hu[t] - cu[t];
            const vb up = at[t] > 0, dn = at[t] < 0, on = W::bor(up, dn);
            // pinned (ZU holds delta = du - d, see as_apply): multiplier as_gamma delta for an upper, -as_gamma delta for a lower bound;
            // released when negative
            const vb keep = W::band(on, W::sel(up, zu[t], vd(0.0) - zu[t]) >= 0.0);
            // free: beyond a bound (or, as_iter_max = 0, closer to it than auto_margin) -> pinned there
            const vb vhi = W::band(!on, !(zu[t] < hi)), vlo = W::band(!on, !(zu[t] > lo));
            na[t] = W::sel(keep, at[t], W::sel(vhi, vi(1), W::sel(vlo, vi(-1), vi(0))));
            same = W::band(same, na[t] == at[t]);
        }
        if (W::all(same)) {
            for (int t = 0; t < RUA; ++t) {       // (pins exist only with the active set on: the moved input bounds are the bounds)
                vi e = W::imin(lane + 64 * t, nzu - 1);
                W::stp(lds, e + m.ZU, W::sel(at[t] > 0, hu[t], lu[t]) - cu[t], !(at[t] == 0));       // onto the bound exactly
            }
            W::sync();
            return 0;
        }
        if (update) {
            for (int t = 0; t < RUA; ++t) a_put(A, m, lds, t, W::imin(lane + 64 * t, nzu - 1), na[t]);
            if constexpr (A_LDS) W::sync();
        }
        return 1;
    }
    // keep the set for the next control step
    static NDP_D void as_store(const RtiParams &P, const LdsMap &m, lp lds, const RtiIo &io, const ActSet &A)
    {
        const int nzu = horizon(P) * NU;
        vi lane = W::lane_here();
        for (int t = 0; t < RUA; ++t) {
            vi e = lane + 64 * t;
            W::gst_i8(io.act, e, a_get(A, m, lds, t, a_elem(P, lane, t)), e < nzu);
        }
    }

    static NDP_D double absmax(lp lds, int off, int n)
    {
        vi lane = W::lane();
        vd mx = 0.0;
        for (int t = 0; t < n; t += 64) {
            vi i = lane + t;
            mx = W::vmax(mx, W::vabs(W::ldp(lds, i + off, i < n)));
        }
        return W::wave_max(mx);
    }

    // Mehrotra predictor-corrector in absolute form; every Newton system is one riccati_sweep with
    // diag += Gamma, grad += gamma on the bounded variables (same algorithm as oracle orc_qp_solve).
    // `failed` is set when no usable step exists (factorisation failure or NaN): the caller then keeps the iterate, as
    // acados' SQP_RTI returns ACADOS_QP_FAILURE before updating the variables; an exhausted iteration budget (status 4
    // as well) still hands over the last interior-point iterate.
    static NDP_D int ipm(const RtiParams &P, const LdsMap &m, const Tables &T, Slots &S, lp lds, int &iters_out, bool &failed,
                         const RtiIo *fio = nullptr /* fine stamps of the first iteration (-DNDP_FINE_STAMPS builds, debug path) */)
    {
        const int N = horizon(P);
        const int nzx = (N + 1) * NX, nzu = N * NU;
        // The loop's own constants (start values, tolerances, refinement switches) are read HERE, through a view of the parameter block
        // the compiler cannot connect with the kernel's arguments: read as plain members it fetches all fourteen words in front of the
        // nominal sweep -- the loop is entered from there -- and with the scalar registers as full as they are, each fetch then waits
        // for itself and is spilled (three load-wait-spill rounds on the hot path, ~300 cycles per step, measured with phase stamps).
        const auto *Q = W::late_params(P);
        const double inv2m = Q->inv2m;
        vi lane = W::lane();
        int status = 0, iters = 0;
        // cold start at dz = 0
        vd musum = 0.0, nrm = 1.0;
        for (int s = 0; s < NSLOT; ++s) {
            vb v = S.valid[s];
            vd lo, hi;
            bounds(S, lds, s, lo, hi);
            S.tl[s] = W::sel(v, W::vmax(-lo, vd(Q->thr0)), vd(1.0));
            S.tu[s] = W::sel(v, W::vmax(hi, vd(Q->thr0)), vd(1.0));
            S.ll[s] = W::sel(v, W::rcp(S.tl[s]) * Q->mu0, vd(0.0));     // reciprocals (v_rcp_f64 + Newton), not IEEE divides:
            S.lu[s] = W::sel(v, W::rcp(S.tu[s]) * Q->mu0, vd(0.0));     // a divide is ~30 VALU instructions, the loop had 14 per slot
            S.pl[s] = 0.0; S.pu[s] = 0.0;
            musum = musum + S.ll[s] * S.tl[s] + S.lu[s] * S.tu[s];
            nrm = W::vmax(nrm, W::vmax(S.ll[s], S.lu[s]));
            vd rdl = W::vabs(-lo - S.tl[s]), rdu = W::vabs(hi - S.tu[s]);
            nrm = W::vmax(nrm, W::sel(v, W::vmax(rdl, rdu), vd(0.0)));
        }
        double mu = W::wave_sum(musum) * inv2m;
        // scale of the residuals: largest magnitude among the slack / multiplier start values, the gradients (q_e: 10 per stage
        // 0..N, r_b: 4 per stage < N), the defects b (10 per stage < N) and dx_0.  One running maximum per lane -- 16 lanes per
        // stage, four stages per round -- and ONE wave reduction (a reduction per block was 61 of them: ~9 k cycles per solve).
        {
            vi e = lane & 15, kq = lane >> 4;
            // (three-slot kernels: branch-free predicated reads, all in flight together -- as exec-masked blocks they were eighteen LDS
            // round trips one after the other; the five-slot kernels, at 510 registers, keep the serial form: 33 reads in flight spill)
            auto rd = [&](vi off, vb p) { if constexpr (NSLOT <= 3) return W::ldz(lds, off, p); else return W::ldp(lds, off, p); };
            for (int t = 0; 4 * t <= N; ++t) {
                vi k = kq + 4 * t;
                nrm = W::vmax(nrm, W::vabs(rd(k * int(CB_STRIDE) + e + (m.CB + int(CB_QE)), (e < 10) && (k <= N))));
                nrm = W::vmax(nrm, W::vabs(rd(k * int(CB_STRIDE) + e + (m.CB + int(CB_RB)), (e < 4) && (k < N))));
                nrm = W::vmax(nrm, W::vabs(rd(k * int(MB_STRIDE) + e + (m.MB + int(MB_B)), (e < 10) && (k < N))));
            }
            nrm = W::vmax(nrm, W::vabs(W::ldp(lds, lane + m.ZX, lane < 10)));
        }
        double norm0 = W::wave_max(nrm);
        double rho = 1.0;
        for (int t = 0; t < nzx + nzu; t += 64) {   // CX and CU are contiguous
            vi i = lane + t;
            W::stp(lds, i + m.CX, vd(0.0), i < nzx + nzu);
        }
        W::sync();
        bool ok = true;
        // corrector as a second solve with the predictor's factorisation (delta_sweep): compile-time horizons (the per-stage
        // -Lam^-1 operands live in registers) on the f64 instruction (in fp32 the interior-point loop converges worse with it)
        constexpr bool DELTA = NC > 0 && NC <= DELTA_MAX_N && W::delta_ok;
        md linv[DELTA ? linv_regs(NC) : 1];
        // Refinement of a Newton system's solution (oracle: refine_solution): a further solve with the factorisation at hand, hence
        // where the second solve exists (DELTA).  Needed only while a STATE bound's barrier term is large: gmaxv, per iteration.
        // Both live in the three-slot kernels only (N <= 27: the reference configuration).  In the five-slot kernels (N = 40, config 5)
        // a third instantiation of the sweep inside the loop brings the registers back over the edge (324 B of scratch per lane) for
        // a case -- active STATE bounds -- that the reference's +-20 m/s box does not produce; there the loop runs as in round 3.
#ifdef NDP_DEV_NO_STIFF      // kernel-development hook: the loop as in round 3 (A/B runs of the headline with and without the extra code)
        constexpr bool STIFF = false;
#else
        constexpr bool STIFF = NSLOT <= 3 && !LEAN;
#endif
#ifdef NDP_DEV_NO_REFINE      // (kernel-development hooks, register studies: NO_REFINE 344 / NO_ROBUST 316 / both 292 of 366 registers)
        constexpr bool REFINE = false;
#else
        constexpr bool REFINE = DELTA && MMA4 && STIFF;
#endif
        const int nu4 = 4 * N;
        for (;;) {
            if (mu <= Q->tol && rho * norm0 <= Q->tol) break;
            if (iters >= Q->iter_max) { status = 4; break; }
            ++iters;
            double sigma_mu = 0.0;
            double gmaxv = 0.0;
            for (int pass = 0; pass < 2; ++pass) {
                vd gmv = 0.0;
                for (int s = 0; s < NSLOT; ++s) {
                    vd sl = pass ? vd(sigma_mu) - S.pl[s] : vd(0.0);
                    vd su = pass ? vd(sigma_mu) - S.pu[s] : vd(0.0);
                    vd rtl = W::rcp(S.tl[s]), rtu = W::rcp(S.tu[s]);
                    vd gl = S.ll[s] * rtl, gu = S.lu[s] * rtu;
                    vd Gam = gl + gu;
                    vd lo = 0.0, hi = 0.0;
                    if (!(DELTA && pass)) bounds(S, lds, s, lo, hi);
                    vd gam = -sl * rtl - S.ll[s] - gl * lo + su * rtu + S.lu[s] - gu * hi;
                    if (STIFF && pass == 0) gmv = W::vmax(gmv, W::sel(S.valid[s] && (lane + 64 * s >= nu4), Gam, vd(0.0)));
                    if (DELTA && pass) {
                        // same diagonal as the predictor; the gradient slot takes the CHANGE of the gradient only
                        W::stp(lds, S.de_off[s] + int(SL_GE), su * rtu - sl * rtl, S.valid[s]);
                    } else {
                        vd dbase = P.dt * W::ld(lds, S.lb_off[s] + int(SL_DW));
                        W::stp(lds, S.de_off[s], dbase + Gam, S.valid[s]);
                        W::stp(lds, S.de_off[s] + int(SL_GE), W::ld(lds, S.de_off[s] + int(SL_GB)) + gam, S.valid[s]);
                    }
                }
                W::sync();
                NDP_FINE(if (fio && fio->dbg && iters == 1) fstamp(*fio, m, 6 + 3 * pass);)
                if (STIFF && pass == 0 && Q->refine > 0) gmaxv = W::wave_max(gmv);
                if (DELTA && pass) {
                    DeltaTabs DT;                    // (a dozen integer instructions: built here, not held across the factorisation sweep)
                    build_delta_tabs(m, DT);
                    delta_sweep(P, m, T, DT, lds, linv);
                }
#ifndef NDP_DEV_NO_ROBUST
                else if (STIFF && NDP_RARELY(Q->refine > 0 && gmaxv > Q->refine_gamma)) ok = riccati_sweep<DELTA, STIFF>(P, m, T, lds, nullptr, linv) && ok;
#endif
                else ok = riccati_sweep<DELTA>(P, m, T, lds, nullptr, linv) && ok;
                if constexpr (REFINE) {
                    if (NDP_RARELY(ok && Q->refine > 0 && gmaxv > Q->refine_gamma)) {
                        if (pass) {
                            // the corrector's second solve left only the CHANGE of the bounded rows' gradient in the cost blocks:
                            // put the corrector's full gradient there, as the refinement needs the gradient of the whole quadratic
                            for (int s = 0; s < NSLOT; ++s) {
                                vd sl = vd(sigma_mu) - S.pl[s], su = vd(sigma_mu) - S.pu[s];
                                vd rtl = W::rcp(S.tl[s]), rtu = W::rcp(S.tu[s]);
                                vd gl = S.ll[s] * rtl, gu = S.lu[s] * rtu;
                                vd lo, hi;
                                bounds(S, lds, s, lo, hi);
                                vd gam = -sl * rtl - S.ll[s] - gl * lo + su * rtu + S.lu[s] - gu * hi;
                                W::stp(lds, S.de_off[s] + int(SL_GE), W::ld(lds, S.de_off[s] + int(SL_GB)) + gam, S.valid[s]);
                            }
                            W::sync();
                        }
                        DeltaTabs DT;
                        build_delta_tabs(m, DT);
                        double corr = 0.0;
                        for (int rf = 0; rf < Q->refine; ++rf) {
                            refine_gradient(P, m, lds);
                            corr = delta_sweep<true>(P, m, T, DT, lds, linv);
                        }
                        // a LAST correction still visible at the parity bar: the solves of this system cannot be trusted (barrier
                        // terms beyond what fp64 carries through the recursion) -- a QP failure, said so, not a silent answer
                        // (the corrector's solve -- the one the iterate is updated from; the predictor only steers the centring)
                        if (pass && !(corr <= REFINE_FAIL)) ok = false;
                    }
                }
                NDP_FINE(if (fio && fio->dbg && iters == 1) fstamp(*fio, m, 7 + 3 * pass);)
                if (!ok) break;
                vd amin = 1.0;
                vd Ddtl[NSLOT], Ddtu[NSLOT], Ddll[NSLOT], Ddlu[NSLOT];   // this pass's step of the slacks / multipliers (not kept across a sweep)
                for (int s = 0; s < NSLOT; ++s) {
                    vb v = S.valid[s];
                    vd zn = W::ld(lds, S.zoff[s]);
                    vd sl = pass ? vd(sigma_mu) - S.pl[s] : vd(0.0);
                    vd su = pass ? vd(sigma_mu) - S.pu[s] : vd(0.0);
                    vd lo, hi;
                    bounds(S, lds, s, lo, hi);
                    vd dtl = zn - lo - S.tl[s], dtu = hi - zn - S.tu[s];
                    vd rtl = W::rcp(S.tl[s]), rtu = W::rcp(S.tu[s]);
                    vd dll = sl * rtl - S.ll[s] - S.ll[s] * rtl * dtl;
                    vd dlu = su * rtu - S.lu[s] - S.lu[s] * rtu * dtu;
                    dtl = W::sel(v, dtl, vd(0.0)); dtu = W::sel(v, dtu, vd(0.0));
                    dll = W::sel(v, dll, vd(0.0)); dlu = W::sel(v, dlu, vd(0.0));
                    Ddtl[s] = dtl; Ddtu[s] = dtu; Ddll[s] = dll; Ddlu[s] = dlu;
                    // ratio test: lanes whose direction does not shrink the variable see a clamped denominator (-1) and are
                    // discarded by the select, so no reciprocal of zero is formed
                    amin = W::vmin(amin, W::sel(dtl < 0.0, -S.tl[s] * W::rcp(W::vmin(dtl, vd(-1e-300))), vd(1.0)));
                    amin = W::vmin(amin, W::sel(dtu < 0.0, -S.tu[s] * W::rcp(W::vmin(dtu, vd(-1e-300))), vd(1.0)));
                    amin = W::vmin(amin, W::sel(dll < 0.0, -S.ll[s] * W::rcp(W::vmin(dll, vd(-1e-300))), vd(1.0)));
                    amin = W::vmin(amin, W::sel(dlu < 0.0, -S.lu[s] * W::rcp(W::vmin(dlu, vd(-1e-300))), vd(1.0)));
                }
                double alpha = W::wave_min(amin);
                NDP_FINE(if (fio && fio->dbg && iters == 1) fstamp(*fio, m, 8 + 3 * pass);)
                if (pass == 0) {
                    vd acc = 0.0;
                    for (int s = 0; s < NSLOT; ++s) {
                        acc = acc + (S.ll[s] + Ddll[s] * alpha) * (S.tl[s] + Ddtl[s] * alpha)
                                  + (S.lu[s] + Ddlu[s] * alpha) * (S.tu[s] + Ddtu[s] * alpha);
                        S.pl[s] = Ddll[s] * Ddtl[s];
                        S.pu[s] = Ddlu[s] * Ddtu[s];
                    }
                    const double mu_aff = W::wave_sum(acc) * inv2m;
                    const double r = mu_aff / mu;
                    // Mehrotra's centring target, kept from undershooting the tolerance (slacks are formed by subtraction:
                    // their relative accuracy, and the Newton systems', is eps / t) -- same rule as the oracle
                    sigma_mu = fmax_u(r * r * r * mu, Q->mu_floor * Q->tol);
                } else {
                    if (alpha < 1.0) alpha *= Q->tau;
                    for (int t = 0; t < nzx + nzu; t += 64) {   // (ZX,ZU) and (CX,CU) are laid out alike
                        vi i = lane + t;
                        vb p = i < nzx + nzu;
                        vd cur = W::ldp(lds, i + m.CX, p), nw = W::ldp(lds, i + m.ZX, p);
                        W::stp(lds, i + m.CX, cur + (nw - cur) * alpha, p);
                    }
                    vd acc = 0.0;
                    for (int s = 0; s < NSLOT; ++s) {
                        S.tl[s] = S.tl[s] + Ddtl[s] * alpha; S.tu[s] = S.tu[s] + Ddtu[s] * alpha;
                        S.ll[s] = S.ll[s] + Ddll[s] * alpha; S.lu[s] = S.lu[s] + Ddlu[s] * alpha;
                        acc = acc + S.ll[s] * S.tl[s] + S.lu[s] * S.tu[s];
                    }
                    mu = W::wave_sum(acc) * inv2m;
                    rho *= (1.0 - alpha);
                }
            }
            NDP_FINE(if (fio && fio->dbg && iters == 1) fstamp(*fio, m, 12);)
            if (!ok) { status = 4; failed = true; break; }
            if (!(mu == mu)) { status = 1; failed = true; break; }
        }
        iters_out += iters;
        return status;
    }

    static NDP_D double fmax_u(double a, double b) { return a > b ? a : b; }

    // fine-grained stamp: waits until `dep` (if given) is available, then reads the clock (debug path only)
    static NDP_D void fstamp(const RtiIo &io, const LdsMap &m, int idx, vd dep = vd(0.0))
    {
        vd t = W::clock_after(dep);
        W::gst(io.dbg, W::lane() * 0 + (m.total + 16 + idx), t, W::lane() == 0);
    }

    // debug-path phase stamps (shader clock) written behind the LDS image dump; no-op when io.dbg is null
    static NDP_D void stamp(const RtiIo &io, const LdsMap &m, int idx)
    {
        if (NDP_RARELY(io.dbg)) {
            vd t = W::clock();
            W::gst(io.dbg, W::lane() * 0 + (m.total + idx), t, W::lane() == 0);
        }
        if (NDP_RARELY(io.stamps)) {
            vd t = W::clock();
            W::gst(io.stamps, W::lane() * 0 + idx, t, W::lane() == 0);
        }
    }

    // ---------------------------------------------------------------- the control step
    // first-iteration loads, split out so a caller can put independent work between issue and use
    static NDP_D void issue_first(const RtiParams &P, const RtiIo &io, InBuf &inb, vd &x0v)
    {
        vi lane = W::lane();
        x0v = W::gldu(io.x0, W::imin(lane, NX - 1));
        RtiIo g = io;
        g.f_in_lds = 0;                       // decide on f from the pointer only (fused callers pass f = null here)
        issue_inputs(P, g, inb, true);
    }

    static NDP_D void run(const RtiParams &P, const RtiIo &io, lp lds)
    {
        InBuf inb;
        vd x0v;
        issue_first(P, io, inb, x0v);
        run<false>(P, io, lds, inb, x0v);
    }


    // DEFER (producer launch of the work list): when the first QP shows up whose first solve -- with the kept set's pins -- does not
    // settle it, return true at once: nothing of this instance has been written to global memory then, and the consumer launch redoes
    // the step from the same inputs with DEFER = false.  Neither the interior-point code nor the active set's re-solve loop is
    // instantiated: the producer stays the straight-line kernel it was (160 registers; with the loop's back edge 464 and a scratch slot).
    // The five-slot producer (N = 40, 2 RTI iterations: 256 + 20 registers) does not even apply a kept set's pins -- an instance that
    // carries one is handed over before its first sweep (with the pins' code in it the kernel needed a scratch slot).
    // IPM_RARE (the in-place kernels): the interior-point loop is the exception -- see NDP_RARELY.  (The work list's consumer runs
    // nothing else.)
    template <bool DEFER, bool IPM_RARE = false>
    static NDP_D bool run(const RtiParams &P, const RtiIo &io, lp lds, InBuf &inb, vd x0v)
    {
        const int N = horizon(P);
        const LdsMap m = make_map(N);
        const int nzx = (N + 1) * NX, nzu = N * NU;
        stamp(io, m, 0);
        vi lane = W::lane();
        int status = 0, iters = 0;
        typename W::late_t late_prev = W::late_none();
        // the global loads are already in flight (issue_first); the index tables (pure VALU) are built under them
        // fused downwash: f sits in the staging slot already; the slot is recycled by the sweep, so keep a register copy
        vd fkeep[RF];
        for (int t = 0; t < RF; ++t) {
            vi i = lane + 64 * t;
            fkeep[t] = io.f_in_lds ? W::ldp(lds, i + m.TF, i < (N + 1) * 3) : vd(0.0);
        }
        Tables T;
        // (requesting the tables before the caller's MLP phase instead keeps 36 more registers live across it: measured
        // 29.7 us per step against 25.0)
        if (HT) load_tables(io.tables, T);      // in flight while the inputs are committed and the cost / dynamics blocks built
        else build_tables(m, T);
        stamp(io, m, 1);
        const int n_rti = NR ? NR : P.n_rti;
        // QP_AUTO's active set (as_check), kept from the previous control step and across this step's RTI iterations
        const bool as_on = ASET && P.qp_mode == QP_AUTO && P.as_iter_max > 0;
        ActSet A;
        if (as_on) as_issue(P, io, A);       // (in flight under the cost / linearisation phases)
        else for (int t = 0; t < RUA; ++t) A.a[t] = vi(0);
        int sweeps = 0, cond_kept = 0;
        // Several RTI iterations per step at a compile-time count (NR >= 2, config 5): the iteration loop stays a LOOP (its body is
        // 17 k instructions), so inputs requested in front of it and replaced inside it are loop-carried values -- 23 doubles per lane
        // held (in scratch memory, as it turned out) across the first iteration's sweeps and interior-point loop for nothing.  Those
        // kernels request every iteration's inputs at the top of the iteration into a buffer that dies at the commit; the
        // caller's early request (issue_first) is simply not used and disappears.
        constexpr bool LOCAL_IN = NR >= 2;
        for (int it = 0; it < n_rti; ++it) {
            InBuf lbuf;
            if constexpr (LOCAL_IN) {
                RtiIo g = io;
                g.f_in_lds = 0;
                x0v = W::gldu(io.x0, W::imin(lane_it(), NX - 1));
                issue_inputs(P, g, lbuf, it == 0);
            } else if (it > 0) issue_inputs(P, io, inb, false);
            InBuf &ib = LOCAL_IN ? lbuf : inb;
            if (io.f_in_lds)
                for (int t = 0; t < RF; ++t) ib.f[t] = fkeep[t];
            commit_inputs(P, m, ib, lds, it == 0, io);
            if (A_LDS && it == 0) { if (as_on) as_park(P, m, lds, A); else as_clear(P, m, lds, A); }
            stamp(io, m, 2);
            build_cost(P, m, lds);
            stamp(io, m, 3);
            // late force: one lane per stage holds f_k (3 floats); requested here, used after the linearisation
            vd fl[3] = {vd(0.0), vd(0.0), vd(0.0)};
            bool late_ok = true;
            if (io.f_late && P.use_fd) {
                vi k3 = W::imin(lane, N - 1) * 3;
                if (io.late_ready) {
                    for (int i = 0; i < 3; ++i) fl[i] = W::gldfu(io.f_late, k3 + i);
                } else if ((io.late_slow ? (W::count(io.late_slow), true) : true) &&
                           (late_ok = W::wait_ge(io.late_flag, io.late_flag2, io.late_want, io.late_timeout_us))) {
                    for (int i = 0; i < 3; ++i) fl[i] = W::gldf_fresh(io.f_late, k3 + i);
                } else if (io.late_missed) {
                    W::count(io.late_missed);
                }
            }
            linearize(P, m, lds);
            if (io.f_late && P.use_fd && late_ok) {
                // b_k += [h^2/2 f_k/m ; h f_k/m ; 0]  (ndp_nmpc_body_rate_ctl.py:155-157 through one RK4 step: exact, the force is
                // constant over the step); lanes past the last stage repeat stage N-1's correction on a private copy (dump slot)
                W::sync();
                vi kk = W::imin(lane, N - 1);
                vi mb = kk * int(MB_STRIDE) + m.MB + int(MB_B);
                vb mine = lane < N;
                for (int i = 0; i < 3; ++i) {
                    vd a = fl[i] * P.inv_mass;
                    vd bp = W::ld(lds, mb + i), bv = W::ld(lds, mb + 3 + i);
                    W::stp(lds, mb + i, bp + a * (0.5 * P.dt * P.dt), mine);
                    W::stp(lds, mb + 3 + i, bv + a * P.dt, mine);
                }
            }
            if (!late_ok && !status) status = 5;
            if (io.f_late && it == 0) late_prev = W::late_count(io.late_cnt, io.late_group, io.late_group_size);   // (looked at when the step is over)
            stamp(io, m, 4);
            // solve_for_x0: dx_0 = x0 - x_0  (nmpc_body_rate_ctl.py:107)
            W::stp(lds, lane + m.ZX, x0v - W::ldp(lds, lane + m.XI, lane < NX), lane < NX);
            W::sync();
            if (NDP_RARELY(io.dbg && it == 0)) {   // test hook: dump the linearisation + cost blocks
                for (int t = 0; t < m.total; t += 64) {
                    vi i = lane + t;
                    W::gst(io.dbg, i, W::ldp(lds, i, i < m.total), i < m.total);
                }
            }
            bool done = false, failed = false;
            bool pinned = false;                       // does the sweep at hand carry pins?  (then as_apply has moved the defects: their base values are in CX)
            int st = 0;
            if (DEFER || P.qp_mode == QP_AUTO) {
                // Equality-constrained minimiser inside the box => it IS the QP solution (all multipliers 0).  Otherwise, and when the
                // previous step left a set: active-set iterations on the input bounds (as_check) -- each one this same sweep again.
                if (NDP_RARELY(as_on && as_any(P, m, lds, A))) {    // warm start: the kept set's pins
                    if (DEFER && A_LDS) return true;                // (the five-slot producer hands an instance with a kept set straight over: see DEFER)
                    as_apply(P, m, lds, A, pinned);
                }
                const int sweeps0 = sweeps;            // (as_iter_max counts per QP, `sweeps` over the step's RTI iterations)
                for (;;) {
                    stamp(io, m, 5);
                    bool ok = false, condensed = false;
                    if constexpr (COND != 0) {
                        if (!pinned && sweeps == sweeps0) {       // the QP's first solve: condensed, fp32 / bf16 (a failed factorisation: the sweep)
                            condensed = W::template cond_solve<COND>(P, m, lds, N,
                                                                     [&](int r, int c) { return m_entry(m, r, c); }, [&](int r, int c) { return c_entry(m, r, c); });
                            ok = condensed;
                        }
                    }
                    if (!condensed) ok = riccati_sweep(P, m, T, lds, &io);
                    stamp(io, m, 7);
                    ++sweeps;
                    if (NDP_RARELY(!ok)) {
                        // the plain sweep failed: no usable step (as in rounds 1-5); a sweep WITH pins failed: the interior-point loop's case
                        if (!pinned) { st = 4; failed = true; done = true; }
                        break;
                    }
                    if (P.qp_mode != QP_AUTO) break;
                    // (the constraint slots are built only at the interior-point loop's door: keeping ~90 more registers live across the
                    // sweep forces the MFMA accumulators into AGPRs with copies on every dependency)
                    const int verdict = as_check(P, m, lds, A, as_on, pinned);
                    if (verdict == 0) {                            // then the step is the sweep's solution, read where it lies (ZX|ZU)
                        if (COND != 0 && condensed) cond_kept += COND_ACCEPTED;
                        done = true;
                        break;
                    }
                    if constexpr (COND != 0) {
                        if (condensed && verdict != 1) {           // outside the box in a way the active set does not take up: the fp64 solve, then the usual course
                            if (!pinned) as_save_b(P, m, lds);     // (keeps `pinned` = "the base defects are in CX")
                            pinned = true;                         // (marks "not the first solve": no pins exist, as_check's general part handles that)
                            continue;
                        }
                    }
                    if (!DEFER && NDP_RARELY(as_on && verdict == 1 && sweeps - sweeps0 <= P.as_iter_max)) {
                        as_apply(P, m, lds, A, pinned);            // (an update that only releases may leave no pin: the general test handles an empty set too)
                        continue;
                    }
                    break;
                }
            }
            const int zsrc = done ? m.ZX : m.CX;       // ZX|ZU and CX|CU are laid out alike
            if (IPM_RARE ? NDP_RARELY(!done) : !done) {
                if (DEFER) return true;
                if (as_on) {                           // the interior-point loop's answer carries no set: the next step starts cold
                    if (NDP_RARELY(pinned)) as_restore_b(P, m, lds);
                    as_clear(P, m, lds, A);
                    if (io.act) as_store(P, m, lds, io, A);
                }
                if constexpr (NSLOT <= 3) { if (io.ipm_ctr && it == 0) W::count64(io.ipm_ctr); }   // (the five-slot kernels sit at the register limit and have no automatic rule)
                Slots S;
                build_slots(P, m, S);
                load_bounds(m, S, lds);
                st = ipm(P, m, T, S, lds, iters, failed, &io);
            }
            if (st && !status) status = st;
            // full step, no line search; XI|UI and CX|CU are laid out alike.
            // All LDS reads of the step are issued before the first add (one wait, not one per 64 elements); after the
            // last RTI iteration the new iterate goes straight to global memory as well.
            // No usable step (factorisation failure, NaN): the iterate stays as it is and u_0 is read from it.
            {
                constexpr int RZ = RX + RU;     // rounds that cover X|U contiguously: nzx + nzu <= 64 * RZ
                const bool last = it + 1 == n_rti;
                const vi lane = lane_it();      // (shadows the function's: the write-back's global addresses are formed HERE, see lane_it)
                vd xa[RZ], xc[RZ];
                for (int t = 0; t < RZ; ++t) {
                    vi i = W::imin(lane + 64 * t, nzx + nzu - 1);
                    xa[t] = W::ld(lds, i + m.XI);
                    xc[t] = W::ld(lds, i + zsrc);
                }
                vb bad = lane < 0;
                for (int t = 0; t < RZ; ++t) bad = W::bor(bad, !(xc[t] == xc[t]));
                if (NDP_RARELY(!failed && W::any(bad))) {
                    failed = true;
                    if (!status) status = 1;
                }
                for (int t = 0; t < RZ; ++t) {
                    // lanes past the end repeat the last element: identical duplicate stores, no predicates
                    vi i = W::imin(lane + 64 * t, nzx + nzu - 1);
                    vd xn = failed ? xa[t] : xa[t] + xc[t];
                    if (!last) W::st(lds, i + m.XI, xn);   // the next RTI iteration linearises there; after the last one the LDS image dies
                    if (last) {
                        // X and U are separate global arrays: element i < nzx goes to X[i], else to U[i - nzx]
                        W::gst2(io.X, io.U, i, nzx, xn);
                        if (io.Xm) W::gst2(io.Xm, io.Um, i, nzx, xn);   // wave-uniform: the host step's mirror of the iterate
                        if (t == (nzx >> 6)) {   // the round that holds u_0 = U[0..3] (wave-uniform test)
                            vb pu = (i >= nzx) && (i < nzx + NU);
                            W::gst(io.u0, i - nzx, xn, pu);
                            if (NDP_RARELY(io.cmd != nullptr)) W::cmd_store(io.cmd, io.thrust_keep, io.kthr, io.cmd_mass, i - nzx, xn, pu);
                            if (W::any(pu && !(xn == xn))) status = 1;
                        }
                        if (nzx + NU > 64 * ((nzx >> 6) + 1) && t == (nzx >> 6) + 1) {   // u_0 straddles two rounds
                            vb pu = (i >= nzx) && (i < nzx + NU);
                            W::gst(io.u0, i - nzx, xn, pu);
                            if (NDP_RARELY(io.cmd != nullptr)) W::cmd_store(io.cmd, io.thrust_keep, io.kthr, io.cmd_mass, i - nzx, xn, pu);
                            if (W::any(pu && !(xn == xn))) status = 1;
                        }
                    }
                }
            }
            W::sync();
        }
        stamp(io, m, 8);
        W::gsti(io.status, status);
        W::gsti(io.iters, iters + ((sweeps + cond_kept) << int(ITERS_SWEEP_SHIFT)));
        // the kept set changed <=> a QP took more than its one sweep (a set that holds reproduces itself in the first one)
        if (NDP_RARELY(as_on && io.act && sweeps != n_rti)) as_store(P, m, lds, io, A);
        if (io.f_late) W::late_publish(late_prev, io.late_gsize, io.late_done_word);
        return false;
    }
};

// ---------------------------------------------------------------- host-built index tables
// One-lane scalar backend: runs build_tables for a single lane on the host so that the serialised tables are
// produced by the very code the in-kernel path (HT = false, used by the emulator tests) executes.
struct LaneW {
    using vd = double;
    using vi = int;
    using vb = bool;
    struct vd4 { double r[4]; };
    using md = double;
    using md4 = vd4;
    static constexpr bool packed_k = false, delta_ok = true, has_mma4 = true;
    using lds_ptr = double *;
    // host-only on purpose (no device attribute): only fill_tables, a host function, instantiates code that calls them
    static int &cur() { static thread_local int l = 0; return l; }
    static int &layout() { static thread_local int l = 0; return l; }   // 0: f64 instruction, 1: f32 / bf16 instructions (see lane_preds)
    static vi lane() { return cur(); }
    static vi lcol(vi lane) { const int jt = lane & 15; return layout() ? (jt >> 2) + 4 * (jt & 3) : jt; }
    static vd sel(vb p, vd a, vd b) { return p ? a : b; }
    static vi sel(vb p, vi a, vi b) { return p ? a : b; }
};

enum { TB_WORDS = 36 * 64 };

inline void fill_tables(int N, int *out /* [TB_WORDS] */, int layout = 0)
{
    using Prog = RtiWave<LaneW, 1>;
    const LdsMap m = make_map(N);
    LaneW::layout() = layout;
    for (int i = 0; i < TB_WORDS; ++i) out[i] = 0;
    for (int lane = 0; lane < 64; ++lane) {
        LaneW::cur() = lane;
        typename Prog::Tables T;
        Prog::build_tables(m, T);
        Prog::for_each_int(T, [&](int i, int &fld) { out[Prog::tb_word(i, lane)] = fld; });
    }
}

}  // namespace ndp
