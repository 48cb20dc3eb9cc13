// ndp_hip.hip -- gfx950 kernels + the C-ABI of include/ndp_nmpc.h.
//
// Kernels
//   rti_kernel  : one wavefront per OCP instance runs the whole SQP-RTI step (rti_wave.hpp) out of its
//                 LDS slice; 4 waves (= 4 instances) per 256-thread workgroup, one per SIMD of a CU.
//                 FUSED: the downwash MLP tile (mlp_tile) runs in front of linearise inside the same launch.
//                 QMODE 1 / 2: producer / consumer of the work list (instances whose QP needs the interior point).
//   mlp_kernel  : DownwashNN.update + r_horiz gate for all (instance, horizon row) pairs; the four
//                 layers are chained through v_mfma_f32_32x32x2_f32 accumulators (no LDS round trip).
//   mlp_stream_kernel + prefetch_gate_kernel / prefetch_done_kernel : the downwash of the NEXT tick on a second stream, LDS-free
//                 (weights out of L2), ordered against the control step by per-tile epochs (LateArgs / PF_* words).
//   peer_publish_kernel + peer_epoch_kernel : the per-tick neighbour exchange through peer-mapped windows (peer_epoch.hpp).
//   ref_window_kernel, ref_list_*_kernel, throttle / actuator / plant kernels : the rows either side of the step (f1, f3, f4).
// The iterate (X, U) of a handle lives in HBM and stays there between steps; host-array steps (ndp_step / ndp_step_begin)
// read their inputs from, and mirror their outputs to, page-locked host slots over PCIe (zero-copy) -- see step_begin_locked.
// There is no CPU fallback: every entry point fails (<0) if HIP is unusable.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <dlfcn.h>
#include <sched.h>
#include <stdio.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "wave_gfx950.hpp"   // defines the device qualifiers, must precede rti_wave.hpp
#include "cfg_params.hpp"
#include "cond_qp.hpp"
#define NDP_PEER_FN __host__ __device__ inline
#include "peer_epoch.hpp"

namespace ndp {

// Individually rounded multiply / add.  hipcc contracts a*b+c into an FMA by default, and HIP's __dmul_rn/__dadd_rn
// are plain operators that get re-fused; the reference evaluates the gate, the Kalman filter and the alpha filter
// in Python/numpy doubles without fusion, so these few expressions are built from non-contractable operations.
__device__ __forceinline__ double nc_mul(double a, double b)
{
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ double nc_add(double a, double b)
{
#pragma clang fp contract(off)
    return a + b;
}

// ------------------------------------------------------------------------------------------ RTI kernel
struct BatchPtrs {
    const double *kc;
    const int *tables;      // host-built index tables (fill_tables)
    const double *x0, *xr, *ur;
    const float *f;
    double *X, *U, *u0;
    int *status, *iters;
    double *Xm, *Um;        // mirror of the new iterate ([B][N+1][10] | [B][N][4], page-locked host memory) or null
    double *dbg;
    double *stamps;         // [B][NDP_NSTAMP] phase stamps of every instance (ndp_debug_stamps), or null
    size_t xr_pitch, ur_pitch;   // doubles from one instance's reference window to the next: (N+1) 10 / 4 N for dense [B][N+1][10] / [B][N][4]
                                 // arrays; RingGeom::px / pu when the windows are read straight out of the reference list (ndp_tick)
    size_t x0_pitch;             // doubles from one instance's x0 to the next (10)
    // ndp_tick: the actuator command written beside u0 (RtiIo::cmd): cmd[B][4], k_throttle[B] (the estimator's state row), the thrust
    // kept for the next estimator update [B]; null cmd = a plain control step
    double *cmd;
    const double *kthr;
    double *thrust_keep;
    double cmd_mass;
    int f_f64;                   // 1: f holds doubles, [B][N+1][3] (ndp_step_ex_f64)
    signed char *act;            // [B][act_pitch(N)] the instances' kept active sets (RtiIo::act), or null: no warm start of the QP's active set
};

struct MlpArgs {            // fused downwash (null frag = not fused)
    const float *frag;
    const double *other;    // neighbour windows: row (instance) r starts at other + r * (N+1) * other_stride, node k at + k * other_stride
    const double *ego_xy;   // [B][2] or null (gate always open)
    float *force_out;       // [B][N+1][3] copy of the predicted force for callers
    double r2;
    int other_stride;       // doubles per node of `other`: 10 (a full reference window) or 6 (positions + velocities only, what the MLP reads)
    const int *other_index; // [B] row of `other` that holds instance i's neighbour (multi-GPU: a row of the gathered buffer);
                            // < 0 = no neighbour (force 0: the plain NMPC followers of a formation); null = row i
    int other_sys;          // 1: `other` is another process's / GPU's memory mapped through ndp_peer_open -- read it with system-scope loads
    size_t other_pitch;     // doubles from one row of `other` to the next: (N+1) other_stride when dense; RingGeom::px for windows in the list
    size_t ego_pitch;       // doubles from one instance's ego xy to the next: 2 ([B][2]), or 10 when the gate reads the odometry rows x0[B][10]
};

// Neighbour windows that live in ANOTHER agent's memory (peer windows over xGMI) are read with system-scope loads: such lines are
// not kept coherent in this GPU's L2s, and what a kernel boundary invalidates depends on the fence scope the runtime put on the
// dispatch packet (agent scope between back-to-back launches of one queue).  A system-scope load always fetches from the owner's
// memory -- two 8-byte loads per lane and launch instead of one 16-byte load, nothing else changes.  Local windows: plain loads.
typedef double ndp_d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double ld_other(const double *p, int sys)
{
    return sys ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : *p;
}
__device__ __forceinline__ ndp_d2 ld_other2(const double *p, int sys)
{
    if (sys) {
        ndp_d2 r;
        r[0] = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        r[1] = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return r;
    }
    return *(const ndp_d2 *)p;
}

// Work list of instances whose QP needs the interior-point loop (batches with more instances than SIMDs).  An
// interior-point solve costs ~18 Riccati sweeps against 1 for the early exit, so with several instances per SIMD one such
// instance per workgroup leaves the other three SIMDs of its CU idle for most of the launch.  Instead the step is split:
//   producer launch  (QMODE 1): every wave runs the cheap part of its own instance; an instance whose equality-constrained
//                               minimiser is not inside the box is appended to the list (one atomic per such instance)
//                               and NOT touched otherwise;
//   consumer launch  (QMODE 2): wave j solves list entry j from scratch with the interior-point loop; waves past the end
//                               of the list exit at once -- the listed instances are spread evenly over all SIMDs.
// The list counter is zeroed by a one-wave launch behind the consumer (queue_reset_kernel).  (Round 3 first let the consumer do
// it -- every consumer workgroup counted itself with an atomic, the last one reset -- and paid for it: agent-scope atomics on one
// address are served at the memory side at 30-60 ns each and serialise, 61 us at batch 4096 where the memset node it replaced had
// cost 4.6 us.)  (An in-kernel queue -- finished waves popping
// other instances' solves -- was built first: as a second inlined copy of the unrolled step it wrecked the register
// allocation of both copies, as a called function it lost the scalar registers; either way 2.3x slower than this.)
struct QueueArgs {
    unsigned *count;        // entries of ids
    int *ids;               // [B]
    unsigned long long *ipm_total;   // [0] monotonic: instances that needed the interior-point loop (in place: counted by the kernel; work
                                     // list: added up by the reset launch); [1] monotonic: control steps executed (one count per launch) --
                                     // the pair the handle's automatic work-list rule looks at (queue_policy)
};

typedef __attribute__((address_space(3))) float *lds_f32;
typedef const __attribute__((address_space(3))) float *lds_cf32;
__device__ __forceinline__ void stage_fragments(const float *__restrict__ fr, lds_f32 dst, int tid, int nthreads);
__device__ __forceinline__ void mlp_tile(lds_cf32 fr, const float zb[3], int lane, float o[3]);
__device__ __forceinline__ bool gate_open(const double *other_inst, const double *ego_xy_inst, double r2);

__device__ __forceinline__ void bind_instance(RtiIo &io, const BatchPtrs &bp, int inst, int N)
{
    const size_t nx = (size_t)(N + 1) * NX, nu = (size_t)N * NU, nf = (size_t)(N + 1) * 3;
    io.x0 = bp.x0 + (size_t)inst * bp.x0_pitch;
    io.xr = bp.xr + inst * bp.xr_pitch;
    io.ur = bp.ur + inst * bp.ur_pitch;
    io.f = bp.f ? bp.f + inst * nf * (bp.f_f64 ? 2 : 1) : nullptr;
    io.f_is_f64 = bp.f_f64;
    io.X = bp.X + inst * nx;
    io.U = bp.U + inst * nu;
    io.Xm = bp.Xm ? bp.Xm + inst * nx : nullptr;
    io.Um = bp.Xm ? bp.Um + inst * nu : nullptr;
    io.u0 = bp.u0 + (size_t)inst * NU;
    io.status = bp.status + inst;
    io.iters = bp.iters + inst;
    io.dbg = bp.dbg;
    io.f_in_lds = 0;
    io.kc = bp.kc;
    io.tables = bp.tables;
    io.stamps = bp.stamps ? bp.stamps + (size_t)inst * NDP_NSTAMP : nullptr;
    io.act = bp.act ? bp.act + (size_t)inst * act_pitch(N) : nullptr;
    if (NDP_RARELY(bp.cmd != nullptr)) {
        io.cmd = bp.cmd + (size_t)inst * NU;
        io.thrust_keep = bp.thrust_keep + inst;
        io.kthr = bp.kthr[inst];            // (requested here, used at the step's very end)
        io.cmd_mass = bp.cmd_mass;
    }
}

// Downwash predicted one tick ahead by mlp_stream_kernel on a second stream (ndp_downwash_prefetch_device), consumed by the
// control-step launch of the tick (ndp_step_device_prefetched).  Two chains of launches that order themselves on the device:
//   second stream : prefetch_gate_kernel (one wave: number m = previous + 1; waits until control step m - 2, the last reader of
//                   force slot m & 1, holds its values; publishes m in PF_CUR_M) -> mlp_stream_kernel (reads m with a plain load --
//                   it was written by the launch before it in its own stream; every wave writes its 32 rows of slot m & 1 with
//                   write-through stores and then its tile's epoch word := m)
//   main stream   : control step t = (completed control-step groups) / groups + 1 (plain load: only control steps, in this
//                   stream, advance it); waits late (after its cost phase) for the one or two tile epochs that cover its rows
//                   to reach t, loads its forces past the L2, and counts itself done-reading (WaveGfx950::late_count):
//                   PF_RTI_C1 + g  workgroups counted into group g = workgroup index mod groups, PF_RTI_C2 groups completed --
//                   launch t has read its slot completely at t * groups.
// The usual case costs the control step nothing at agent scope: prefetch_done_kernel, behind every downwash launch in its stream,
// publishes PF_MLP_DONE = m; a control step that finds PF_MLP_DONE >= t when it STARTS (plain load, fresh after the launch
// boundary) knows its slot was in memory before it began and reads it with ordinary cached loads.  Only a control step that
// started before its prediction was complete takes the epoch path.
// No word is shared by many waves at agent scope: the eight XCDs' L2s are not coherent with each other, agent-scope loads and
// atomics are served at the memory side and serialise per address (30-60 ns each: 1024 waves on one flag word cost 7 us per wave,
// one counting atomic per wave 18 us per launch).  Every word has its own 4 KB (PF_STRIDE words: its own memory channel).
// Nothing is baked into a launch, so captured launches replay correctly.  PF_MISSED: control-step waves whose wait timed out
// (zero force, status 5); PF_GATE_TIMEOUT: gate waits that timed out.
enum { PF_GROUPS = 8, PF_STRIDE = 512, PF_CUR_M = 0, PF_RTI_C1 = 1 * PF_STRIDE, PF_RTI_C2 = 9 * PF_STRIDE,
       PF_MISSED = 10 * PF_STRIDE, PF_GATE_TIMEOUT = 11 * PF_STRIDE, PF_MLP_DONE = 12 * PF_STRIDE, PF_SLOW = 13 * PF_STRIDE, PF_EPOCH = 14 * PF_STRIDE /* [2][ntiles] */ };
struct LateArgs {
    unsigned long long *proto;     // null = not a prefetched-force launch
    const float *F[2];             // the two force slots, [B][N+1][3] each
    unsigned timeout_us;
    unsigned groups_rti, ntiles;
};
__device__ __host__ inline unsigned pf_group_size(unsigned n, unsigned groups, unsigned g) { return n / groups + (g < n % groups ? 1u : 0u); }

struct ThrCfg { double a1, a2, hm, g, R, Q0, Q1, mass; };

// layout of the reference list in HBM (see the f1 list kernels below)
struct RingGeom {
    int step, np1;                 // list entries per node spacing; N + 1
    __host__ __device__ int ring() const { return step * (np1 - 1) + 1; }
    __host__ __device__ size_t px() const { return (size_t)step * 2 * np1 * 10; }     // doubles per vehicle, x ring
    __host__ __device__ size_t pu() const { return (size_t)step * 2 * np1 * 4; }
    __host__ __device__ size_t slot(unsigned long long j) const { return (size_t)(j % step) * 2 * np1 + (size_t)((j / step) % np1); }
};

// ndp_tick in ONE launch (rti_kernel<..., TICK = true>): what tick_pre_kernel does -- the reference list's newest entry, which is node
// N of this tick's window, and the hover-throttle estimator's update -- done by the control step's own wave in front of its work, so
// that a control tick is a single dispatch.  (As a launch of its own that part cost 7-8.5 us + a 4.5 us gap per tick in the kernel
// trace against 24.8 us for the control step: a third of the tick for 112 bytes per vehicle.)
enum { SEGC_SLOT = 32, SEGC_PER = 72 };      // doubles per slot / per vehicle of the tick's segment cache (tick_early)
struct TickArgs {
    const double *coeff, *tcum, *tseg, *fpt;   // the trajectories (ndp_ref_set_trajectory)
    const double *segc;                        // [B][SEGC_PER] the vehicles' current / next segment records (see tick_early): the copy this launch READS
    double *segc_wr;                           // ... and the copy it WRITES (every vehicle's record, re-filled or carried over): the next tick's `segc`
    int n_seg;
    const double *t;                           // [B] trajectory time of the tick, or null: t_all for every vehicle
    double t_all;
    int advance;                               // 0: the list is not advanced in this tick
    double toff, mass, g;                      // T_horizon; flatness constants
    unsigned long long j_new;                  // absolute index of the list entry the new point becomes
    size_t new_slot;                           // rg.slot(j_new), from the host (two 64-bit divisions otherwise, in front of the barrier)
    RingGeom rg;
    double *rx, *ru;
    ThrCfg thr;                                // estimator
    double *st;
    const double *vz;
    size_t vz_pitch;
    const double *throttle;
    int est;
};
// what tick_new_point works on, all of it requested at the kernel's very top (tick_early): the time, and the lane's share of the
// vehicle's two cached segment records (see tick_early)
typedef double tick_d2 __attribute__((ext_vector_type(2)));
// h0 / h1: (time_cum[i], time_cum[i + 1]), (time_seg[i], i) of slot 0 / 1; ca / cn: the lane's 8 coefficients in slot 0 / 1; tf: (end of the
// trajectory, the lane's component of final_pt)
struct TickEarly { double tv; tick_d2 h0[2], h1[2], ca[4], cn[4], tf; int v; double own, ownc; };   // own / ownc: word `lane` / constant `lane` of the EGO's record (carried over)
__device__ __forceinline__ TickEarly tick_early(const TickArgs &ta, int inst, int orow, int lane);
__device__ __forceinline__ double tick_new_point(const TickArgs &ta, const TickEarly &te, int inst, int lane, bool store, double xv[10],
                                                 double uv[4], double nbv[6], int &refill, double &cfill, double *stamps);
__device__ __forceinline__ void tick_arrived(TickEarly &te);
__device__ __forceinline__ void tick_cache_store(const TickArgs &ta, const TickEarly &te, int inst, int lane, int refill, double fill, double cfill);
__device__ __forceinline__ double tick_estimator(const TickArgs &ta, int inst, int B, int lane);

struct KernArgs {
    RtiParams P;
    BatchPtrs bp;
    int B, lds_per_wave;
    MlpArgs ma;
    QueueArgs qa;
    LateArgs la;
    TickArgs ta;            // read by the TICK instantiations only
};

// Arguments a kernel needs LATE (the tick's estimator constants, the list's geometry for the new entry's store, the trajectory arrays
// of the rare slow path), fetched where they are used.  Read as ordinary members of `ka` the compiler requests every argument at the
// kernel's top and keeps it in scalar registers until its use: the tick kernels ran out of them (160 spills to / 740 reloads from
// vector-register lanes against 19 / 82 in the plain step -- with one wave per SIMD every one of those is time on the clock).  The
// pointer goes through an empty asm so that the loads cannot move up; they hit the scalar cache (the block was touched at the top).
struct KernargLate {
    const __attribute__((address_space(4))) unsigned *kp;
    __device__ __forceinline__ KernargLate()
    {
        kp = (const __attribute__((address_space(4))) unsigned *)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(kp));
    }
    template <class T> __device__ __forceinline__ T get(unsigned off) const
    {
        static_assert(sizeof(T) % 4 == 0 && std::is_trivially_copyable<T>::value, "plain words only");
        unsigned w[sizeof(T) / 4];
#pragma unroll
        for (unsigned i = 0; i < sizeof(T) / 4; ++i) w[i] = kp[off / 4 + i];
        T t;
        __builtin_memcpy(&t, w, sizeof(T));
        return t;
    }
};
static_assert(offsetof(KernArgs, P) == 0, "WaveGfx950::late_params reads the parameter block at the start of the argument segment");
static_assert(std::is_standard_layout<KernArgs>::value && std::is_trivially_copyable<KernArgs>::value,
              "KernargLate addresses members of the one kernel argument by offsetof");
#define NDP_TA_LATE(L, f) ((L).template get<decltype(TickArgs::f)>((unsigned)(offsetof(KernArgs, ta) + offsetof(TickArgs, f))))
// (A pointer fetched this way has lost what the compiler knows of pointers in the argument block -- that they point to global memory --
// and is dereferenced with FLAT instructions, which also count as LDS operations and turn the waits behind them into full drains:
// fine on the rare paths; the common one goes through gptr.)
template <class T> using gptr = __attribute__((address_space(1))) T *;

// FUSED: the wave first predicts its own instance's disturbance force (gate + MLP over the N+1 <= 32 horizon rows,
// one 32x32 f32 MFMA tile) and leaves it in the LDS staging slot the RTI program reads f from -- no second
// launch and no trip of f through HBM.
// QMODE: 0 = the whole step in place; 1 / 2 = producer / consumer of the interior-point work list (see QueueArgs); 3 = in place, the
// lean program (RtiWave's LEAN: no stiff sweeps) -- the late-force step that shares the SIMDs with the next tick's downwash launch.
#ifndef NDP_RTI_ATTR       // kernel-development hook: extra attributes of rti_kernel (e.g. a register cap for occupancy studies)
#define NDP_RTI_ATTR
#endif
// TICK: the launch is a whole control tick of ndp_tick (see TickArgs): the wave makes its window's newest node -- and its neighbour's --
// itself and runs the estimator; instantiated for the reference configuration's in-place and producer forms only.
template <int NSLOT, int WAVES, bool FUSED, int NC = 0, int PREC = 0, int NRC = (NC ? 1 : 0), int QMODE = 0, bool TICK = false>
__global__ __launch_bounds__(64 * WAVES) NDP_RTI_ATTR void rti_kernel(KernArgs ka)
{
    static_assert(!(FUSED && QMODE == 2), "the consumer reads the force the producer left in global memory");
    static_assert(!TICK || (QMODE <= 1 && NC > 0 && PREC == 0), "the one-launch tick exists for the compile-time horizon's in-place and producer forms");
    extern __shared__ __attribute__((aligned(16))) double smem[];
#ifndef NDP_NO_KERNARG_WARM
    {   // The argument block is ~1.2 KB = 19 scalar-cache lines, and the compiler fetches each field next to its first use, waiting for it
        // there: every first touch of a line is a memory round trip of its own, one behind the other through the whole prologue.
        // One dword of every line, requested together at the very top: one round trip, the later fetches hit the scalar cache.
        typedef const __attribute__((address_space(4))) unsigned *kptr;
        kptr kp = (kptr)__builtin_amdgcn_kernarg_segment_ptr();
        unsigned acc = 0;
#pragma unroll
        for (unsigned o = 0; o < (unsigned)sizeof(KernArgs); o += 64) acc |= kp[o / 4];
        asm volatile("" : : "s"(acc));
    }
#endif
    const RtiParams &P = ka.P;
    const BatchPtrs &bp = ka.bp;
    const MlpArgs &ma = ka.ma;
    const QueueArgs &qa = ka.qa;
    const int B = ka.B;
    const int wave = (int)(threadIdx.x >> 6);
    __shared__ unsigned wg_done;     // prefetched-force launches: the workgroup's waves that hold their force values (see LateArgs)
    if (!FUSED && (QMODE == 0 || QMODE == 3) && ka.la.proto) {
        if (threadIdx.x == 0) wg_done = 0;
        __syncthreads();             // (before any wave of a ragged last workgroup leaves)
    }
    int inst_raw = __builtin_amdgcn_readfirstlane((int)blockIdx.x * WAVES + wave);
    if (QMODE == 2) {         // list entry -> instance; past the end of the list: nothing to do
        const int n = (int)*qa.count;
        if (inst_raw >= n) return;
        inst_raw = __builtin_amdgcn_readfirstlane(qa.ids[inst_raw]);
    }
    const bool active = inst_raw < B;
    if (!FUSED && !active) return;
    const int inst = active ? inst_raw : B - 1;   // fused: idle waves of the last workgroup still take part in the barriers
    const int N = NC ? NC : P.N;
    const size_t nf = (size_t)(N + 1) * 3;
    RtiIo io;
    bind_instance(io, bp, inst, N);
    if (NSLOT <= 3 && (QMODE == 0 || QMODE == 3)) io.ipm_ctr = qa.ipm_total;
    if (NSLOT <= 3 && QMODE != 2 && blockIdx.x == 0 && threadIdx.x == 0 && qa.ipm_total)
        __hip_atomic_fetch_add(qa.ipm_total + 1, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!FUSED && (QMODE == 0 || QMODE == 3) && ka.la.proto) {
        // this launch is control step number t; its force was written into slot t & 1 by downwash launch t
        const LateArgs &la = ka.la;
        const unsigned long long t = la.proto[PF_RTI_C2] / la.groups_rti + 1;       // (plain load: see LateArgs)
        const unsigned g = blockIdx.x % la.groups_rti;
        const unsigned np1 = (unsigned)N + 1, row0 = (unsigned)inst * np1;
        io.f_late = la.F[t & 1] + (size_t)inst * nf;
        io.late_flag = la.proto + PF_EPOCH + (t & 1) * la.ntiles + row0 / 32;
        io.late_flag2 = la.proto + PF_EPOCH + (t & 1) * la.ntiles + (row0 + np1 - 1) / 32;
        io.late_want = t;
        io.late_ready = la.proto[PF_MLP_DONE] >= t ? 1 : 0;                         // (plain load)
        io.late_timeout_us = la.timeout_us;
        io.late_missed = reinterpret_cast<int *>(la.proto + PF_MISSED);
        io.late_slow = reinterpret_cast<int *>(la.proto + PF_SLOW);
        io.late_cnt = reinterpret_cast<unsigned *>(la.proto + PF_RTI_C1 + PF_STRIDE * g);
        io.late_done_word = la.proto + PF_RTI_C2;
        io.late_gsize = pf_group_size(gridDim.x, la.groups_rti, g);
        // (LDS offset + 1: the word may well sit at offset 0, and null means "no workgroup-level counter")
        io.late_group = (void *)((size_t)(unsigned)(size_t)(__attribute__((address_space(3))) unsigned *)&wg_done + 1);
        const int left = B - (int)blockIdx.x * WAVES;
        io.late_group_size = (unsigned)(left < WAVES ? left : WAVES);
    }
    const int lpw = NC ? ((lds_doubles(NC) + 1) & ~1) : ka.lds_per_wave;
    WaveGfx950::lds_ptr lds = (WaveGfx950::lds_ptr)(smem + (size_t)wave * lpw);
    // PREC 0: the product path (f64 matrix instruction); 1 / 2: operand-rounding studies on it; 3 / 4: the sweeps on the real
    // fp32 / bf16-input matrix instructions (BASELINE config 5)
    using WB = std::conditional_t<PREC == 3, WaveGfx950F32, std::conditional_t<PREC == 4, WaveGfx950BF16, WaveGfx950>>;
    // PREC 5 / 6: config 5's CONDENSED study (cond_qp.hpp) -- the f64 program with every QP's first solve in condensed form on the fp32 / bf16 instructions
    using Prog = RtiWave<WB, NSLOT, NC, true, NRC, (PREC >= 3 ? 0 : PREC), QMODE == 3, (PREC == 5 ? 1 : (PREC == 6 ? 2 : 0))>;   // compile-time horizon and iteration count (NC = 0: both at run time)
    if (NDP_RARELY(io.stamps && (threadIdx.x & 63u) == 0)) {    // profiling hook: real time (100 MHz) and shader clock at entry -> the clock the launch ran at
        io.stamps[12] = (double)__builtin_amdgcn_s_memrealtime();
        io.stamps[14] = (double)__builtin_amdgcn_s_memtime();
    }
    // fused, neighbour rows picked through other_index: the row number is the head of a dependent load chain (index -> window
    // address -> window loads).  Fetch it before anything else is in the wave's in-order load queue and consume it here, so
    // that the one unavoidable wait covers one load, not the seventeen input loads requested next.
    // wg_nb: does ANY instance of this workgroup have a neighbour?  (Four scalar loads of the workgroup's own index entries, the
    // same in every wave: no barrier.)  A workgroup of plain NMPC followers -- a rank's local order puts them behind its leaders,
    // dist.config4_gids -- then skips the 70 KB weight transfer, both barriers and the network's input loads altogether.
    int orow = inst;
    bool wg_nb = true;
    if (FUSED && ma.other_index) {
        int any = 0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            const int iw = (int)blockIdx.x * WAVES + w;
            any |= (ma.other_index[iw < B ? iw : B - 1] >= 0) ? 1 : 0;
        }
        orow = __builtin_amdgcn_readfirstlane(ma.other_index[inst]);
        asm volatile("" : : "s"(orow));
        wg_nb = __builtin_amdgcn_readfirstlane(any) != 0;
    }
    const bool advance = TICK && ka.ta.advance != 0;
    TickEarly te;                            // (left as it is without `advance`: nothing looks at it then)
    if (TICK && NDP_RARELY(io.stamps && (threadIdx.x & 63u) == 0)) io.stamps[16] = (double)__builtin_amdgcn_s_memtime();   // neighbour index known
    if (TICK && advance) {
        te = tick_early(ka.ta, inst, FUSED && wg_nb ? orow : -1, (int)(threadIdx.x & 63u));
        __builtin_amdgcn_sched_barrier(0);   // (these two loads lead the wave's in-order load queue)
    }
    if (TICK && NDP_RARELY(io.stamps && (threadIdx.x & 63u) == 0)) io.stamps[22] = (double)__builtin_amdgcn_s_memtime();   // cache loads issued
    typename Prog::InBuf inb;
    double x0v;
    Prog::issue_first(P, io, inb, x0v);      // every global input of the RTI step is now in flight (hidden under the MLP when fused)
    __builtin_amdgcn_sched_barrier(0);       // do not let the scheduler sink those loads behind the MLP
    if (TICK && NDP_RARELY(io.stamps && (threadIdx.x & 63u) == 0)) io.stamps[23] = (double)__builtin_amdgcn_s_memtime();   // input loads issued
    // TICK: the newest list entry of this vehicle (x_new / u_new) and the position / velocity part of the neighbour's (nb_new).  Row N of
    // both windows is NOT read from the list in this launch (the neighbour's wave writes its entry while this one runs): the ego's goes
    // into the staged window through RtiIo::xrN, the pair into the network's input below.
    double x_new[10], u_new[4], nb_new[6], seg_fill = 0.0, seg_cfill = 0.0;
    int seg_refill = 0;
    if (TICK && !(FUSED && wg_nb) && advance) {      // (fused with neighbours: made below, under the weight transfer)
        tick_arrived(te);
        seg_fill = tick_new_point(ka.ta, te, inst, (int)(threadIdx.x & 63u), active, x_new, u_new, nb_new, seg_refill, seg_cfill, io.stamps);
#pragma unroll
        for (int i = 0; i < 10; ++i) io.xrN[i] = x_new[i];
        io.have_xrN = 1;
    }
    if (FUSED && !wg_nb) {          // no instance of the workgroup has a neighbour: zero force, nothing of the network runs
        if (!active) return;
        const int lane = (int)(threadIdx.x & 63u);
        const LdsMap m = make_map(N);
        if (lane < 3 * (N + 1)) {
            lds[m.TF + lane] = 0.0;
            ma.force_out[inst * nf + lane] = 0.0f;
        }
        if (3 * (N + 1) > 64 && lane + 64 < 3 * (N + 1)) {
            lds[m.TF + lane + 64] = 0.0;
            ma.force_out[inst * nf + lane + 64] = 0.0f;
        }
        WaveGfx950::sync();
        io.f = nullptr;
        io.f_in_lds = 1;
    } else if (FUSED) {
        const int lane = (int)(threadIdx.x & 63u), j = lane & 31, h = lane >> 5;
        const int np1 = N + 1;
        const int st = ma.other_stride;
        const double *oth = ma.other + (size_t)(orow < 0 ? 0 : orow) * ma.other_pitch;
        // the gate's four numbers are only REQUESTED here; the comparison comes after the barrier (consuming them here would
        // park the wave on the whole in-order load queue -- s_waitcnt vmcnt(0) -- before the weight transfer is even issued)
        const double *exy = ma.ego_xy ? ma.ego_xy + (size_t)inst * ma.ego_pitch : oth;
        const int osys = ma.other_sys;
        const double g_ox = ld_other(oth, osys), g_oy = ld_other(oth + 1, osys), g_ex = exy[0], g_ey = exy[1];
        const int jr = j < np1 ? j : np1 - 1;
        float zb[3], o[3];
        // Network input (downwash_nn.py:22-23): columns 0..5 of (other - ego reference), rows 0..N, subtracted in fp64.  Lane
        // (j, h) of the tile wants row j, columns 2s + h -- read that way it is an 8-byte load at an 80-byte lane stride (ten
        // cache lines per quarter wave, six instructions).  Instead ONE 16-byte load per array covers a row's six columns with
        // three adjacent lanes (lane l: row l / 3, columns 2 (l % 3), 2 (l % 3) + 1: 63 lanes for N = 20), and the tile's
        // layout is made by a cross-lane gather of the fp32 differences (ds_bpermute: the LDS crossbar, no LDS storage).
        typedef double d2_t __attribute__((ext_vector_type(2)));
        constexpr int ZR = NC ? (3 * (NC + 1) + 63) / 64 : 2;      // load rounds: 3 (N+1) lanes, N + 1 <= 32
        d2_t dv[ZR], ev[ZR];
#pragma unroll
        for (int t = 0; t < ZR; ++t) {
            const int l3 = lane + 64 * t, r3 = l3 / 3, c3 = l3 - 3 * r3, rc = r3 < np1 ? r3 : np1 - 1;
            dv[t] = ld_other2(oth + (size_t)rc * st + 2 * c3, osys);
            ev[t] = *(const d2_t *)(io.xr + (size_t)rc * NX + 2 * c3);
        }
        const LdsMap m = make_map(N);
        if (NDP_RARELY(io.dbg && lane == 0)) io.dbg[m.total + 9] = (double)__builtin_amdgcn_s_memtime();
        if (NDP_RARELY(io.stamps && lane == 0)) io.stamps[9] = (double)__builtin_amdgcn_s_memtime();
        // the whole workgroup's LDS is still unused: park the weight fragments there for the MLP phase
        lds_f32 wl = (lds_f32)smem;
        if (TICK) tick_arrived(te);               // (see there: the one wait of the prologue, in FRONT of the weight transfer; not under
                                                  // `advance`: a path around it leaves the values pending in the compiler's books)
        stage_fragments(ma.frag, wl, (int)threadIdx.x, 64 * WAVES);
        if (TICK && advance) {                    // the polynomial work runs while the weights stream into LDS
            seg_fill = tick_new_point(ka.ta, te, inst, lane, active, x_new, u_new, nb_new, seg_refill, seg_cfill, io.stamps);
#pragma unroll
            for (int i = 0; i < 10; ++i) io.xrN[i] = x_new[i];
            io.have_xrN = 1;
#pragma unroll
            for (int t = 0; t < ZR; ++t) {        // row N of the network's input: the two new entries (downwash_nn.py:22: columns 0..5)
                const int l3 = lane + 64 * t, r3 = l3 / 3, c3 = l3 - 3 * r3;
                if (r3 == N) {
                    dv[t][0] = c3 == 0 ? nb_new[0] : (c3 == 1 ? nb_new[2] : nb_new[4]);
                    dv[t][1] = c3 == 0 ? nb_new[1] : (c3 == 1 ? nb_new[3] : nb_new[5]);
                    ev[t][0] = c3 == 0 ? x_new[0] : (c3 == 1 ? x_new[2] : x_new[4]);
                    ev[t][1] = c3 == 0 ? x_new[1] : (c3 == 1 ? x_new[3] : x_new[5]);
                }
            }
        }
        __syncthreads();
        const double g_o[2] = {g_ox, g_oy}, g_e[2] = {g_ex, g_ey};
        const bool open = orow >= 0 && (ma.ego_xy ? gate_open(g_o, g_e, ma.r2) : true);
        {
            float fx[ZR], fy[ZR];
#pragma unroll
            for (int t = 0; t < ZR; ++t) { fx[t] = (float)(dv[t][0] - ev[t][0]); fy[t] = (float)(dv[t][1] - ev[t][1]); }
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int src = 3 * jr + s, sl = (src & 63) << 2;
                float vx = __int_as_float(__builtin_amdgcn_ds_bpermute(sl, __float_as_int(fx[0])));
                float vy = __int_as_float(__builtin_amdgcn_ds_bpermute(sl, __float_as_int(fy[0])));
                if (ZR > 1) {
                    const float wx = __int_as_float(__builtin_amdgcn_ds_bpermute(sl, __float_as_int(fx[ZR - 1])));
                    const float wy = __int_as_float(__builtin_amdgcn_ds_bpermute(sl, __float_as_int(fy[ZR - 1])));
                    if (src >= 64) { vx = wx; vy = wy; }
                }
                zb[s] = h ? vy : vx;
            }
        }
        if (NDP_RARELY(io.stamps && lane == 0)) io.stamps[11] = (double)__builtin_amdgcn_s_memtime();
        // gate closed (or no neighbour): the force is zero and the reference does not evaluate the network either
        // (ndp_nmpc_leader_node.py:66-76).  The test is the same in every lane: a wave-uniform branch around the tile.
        o[0] = o[1] = o[2] = 0.0f;
        if (__builtin_amdgcn_readfirstlane((int)open)) mlp_tile(wl, zb, lane, o);
        __syncthreads();                              // every wave is done with the weights before LDS becomes RTI state
        if (!active) return;
        if (NDP_RARELY(io.dbg && lane == 0)) io.dbg[m.total + 10] = (double)__builtin_amdgcn_s_memtime();
        if (NDP_RARELY(io.stamps && lane == 0)) io.stamps[10] = (double)__builtin_amdgcn_s_memtime();
        if (j < np1 && h == 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float v = open ? o[c] : 0.0f;      // ndp_nmpc_leader_node.py:75-76
                lds[m.TF + j * 3 + c] = (double)v;       // fp32 value promoted to fp64
                ma.force_out[inst * nf + j * 3 + c] = v; // also what the consumer launch of the work list reads
            }
        }
        WaveGfx950::sync();
        io.f = nullptr;
        io.f_in_lds = 1;
    }
    if (TICK && advance && active) tick_cache_store(ka.ta, te, inst, (int)(threadIdx.x & 63u), seg_refill, seg_fill, seg_cfill);   // (requested in the prologue: long there)
    if (TICK && ka.ta.est && active) io.kthr = tick_estimator(ka.ta, inst, B, (int)(threadIdx.x & 63u));
    const bool deferred = Prog::template run<QMODE == 1, QMODE == 0 || QMODE == 3>(P, io, lds, inb, x0v);
    if (NDP_RARELY(io.stamps && (threadIdx.x & 63u) == 0)) {
        io.stamps[13] = (double)__builtin_amdgcn_s_memrealtime();
        io.stamps[15] = (double)__builtin_amdgcn_s_memtime();
    }
    if (QMODE == 1 && deferred && (threadIdx.x & 63u) == 0) {
        const unsigned s = __hip_atomic_fetch_add(qa.count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        qa.ids[s] = inst;
    }
}

// test hook: one v_mfma_f64_16x16x4_f64 / v_mfma_f64_4x4x4_4b_f64 with caller-chosen per-lane operands (pins the register maps)
__global__ void mfma_probe_kernel(const double *a, const double *b, const double *c, double *d)
{
    const int l = (int)threadIdx.x;
    WaveGfx950::vd4 acc;
    for (int r = 0; r < 4; ++r) acc.r[r] = c[r * 64 + l];
    acc = WaveGfx950::mfma(a[l], b[l], acc);
    for (int r = 0; r < 4; ++r) d[r * 64 + l] = acc.r[r];
    d[256 + l] = WaveGfx950::readlane(a[l], 37) + WaveGfx950::wave_sum(b[l]) + WaveGfx950::wave_min(a[l]) + WaveGfx950::wave_max(a[l]);
    // the four-block v_mfma_f64_4x4x4_4b_f64 on the same operands (accumulator: c's first register) and the four row broadcasts
    d[320 + l] = WaveGfx950::mfma4(a[l], b[l], c[l]);
    d[384 + l] = WaveGfx950::rowb<0>(a[l]);
    d[448 + l] = WaveGfx950::rowb<1>(a[l]);
    d[512 + l] = WaveGfx950::rowb<2>(a[l]);
    d[576 + l] = WaveGfx950::rowb<3>(a[l]);
    // the row rotations that bring the packed -Lam^-1 of a stage to block 3 (rti_wave.hpp: linv_get)
    d[640 + l] = WaveGfx950::rowror4<1>(a[l]);
    d[704 + l] = WaveGfx950::rowror4<2>(a[l]);
    d[768 + l] = WaveGfx950::rowror4<3>(a[l]);
}

// test hook: one v_mfma_f32_16x16x4_f32 (mode 0) or one v_mfma_f32_16x16x16_bf16 (mode 1: four packed contraction steps)
// through the config-5 backends, caller-chosen per-lane operands a[4][64], b[4][64] (mode 0 uses row 0), c[4][64] -> d[4][64]
__global__ void mfma_probe32_kernel(const float *a, const float *b, const float *c, float *d, int mode)
{
    const int l = (int)threadIdx.x;
    WaveGfx950F32::md4 acc;
    for (int r = 0; r < 4; ++r) acc.r[r] = c[r * 64 + l];
    if (mode == 0) acc = WaveGfx950F32::mfma(a[l], b[l], acc);
    else {
        float av[4], bv[4];
        for (int i = 0; i < 4; ++i) { av[i] = a[i * 64 + l]; bv[i] = b[i * 64 + l]; }
        acc = WaveGfx950BF16::mfma_k(av, bv, 4, acc);
    }
    for (int r = 0; r < 4; ++r) d[r * 64 + l] = acc.r[r];
    // the row sum of the config-5 layout: lanes 4 apart inside a 16-lane row
    double x = (double)a[l];
    x = x + WaveGfx950F32::csum1(x);
    x = x + WaveGfx950F32::csum2(x);
    d[256 + l] = (float)x;
}

// behind the work list's consumer launch: the list is empty again for the next step's producer
__global__ void queue_reset_kernel(unsigned *count, unsigned long long *ipm_total)
{
    if (threadIdx.x == 0) {
        *ipm_total += *count;
        *count = 0u;
    }
}

// ------------------------------------------------------------------------------------------ MLP kernel
// nn_net.py:7-18: Linear(6,128) ReLU Linear(128,64) ReLU Linear(64,128) ReLU Linear(128,3), fp32.
// One wave = 32 horizon rows (columns of the MFMA tile); activations stay transposed [feature][row] in
// the accumulators: the 32x32 f32 accumulator holds feature (r&3)+8(r>>2)+4(lane>>5) of row lane&31 in
// register r, which is exactly the B-operand shape of the next layer's 32x32x2 step when that step
// contracts the feature pair {f0(r), f0(r)+4}.  The weights are pre-permuted on the host into that
// "fragment order" (one 64-float record per MFMA), so A operands are coalesced 256-byte loads.
typedef float f16_t __attribute__((ext_vector_type(16)));

// Fragment blob (built on the host by make_fragments, parked in LDS during the MLP phase), in float units:
//   FR_L1  12 x 64 f32     layer-1 A operands for v_mfma_f32_32x32x2_f32 (K = 6 inputs)
//   FR_B1/B2/B3, FR_W4 ([128 features][4]: w0 w1 w2 0), FR_B4
//   FR_HF  layers 2 and 3 as fp16 pairs: 32 records (16 per layer) x 2 splits (hi, lo * 2^11) x 64 lanes x 8 halves
// The blob is moved by LDS-DMA in 1-KB pieces (one global_load_lds_dwordx4 per wave): its size is a multiple of 256 floats.
enum { FR_L1 = 0, FR_B1 = FR_L1 + 12 * 64, FR_B2 = FR_B1 + 128, FR_B3 = FR_B2 + 64, FR_W4 = FR_B3 + 128,
       FR_B4 = FR_W4 + 4 * 128, FR_HF = FR_B4 + 4, FR_REC = 2 * 64 * 8 / 2 /* floats per record */,
       FR_USED = FR_HF + 32 * FR_REC,
       // (a multiple of 8 pieces: every wave of a 1- / 2- / 4- / 8-wave workgroup moves the SAME number of them -- see stage_fragments)
       FR_CHUNKS = (FR_USED + 2047) / 2048 * 8, FR_TOTAL = FR_CHUNKS * 256 };
static_assert(FR_HF % 4 == 0 && FR_W4 % 4 == 0, "16-byte alignment of the LDS image");

__device__ __forceinline__ int f0(int r) { return (r & 3) + 8 * (r >> 2); }

// The four layers for one 32-row tile held by one wave.  zb[s] = input feature 2s + (lane>>5) of row lane&31;
// returns the three outputs of row lane&31 in o[] (both half-waves hold the full sums).

// The workgroup copies the fragment blob (FR_TOTAL floats, L2-resident) into LDS by LDS-DMA: each wave issues one
// global_load_lds_dwordx4 per 1-KB piece (64 lanes x 16 B, lane-linear destination), nothing passes through VGPRs and
// all pieces are in flight at once; the caller's __syncthreads() (which waits vmcnt(0)) retires them.  Through
// registers (global_load_dwordx4 + ds_write_b128 per thread and pass) the same copy took 7.5k cycles per workgroup.
// Streaming the weights per wave straight from L2 instead made 1024 waves fetch the same lines in lockstep (channel
// hot-spotting: the MLP tile took 33k cycles at B = 1024 against 25k alone).
__device__ __forceinline__ void stage_fragments(const float *__restrict__ fr, lds_f32 dst, int tid, int nthreads)
{
    // the wave index is uniform: keep the piece loop scalar (derived from threadIdx it would run under an exec mask), a
    // compile-time number of rounds with a uniform guard on the last one
    // The number of pieces per wave must not depend on the wave: with a guarded last round the compiler cannot count the transfers
    // in flight and makes the NEXT wait of the wave -- whatever it is for -- a wait for all of them (s_waitcnt vmcnt(0)); the one-launch
    // tick's polynomial work, meant to run under the transfer, then started behind it (+2 600 cycles per tick).
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, nw = nthreads >> 6;
#pragma unroll
    for (int i = 0; i < FR_CHUNKS; ++i) {
        if (i * nw >= FR_CHUNKS) break;
        const int c = wave + i * nw;
        if ((i + 1) * nw <= FR_CHUNKS || c < FR_CHUNKS)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(fr + c * 256 + lane * 4),
                                             (__attribute__((address_space(3))) void *)(dst + c * 256), 16, 0, 0);
    }
}

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
struct Split2 { h16x8 hi, lo; };
#define NDP_LO_SCALE 2048.0f            // 2^11: the low parts are carried scaled so that they stay fp16-normal
#define NDP_LO_INV (1.0f / 2048.0f)
#define NDP_H16_CAP 65000.0f            // activations are capped below the fp16 overflow threshold (see split2)

// x = hi + lo / 2^11 with two fp16 terms (11 + 11 significand bits; fp32 has 24): hi = fp16(x), the residual
// x - hi is exact in fp32 and lo = fp16(residual * 2^11).  Relative error of the pair 2^-22.
__device__ __forceinline__ void split2(const f16_t &v, int s, Split2 &o)
{
    typedef float f2_t __attribute__((ext_vector_type(2)));
    typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int j = 0; j < 8; j += 2) {            // two registers at a time: packed f32 subtract / multiply, packed conversions
        const f2_t x = {v[8 * s + j], v[8 * s + j + 1]};
        const h2_t hh = __builtin_convertvector(x, h2_t);
        const f2_t r = (x - __builtin_convertvector(hh, f2_t)) * NDP_LO_SCALE;
        const h2_t ll = __builtin_convertvector(r, h2_t);
        o.hi[j] = hh[0]; o.hi[j + 1] = hh[1];
        o.lo[j] = ll[0]; o.lo[j + 1] = ll[1];
    }
}

// ReLU of a hidden layer that feeds an fp16 split, capped at NDP_H16_CAP: one v_med3_f32, the price of a plain
// v_max_f32.  The cap only acts on inputs ~1000x outside the training envelope, where the reference returns a finite
// meaningless force; uncapped, the fp16 conversion would overflow to inf and turn that into NaN.
__device__ __forceinline__ float relu_cap(float x) { return __builtin_amdgcn_fmed3f(x, 0.0f, NDP_H16_CAP); }

// one (output tile, 16-deep k-step): W x = W_hi x_hi + (W_hi x_lo + W_lo x_hi) / 2^11; the dropped W_lo x_lo term is
// below 2^-22 of the result.  Three v_mfma_f32_32x32x16_f16 (products exact in the fp32 accumulators), the cross terms
// in their own accumulator.  16x the f32 MFMA rate per instruction, so the three still run 5x faster than the exact
// f32 form; measured error on the reference fixture 3.7e-6 (bar 1e-5).
__device__ __forceinline__ void mm3(const Split2 &w, const Split2 &x, f16_t &acc_hi, f16_t &acc_lo)
{
    acc_lo = __builtin_amdgcn_mfma_f32_32x32x16_f16(w.lo, x.hi, acc_lo, 0, 0, 0);
    acc_lo = __builtin_amdgcn_mfma_f32_32x32x16_f16(w.hi, x.lo, acc_lo, 0, 0, 0);
    acc_hi = __builtin_amdgcn_mfma_f32_32x32x16_f16(w.hi, x.hi, acc_hi, 0, 0, 0);
}

__device__ __forceinline__ void load_w(lds_cf32 fr, int rec, int lane, Split2 &w)
{
    const __attribute__((address_space(3))) h16x8 *p = (const __attribute__((address_space(3))) h16x8 *)(fr + FR_HF) + rec * 128 + lane;
    w.hi = p[0]; w.lo = p[64];
}

// The four layers for one 32-row tile held by one wave.  zb[s] = input feature 2s + (lane>>5) of row lane&31;
// returns the three outputs of row lane&31 in o[] (both half-waves hold the full sums).
// Measured (scripts/ubench/mfma_f16_valu_overlap.hip, profiles/r02_ubench_mfma_f16_valu_overlap.txt): a wave's f32 VALU work is
// NOT hidden behind its own v_mfma_f32_32x32x16_f16 -- 35 cycles per instruction alone, 35 + 6 + 2.5 per v_fma_f32 issued
// behind it -- so a software-pipelined form of this tile (conversions of one layer issued between the matrix instructions
// of the next) ran no faster than this layer-by-layer form (9.46 k against 9.18 k cycles); what counts is the instruction
// count.  No scheduling fences here: the compiler's own order is 0.84 k cycles shorter than a fenced one.
// Activations stay transposed [feature][row] in the accumulators.  Registers 8s..8s+7 of a 32x32 accumulator,
// converted to fp16 pairs, ARE the B operand of k-step s of the next layer (feature 16s + 8(j>>2) + 4(lane>>5) + (j&3) in
// element j); the weights are stored in that k order.
__device__ __forceinline__ void mlp_tile(lds_cf32 fr, const float zb[3], int lane, float o[3])
{
    typedef float f4_t __attribute__((ext_vector_type(4)));
    const int h = lane >> 5;
    Split2 x1[4][2], x2[2][2];
    f16_t h3[4];
    float bc[16];
    // layer 1 (6 -> 128): exact f32 MFMA, K = 2 per instruction
#pragma unroll
    for (int ot = 0; ot < 4; ++ot) {
#pragma unroll
        for (int r = 0; r < 16; ++r) bc[r] = fr[FR_B1 + ot * 32 + f0(r) + 4 * h];
        f16_t acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = bc[r];        // the bias rides in the accumulator
#pragma unroll
        for (int s = 0; s < 3; ++s)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fr[FR_L1 + (ot * 3 + s) * 64 + lane], zb[s], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = relu_cap(acc[r]);
        split2(acc, 0, x1[ot][0]);
        split2(acc, 1, x1[ot][1]);
    }
    // layers 2 (128 -> 64) and 3 (64 -> 128) as one stream of 32 weight records, the next record requested before
    // the current record's six MFMAs issue
    Split2 wc, wn;
    load_w(fr, 0, lane, wc);
    f16_t acc, accl;
#pragma unroll
    for (int rec = 0; rec < 32; ++rec) {
        const bool l2 = rec < 16;
        const int q = l2 ? rec : rec - 16;
        const int ot = l2 ? q / 8 : q / 4, it = l2 ? (q / 2) % 4 : (q / 2) % 2, s = q % 2;
        const bool first = l2 ? (q % 8 == 0) : (q % 4 == 0), last = l2 ? (q % 8 == 7) : (q % 4 == 3);
        if (rec + 1 < 32) load_w(fr, rec + 1, lane, wn);
        if (first) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[r] = fr[(l2 ? FR_B2 : FR_B3) + ot * 32 + f0(r) + 4 * h];   // the bias rides in the accumulator
                accl[r] = 0.0f;
            }
        }
#ifdef NDP_DEV_HALF_TILE        // (measurement only: half of the tile's matrix instructions, wrong forces -- what a tile shared by two waves would cost at best)
        if (!(rec & 1))
#endif
        mm3(wc, l2 ? x1[it][s] : x2[it][s], acc, accl);
        if (last) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = fmaf(accl[r], NDP_LO_INV, acc[r]);
                acc[r] = l2 ? relu_cap(v) : fmaxf(v, 0.0f);
            }
            if (l2) { split2(acc, 0, x2[ot][0]); split2(acc, 1, x2[ot][1]); }
            else h3[ot] = acc;
        }
        wc = wn;
    }
    // last layer (128 -> 3) on the VALU in f32: each half-wave owns 64 of the 128 features of its row; weights come as
    // one 16-byte record per feature, the records of the next 16 features requested before the current ones are used
    const __attribute__((address_space(3))) f4_t *w4 = (const __attribute__((address_space(3))) f4_t *)(fr + FR_W4);
    f4_t qc[16], qn[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) qc[r] = w4[f0(r) + 4 * h];
    o[0] = o[1] = o[2] = 0.0f;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        if (it + 1 < 4) {
#pragma unroll
            for (int r = 0; r < 16; ++r) qn[r] = w4[(it + 1) * 32 + f0(r) + 4 * h];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int c = 0; c < 3; ++c) o[c] = fmaf(qc[r][c], h3[it][r], o[c]);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) qc[r] = qn[r];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) o[c] = o[c] + __shfl_xor(o[c], 32, 64) + fr[FR_B4 + c];
}

// gate of ndp_nmpc_leader_node.py:65-68: other.x[0] xy against ego ODOMETRY xy, strict '<'.  Individually rounded
// mul/add (nc_mul / nc_add): the reference evaluates this in Python doubles and must agree at the rim.
__device__ __forceinline__ bool gate_open(const double *other_inst, const double *ego_xy_inst, double r2)
{
    const double dx = other_inst[0] - ego_xy_inst[0];
    const double dy = other_inst[1] - ego_xy_inst[1];
    return nc_add(nc_mul(dx, dx), nc_mul(dy, dy)) < r2;
}

// Standalone form (DownwashNN.update for arbitrary row counts): one 32-row tile per wave, no tile loop --
// a loop would make every weight load loop-invariant and the compiler then tries to keep 17k weights in registers.
__global__ __launch_bounds__(256) void mlp_kernel(const float *__restrict__ fr, const double *__restrict__ other,
                                                  const double *__restrict__ ego, const double *__restrict__ ego_xy,
                                                  float *__restrict__ fout, int rows, int np1, double r2,
                                                  int other_stride, const int *__restrict__ other_index, int other_sys,
                                                  size_t other_pitch, size_t ego_pitch, size_t ego_xy_pitch)     // doubles per row of other / per instance of ego, ego_xy (see MlpArgs)
{
    extern __shared__ __attribute__((aligned(16))) float wsm[];
    const int lane = (int)(threadIdx.x & 63u), wave = (int)(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int ntiles = (rows + 31) / 32;
    const int tile = (int)blockIdx.x * 4 + wave;
    lds_f32 wl = (lds_f32)wsm;
    stage_fragments(fr, wl, (int)threadIdx.x, 256);
    __syncthreads();
    if (tile >= ntiles) return;
    const int row = tile * 32 + j;
    const bool valid = row < rows;
    const int rowc = valid ? row : rows - 1;
    const int inst = rowc / np1, k = rowc - inst * np1;
    const int orow = other_index ? other_index[inst] : inst;          // see MlpArgs
    const double *oth = other + (size_t)(orow < 0 ? 0 : orow) * other_pitch;
    bool open = valid && orow >= 0;
    if (ego_xy) {
        const double oxy[2] = {ld_other(oth, other_sys), ld_other(oth + 1, other_sys)};
        open = open && gate_open(oxy, ego_xy + (size_t)inst * ego_xy_pitch, r2);
    }
    // downwash_nn.py:22-23: (other - ego)[:, 0:6] in fp64, cast to fp32
    float zb[3], o[3];
#pragma unroll
    for (int s = 0; s < 3; ++s)
        zb[s] = (float)(ld_other(oth + (size_t)k * other_stride + 2 * s + h, other_sys) - ego[(size_t)inst * ego_pitch + (size_t)k * NX + 2 * s + h]);
    mlp_tile(wl, zb, lane, o);
    if (valid && h == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) fout[(size_t)row * 3 + c] = open ? o[c] : 0.0f;   // :75-76 zeros when gated off
    }
}

// ---- LDS-free form of the tile (measurement: downwash of the NEXT tick in a second stream beside the control-step kernel, which
// holds all of a CU's LDS and 320 of its 512 registers per lane).  The weight records come straight from global memory (L2),
// NPRE records requested ahead; biases, first- and last-layer weights likewise.  Same arithmetic, same fragment blob.
typedef const float *__restrict__ g_cf32;
__device__ __forceinline__ void load_w_g(g_cf32 fr, int rec, int lane, Split2 &w)
{
    const h16x8 *p = reinterpret_cast<const h16x8 *>(fr + FR_HF) + rec * 128 + lane;
    w.hi = p[0]; w.lo = p[64];
}

__device__ __forceinline__ void mlp_tile_stream(g_cf32 fr, const float zb[3], int lane, float o[3])
{
    typedef float f4_t __attribute__((ext_vector_type(4)));
    constexpr int NPRE = 2;
    const int h = lane >> 5;
    Split2 wq[NPRE];
#pragma unroll
    for (int i = 0; i < NPRE; ++i) load_w_g(fr, i, lane, wq[i]);        // in flight under layer 1
    Split2 x1[4][2], x2[2][2];
    const f4_t *w4 = reinterpret_cast<const f4_t *>(fr + FR_W4);
    o[0] = o[1] = o[2] = 0.0f;
#pragma unroll
    for (int ot = 0; ot < 4; ++ot) {
        f16_t acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = fr[FR_B1 + ot * 32 + f0(r) + 4 * h];
#pragma unroll
        for (int s = 0; s < 3; ++s)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fr[FR_L1 + (ot * 3 + s) * 64 + lane], zb[s], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = relu_cap(acc[r]);
        split2(acc, 0, x1[ot][0]);
        split2(acc, 1, x1[ot][1]);
        __builtin_amdgcn_sched_barrier(0);     // keep the phases apart: left alone the scheduler hoists every later load up front (254 registers)
    }
    f16_t acc, accl;
#pragma unroll
    for (int rec = 0; rec < 32; ++rec) {
        const bool l2 = rec < 16;
        const int q = l2 ? rec : rec - 16;
        const int ot = l2 ? q / 8 : q / 4, it = l2 ? (q / 2) % 4 : (q / 2) % 2, s = q % 2;
        const bool first = l2 ? (q % 8 == 0) : (q % 4 == 0), last = l2 ? (q % 8 == 7) : (q % 4 == 3);
        const Split2 wc = wq[rec % NPRE];
        if (rec + NPRE < 32) load_w_g(fr, rec + NPRE, lane, wq[rec % NPRE]);
        if (first) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[r] = fr[(l2 ? FR_B2 : FR_B3) + ot * 32 + f0(r) + 4 * h];
                accl[r] = 0.0f;
            }
        }
        mm3(wc, l2 ? x1[it][s] : x2[it][s], acc, accl);
        if (last) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = fmaf(accl[r], NDP_LO_INV, acc[r]);
                acc[r] = l2 ? relu_cap(v) : fmaxf(v, 0.0f);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (l2) { split2(acc, 0, x2[ot][0]); split2(acc, 1, x2[ot][1]); }
            else {      // last layer (128 -> 3) on this output tile at once: its 32 hidden features need not stay live
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const f4_t qv = w4[ot * 32 + f0(r) + 4 * h];
#pragma unroll
                    for (int c = 0; c < 3; ++c) o[c] = fmaf(qv[c], acc[r], o[c]);
                }
            }
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) o[c] = o[c] + __shfl_xor(o[c], 32, 64) + fr[FR_B4 + c];
}

// one wave = one 32-row tile, four per workgroup, no LDS and no barrier, ~180 registers per lane (control step: 320 -- the two fit on
// a SIMD together; ensure_prefetch checks it).  proto != null: launch m of the prefetch protocol (see LateArgs).
__global__ __launch_bounds__(256)
void mlp_stream_kernel(const float *__restrict__ fr, const double *__restrict__ other, const double *__restrict__ ego,
                       const double *__restrict__ ego_xy, float *__restrict__ fout, float *__restrict__ fout1, int rows, int np1, double r2,
                       int other_stride, const int *__restrict__ other_index, unsigned long long *proto, int other_sys)
{
    const int lane = (int)(threadIdx.x & 63u), wave = (int)(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int ntiles = (rows + 31) / 32;
    const int tile = (int)blockIdx.x * 4 + wave;
    if (tile >= ntiles) return;
    unsigned long long m = 0;
    if (proto) {
        m = proto[PF_CUR_M];                        // written by the gate launch in front of this one (plain load)
        if (m & 1) fout = fout1;
    }
    const int row = tile * 32 + j;
    const bool valid = row < rows;
    const int rowc = valid ? row : rows - 1;
    const int inst = rowc / np1, k = rowc - inst * np1;
    const int orow = other_index ? other_index[inst] : inst;
    const double *oth = other + (size_t)(orow < 0 ? 0 : orow) * np1 * other_stride;
    bool open = valid && orow >= 0;
    if (ego_xy) {
        const double oxy[2] = {ld_other(oth, other_sys), ld_other(oth + 1, other_sys)};
        open = open && gate_open(oxy, ego_xy + inst * 2, r2);
    }
    float zb[3], o[3];
#pragma unroll
    for (int s = 0; s < 3; ++s)
        zb[s] = (float)(ld_other(oth + (size_t)k * other_stride + 2 * s + h, other_sys) - ego[(size_t)rowc * NX + 2 * s + h]);
    mlp_tile_stream(fr, zb, lane, o);
    if (!proto) {
        if (valid && h == 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) fout[(size_t)row * 3 + c] = open ? o[c] : 0.0f;
        }
        return;
    }
    // the rows go out past this XCD's L2 (agent-scope stores: the reader runs on other XCDs, now), then -- once they are complete --
    // the tile's epoch word
    if (valid && h == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c)
            __hip_atomic_store(reinterpret_cast<unsigned *>(fout) + (size_t)row * 3 + c, __float_as_uint(open ? o[c] : 0.0f),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // The rows were written THROUGH to memory (sc1 stores): what is needed before the epoch word goes out is their COMPLETION, i.e.
    // s_waitcnt vmcnt(0) -- stores of one wave to different addresses / channels may complete out of order.  Not an agent-scope
    // release: its buffer_wbl2 writes back the XCD's whole L2, the control step's dirty iterate included (one per wave, 672 per
    // launch, made the control step beside it 50 % slower).  And not a workgroup-scope release fence alone: on gfx950 it emits NO
    // vmcnt wait (the ISA had the three row stores followed directly by the epoch store) -- the wait is spelled out;
    // the fence stays as the compiler-level barrier.  scripts/isa_audit.py checks the instruction is there.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0)
        __hip_atomic_store(proto + PF_EPOCH + (m & 1) * (unsigned)ntiles + tile, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One wave, in front of downwash launch m on the second stream: takes the number, waits until the control step that read force
// slot m & 1 last (number m - 2) holds its values, publishes the number for the launch behind it.  A single wave polling gently
// costs nothing (waves of the downwash kernel polling on every CU kept registers the next control-step workgroups needed).
__global__ void prefetch_gate_kernel(unsigned long long *proto, unsigned timeout_us, unsigned groups_rti)
{
    if (threadIdx.x != 0) return;
    const unsigned long long m = proto[PF_CUR_M] + 1;
    if (m > 2) {
        const unsigned long long want = (m - 2) * groups_rti;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__hip_atomic_load(proto + PF_RTI_C2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
            __builtin_amdgcn_s_sleep(16);
            if (__builtin_amdgcn_s_memrealtime() - t0 > 100ull * timeout_us) {
                atomicAdd(reinterpret_cast<int *>(proto + PF_GATE_TIMEOUT), 1);
                break;
            }
        }
    }
    proto[PF_CUR_M] = m;
}

// behind downwash launch m in its stream: that launch is complete
__global__ void prefetch_done_kernel(unsigned long long *proto)
{
    if (threadIdx.x == 0) __hip_atomic_store(proto + PF_MLP_DONE, proto[PF_CUR_M], __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// host: blob (W1 b1 W2 b2 W3 b3 W4 b4, row-major [out][in]) -> fragment order
static uint16_t f16_rn(float x)
{   // round-to-nearest-even f32 -> fp16 bits, subnormals included (the weights and their residuals are finite and small)
    uint32_t u;
    memcpy(&u, &x, 4);
    const uint16_t sign = (uint16_t)((u >> 16) & 0x8000u);
    const uint32_t ax = u & 0x7FFFFFFFu;
    if (ax >= 0x47800000u) return (uint16_t)(sign | 0x7C00u);               // >= 65536: inf (never for these data)
    if (ax < 0x33000000u) return sign;                                       // < 2^-25: 0
    const int e = (int)(ax >> 23) - 127;
    uint32_t m = (ax & 0x7FFFFFu) | 0x800000u;                               // 24-bit significand
    const int shift = e < -14 ? 13 + (-14 - e) : 13;                         // bits to drop (subnormal: more)
    const uint32_t keep = m >> shift, rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
    uint32_t r = keep + ((rem > half || (rem == half && (keep & 1u))) ? 1u : 0u);
    // normal: r has the hidden bit at position 10; adding the exponent field lets a mantissa carry roll into it
    const uint32_t bits = e < -14 ? r : (uint32_t)((e + 15 - 1) << 10) + r;
    return (uint16_t)(sign | bits);
}
static float f16_f32(uint16_t b)
{
    const uint32_t sign = (uint32_t)(b & 0x8000u) << 16, ex = (b >> 10) & 0x1Fu, mant = b & 0x3FFu;
    float x;
    if (ex == 0) {
        x = (float)mant * 5.9604644775390625e-08f;                           // 2^-24
        uint32_t u; memcpy(&u, &x, 4); u |= sign; memcpy(&x, &u, 4);
        return x;
    }
    const uint32_t u = sign | ((ex + 112u) << 23) | (mant << 13);
    memcpy(&x, &u, 4);
    return x;
}

static void make_fragments(const float *blob, std::vector<float> &fr)
{
    const float *W1 = blob, *b1 = W1 + 128 * 6, *W2 = b1 + 128, *b2 = W2 + 64 * 128;
    const float *W3 = b2 + 64, *b3 = W3 + 128 * 64, *W4 = b3 + 128, *b4 = W4 + 3 * 128;
    fr.assign(FR_TOTAL, 0.0f);
    for (int ot = 0; ot < 4; ++ot)
        for (int s = 0; s < 3; ++s)
            for (int l = 0; l < 64; ++l) fr[FR_L1 + (ot * 3 + s) * 64 + l] = W1[(ot * 32 + (l & 31)) * 6 + 2 * s + (l >> 5)];
    for (int i = 0; i < 128; ++i) fr[FR_B1 + i] = b1[i];
    for (int i = 0; i < 64; ++i) fr[FR_B2 + i] = b2[i];
    for (int i = 0; i < 128; ++i) fr[FR_B3 + i] = b3[i];
    for (int f = 0; f < 128; ++f)
        for (int c = 0; c < 3; ++c) fr[FR_W4 + f * 4 + c] = W4[c * 128 + f];
    for (int i = 0; i < 3; ++i) fr[FR_B4 + i] = b4[i];
    // layers 2, 3: record = (out tile, in tile, k-step); lane (r = l&31, h = l>>5) element j holds
    // W[ot*32 + r][it*32 + 16 s + 8 (j>>2) + 4 h + (j&3)] split into hi + lo / 2^11
    uint16_t *hf = reinterpret_cast<uint16_t *>(fr.data() + FR_HF);
    for (int rec = 0; rec < 32; ++rec) {
        const bool l2 = rec < 16;
        const int q = l2 ? rec : rec - 16;
        const int ot = l2 ? q / 8 : q / 4, it = l2 ? (q / 2) % 4 : (q / 2) % 2, s = q % 2;
        const float *W = l2 ? W2 : W3;
        const int nin = l2 ? 128 : 64;
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 8; ++j) {
                const int kin = it * 32 + 16 * s + 8 * (j >> 2) + 4 * (l >> 5) + (j & 3);
                const float w = W[(ot * 32 + (l & 31)) * nin + kin];
                const uint16_t hi = f16_rn(w);
                const float r1 = w - f16_f32(hi);
                const uint16_t lo = f16_rn(r1 * NDP_LO_SCALE);
                uint16_t *rp = hf + (size_t)rec * 2 * 512;
                rp[0 * 512 + l * 8 + j] = hi;
                rp[1 * 512 + l * 8 + j] = lo;
            }
    }
}

// ------------------------------------------------------------------------------------------ f3 kernels
// Hover-throttle estimator (2-state Kalman filter on [f_collect, k_throttle] + Tustin differentiator), one thread
// per vehicle.  Elementwise and HBM-bound: state is SoA ([8][B] doubles) so every access is a coalesced 512-B wave
// load/store; 152 algorithmic bytes per vehicle and tick.  Operation order follows the reference's numpy
// expressions (hover_throttle_estimator.py:38-51) so results agree to rounding.

// Streaming (non-temporal) accesses of the rows' kernels: outputs nobody reads again in the same launch, inputs read once.  With plain
// stores ref_window_kernel ran at 0.45 of the HBM roof although its loads + arithmetic alone take 75 us and its arithmetic + stores
// alone 111 us of the 187 (knock-out builds, round 5): the 616 MB of window rows went through L2 as ordinary dirty lines and every
// wave's dependent load rounds queued behind them.  `nt`: 137 us (0.62).
typedef double nt_d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void st_stream(double2 *p, const double2 &v)
{
    const nt_d2 t = {v.x, v.y};
    __builtin_nontemporal_store(t, reinterpret_cast<nt_d2 *>(p));
}
__device__ __forceinline__ void st_stream(double *p, double v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ double2 ld_stream(const double2 *p)
{
    const nt_d2 t = __builtin_nontemporal_load(reinterpret_cast<const nt_d2 *>(p));
    return make_double2(t.x, t.y);
}

// one estimator update of vehicle v (state SoA [8][S]); returns k_throttle
__device__ __forceinline__ double throttle_update_one(const ThrCfg &c, double *__restrict__ st, size_t S, int v, double vzv, double th)
{
    const double az = nc_add(nc_mul(c.a1, st[7 * S + v]), nc_mul(c.a2, vzv - st[6 * S + v]));   // differentiator.py:21
    st[6 * S + v] = vzv;
    st[7 * S + v] = az;
    double x1 = st[1 * S + v];
    if (0.1 < th && th < 1.0) {                                            // hover_throttle_estimator.py:40
        const double z = az + c.g;
        const double P11 = st[5 * S + v];
        // numpy evaluates these products without fused multiply-add: keep individually rounded operations
        const double p01 = nc_mul(th, P11), p10 = nc_mul(P11, th);
        const double p00 = nc_add(nc_mul(p01, th), c.Q0), p11 = nc_add(P11, c.Q1);
        const double inv = 1.0 / nc_add(nc_mul(nc_mul(c.hm, p00), c.hm), c.R);
        const double K0 = nc_mul(nc_mul(p00, c.hm), inv), K1 = nc_mul(nc_mul(p10, c.hm), inv);
        const double x0p = nc_mul(th, x1);
        const double innov = z - nc_mul(c.hm, x0p);
        st[0 * S + v] = nc_add(x0p, nc_mul(K0, innov));
        x1 = nc_add(x1, nc_mul(K1, innov));
        st[1 * S + v] = x1;
        const double i00 = 1.0 - nc_mul(K0, c.hm), i10 = -nc_mul(K1, c.hm);
        st[2 * S + v] = nc_mul(i00, p00);
        st[3 * S + v] = nc_mul(i00, p01);
        st[4 * S + v] = nc_add(nc_mul(i10, p00), p10);
        st[5 * S + v] = nc_add(nc_mul(i10, p01), p11);
    }
    return x1;
}

__global__ __launch_bounds__(256) void throttle_kernel(ThrCfg c, double *__restrict__ st, const double *__restrict__ vz,
                                                       const double *__restrict__ throttle, double *__restrict__ k_out, int B)
{
    const int v = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (v >= B) return;
    k_out[v] = throttle_update_one(c, st, (size_t)B, v, vz[v], throttle[v]);
}

__global__ __launch_bounds__(256) void throttle_reset_kernel(double *st, double k_init, int B)
{
    const int v = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (v >= B) return;
    const size_t S = (size_t)B;
    st[0 * S + v] = 0.0; st[1 * S + v] = k_init;
    st[2 * S + v] = 1.0; st[3 * S + v] = 0.0; st[4 * S + v] = 0.0; st[5 * S + v] = 1.0;
    st[6 * S + v] = 0.0; st[7 * S + v] = 0.0;
}

// nmpc_u_2_att_tgt (nmpc_node.py:273-283): body rates pass through, thrust = c * mass / k_throttle (0 if k == 0)
__device__ __forceinline__ double thrust_cmd(double c, double mass, double k) { return k != 0.0 ? nc_mul(c, mass) / k : 0.0; }
__global__ __launch_bounds__(256) void actuator_kernel(const double *__restrict__ u0, const double *__restrict__ k,
                                                       double *__restrict__ cmd, double mass, int B)
{
    const int v = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (v >= B) return;
    const double2 a = reinterpret_cast<const double2 *>(u0)[2 * v], b = reinterpret_cast<const double2 *>(u0)[2 * v + 1];
    const double kk = k[v];
    double2 o0 = a, o1 = b;
    o1.y = thrust_cmd(b.y, mass, kk);
    reinterpret_cast<double2 *>(cmd)[2 * v] = o0;
    reinterpret_cast<double2 *>(cmd)[2 * v + 1] = o1;
}

// ------------------------------------------------------------------------------------------ f2 kernels
// AlphaFilter per instance and axis (alpha_filter.py:19; individually rounded like the Python expression)
__global__ __launch_bounds__(256) void relay_formation_kernel(double alpha, double *__restrict__ st, const double *__restrict__ form, int B)
{
    const int v = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (v >= B) return;
    const bool init = st[v * 4 + 3] != 0.0;
    const double oma = 1.0 - alpha;
    for (int a = 0; a < 3; ++a) {
        const double u = form[v * 3 + a];
        const double y = init ? st[v * 4 + a] : u;
        st[v * 4 + a] = nc_add(nc_mul(alpha, y), nc_mul(oma, u));
    }
    st[v * 4 + 3] = 1.0;
}

// follower reference = leader window with the filtered offset added to the positions.  HBM-bound: 3360 B per instance at N = 20.
// A dense copy in 16-byte pieces (piece p: row p / 5, doubles 2 (p % 5) and + 1 of it), the offset added to pieces 0 (x, y) and 1 (z):
// every instruction of a wave covers 1 KB of contiguous memory on both sides, the output streams (st_stream).  As one thread per ROW
// -- five pieces at an 80-byte lane stride -- each instruction touched 40 lines a fifth each: 195 us for 262 144 windows (0.58 of the
// roof); that form with streaming loads / stores: 685 us, every partial line fetched again by each of its five instructions.
enum { RELAY_UNROLL = 4 };
__global__ __launch_bounds__(256) void relay_reference_kernel(const double *__restrict__ st, const double *__restrict__ xr_lead,
                                                              double *__restrict__ xr_out, int rows, int np1)
{
    const size_t total = (size_t)rows * 5;
    const size_t p0 = (size_t)blockIdx.x * (256 * RELAY_UNROLL) + threadIdx.x;
    const double2 *src = reinterpret_cast<const double2 *>(xr_lead);
    double2 *dst = reinterpret_cast<double2 *>(xr_out);
    double2 v[RELAY_UNROLL], o[RELAY_UNROLL];
#pragma unroll
    for (int j = 0; j < RELAY_UNROLL; ++j) {
        const size_t p = p0 + (size_t)j * 256, pc = p < total ? p : total - 1;
        const int r = (int)(pc / 5), c = (int)(pc - (size_t)r * 5), inst = r / np1;
        v[j] = src[pc];
        o[j] = make_double2(0.0, 0.0);
        if (c == 0) o[j] = *reinterpret_cast<const double2 *>(st + (size_t)inst * 4);        // (ox, oy)
        else if (c == 1) o[j].x = st[(size_t)inst * 4 + 2];                                   // (oz, -)
    }
#pragma unroll
    for (int j = 0; j < RELAY_UNROLL; ++j) {
        const size_t p = p0 + (size_t)j * 256;
        const int c = (int)(p % 5);
        double2 w = v[j];
        if (c == 0) { w.x += o[j].x; w.y += o[j].y; }
        else if (c == 1) w.x += o[j].x;
        if (p < total) st_stream(dst + p, w);
    }
}

// ------------------------------------------------------------------------------------------ f4 kernel
// plant: the OCP's own dynamics (nmpc_body_rate_ctl.py:147-158 + f/mass), RK4 substeps, quaternion renormalised
__device__ __forceinline__ void plant_f(const double *x, const double *u, const double *acc, double *d)
{
    const double qw = x[6], qx = x[7], qy = x[8], qz = x[9];
    d[0] = x[3]; d[1] = x[4]; d[2] = x[5];
    d[3] = 2.0 * (qx * qz + qw * qy) * u[3] + acc[0];
    d[4] = 2.0 * (qy * qz - qw * qx) * u[3] + acc[1];
    d[5] = (1.0 - 2.0 * qx * qx - 2.0 * qy * qy) * u[3] + acc[2];
    d[6] = (-u[0] * qx - u[1] * qy - u[2] * qz) * 0.5;
    d[7] = (u[0] * qw + u[2] * qy - u[1] * qz) * 0.5;
    d[8] = (u[1] * qw - u[2] * qx + u[0] * qz) * 0.5;
    d[9] = (u[2] * qw + u[1] * qx - u[0] * qy) * 0.5;
}

__global__ __launch_bounds__(256) void plant_kernel(double *__restrict__ x, const double *__restrict__ u, const double *__restrict__ f,
                                                    double h, int sub, double inv_mass, double g, int B)
{
    const int v = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (v >= B) return;
    double xv[10], uv[4], acc[3] = {0.0, 0.0, -g};
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const double2 t = reinterpret_cast<const double2 *>(x)[(size_t)v * 5 + i];
        xv[2 * i] = t.x; xv[2 * i + 1] = t.y;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const double2 t = reinterpret_cast<const double2 *>(u)[(size_t)v * 2 + i];
        uv[2 * i] = t.x; uv[2 * i + 1] = t.y;
    }
    if (f) { acc[0] = f[v * 3] * inv_mass; acc[1] = f[v * 3 + 1] * inv_mass; acc[2] = f[v * 3 + 2] * inv_mass - g; }
    for (int s = 0; s < sub; ++s) {
        double k1[10], k2[10], k3[10], k4[10], xs[10];
        plant_f(xv, uv, acc, k1);
#pragma unroll
        for (int i = 0; i < 10; ++i) xs[i] = xv[i] + 0.5 * h * k1[i];
        plant_f(xs, uv, acc, k2);
#pragma unroll
        for (int i = 0; i < 10; ++i) xs[i] = xv[i] + 0.5 * h * k2[i];
        plant_f(xs, uv, acc, k3);
#pragma unroll
        for (int i = 0; i < 10; ++i) xs[i] = xv[i] + h * k3[i];
        plant_f(xs, uv, acc, k4);
#pragma unroll
        for (int i = 0; i < 10; ++i) xv[i] += h / 6.0 * (k1[i] + 2.0 * k2[i] + 2.0 * k3[i] + k4[i]);
    }
    const double n = sqrt(xv[6] * xv[6] + xv[7] * xv[7] + xv[8] * xv[8] + xv[9] * xv[9]);
#pragma unroll
    for (int i = 6; i < 10; ++i) xv[i] /= n;
#pragma unroll
    for (int i = 0; i < 5; ++i) reinterpret_cast<double2 *>(x)[(size_t)v * 5 + i] = make_double2(xv[2 * i], xv[2 * i + 1]);
}

// ------------------------------------------------------------------------------------------ f1 kernel
// Reference window generation (the step before the path): per vehicle a piecewise polynomial trajectory
// (TrajCoefficients.msg) is evaluated at the N+1 node times t + k*dt and pushed through the differential-flatness map;
// what NMPCRefPublisher.get_nmpc_pts returns (pt_pub/pt_publisher.py:79-103, base_pt_publisher.py:81-133,
// diff_flatness :188-248, traj_full_pt_2_x_u :115-146).  One thread per (vehicle, node): reads its segment's 28
// coefficients (224 contiguous bytes, shared by the neighbouring nodes of the vehicle), writes 80 + 32 contiguous bytes.
struct RefCfg { int B, N, n_seg; double dt, mass, g, toff; };   // toff: added to every vehicle's node-0 time (rollouts)

// Value and first ND - 1 derivatives of sum_i c[i] s^i at s by repeated synthetic division (the Taylor shift): pass k divides the
// previous pass's quotient by (x - s) once more and leaves p^(k)(s) / k! -- NC_ - 1 - k fused multiply-adds, 22 for the four values of
// a septic against 43 when every derivative is a Horner pass of its own with the factors i (i-1) .. applied on the way
// (get_poly_params + _get_output_value, base_pt_publisher.py:102-133, polym_optimizer.py:104-139, evaluate every power, factor and
// term separately: ~160 multiplies and adds per reference point).  out[k] = p^(k)(s) / k!.
// Measured (round 3): folding factors and 1 / tseg^d into per-derivative coefficient blocks on the host (1 operation per
// coefficient, but 85 instead of 28 loads per point) made the kernel SLOWER: the loads cost more than the arithmetic saved.
template <int NC_, int ND>
__device__ __forceinline__ void taylor_shift(const double *__restrict__ c, double s, double out[ND])
{
    double b[NC_];
#pragma unroll
    for (int i = 0; i < NC_; ++i) b[i] = c[i];
#pragma unroll
    for (int k = 0; k < ND; ++k) {
#pragma unroll
        for (int i = NC_ - 2; i >= k; --i) b[i] = fma(b[i + 1], s, b[i]);
        out[k] = b[k];
    }
}

// ---- one reference point, in three pieces shared by every kernel that makes one (so that they all make the SAME point, bit for bit:
// the control step that computes its window's newest node itself -- rti_kernel<..., TICK> -- must agree with the list kernels):
//   seg_locate   which polynomial segment holds trajectory time t (base_pt_publisher.py:93-100)
//   traj_chain   one of the 14 polynomial values of a trajectory point: p / v / a / j of one axis, yaw, yaw rate (:102-133)
//   flatness_xu  the differential-flatness map of the point (pt_publisher.py:188-248) and the x / u packing (:115-146)
// seg_hint (or null): the vehicle's segment at its previous point -- control ticks move forward 20 ms at a time, so it is nearly always
// still the one: two loads confirm it instead of a search over time_cum.  Returns -1 past the end of the trajectory.
__device__ __forceinline__ int seg_locate(int n_seg, const double *__restrict__ tc, double t, int hint)
{
    if (t >= tc[n_seg]) return -1;                        // base_pt_publisher.py:93-94: hover at final_pt after the end
    int idx = hint < 0 ? 0 : (hint >= n_seg ? n_seg - 1 : hint);
    if ((idx == 0 || !(tc[idx] > t)) && tc[idx + 1] > t) return idx;
    // :100: first i with time_cum[i] > t, minus one -- time_cum ascends, so that is (entries of 0 .. n_seg-1 not above t) - 1.
    // Counted eight independent loads at a time: a search loop is a chain of dependent global loads, ~0.6 us each.
    idx = 0;
    for (int i = 0; i < n_seg; i += 8) {
        double v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = tc[i + j < n_seg ? i + j : n_seg];
#pragma unroll
        for (int j = 0; j < 8; ++j) idx += (i + j < n_seg && !(v[j] > t)) ? 1 : 0;
    }
    return idx > 0 ? idx - 1 : 0;
}

// value number c of a trajectory point at normalised segment time s; its = 1 / time_seg: c = 3 d + axis (d = derivative 0..3,
// axis 0..2) for c < 12, c = 12: yaw, c = 13: yaw rate.  ca = the value's own polynomial: chain_base(c) doubles into the segment's
// record of 28 coefficients, x(8) y(8) z(8) yaw(4).
__device__ __forceinline__ int chain_base(int c) { return c >= 12 ? 24 : 8 * (c % 3); }
// p, v, a, j of one axis (ca: its 8 coefficients) / yaw, yaw rate (ca: the 4 yaw coefficients): derivative d carries d! / time_seg^d
__device__ __forceinline__ void traj_axis(const double *__restrict__ ca, double s, double its, double out[4])
{
#pragma clang fp contract(off)
    double tl[4];
    taylor_shift<8, 4>(ca, s, tl);
    const double its2 = its * its;
    out[0] = tl[0]; out[1] = tl[1] * its; out[2] = tl[2] * (2.0 * its2); out[3] = tl[3] * (6.0 * (its2 * its));
}
__device__ __forceinline__ void traj_yaw(const double *__restrict__ ca, double s, double its, double out[2])
{
#pragma clang fp contract(off)
    double tl[2];
    taylor_shift<4, 2>(ca, s, tl);
    out[0] = tl[0]; out[1] = tl[1] * its;
}
__device__ __forceinline__ double traj_chain(const double *__restrict__ ca, int c, double s, double its)
{
    if (c >= 12) {
        double y[2];
        traj_yaw(ca, s, its, y);
        return c == 12 ? y[0] : y[1];
    }
    double v[4];
    traj_axis(ca, s, its, v);
    const int d = c / 3;
    return d == 0 ? v[0] : (d == 1 ? v[1] : (d == 2 ? v[2] : v[3]));
}

// ---- f64 helpers of the flatness map.  An IEEE divide or square root is a ~30-instruction dependent chain on gfx950 and the map
// has four and three of them, one behind the other, plus a library sincos (~150 instructions with its large-argument path): measured
// in the one-launch tick, where a single wave runs the map with nothing to overlap it, ~4 000 cycles of a 7 300-cycle prologue; in the
// list / window kernels the same chains are why "HBM-bound" kernels sat at 0.4 of the HBM roof.  These are seed + Newton forms
// (v_rcp_f64 / v_rsq_f64: 2^-26 relative or better; two steps -> ~1e-16, not correctly rounded) and a Cody-Waite sincos with the
// fdlibm kernel polynomials (|error| < 1 ulp for |x| < 1e5) -- deterministic, shared by every kernel that makes a reference point.
__device__ __forceinline__ double rcp_n(double a)
{
    double r = __builtin_amdgcn_rcp(a);
    r = fma(fma(-a, r, 1.0), r, r);
    r = fma(fma(-a, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ double rsqrt_n(double a)
{
#pragma clang fp contract(off)
    double y = __builtin_amdgcn_rsq(a);
    y = fma(y * 0.5, fma(-a * y, y, 1.0), y);
    y = fma(y * 0.5, fma(-a * y, y, 1.0), y);
    return y;
}
__device__ __forceinline__ void sincos_n(double x, double *sn, double *cs)
{
#pragma clang fp contract(off)
    const double k = rint(x * 6.36619772367581382433e-01);            // x * 2 / pi
    double r = fma(-k, 1.57079632673412561417e+00, x);                // pi / 2 in three pieces (fdlibm e_rem_pio2: pio2_1, pio2_2, pio2_3)
    r = fma(-k, 6.07710050630396597660e-11, r);
    r = fma(-k, 2.02226624871116645580e-21, r);
    const double z = r * r;
    // fdlibm k_sin / k_cos on |r| <= pi / 4
    const double ps = fma(z, fma(z, fma(z, fma(z, fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08), 2.75573137070700676789e-06),
                                        -1.98412698298579493134e-04), 8.33333333332248946124e-03), -1.66666666666666324348e-01);
    const double s0 = fma(z * r, ps, r);
    const double pc = fma(z, fma(z, fma(z, fma(z, fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09), -2.75573143513906633035e-07),
                                        2.48015872894767294178e-05), -1.38888888888741095749e-03), 4.16666666666666019037e-02);
    const double c0 = fma(z * z, pc, fma(-0.5, z, 1.0));
    const int q = (int)k & 3;
    const double s1 = (q & 1) ? c0 : s0, c1 = (q & 1) ? s0 : c0;
    *sn = (q & 2) ? -s1 : s1;
    *cs = ((q + 1) & 2) ? -c1 : c1;
}

// Every product-sum below is written out (fma where one is wanted, contraction off otherwise): the compiler's own choice of which
// multiplies to fuse depends on the surrounding code, and two kernels must not differ in the last bit of a reference point.
__device__ __forceinline__ void flatness_xu(double mass, double g, const double pvaj[12], double yaw, double yawd, double xv[10], double uv[4])
{
#pragma clang fp contract(off)
    const double td[3] = {pvaj[6], pvaj[7], pvaj[8] + g};
    const double tn2 = fma(td[0], td[0], fma(td[1], td[1], td[2] * td[2]));
    const double rtn = rsqrt_n(tn2), tn = tn2 * rtn;
    const double zb[3] = {td[0] * rtn, td[1] * rtn, td[2] * rtn};
    double sy, cy;
    sincos_n(yaw, &sy, &cy);
    // z_b x x_c with x_c = [cos yaw, sin yaw, 0]
    const double zx[3] = {-(zb[2] * sy), zb[2] * cy, fma(zb[0], sy, -(zb[1] * cy))};
    const double rnzx = rsqrt_n(fma(zx[0], zx[0], fma(zx[1], zx[1], zx[2] * zx[2])));
    const double yb[3] = {zx[0] * rnzx, zx[1] * rnzx, zx[2] * rnzx};
    const double xb[3] = {fma(yb[1], zb[2], -(yb[2] * zb[1])), fma(yb[2], zb[0], -(yb[0] * zb[2])), fma(yb[0], zb[1], -(yb[1] * zb[0]))};
    const double zj = fma(zb[0], pvaj[9], fma(zb[1], pvaj[10], zb[2] * pvaj[11]));
    double ho[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) ho[i] = rtn * fma(-zj, zb[i], pvaj[9 + i]);      // mass / u1 = 1 / |t_des|
    const double wp = -fma(ho[0], yb[0], fma(ho[1], yb[1], ho[2] * yb[2]));
    const double wq = fma(ho[0], xb[0], fma(ho[1], xb[1], ho[2] * xb[2]));
    const double wr = yawd * zb[2];
    // tf.transformations.quaternion_from_matrix on [x_b y_b z_b] (ROS geometry; restated): R[i][0..2] = xb[i], yb[i], zb[i]
    const double R[3][3] = {{xb[0], yb[0], zb[0]}, {xb[1], yb[1], zb[1]}, {xb[2], yb[2], zb[2]}};
    // (q[3] = w; i = index of the largest diagonal entry, j = i + 1, k = i + 2 mod 3.  Written with selects over the three cases:
    // dynamically indexed local arrays live in scratch memory on this target.)
    double q0, q1, q2, q3, tt = R[0][0] + R[1][1] + R[2][2] + 1.0;
    if (tt > 1.0) {
        q3 = tt; q2 = R[1][0] - R[0][1]; q1 = R[0][2] - R[2][0]; q0 = R[2][1] - R[1][2];
    } else {
        const bool c1 = R[1][1] > R[0][0];
        const bool c2 = R[2][2] > (c1 ? R[1][1] : R[0][0]);
        // case (i, j, k) = (0,1,2), (1,2,0), (2,0,1)
        const double t0 = R[0][0] - (R[1][1] + R[2][2]) + 1.0, t1 = R[1][1] - (R[2][2] + R[0][0]) + 1.0, t2 = R[2][2] - (R[0][0] + R[1][1]) + 1.0;
        const double s01 = R[0][1] + R[1][0], s12 = R[1][2] + R[2][1], s20 = R[2][0] + R[0][2];
        const double d21 = R[2][1] - R[1][2], d02 = R[0][2] - R[2][0], d10 = R[1][0] - R[0][1];
        tt = c2 ? t2 : (c1 ? t1 : t0);
        q0 = c2 ? s20 : (c1 ? s01 : t0);       // q[i] = tt, q[j] = R[i][j] + R[j][i], q[k] = R[k][i] + R[i][k]
        q1 = c2 ? s12 : (c1 ? t1 : s01);
        q2 = c2 ? t2 : (c1 ? s12 : s20);
        q3 = c2 ? d10 : (c1 ? d02 : d21);      // q[3] = R[k][j] - R[j][k]
    }
    const double qs = 0.5 * rsqrt_n(tt);
    // [qw, qx, qy, qz] (pt_publisher.py:237-240, :115-128); u = [p, q, r, collective_force / mass] (:138-145)
    xv[0] = pvaj[0]; xv[1] = pvaj[1]; xv[2] = pvaj[2]; xv[3] = pvaj[3]; xv[4] = pvaj[4]; xv[5] = pvaj[5];
    xv[6] = q3 * qs; xv[7] = q0 * qs; xv[8] = q1 * qs; xv[9] = q2 * qs;
    uv[0] = wp; uv[1] = wq; uv[2] = wr; uv[3] = tn;            // collective_force / mass = (|t_des| mass) / mass (:138-145)
}

// One reference point: trajectory of vehicle b at trajectory time t -> x[10] = [p, v, qw, qx, qy, qz], u[4] = [wx, wy, wz, c]
// (get_traj_pt, base_pt_publisher.py:81-133; diff_flatness, pt_publisher.py:188-248; traj_full_pt_2_x_u, :115-146)
__device__ __forceinline__ void ref_point(const RefCfg &cf, const double *__restrict__ coeff, const double *__restrict__ tcum,
                                          const double *__restrict__ tseg, const double *__restrict__ fpt, int b, double t,
                                          double xv[10], double uv[4], int *__restrict__ seg_hint = nullptr)
{
    const double *tc = tcum + (size_t)b * (cf.n_seg + 1);
    double pvaj[12], yaw = 0.0, yawd = 0.0;
#pragma unroll
    for (int i = 0; i < 12; ++i) pvaj[i] = 0.0;
    const int idx = seg_locate(cf.n_seg, tc, t, seg_hint ? seg_hint[b] : 0);
    if (idx < 0) {
        for (int i = 0; i < 3; ++i) pvaj[i] = fpt[(size_t)b * 3 + i];
    } else {
        if (seg_hint) seg_hint[b] = idx;
        const double its = rcp_n(tseg[(size_t)b * cf.n_seg + idx]);
        // the record as 14 16-byte loads (it starts at a multiple of 224 bytes): the load unit's time per instruction does not depend
        // on the width, and these rows are bound by the number of load instructions (see ref_window_kernel)
        double rec[28];
        {
            const double2 *r2 = reinterpret_cast<const double2 *>(coeff + ((size_t)b * cf.n_seg + idx) * 28);
#pragma unroll
            for (int i = 0; i < 14; ++i) { const double2 v = r2[i]; rec[2 * i] = v.x; rec[2 * i + 1] = v.y; }
        }
        double s;
        {
#pragma clang fp contract(off)
            s = (t - tc[idx]) * its;                      // :102-103
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            double v[4];
            traj_axis(rec + 8 * a, s, its, v);
            pvaj[a] = v[0]; pvaj[3 + a] = v[1]; pvaj[6 + a] = v[2]; pvaj[9 + a] = v[3];
        }
        double y[2];
        traj_yaw(rec + 24, s, its, y);
        yaw = y[0]; yawd = y[1];
    }
    flatness_xu(cf.mass, cf.g, pvaj, yaw, yawd, xv, uv);
}

#define REF_ROWS 64     // rows (vehicle, node) per workgroup = one wave: small batches spread over all CUs
// What binds it (round 5, knock-out builds at 262 144 windows, 187 us as it stood): loads + arithmetic alone 75 us, arithmetic + stores
// alone 111 us (5.5 TB/s of writes: the device's fill ceiling), loads + stores WITHOUT the arithmetic 168 us -- the two memory phases
// did not overlap across waves: the 616 MB of output went through L2 as ordinary dirty lines and the dependent load rounds of the
// other waves (time -> segment -> coefficients) queued behind them.  Streaming stores (st_stream): 127-137 us, 0.62-0.68 of the roof.
// Not what binds it, each measured: the number of load instructions (coefficients as 14 16-byte loads: +2.6 %; from the scalar
// cache instead, a timing experiment: nothing), the number of dependent rounds (a branch-free scan that merges two of the three rounds
// but requests six more time_cum entries: 13 % SLOWER), occupancy (64 registers for 8 waves spills and is slower).  ONE 5 KB staging
// buffer used twice (x rows, then u rows): 7 KB per wave had capped a CU at 22 workgroups.
__global__ __launch_bounds__(REF_ROWS)
void ref_window_kernel(RefCfg cf, const double *__restrict__ coeff, const double *__restrict__ tcum,
                       const double *__restrict__ tseg, const double *__restrict__ fpt,
                       const double *__restrict__ tq, double *__restrict__ xr, double *__restrict__ ur)
{
    // Each lane produces 80 + 32 contiguous bytes; written directly that is a 16-byte store at an 80-byte lane stride
    // (one fifth of every cache line per instruction).  The wave's rows are contiguous in xr (and, minus the node-N
    // rows, in ur), so the outputs are transposed through LDS and leave as dense 1024-byte wave stores.
    __shared__ __attribute__((aligned(16))) double sx[REF_ROWS * 10];
    const int lane = (int)threadIdx.x;
    const int row0 = (int)blockIdx.x * REF_ROWS;
    const int np1 = cf.N + 1, nrows = cf.B * np1;
    const int row = row0 + lane < nrows ? row0 + lane : nrows - 1;      // tail lanes recompute the last row, never store
    const int b = row / np1, k = row - b * np1;
    const double t = (tq ? tq[b] : 0.0) + cf.toff + k * cf.dt;
    double xv[10], uv[4];
    ref_point(cf, coeff, tcum, tseg, fpt, b, t, xv, uv);
#pragma unroll
    for (int i = 0; i < 5; ++i) reinterpret_cast<double2 *>(sx)[lane * 5 + i] = make_double2(xv[2 * i], xv[2 * i + 1]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const int rows_here = nrows - row0 < REF_ROWS ? nrows - row0 : REF_ROWS;
    // 16 bytes per lane and store (rows are 80 / 32 bytes: both arrays stay 16-byte aligned at every row)
    double2 *xg = reinterpret_cast<double2 *>(xr + (size_t)row0 * 10);
    const double2 *s2 = reinterpret_cast<const double2 *>(sx);
    for (int i = lane; i < rows_here * 5; i += REF_ROWS) st_stream(xg + i, s2[i]);
    // ur has no node-N rows: the number of u rows before row (b, k) is b N + k = row - b
    const int ufirst = row0 - row0 / np1;
    const int uslot = (row - b) - ufirst;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_barrier();                                        // every lane has read its x pieces: the buffer is free
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (k < cf.N && row0 + lane < nrows) {
        reinterpret_cast<double2 *>(sx)[uslot * 2] = make_double2(uv[0], uv[1]);
        reinterpret_cast<double2 *>(sx)[uslot * 2 + 1] = make_double2(uv[2], uv[3]);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const int rend = row0 + rows_here;
    const int nu = (rend - rend / np1) - ufirst;                                 // u rows among [row0, rend)
    double2 *ug = reinterpret_cast<double2 *>(ur + (size_t)ufirst * 4);
    for (int i = lane; i < nu * 2; i += REF_ROWS) st_stream(ug + i, s2[i]);
}

// ---- f1, the reference's own bookkeeping: NMPCRefPublisher keeps a list of `ring` = step N + 1 reference points per vehicle,
// ts_nmpc apart (pt_publisher.py:36-38, params/nmpc_params.py:40-43; step = 5); every control tick drops the oldest and appends
// the point at ros_t + T_horizon (:78-97); the controller's window is every step-th entry (:99-103).
// Device layout (round 5): the window of tick n is the list entries with ABSOLUTE index n, n + step, .., n + step N (entry j =
// the j-th point ever put into the list) -- all of one residue class mod step.  So the list is kept PHASE-MAJOR, a short ring of
// N + 1 positions per phase, every entry stored twice, N + 1 positions apart:
//     x ring [B][step][2 (N+1)][10]      u ring [B][step][2 (N+1)][4]
//     entry j -> phase j % step, positions (j / step) % (N+1) and + (N+1)
// and every window is N + 1 CONTIGUOUS x rows (N u rows) starting at position (n / step) % (N+1) of phase n % step: the control
// step reads its reference window -- and a neighbour's -- straight out of the list (instance pitch = RingGeom::px / pu doubles),
// there is no window copy on the control tick's path, and the stand-alone window call is a dense copy.  `n` lives on the host
// (ndp_handle::list_n) and is baked into each launch's arguments.

__device__ __forceinline__ void ring_store(const RingGeom &rg, double *__restrict__ rx, double *__restrict__ ru, int b,
                                           unsigned long long j, const double xv[10], const double uv[4])
{
    const size_t s = rg.slot(j);
    double2 *x0 = reinterpret_cast<double2 *>(rx + (size_t)b * rg.px() + s * 10), *x1 = x0 + (size_t)rg.np1 * 5;
    double2 *u0 = reinterpret_cast<double2 *>(ru + (size_t)b * rg.pu() + s * 4), *u1 = u0 + (size_t)rg.np1 * 2;
#pragma unroll
    for (int c = 0; c < 5; ++c) { const double2 v = make_double2(xv[2 * c], xv[2 * c + 1]); x0[c] = v; x1[c] = v; }
#pragma unroll
    for (int c = 0; c < 2; ++c) { const double2 v = make_double2(uv[2 * c], uv[2 * c + 1]); u0[c] = v; u1[c] = v; }   // (plain: 80- / 32-byte pieces, partial lines -- streamed they cost the list advance a quarter of its rate)
}

// Fills list entries: point i of vehicle b at trajectory time (tq ? tq[b] : 0) + toff + i * tstep becomes entry j0 + i;
// dup0 also makes point 0 entry j0 - 1 (_gen_long_list_w_traj's duplicate, :73-74).
__global__ __launch_bounds__(256) void ref_list_fill_kernel(RefCfg cf, const double *__restrict__ coeff, const double *__restrict__ tcum,
                                                            const double *__restrict__ tseg, const double *__restrict__ fpt,
                                                            const double *__restrict__ tq, double tstep, int npts, unsigned long long j0,
                                                            RingGeom rg, int dup0, double *__restrict__ rx, double *__restrict__ ru)
{
    const int id = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (id >= cf.B * npts) return;
    const int b = id / npts, i = id - b * npts;
    double xv[10], uv[4];
    // one point per vehicle = the per-tick advance: the segment hint applies (it lives behind final_pt, see ndp_ref_set_trajectory)
    int *hint = npts == 1 ? reinterpret_cast<int *>(const_cast<double *>(fpt + (size_t)cf.B * 3 + (size_t)cf.B * SEGC_PER)) : nullptr;
    ref_point(cf, coeff, tcum, tseg, fpt, b, (tq ? tq[b] : 0.0) + cf.toff + i * tstep, xv, uv, hint);
    ring_store(rg, rx, ru, b, j0 + (unsigned long long)i, xv, uv);
    if (dup0 && i == 0) ring_store(rg, rx, ru, b, j0 - 1, xv, uv);
}

// gen_fix_pt_ref (pt_publisher.py:40-55): every entry = the odometry state, u = [0, 0, 0, c_hover]; one thread per stored row
__global__ __launch_bounds__(256) void ref_list_fix_kernel(const double *__restrict__ x_odom, double c_hover, int B, RingGeom rg,
                                                           double *__restrict__ rx, double *__restrict__ ru)
{
    const int per = rg.step * 2 * rg.np1;
    const int id = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (id >= B * per) return;
    const int b = id / per;
    const double2 *s = reinterpret_cast<const double2 *>(x_odom) + (size_t)b * 5;
    double2 *dx = reinterpret_cast<double2 *>(rx) + (size_t)id * 5, *du = reinterpret_cast<double2 *>(ru) + (size_t)id * 2;
#pragma unroll
    for (int c = 0; c < 5; ++c) dx[c] = s[c];
    du[0] = make_double2(0.0, 0.0);
    du[1] = make_double2(0.0, c_hover);
}

// get_nmpc_ref_from_long_list (:99-103) as a stand-alone call: the window of tick n -> xr[B][N+1][10], ur[B][N][4].  Both sides
// are contiguous per vehicle (see RingGeom): a dense copy, 16 bytes per lane -- 5 (N+1) + 2 N pieces per vehicle.
enum { WIN_UNROLL = 4 };      // 16-byte pieces per thread, a block apart: four loads in flight per lane before the first store
__global__ __launch_bounds__(256) void ref_list_window_kernel(const double *__restrict__ rx, const double *__restrict__ ru, RingGeom rg,
                                                              unsigned long long n, int B, double *__restrict__ xr, double *__restrict__ ur)
{
    const int N = rg.np1 - 1, nxp = 5 * rg.np1, per = nxp + 2 * N;
    const size_t total = (size_t)B * per, s = rg.slot(n);
    const size_t id0 = (size_t)blockIdx.x * (256 * WIN_UNROLL) + threadIdx.x;
    double2 v[WIN_UNROLL];
    double2 *dst[WIN_UNROLL];
#pragma unroll
    for (int j = 0; j < WIN_UNROLL; ++j) {
        const size_t id = id0 + (size_t)j * 256;
        const size_t idc = id < total ? id : total - 1;
        const int b = (int)(idc / per), e = (int)(idc - (size_t)b * per);
        const double2 *src = e < nxp ? reinterpret_cast<const double2 *>(rx + (size_t)b * rg.px() + s * 10) + e
                                     : reinterpret_cast<const double2 *>(ru + (size_t)b * rg.pu() + s * 4) + (e - nxp);
        dst[j] = id < total ? (e < nxp ? reinterpret_cast<double2 *>(xr) + (size_t)b * nxp + e
                                       : reinterpret_cast<double2 *>(ur) + (size_t)b * 2 * N + (e - nxp)) : nullptr;
        v[j] = *src;
    }
#pragma unroll
    for (int j = 0; j < WIN_UNROLL; ++j)
        if (dst[j]) st_stream(dst[j], v[j]);
}


// ------------------------------------------------------------------------------------------ the node's control tick (ndp_tick)
// nmpc_node.py:211-231 for every vehicle of the handle, on the device, references resident: per tick the host hands over the
// odometry states (80 B per vehicle) and a few scalars; everything else the tick needs is already in HBM.
//   tick_pre_kernel  (one thread per vehicle): the reference list's advance -- the point at t + T_horizon becomes the list's newest
//                    entry (get_nmpc_pts, pt_publisher.py:79-97), which is also node N of this tick's window -- and the hover-
//                    throttle estimator's update (hover_throttle_callback, nmpc_node.py:251-253) from vz and the thrust command of
//                    the previous tick
//   rti_kernel       the control step: x0 = the odometry rows, xr / ur / the neighbour's window straight out of the list; its last
//                    store is nmpc_u_2_att_tgt (:273-283): [wx, wy, wz, c mass / k_throttle] where the host reads it (RtiIo::cmd),
//                    the thrust kept on the device for the next estimator update.  (A third launch for that -- tick_post_kernel,
//                    the first form -- cost 4.6 us per tick in the trace for 32 bytes per vehicle.)
struct TickPre {
    RefCfg cf;
    const double *coeff, *tcum, *tseg, *fpt;
    int *seg_hint;                     // [B] the segment each vehicle's last point lay in (ref_point)
    const double *t;                   // [B] trajectory time of the tick, or null: t_all for every vehicle
    double t_all;
    int advance;                       // 0: the list is not advanced
    unsigned long long j_new;          // absolute index of the entry the new point becomes
    RingGeom rg;
    double *rx, *ru;
    ThrCfg thr;
    double *st;                        // estimator state, SoA [8][B]
    const double *vz;                  // vz of vehicle b at vz[b * vz_pitch]: a [B] array (pitch 1) or column 5 of the odometry rows (pitch 10)
    size_t vz_pitch;
    const double *throttle;            // [B]: the thrust command sent last tick (caller's array, or the one the control step kept)
    int est;                           // run the estimator this tick
    // ndp_xchg_tick_begin: the advanced window's position / velocity columns -> pv[b][N+1][6] in the same launch (one launch less on a
    // path that is bound by the host's launches); the window starts at list slot pv_slot, its node N is the point made here
    double *pv = nullptr;
    size_t pv_slot = 0;
};

__global__ __launch_bounds__(64) void tick_pre_kernel(TickPre a)
{
    const int b = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (b >= a.cf.B) return;
    double vzv = 0.0, th = 0.0;
    if (a.est) { vzv = a.vz[(size_t)b * a.vz_pitch]; th = a.throttle[b]; }      // (requested before the polynomial work)
    if (a.advance) {
        double xv[10], uv[4];
        const int N = a.rg.np1 - 1;
        const double2 *src = reinterpret_cast<const double2 *>(a.rx + (size_t)b * a.rg.px() + a.pv_slot * 10);
        double2 *dst = reinterpret_cast<double2 *>(a.pv) + (size_t)b * a.rg.np1 * 3;
        if (a.pv) {                    // nodes 0 .. N-1 lie in the list since earlier ticks: copied under the polynomial work
#pragma unroll 4
            for (int k = 0; k < N; ++k) {
                const double2 v0 = src[k * 5], v1 = src[k * 5 + 1], v2 = src[k * 5 + 2];
                dst[k * 3] = v0; dst[k * 3 + 1] = v1; dst[k * 3 + 2] = v2;
            }
        }
        ref_point(a.cf, a.coeff, a.tcum, a.tseg, a.fpt, b, (a.t ? a.t[b] : a.t_all) + a.cf.toff, xv, uv, a.seg_hint);
        ring_store(a.rg, a.rx, a.ru, b, a.j_new, xv, uv);
        if (a.pv) {                    // node N: the point itself (not read back)
            dst[N * 3] = make_double2(xv[0], xv[1]); dst[N * 3 + 1] = make_double2(xv[2], xv[3]); dst[N * 3 + 2] = make_double2(xv[4], xv[5]);
        }
    }
    if (a.est) (void)throttle_update_one(a.thr, a.st, (size_t)a.cf.B, b, vzv, th);
}

// ---- the one-launch tick's prologue (rti_kernel<..., TICK>, see TickArgs), one wave per vehicle
// Lanes 0..13 evaluate the 14 polynomial values of the ego's new point (value c on lane c: the same traj_chain a list kernel calls),
// lanes 16..21 the position / velocity values of the neighbour's (orow >= 0); the values are collected with v_readlane and every lane
// runs the flatness map on them (uniform values: as long as one lane's work).  Lanes 0..13 then store the entry into the list (both
// copies), for the ticks to come.  x_new / u_new / nb_new are the same in every lane.
__device__ __forceinline__ double uniform_lane(double v, int l)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

// A per-vehicle cache of the trajectory's CURRENT and NEXT segment records, 2 x 32 doubles at segc + SEGC_PER v:
//     [0] time_cum[i]  [1] time_cum[i + 1]  [2] time_seg[i]  [3] i  [4 .. 31] the 28 coefficients of segment i
// and behind them [64 + 2 a], [65 + 2 a] = (time_cum[n_seg], final_pt[a]) for the axes a = 0..2 (one 16-byte load per lane)
// Found through the trajectory arrays a point costs two dependent memory round trips (segment index -> record), ~1 700 cycles
// each and nothing in the wave to hide them under; the cache's address depends on the vehicle only, so its loads are the launch's
// first and arrive under the weight transfer.  A vehicle moves on to its next segment every time_seg / 20 ms ticks (and in a batch
// of a thousand some vehicle does in every tick): that is slot 1, valid from the moment slot 0 was; the wave that crosses re-fills
// both slots -- one load per lane, requested in the prologue, stored behind the MLP phase (tick_cache_store), off everybody's
// critical path.  Anything else (the first tick after ndp_ref_set_trajectory -- the cache starts as NaNs --, a jump in time) takes
// seg_locate and the trajectory arrays, and re-fills the cache the same way.
__device__ __forceinline__ TickEarly tick_early(const TickArgs &ta, int inst, int orow, int lane)
{
    TickEarly te;
    const int c = (lane & 15) < 14 ? (lane & 15) : 13, q = (4 + chain_base(c)) >> 1;
    te.v = ((lane >> 4) & 1) && orow >= 0 ? orow : inst;
    // The load unit takes 16 cycles per instruction and wave of 64 (four lanes a cycle, whatever the width), and the four waves of a
    // compute unit share it: as 35 8-byte loads by all 64 lanes these requests alone kept it busy for 2 200 cycles.  Hence 16-byte
    // loads (13 of them) and only the 24 lanes whose values are looked at (lanes 0..13 ego, 16..21 neighbour).
    const tick_d2 *s0 = reinterpret_cast<const tick_d2 *>(ta.segc + (size_t)te.v * SEGC_PER);
    te.tv = ta.t_all;
    if (lane < 24) {
        te.h0[0] = s0[0]; te.h0[1] = s0[1]; te.h1[0] = s0[SEGC_SLOT / 2]; te.h1[1] = s0[SEGC_SLOT / 2 + 1];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int o = q + (c >= 12 ? (k & 1) : k);
            te.ca[k] = s0[o]; te.cn[k] = s0[SEGC_SLOT / 2 + o];
        }
        te.tf = s0[SEGC_SLOT + (c < 3 ? c : 0)];
    }
    // the ego's whole record, word `lane` (and constant `lane` in lanes 0..5): carried over into the copy the NEXT tick reads
    te.own = ta.segc[(size_t)inst * SEGC_PER + lane];
    te.ownc = ta.segc[(size_t)inst * SEGC_PER + 2 * SEGC_SLOT + (lane < 6 ? lane : 5)];
    if (lane < 24) {
        // (written as a branch: as a select the compiler picks between two ADDRESSES -- the argument's copy parked in scratch memory
        // for it -- and loads through a flat pointer)
        if (ta.t) te.tv = ta.t[te.v];
    }
    return te;
}

// Consume everything tick_early requested, HERE.  While an LDS-DMA transfer (global_load_lds) is in flight the compiler cannot use the
// in-order load counter: the instruction counts as "may touch memory AND LDS", and from its issue to the next full drain every wait of
// the wave -- for whatever value -- is emitted as s_waitcnt vmcnt(0), i.e. a wait for the whole 72-KB weight transfer.  Left to its
// first use inside tick_new_point the cached records therefore "arrived" only when the transfer was complete (6 400 cycles after
// entry; requested at 1 000) and the polynomial work ran BEHIND the transfer instead of under it.  Waiting for them in front of the
// transfer costs the transfer a later start (the records' own latency) and takes the polynomial work off the critical path.
__device__ __forceinline__ void tick_arrived(TickEarly &te)
{
    asm volatile("" : "+v"(te.tv), "+v"(te.h0[0]), "+v"(te.h0[1]), "+v"(te.h1[0]), "+v"(te.h1[1]), "+v"(te.tf), "+v"(te.own), "+v"(te.ownc));
    asm volatile("" : "+v"(te.ca[0]), "+v"(te.ca[1]), "+v"(te.ca[2]), "+v"(te.ca[3]), "+v"(te.cn[0]), "+v"(te.cn[1]), "+v"(te.cn[2]), "+v"(te.cn[3]));
}

// returns (in every lane) the value lane l must store into the ego's cache word l behind the MLP phase, valid if refill != 0
// store: false in the idle waves of a ragged last workgroup (they shadow the last instance for the barriers' sake and must not write)
__device__ __forceinline__ double tick_new_point(const TickArgs &ta, const TickEarly &te, int inst, int lane, bool store, double xv[10],
                                                 double uv[4], double nbv[6], int &refill, double &cfill, double *stamps)
{
    // profiling hook (ndp_debug_stamps): slots 17.. = the prologue's own timeline; each stamp waits for the value it names
    auto stamp_after = [&](int idx, double dep) {
        if (NDP_RARELY(stamps != nullptr)) {
            unsigned long long tk;
            asm volatile("s_nop 0\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tk) : "v"(dep) : "memory");
            if (lane == 0) stamps[idx] = (double)tk;
        }
    };
    const int c = (lane & 15) < 14 ? (lane & 15) : 13;
    const int v = te.v;
    const int S = ta.n_seg;
    const double lo0 = te.h0[0][0], hi0 = te.h0[0][1], ts0 = te.h0[1][0], i0 = te.h0[1][1];
    const double lo1 = te.h1[0][0], hi1 = te.h1[0][1], ts1 = te.h1[1][0], i1 = te.h1[1][1];
    stamp_after(17, lo0 + te.cn[3][1] + te.tf[1]);                    // the cached records are there
    const double t = te.tv + ta.toff;
    stamp_after(18, t);                                               // the time is there
    bool past = t >= te.tf[0];                                        // base_pt_publisher.py:93-94: hover at final_pt after the end
    // (the same tests as seg_locate's: segment 0 also serves times in front of time_cum[0]; a NaN bound -- the empty cache -- fails all three)
    const bool in0 = (i0 == 0.0 || !(lo0 > t)) && hi0 > t, in1 = !in0 && !(lo1 > t) && hi1 > t;
    const bool slow = lane < 24 && !past && !in0 && !in1;            // (lanes 24..63 hold nothing: tick_early)
    int idx = (int)(in0 ? i0 : i1);
    double tcs = in0 ? lo0 : lo1, tsg = in0 ? ts0 : ts1, fp = te.tf[1];
    // (opaque to the optimiser: left visible as "a loaded value or, on the slow path, another load", it parks the cached values in
    // scratch memory to select between ADDRESSES and load through a flat pointer)
    asm volatile("" : "+v"(tcs), "+v"(tsg), "+v"(fp));
    double ca[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) ca[i] = in1 ? te.cn[i >> 1][i & 1] : te.ca[i >> 1][i & 1];
    if (slow) {                                                       // (rare; per lane) through the trajectory arrays
        const KernargLate L;
        const double *tc = NDP_TA_LATE(L, tcum) + (size_t)v * (S + 1);
        const int cb = chain_base(c);
        idx = seg_locate(S, tc, t, -1);
        if (idx < 0) {                                                // (an empty cache does not know where the trajectory ends)
            past = true;
            idx = S - 1;
            fp = NDP_TA_LATE(L, fpt)[(size_t)v * 3 + (c < 3 ? c : 0)];
        } else {
            tcs = tc[idx]; tsg = NDP_TA_LATE(L, tseg)[(size_t)v * S + idx];
            const double *r = NDP_TA_LATE(L, coeff) + ((size_t)v * S + idx) * 28 + cb;
#pragma unroll
            for (int i = 0; i < 8; ++i) ca[i] = r[c >= 12 ? (i & 3) : i];
        }
    }
    double val = 0.0;
    if (past) {
        if (c < 3) val = fp;
    } else {
        const double its = rcp_n(tsg);
        double s;
        {
#pragma clang fp contract(off)
            s = (t - tcs) * its;
        }
        val = traj_chain(ca, c, s, its);
    }
    // the ego's cache: re-filled by this wave when its point did not come out of slot 0 (lane 0 belongs to the ego's group)
    refill = __builtin_amdgcn_readlane((in1 || slow) ? 1 : 0, 0);
    double fill = 0.0;
    cfill = te.ownc;
    if (refill) {
        const int ie = __builtin_amdgcn_readlane(idx, 0);
        const int sl = lane >> 5, f = lane & 31, i = ie + sl < S ? ie + sl : S - 1;
        const KernargLate L;
        const double *tce = NDP_TA_LATE(L, tcum) + (size_t)inst * (S + 1), *fpt = NDP_TA_LATE(L, fpt);
        cfill = lane < 6 ? ((lane & 1) ? fpt[(size_t)inst * 3 + (lane >> 1)] : tce[S]) : 0.0;   // (constants of the trajectory; stored by tick_cache_store)
        fill = f == 0 ? tce[i] : (f == 1 ? tce[i + 1] : (f == 2 ? NDP_TA_LATE(L, tseg)[(size_t)inst * S + i] : (f == 3 ? (double)i
                 : NDP_TA_LATE(L, coeff)[((size_t)inst * S + i) * 28 + (f - 4)])));
        if (ie + sl >= S && f == 1) fill = -1.0e300;                  // no segment behind the last one: slot 1 never matches (hi <= any t)
    }
    stamp_after(19, val);                                             // the lane's polynomial value
    double pvaj[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) pvaj[i] = uniform_lane(val, i);
    const double yaw = uniform_lane(val, 12), yawd = uniform_lane(val, 13);
#pragma unroll
    for (int i = 0; i < 6; ++i) nbv[i] = uniform_lane(val, 16 + i);
    stamp_after(20, pvaj[11] + nbv[5] + yawd);                        // collected over the lanes
    // where the entry goes (lane l: element l of x | u): the list's geometry is a late argument, requested HERE so that its fetch passes
    // under the flatness map instead of standing between the map and the barrier (-390 cycles per tick)
    gptr<double> dst;
    size_t mirror;
    {
        const KernargLate L;
        const size_t sl = NDP_TA_LATE(L, new_slot);
        const RingGeom rg = NDP_TA_LATE(L, rg);
        dst = (gptr<double>)(lane < 10 ? NDP_TA_LATE(L, rx) + (size_t)inst * rg.px() + sl * 10 + lane
                                       : NDP_TA_LATE(L, ru) + (size_t)inst * rg.pu() + sl * 4 + (lane - 10));
        mirror = (size_t)rg.np1 * (lane < 10 ? 10 : 4);
    }
    flatness_xu(ta.mass, ta.g, pvaj, yaw, yawd, xv, uv);
    stamp_after(21, xv[9] + uv[0]);                                   // flatness map done
    // the entry, for the windows of the ticks to come: element l of x | u from lane l
    // (v_writelane of the uniform values: written as a chain of selects on the lane id the compiler builds a table in scratch memory)
    int elo = 0, ehi = 0;
#pragma unroll
    for (int i = 0; i < 14; ++i) {
        const double w = i < 10 ? xv[i] : uv[i - 10];
        const int wl = __builtin_amdgcn_readfirstlane(__double2loint(w)), wh = __builtin_amdgcn_readfirstlane(__double2hiint(w));
        asm("v_writelane_b32 %0, %1, %2" : "+v"(elo) : "s"(wl), "n"(i));
        asm("v_writelane_b32 %0, %1, %2" : "+v"(ehi) : "s"(wh), "n"(i));
    }
    const double e = __hiloint2double(ehi, elo);
    if (store && lane < 14) {
        dst[0] = e;
        dst[mirror] = e;
    }
    return fill;
}

// The cache has two copies.  A launch READS one (its own record and its neighbour's, at entry) and WRITES the other -- every vehicle's
// record, every advancing tick: re-filled when the vehicle crossed into its next segment, carried over otherwise -- and the host swaps
// them between ticks.  (With one copy written in place, the neighbour's wave -- another workgroup, possibly another XCD, possibly a later
// round of a batch larger than the device -- could read a record while its owner re-filled it in the same launch: old header, new
// coefficients.  Nothing orders two workgroups of one launch; a kernel boundary orders everything.)
__device__ __forceinline__ void tick_cache_store(const TickArgs &ta, const TickEarly &te, int inst, int lane, int refill, double fill, double cfill)
{
    ta.segc_wr[(size_t)inst * SEGC_PER + lane] = refill ? fill : te.own;
    if (lane < 6) ta.segc_wr[(size_t)inst * SEGC_PER + 2 * SEGC_SLOT + lane] = cfill;
}

// hover_throttle_callback (nmpc_node.py:251-253) of this vehicle, by lane 0; returns k_throttle to every lane
__device__ __forceinline__ double tick_estimator(const TickArgs &ta, int inst, int B, int lane)
{
    double k = 0.0;
    const KernargLate L;
    if (lane == 0) k = throttle_update_one(NDP_TA_LATE(L, thr), NDP_TA_LATE(L, st), (size_t)B, inst, NDP_TA_LATE(L, vz)[(size_t)inst * NDP_TA_LATE(L, vz_pitch)],
                                           NDP_TA_LATE(L, throttle)[inst]);
    return uniform_lane(k, 0);
}

// ------------------------------------------------------------------------------------------ peer windows: per-tick publish
// peer_epoch.hpp's protocol on the device.  TWO launches per control tick and rank, in front of the control-step launch:
//  peer_publish_kernel (<= 256 blocks)
//   thread 0 of block 0   : reader role -- acknowledge tick t-1 in the neighbour's header (its slot may be overwritten now)
//   thread 0 of each block: owner role  -- wait until the own slot t & 1 is free (the reader's acknowledgement of tick t-2)
//   all threads           : copy this tick's windows (src, the reference generator's output) into the own slot, plain stores;
//                           the end of the launch is what makes them visible system-wide
//  peer_epoch_kernel (one wave)
//   epoch[t & 1] := t (release, system scope), then -- reader role -- wait for the neighbour's epoch of tick t.  When this launch
//   has completed, the control-step kernel launched next on the same stream may read the neighbour's slot t & 1 (kernel
//   boundary = system-scope acquire).
// (One launch that counts its finished blocks with an atomic and lets the last one publish was the first form: agent-scope atomics
// on one address serialise at 30-60 ns each -- 8.7 us for 1.7 MB of windows against 4.7 us this way, 15-114 us against 7-9 us
// for 20 MB depending on the block count; scripts/ubench/publish_copy.hip.)
struct PeerDevMem {
    typedef unsigned long long u64;
    static __device__ __forceinline__ u64 load(const u64 *p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM); }
    static __device__ __forceinline__ u64 peek(const u64 *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
    static __device__ __forceinline__ void store(u64 *p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
    static __device__ __forceinline__ u64 now_us() { return __builtin_amdgcn_s_memrealtime() / 100; }   // 100 MHz constant clock
};

struct PeerPubArgs {
    const double *src;              // [n] doubles: this rank's windows of the tick
    unsigned long long *own;        // this rank's buffer (header + two slots)
    unsigned long long *nb;         // the neighbour rank's buffer, mapped (== own with one rank)
    size_t n;
    int slot;                       // the slot parity the host baked into the control-step launch that follows
    unsigned timeout_us;
};

// No LDS and no barrier: the launch may have to run BESIDE a control step whose workgroups hold the CU's whole LDS (the one-tick-ahead
// form), where a workgroup that asks for any would wait for a control-step workgroup to leave.  Every wave reads the tick and waits for the
// slot by itself (the same two words).
__global__ __launch_bounds__(256) void peer_publish_kernel(PeerPubArgs a)
{
    typedef PeerProto<PeerDevMem> PP;
    typedef unsigned long long u64;
    u64 t = 0;
    if ((threadIdx.x & 63u) == 0) {
        t = PP::next_tick(a.own);                    // (the epochs only change in peer_epoch_kernel, behind this launch)
        if (blockIdx.x == 0 && threadIdx.x == 0) PP::ack_previous(a.nb, t);
        const bool freed = PP::wait_slot_free(a.own, t, a.timeout_us);
        if (blockIdx.x == 0 && threadIdx.x == 0 && !freed) a.own[PEER_W_STAT + PEER_STAT_ACK_TIMEOUT] += 1;
    }
    const unsigned par = (unsigned)__builtin_amdgcn_readfirstlane((int)(t & 1));
    double2 *dst = reinterpret_cast<double2 *>(reinterpret_cast<unsigned char *>(a.own) + peer_slot_offset(a.n, (int)par));
    const double2 *src = reinterpret_cast<const double2 *>(a.src);
    const size_t n2 = a.n / 2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
    if ((a.n & 1) && blockIdx.x == 0 && threadIdx.x == 0) reinterpret_cast<double *>(dst)[a.n - 1] = a.src[a.n - 1];
}

__global__ void peer_epoch_kernel(PeerPubArgs a)
{
    typedef PeerProto<PeerDevMem> PP;
    typedef unsigned long long u64;
    if (threadIdx.x != 0) return;
    const u64 t = PP::next_tick(a.own);
    PP::set_epoch(a.own, t);
    a.own[PEER_W_STAT + PEER_STAT_TICKS] = t;
    if ((int)(t & 1) != a.slot) a.own[PEER_W_STAT + PEER_STAT_DESYNC] += 1;
    if (!PP::wait_epoch(a.nb, t, a.timeout_us)) a.own[PEER_W_STAT + PEER_STAT_EPOCH_TIMEOUT] += 1;
}

// ------------------------------------------------------------------------------------------ RCCL exchange: the pack
// rows x [10] reference windows -> rows x [6]: the position / velocity columns, all that travels (downwash_nn.py:22).  One 16-byte
// piece per thread: piece p of row r = columns 2p, 2p + 1.
__global__ __launch_bounds__(256) void pack_pv_kernel(const double *__restrict__ xr, double *__restrict__ pv, size_t rows)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * 3) return;
    const size_t r = i / 3, p = i - r * 3;
    reinterpret_cast<double2 *>(pv)[i] = *reinterpret_cast<const double2 *>(xr + r * NX + 2 * p);
}

// the same for windows that lie in the reference list (ndp_tick): window row k of vehicle b = list row base + b * pitch + k * 10
__global__ __launch_bounds__(256) void pack_pv_list_kernel(const double *__restrict__ base, size_t pitch, int np1, double *__restrict__ pv, size_t B)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * (size_t)np1 * 3) return;
    const size_t r = i / 3, p = i - r * 3, b = r / (size_t)np1, k = r - b * (size_t)np1;
    reinterpret_cast<double2 *>(pv)[i] = *reinterpret_cast<const double2 *>(base + b * pitch + k * NX + 2 * p);
}

}  // namespace ndp

// ------------------------------------------------------------------------------------------ C-ABI
using namespace ndp;

// Address ranges this process mapped from other processes / GPUs (ndp_peer_open): neighbour windows inside one are read with
// system-scope loads (MlpArgs::other_sys).  A handful of entries, looked up once per launch.
struct PeerRange { uintptr_t lo, hi; };
static std::mutex g_peer_mu;
static std::vector<PeerRange> g_peer_ranges;
static int peer_mapped(const void *p)
{
    if (!p) return 0;
    const uintptr_t a = (uintptr_t)p;
    std::lock_guard<std::mutex> lk(g_peer_mu);
    for (const PeerRange &r : g_peer_ranges)
        if (a >= r.lo && a < r.hi) return 1;
    return 0;
}


// RTI_K(...): the rti_kernel instantiation to reference.  -DNDP_DEV_HEADLINE_ONLY (kernel development builds only, never
// the shipped library) collapses every instantiation but the reference configuration's two onto rti_kernel<3, 4, false, 20>, so
// that an experiment on the headline kernel compiles in 20 s instead of 3 min; such a library serves N = 20, n_rti = 1 only.
#ifdef NDP_DEV_HEADLINE_ONLY
template <int NSLOT, int WAVES, bool FUSED, int NC = 0, int PREC = 0, int NRC = (NC ? 1 : 0), int QMODE = 0, bool TICK = false>
#ifndef NDP_DEV_QMODE      // 1: study the work list's producer form (no interior-point code) in place of the in-place kernel
#define NDP_DEV_QMODE 0
#endif
struct RtiK { static constexpr auto fn = rti_kernel<3, (WAVES == 2 && NC == 20 ? 2 : 4), (FUSED && NC == 20 && QMODE == 0), 20, 0, 1, NDP_DEV_QMODE, (TICK && NC == 20 && QMODE == 0)>; };
#define RTI_K(...) (RtiK<__VA_ARGS__>::fn)
#elif defined(NDP_DEV_COND_ONLY)
// compile / register studies of the condensed study kernels: every instantiation collapses onto rti_kernel<5, 1, false, 0, 5 or 6>
template <int NSLOT, int WAVES, bool FUSED, int NC = 0, int PREC = 0, int NRC = (NC ? 1 : 0), int QMODE = 0, bool TICK = false>
struct RtiK { static constexpr auto fn = rti_kernel<5, 1, false, 0, (PREC == 6 ? 6 : 5)>; };
#define RTI_K(...) (RtiK<__VA_ARGS__>::fn)
#elif defined(NDP_DEV_N40_ONLY)
// register studies of config 5's shape (scripts/dev_regs.sh): every instantiation collapses onto rti_kernel<5, 2, false, 40, 0, 2, NDP_DEV_QMODE>
#ifndef NDP_DEV_QMODE
#define NDP_DEV_QMODE 0
#endif
template <int NSLOT, int WAVES, bool FUSED, int NC = 0, int PREC = 0, int NRC = (NC ? 1 : 0), int QMODE = 0, bool TICK = false>
struct RtiK { static constexpr auto fn = rti_kernel<5, 2, false, 40, 0, 2, NDP_DEV_QMODE>; };
#define RTI_K(...) (RtiK<__VA_ARGS__>::fn)
#else
#define RTI_K(...) (rti_kernel<__VA_ARGS__>)
#endif

// ---- host pack threads.  A host-array step first moves the caller's (pageable) arrays into a page-locked mirror the kernel can
// read; at batch 1024 that is 4.2 MB per step -- 56 us for one core, 26-30 us for eight (measured), against ~100 us for the
// kernel that then pulls them over PCIe.  The mirror is filled by a few persistent threads and the caller, a chunk
// (<= PACK_CHUNK bytes) at a time.
// len bytes as they lie, or -- rows > 0 -- `rows` rows of `len` bytes each taken from rows src_stride bytes apart (the neighbour
// windows: the 6 position / velocity columns of every 10-column row, all the gate and the network read)
struct PackJob { unsigned char *dst; const unsigned char *src; size_t len; size_t rows = 0, src_stride = 0; };
enum : size_t { PACK_CHUNK = (size_t)256 << 10 };
struct PackPool {
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv;
    std::atomic<const PackJob *> jobs{nullptr};
    std::atomic<int> njobs{0}, next{0}, active{0};
    std::unique_ptr<std::atomic<int>[]> done;
    int done_cap = 0;
    uint64_t gen = 0;
    std::atomic<uint64_t> gen_pub{0};   // gen, readable without the lock (the workers' polling phase)
    bool stop = false;
    static void cpu_relax()
    {
#if !defined(__HIP_DEVICE_COMPILE__) && (defined(__x86_64__) || defined(__i386__))
        __asm__ __volatile__("pause");
#endif
    }

    explicit PackPool(int n)
    {
        for (int i = 0; i < n; ++i) th.emplace_back([this] { work(); });
    }
    ~PackPool()
    {
        { std::lock_guard<std::mutex> lk(mu); stop = true; }
        cv.notify_all();
        for (auto &t : th) t.join();
    }
    bool take_one()
    {   // njobs is published last (release) and read first (acquire): a thread that sees a job count sees that list and its flags
        const int n = njobs.load(std::memory_order_acquire);
        if (n == 0) return false;
        const int i = next.fetch_add(1, std::memory_order_acq_rel);
        if (i >= n) return false;
        const PackJob &j = jobs.load(std::memory_order_relaxed)[i];
        if (j.rows == 0) memcpy(j.dst, j.src, j.len);
        else
            for (size_t r = 0; r < j.rows; ++r) memcpy(j.dst + r * j.len, j.src + r * j.src_stride, j.len);
        done[i].store(1, std::memory_order_release);
        return true;
    }
    void drain() { while (take_one()) {} }
    void work()
    {
        uint64_t seen = 0;
        for (;;) {
            // back-to-back steps: the next job list arrives within microseconds -- poll for it a short while (a futex wake-up
            // costs 30-60 us per thread) before going to sleep on the condition variable (a 50 Hz control loop sleeps)
            for (int spin = 0; spin < 20000 && gen_pub.load(std::memory_order_acquire) == seen; ++spin) cpu_relax();
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || gen != seen; });
                if (stop) return;
                seen = gen;
                active.fetch_add(1, std::memory_order_acq_rel);   // under the lock: post() cannot miss a worker that saw this generation
            }
            drain();
            active.fetch_sub(1, std::memory_order_acq_rel);
        }
    }
    // hands the job list to the workers; the caller then consumes done[i] in order (wait_job) and must call finish()
    void post(const PackJob *j, int n)
    {
        if (n > done_cap) { done.reset(new std::atomic<int>[n]); done_cap = n; }
        for (int i = 0; i < n; ++i) done[i].store(0, std::memory_order_relaxed);
        {
            std::lock_guard<std::mutex> lk(mu);
            jobs.store(j, std::memory_order_relaxed);
            next.store(0, std::memory_order_relaxed);
            njobs.store(n, std::memory_order_release);
            ++gen;
            gen_pub.store(gen, std::memory_order_release);
        }
        if (!th.empty() && n > 1) cv.notify_all();
    }
    bool is_done(int i) const { return done[i].load(std::memory_order_acquire) != 0; }
    void wait_job(int i)
    {
        // the caller packs as well while it has nothing to hand to the DMA engine (and does everything when there are no threads)
        while (!is_done(i))
            if (!take_one()) std::this_thread::yield();
    }
    void finish()
    {   // the job list lives on the caller's stack: no worker may still be looking at it when the caller returns
        njobs.store(0, std::memory_order_release);
        while (active.load(std::memory_order_acquire) != 0) std::this_thread::yield();
    }
};

struct ndp_handle {
    ndp_cfg cfg;
    RtiParams P;
    int lds_per_wave = 0;      // doubles
    int waves = 4;             // instances per workgroup
    int n_simd = 1024;         // SIMDs of the device (4 per CU)
    bool use_queue = false;    // interior-point solves through the work list: producer + consumer launch per step (QueueArgs)
    // cfg.work_queue = 0 at (N, n_rti) = (20, 1), batch >= 2 instances per SIMD: the list is switched by what the steps do (queue_policy)
    bool queue_auto = false;
    unsigned long long *hIpm = nullptr;      // page-locked [2]: the device's monotonic counts (interior-point instances, steps executed),
                                             // copied behind every QP_WINDOW-th launch
    unsigned long long ipm_seen = 0, steps_seen = 0;   // the snapshot the last decision was taken on
    unsigned queue_launches = 0;             // launches since the last copy was enqueued
    hipStream_t stream = nullptr;
    // persistent device state
    double *dX = nullptr, *dU = nullptr;
    float *dForce = nullptr, *dFrag = nullptr;
    double *dKC = nullptr;     // constants block of the LDS image (fill_kc)
    int *dTables = nullptr;    // per-lane index tables of the Riccati sweep (fill_tables)
    signed char *dAct = nullptr;   // [B][act_pitch(N)] QP_AUTO's active sets, kept between control steps (RtiIo::act); emptied by reset / set_iterate
    double *dThr = nullptr;    // hover-throttle estimator state, SoA [8][B]
    double *dStamps = nullptr; // [B][NDP_NSTAMP] whole-batch phase stamps (ndp_debug_stamps)
    double *dTraj = nullptr;   // f1: [B][n_seg][28] coeff | [B][n_seg+1] time_cum | [B][n_seg] time_seg | [B][3] final_pt | [B][64] segment cache | int[B] segment hints
    int traj_seg = 0;
    double *dRingX = nullptr, *dRingU = nullptr;   // f1: the reference's sliding list of reference points, phase-major (RingGeom), one allocation (first use)
    unsigned long long list_n = 0;                 // absolute index of the list's oldest entry = control ticks since the list was built
    int segc_par = 0;                              // which copy of the one-launch tick's segment cache the NEXT tick reads (tick_cache_store)
    int list_step = 5;
    // ndp_tick: the node's control tick on the device (rti_kernel<..., TICK>; other shapes: tick_pre_kernel + the control step)
    int *dTickIndex = nullptr;       // [B] neighbour instance of every vehicle (< 0: none), or null: no vehicle has one
    double *dTickThrust = nullptr;   // [B] the thrust command of the previous tick (what hover_throttle_callback reads off body_rate_cmd)
    bool tick_gate = true;           // gate the downwash on |neighbour window node 0 xy - ego odometry xy| < r_horiz (ndp_nmpc_leader_node.py:65-74)
    const double *tick_remote = nullptr;   // ndp_tick_config_remote: the neighbours' windows are rows of the CALLER's buffer (an exchange's gathered /
    int tick_remote_stride = 6;            // peer-mapped windows, [rows][N+1][stride]) instead of this handle's own list
    struct TickSlot { bool busy = false, want_u0 = false; } tslot[2];
    double *dRelay = nullptr;  // follower relay: [B][4] = filtered offset xyz + initialised flag
    double *sThr = nullptr;    // staging of the f1-f4 host entry points: 11 B doubles
    // downwash one tick ahead on a second stream (LateArgs): force slots, protocol words, the stream, its fork / join events
    float *dForceAB[2] = {nullptr, nullptr};
    unsigned long long *dProto = nullptr;
    hipStream_t aux = nullptr;
    hipEvent_t evFork = nullptr, evJoin = nullptr;
    unsigned prefetch_timeout_us = 100000, pf_groups_rti = 1, pf_ntiles = 1;
    unsigned *dQctr = nullptr; // work list: entry count | B instance ids
    int *dQids = nullptr;
    bool have_mlp = false;
    // Host-pointer entry points.  ONE block holds every input of a step (x0 | xr | ur | f | other | ego_xy, each 256-byte
    // aligned) and one its outputs (u0 | status | iters | X | U).  Two slots of page-locked host mirrors of both (HostSlot,
    // allocated at the first host step): the caller's arrays are packed into a slot's input mirror by the handle's pack threads,
    // the kernel reads that mirror over PCIe and writes u0 / status / iterations (and, when asked, a copy of the new iterate)
    // into the slot's output mirror itself -- one launch and one wait per step, no DMA operation, at every batch size (measured,
    // batch 1024: 106 us per step with two steps in flight against 121 us with one H2D copy per step and 147-157 us with the
    // block copied in chunks as it is packed: every asynchronous copy operation costs ~30 us of latency on this platform).
    // With two slots the packing of step i+1 runs while step i's kernel does (ndp_step_begin / ndp_step_end).
    // The persistent iterate dX | dU always lives in HBM.  dIn / dOut: device-side staging of the f1-f4 host entry points
    // (views sx0 ..) and the small outputs of device-pointer steps.
    unsigned char *dIn = nullptr, *dOut = nullptr;
    size_t off_x0 = 0, off_xr = 0, off_ur = 0, off_f = 0, off_other = 0, off_ego = 0, in_bytes = 0;
    size_t off_u0 = 0, off_st = 0, off_it = 0, out_bytes = 0, out_all = 0;
    double *sx0 = nullptr, *sxr = nullptr, *sur = nullptr, *sother = nullptr, *sego = nullptr, *su0 = nullptr, *sdbg = nullptr;
    float *sf = nullptr;
    int *dStatus = nullptr, *dIters = nullptr;
    const int *lastStatus = nullptr, *lastIters = nullptr;   // where the last step wrote them (dStatus / dIters or a slot's output mirror)
    struct HostSlot {
        unsigned char *hIn = nullptr, *hOut = nullptr;
        hipEvent_t evOut = nullptr;                          // the step that uses the slot has completed
        bool busy = false, want_iter = false;
        double *dump = nullptr;
    } slot[2];
    bool slots_ready = false;
    // ndp_track_steps: the completion of every control step marks an event WITHOUT a packet of its own (the dispatch packet's
    // completion signal, hipExtLaunchKernel) -- what another stream orders itself behind (ndp_xchg_begin's after_event)
    bool track_steps = false;
    bool last_step_tracked = false;   // does stepDone[step_seq & 3] belong to the control step launched LAST?
    bool track_pending = false;       // a tracked step on a caller's stream has not been waited for (wait_all)
    hipEvent_t stepDone[4] = {nullptr, nullptr, nullptr, nullptr};
    unsigned step_seq = 0;
    double host_us[4] = {0, 0, 0, 0};   // last host step: packing | enqueue | wait for the results | copy-out  (ndp_debug_host_timing)
    int host_cores = 0, pack_threads = 0;   // what ensure_slots found and started (ndp_debug_host_info)
    int slot_head = 0, slot_tail = 0, slots_busy = 0;   // begin fills slot_head, end drains slot_tail
    std::unique_ptr<struct PackPool> pool;
    // the last foreign stream a *_device call enqueued on: the getters wait for it (hipEvent)
    hipEvent_t evLast = nullptr;
    bool ev_pending = false;
    // timing
    int timing = 0;            // 0 off, n > 0: bracket every n-th launch of each kernel with HIP events
    int64_t launch_no[2] = {0, 0};
    bool timing_open = false;
    struct Ev { hipEvent_t a, b; int kind; };
    std::vector<Ev> events;
    std::mutex mu;
    std::string err;
};

static thread_local std::string g_create_err;

#define NDP_HIP(h, call)                                                                   \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess) {                                                            \
            (h)->err = std::string(#call) + ": " + hipGetErrorString(e_);                  \
            return -(int)e_ - 1000;                                                        \
        }                                                                                  \
    } while (0)

static size_t nxs(const ndp_handle *h) { return (size_t)h->cfg.batch * (h->cfg.N + 1) * NX; }
static size_t nus(const ndp_handle *h) { return (size_t)h->cfg.batch * h->cfg.N * NU; }
static size_t nfs(const ndp_handle *h) { return (size_t)h->cfg.batch * (h->cfg.N + 1) * 3; }
static size_t up256(size_t n) { return (n + 255) & ~(size_t)255; }
// host cores this process may really use: the affinity mask, cut down to the cgroup's CPU quota (cpu.max: "<quota> <period>" or "max ...")
static int usable_cores()
{
    int n = (int)std::thread::hardware_concurrency();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) { const int a = CPU_COUNT(&set); if (a > 0 && a < n) n = a; }
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[32] = {0};
        double period = 0.0;
        if (fscanf(f, "%31s %lf", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0.0) {
            const int c = (int)(atof(q) / period);
            if (c >= 1 && c < n) n = c;
        }
        fclose(f);
    }
    return n < 1 ? 1 : n;
}
static size_t act_bytes(const ndp_handle *h) { return (size_t)h->cfg.batch * (size_t)act_pitch(h->cfg.N); }
// the step's iteration words (RtiIo::iters) -> the caller's interior-point iteration counts
static void copy_ipm_iters(int32_t *dst, const int32_t *src, size_t n)
{
    for (size_t i = 0; i < n; ++i) dst[i] = src[i] & ITERS_IPM_MASK;
}

// the shapes the work-queue form of rti_kernel is instantiated for (compile-time horizon and iteration count)
static bool queue_shape(const ndp_handle *h)
{
    return h->cfg.qp_precision == 0 &&
           ((h->cfg.N == 20 && h->cfg.n_rti == 1 && h->waves == 4) || (h->cfg.N == 40 && h->cfg.n_rti == 2 && h->waves == 2));
}

extern "C" {

int ndp_abi_version(void) { return NDP_ABI_VERSION; }
size_t ndp_cfg_size(void) { return sizeof(ndp_cfg); }

int ndp_default_cfg(ndp_cfg *cfg)
{
    if (!cfg) return -1;
    fill_default_cfg(cfg);
    return 0;
}

const char *ndp_last_error(const ndp_handle *h) { return h ? h->err.c_str() : g_create_err.c_str(); }

int ndp_debug_lds_doubles(int N) { return lds_doubles(N) + DBG_EXTRA; }

int ndp_debug_lds_layout(int N, int *out8)
{
    if (!out8) return -1;
    lds_layout(N, out8);
    return 0;
}

int ndp_debug_mfma_probe(const double *a, const double *b, const double *c, double *d)
{
    double *da = nullptr, *db = nullptr, *dc = nullptr, *dd = nullptr;
    if (hipMalloc((void **)&da, 64 * 8) != hipSuccess || hipMalloc((void **)&db, 64 * 8) != hipSuccess ||
        hipMalloc((void **)&dc, 256 * 8) != hipSuccess || hipMalloc((void **)&dd, 832 * 8) != hipSuccess)
        return -1;
    (void)hipMemcpy(da, a, 64 * 8, hipMemcpyHostToDevice);
    (void)hipMemcpy(db, b, 64 * 8, hipMemcpyHostToDevice);
    (void)hipMemcpy(dc, c, 256 * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(mfma_probe_kernel, dim3(1), dim3(64), 0, 0, da, db, dc, dd);
    const hipError_t e = hipMemcpy(d, dd, 832 * 8, hipMemcpyDeviceToHost);
    (void)hipFree(da); (void)hipFree(db); (void)hipFree(dc); (void)hipFree(dd);
    return e == hipSuccess ? 0 : -2;
}

int ndp_debug_mfma_probe_f32(const float *a, const float *b, const float *c, float *d, int mode)
{
    float *da = nullptr, *db = nullptr, *dc = nullptr, *dd = nullptr;
    if (hipMalloc((void **)&da, 256 * 4) != hipSuccess || hipMalloc((void **)&db, 256 * 4) != hipSuccess ||
        hipMalloc((void **)&dc, 256 * 4) != hipSuccess || hipMalloc((void **)&dd, 320 * 4) != hipSuccess)
        return -1;
    (void)hipMemcpy(da, a, 256 * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(db, b, 256 * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dc, c, 256 * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(mfma_probe32_kernel, dim3(1), dim3(64), 0, 0, da, db, dc, dd, mode);
    const hipError_t e = hipMemcpy(d, dd, 320 * 4, hipMemcpyDeviceToHost);
    (void)hipFree(da); (void)hipFree(db); (void)hipFree(dc); (void)hipFree(dd);
    return e == hipSuccess ? 0 : -2;
}

// ------------------------------------------------------------------------------------------ peer windows (multi-GPU)
// The reference's neighbour exchange is publish / subscribe of the 21x10 float64 reference window (PredXU: nmpc_node.py:116-133
// publishes, ndp_nmpc_leader_node.py:40,60-76 subscribes).  One process per GPU: the publisher keeps its windows in a buffer
// whose IPC handle it hands to the subscriber's process once; the subscriber maps it and its control-step kernel reads the
// neighbour's window straight out of the publisher's HBM over xGMI (peer access) -- no per-step collective, no extra launch.
int ndp_peer_alloc(int device, size_t bytes, void **ptr, unsigned char *handle64)
{
    if (!ptr || !handle64 || bytes == 0) return -1;
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "ndp_peer_*: the handle is passed as 64 bytes");
    if (hipSetDevice(device) != hipSuccess) return -2;
    // Fine-grained device memory: coherent between agents while kernels run (the epoch / acknowledgement words are polled by
    // running kernels of two GPUs, the slots are written here and read there one launch later).  Ordinary (coarse-grained)
    // memory if the runtime refuses, or when NDP_PEER_COARSE=1 asks for it; the protocol's accesses are system-scope either way.
    void *p = nullptr;
    const char *coarse = getenv("NDP_PEER_COARSE");
    if ((coarse && coarse[0] == '1') || hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained) != hipSuccess || !p) {
        (void)hipGetLastError();
        p = nullptr;
        if (hipMalloc(&p, bytes) != hipSuccess) return -3;
    }
    if (hipMemset(p, 0, bytes) != hipSuccess) { (void)hipFree(p); return -3; }      // epochs, acknowledgements, counters start at 0
    if (hipDeviceSynchronize() != hipSuccess) { (void)hipFree(p); return -3; }
    hipIpcMemHandle_t hd;
    if (hipIpcGetMemHandle(&hd, p) != hipSuccess) { (void)hipFree(p); return -4; }
    memcpy(handle64, &hd, 64);
    *ptr = p;
    return 0;
}

int ndp_peer_open(int device, const unsigned char *handle64, void **ptr)
{
    if (!ptr || !handle64) return -1;
    if (hipSetDevice(device) != hipSuccess) return -2;
    hipIpcMemHandle_t hd;
    memcpy(&hd, handle64, 64);
    void *p = nullptr;
    if (hipIpcOpenMemHandle(&p, hd, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); return -3; }
    {   // remember the mapped range: windows inside it are read with system-scope loads (see peer_mapped)
        void *base = nullptr;
        size_t size = 0;
        if (hipMemGetAddressRange(&base, &size, p) != hipSuccess || !base || size == 0) {
            (void)hipGetLastError();
            base = p; size = (size_t)1 << 40;       // extent unknown: err on the side of system-scope loads
        }
        std::lock_guard<std::mutex> lk(g_peer_mu);
        g_peer_ranges.push_back({(uintptr_t)base, (uintptr_t)base + size});
    }
    *ptr = p;
    return 0;
}

int ndp_peer_close(int device, void *ptr)
{
    if (!ptr) return -1;
    if (hipSetDevice(device) != hipSuccess) return -2;
    {
        std::lock_guard<std::mutex> lk(g_peer_mu);
        for (size_t i = 0; i < g_peer_ranges.size(); ++i)
            if ((uintptr_t)ptr >= g_peer_ranges[i].lo && (uintptr_t)ptr < g_peer_ranges[i].hi) { g_peer_ranges.erase(g_peer_ranges.begin() + i); break; }
    }
    return hipIpcCloseMemHandle(ptr) == hipSuccess ? 0 : -3;
}

int ndp_peer_free(int device, void *ptr)
{
    if (!ptr) return -1;
    if (hipSetDevice(device) != hipSuccess) return -2;
    return hipFree(ptr) == hipSuccess ? 0 : -3;
}


// ---- per-tick publish / subscribe through such buffers (peer_epoch.hpp)
int ndp_peer_layout(size_t n_doubles, size_t *buffer_bytes, size_t *slot0_offset, size_t *slot_stride)
{
    if (buffer_bytes) *buffer_bytes = peer_buffer_bytes(n_doubles);
    if (slot0_offset) *slot0_offset = peer_slot_offset(n_doubles, 0);
    if (slot_stride) *slot_stride = peer_slot_bytes(n_doubles);
    return 0;
}

int ndp_peer_publish_device(int device, const void *d_src, size_t n_doubles, void *own_buf, void *nb_buf, int slot,
                            unsigned timeout_us, void *stream)
{
    if (!d_src || !own_buf || !nb_buf || n_doubles == 0 || (slot & ~1)) return -1;
    if (hipSetDevice(device) != hipSuccess) return -2;
    PeerPubArgs a{(const double *)d_src, (unsigned long long *)own_buf, (unsigned long long *)nb_buf, n_doubles, slot, timeout_us};
    size_t blocks = (n_doubles / 2 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 256) blocks = 256;        // all resident at once: every block's first thread may wait on the reader
    hipLaunchKernelGGL(peer_publish_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    hipLaunchKernelGGL(peer_epoch_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int ndp_track_steps(ndp_handle *h, int on)
{
    if (!h) return -1;
    std::lock_guard<std::mutex> lk(h->mu);
    NDP_HIP(h, hipSetDevice(h->cfg.device));
    if (on && !h->stepDone[0])
        for (auto &e : h->stepDone) NDP_HIP(h, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    h->track_steps = on != 0;
    return 0;
}

int ndp_last_step_event(ndp_handle *h, void **event)
{
    if (!h || !event) return -1;
    std::lock_guard<std::mutex> lk(h->mu);
    if (!h->track_steps || h->step_seq == 0) { h->err = "ndp_last_step_event: no tracked step yet (ndp_track_steps first)"; return -14; }
    if (!h->last_step_tracked) {
        h->err = "ndp_last_step_event: the control step launched last carried no completion event (launched while tracking was off, or "
                 "through a path that does not mark one): ordering a gather behind an OLDER step's event could overwrite a buffer the "
                 "last step still reads";
        return -14;
    }
    *event = (void *)h->stepDone[h->step_seq & 3];
    return 0;
}

// ---- The north star's collective issued by the library itself: one RCCL all-gather per control tick of the ranks' position /
// velocity windows, on a HIP stream of its own beside the control-step kernel (ordered by events, no host wait).  RCCL is bound at
// run time (dlopen of the library the process already holds -- torch's -- or the system's): the C-ABI library carries no link-time
// dependency on it and every other entry point works without it.
namespace {
typedef struct { char internal[128]; } rccl_uid;
typedef int (*fn_uid)(rccl_uid *);
typedef int (*fn_init)(void **, int, rccl_uid, int);
typedef int (*fn_ag)(const void *, void *, size_t, int, void *, hipStream_t);
typedef int (*fn_destroy)(void *);
typedef const char *(*fn_errstr)(int);
struct RcclApi {
    void *lib = nullptr;
    fn_uid uid = nullptr; fn_init init = nullptr; fn_ag allgather = nullptr; fn_destroy destroy = nullptr; fn_errstr errstr = nullptr;
};
std::mutex g_rccl_mu;
RcclApi g_rccl;
int rccl_bind(const char *path)
{
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.lib) return 0;
    void *l = nullptr;
    if (path && path[0]) l = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!l) l = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!l) l = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!l) return -20;
    RcclApi a;
    a.lib = l;
    a.uid = (fn_uid)dlsym(l, "ncclGetUniqueId"); a.init = (fn_init)dlsym(l, "ncclCommInitRank");
    a.allgather = (fn_ag)dlsym(l, "ncclAllGather"); a.destroy = (fn_destroy)dlsym(l, "ncclCommDestroy");
    a.errstr = (fn_errstr)dlsym(l, "ncclGetErrorString");
    if (!a.uid || !a.init || !a.allgather || !a.destroy) { dlclose(l); return -21; }
    g_rccl = a;
    return 0;
}
}  // namespace

struct ndp_xchg {
    int device = 0, rank = 0, world = 1;
    void *comm = nullptr;
    hipStream_t cs = nullptr;                 // the exchange's own stream (another priority level: its own hardware queue)
    hipEvent_t evReady = nullptr, evDone = nullptr;
    double *send = nullptr;                   // packed windows of this rank
    size_t send_doubles = 0;
    // the remote tick with the exchange ahead of the control steps (ndp_xchg_tick_begin / _step): begins are numbered 1, 2, ...; begin n
    // lives in slot n % 3 (at most two are ahead of the steps, and step k consumes begin k) and fills whichever gather buffer the
    // caller names -- two buffers (begin i+1 behind step i) or three (begin i+2 behind step i: the gather then never waits for a step)
    hipEvent_t evGather[3] = {nullptr, nullptr, nullptr};     // slot's gather is complete
    unsigned long long win_n[3] = {0, 0, 0};         // the list position its windows belong to
    const void *buf[3] = {nullptr, nullptr, nullptr};   // the gather buffer it fills
    struct Reader { const void *ptr = nullptr; unsigned seq = 0; hipStream_t stream = nullptr; unsigned age = 0; };
    Reader readers[4];                               // per gather buffer: the control step that read it last (seq: its tracked number, 0 = untracked)
    unsigned steps = 0;                              // control steps taken (step k consumes begin k)
    int ahead = 0;                                   // gathers begun and not yet stepped on (0 .. 2)
    // ndp_xchg_tick_async: the exchange stream's launches of a begin (wait, advance + columns, ncclAllGather, event record: ~15 us of
    // host time) are made by a thread of the exchange's own; the caller's begin only describes them (~2 us).  One host thread's
    // launches are what bounds the remote tick one period ahead; with two the device does.
    struct Job {
        bool adv = false;
        TickPre a{};                                  // adv: the advance (+ columns) launch
        const double *pack_base = nullptr;            // !adv: the columns of the window that is there
        size_t pack_pitch = 0, B = 0, rows = 0;
        int np1 = 0, p = 0;
        void *gathered = nullptr;
        hipEvent_t wait_ev = nullptr;
    };
    Job job[3];                                      // begin n's launches: job[n % 3]
    unsigned job_n[3] = {0, 0, 0};                   // ... and n itself
    std::atomic<unsigned> posted{0}, done{0};
    std::atomic<int> async_rc{0};
    std::atomic<bool> stop{false};
    std::thread worker;
    bool async = false;
    std::string err;
};

int ndp_xchg_destroy(ndp_xchg *x);

int ndp_xchg_unique_id(const char *rccl_path, unsigned char *id128)
{
    if (!id128) return -1;
    int rc = rccl_bind(rccl_path);
    if (rc) return rc;
    rccl_uid u;
    if (g_rccl.uid(&u) != 0) return -22;
    memcpy(id128, u.internal, 128);
    return 0;
}

int ndp_xchg_create(int device, int rank, int world, const unsigned char *id128, const char *rccl_path, ndp_xchg **out)
{
    if (!id128 || !out || world < 1 || rank < 0 || rank >= world) return -1;
    *out = nullptr;
    int rc = rccl_bind(rccl_path);
    if (rc) return rc;
    if (hipSetDevice(device) != hipSuccess) return -2;
    std::unique_ptr<ndp_xchg> x(new (std::nothrow) ndp_xchg);
    if (!x) return -4;
    x->device = device; x->rank = rank; x->world = world;
    rccl_uid u;
    memcpy(u.internal, id128, 128);
    if (g_rccl.init(&x->comm, world, u, rank) != 0) return -22;      // collective: every rank calls it
    int lo = 0, hi = 0;
    if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess || hipStreamCreateWithPriority(&x->cs, hipStreamNonBlocking, hi) != hipSuccess ||
        hipEventCreateWithFlags(&x->evReady, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&x->evDone, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&x->evGather[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&x->evGather[1], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&x->evGather[2], hipEventDisableTiming) != hipSuccess) {
        (void)ndp_xchg_destroy(x.release());      // (releases whatever exists: communicator, stream, events)
        return -3;
    }
    *out = x.release();
    return 0;
}

// rows = B_local * (N + 1) windows rows of d_xr ([rows][10] doubles) -> d_gathered ([world * rows][6]); everything `after_stream`
// holds so far comes first (null: the windows are in place, no ordering needed), nothing waits on the host
int ndp_xchg_begin(ndp_xchg *x, const void *d_xr, size_t rows, void *d_gathered, void *after_stream, void *after_event)
{
    if (!x || !d_xr || !d_gathered || rows == 0) return -1;
    if (hipSetDevice(x->device) != hipSuccess) return -2;
    if (x->send_doubles < rows * 6) {
        if (x->send) { (void)hipStreamSynchronize(x->cs); (void)hipFree(x->send); x->send = nullptr; }
        if (hipMalloc((void **)&x->send, rows * 6 * sizeof(double)) != hipSuccess) return -3;
        x->send_doubles = rows * 6;
    }
    // (every event operation is a packet the queue's command processor retires in order: ~3 us each on the stream that also
    // carries the control steps -- callers whose windows are in place already pass no stream)
    if (after_stream && (hipEventRecord(x->evReady, (hipStream_t)after_stream) != hipSuccess || hipStreamWaitEvent(x->cs, x->evReady, 0) != hipSuccess))
        return -3;
    if (after_event && hipStreamWaitEvent(x->cs, (hipEvent_t)after_event, 0) != hipSuccess) return -3;
    const size_t pieces = rows * 3;
    hipLaunchKernelGGL(pack_pv_kernel, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, x->cs, (const double *)d_xr, x->send, rows);
    if (hipGetLastError() != hipSuccess) return -3;
    const int r = g_rccl.allgather(x->send, d_gathered, rows * 6, /* ncclFloat64 */ 8, x->comm, x->cs);
    if (r != 0) { x->err = g_rccl.errstr ? g_rccl.errstr(r) : "ncclAllGather failed"; return -22; }
    return hipEventRecord(x->evDone, x->cs) == hipSuccess ? 0 : -3;
}

// `stream` waits (on the device) for the gather started last
int ndp_xchg_end(ndp_xchg *x, void *stream)
{
    if (!x) return -1;
    return hipStreamWaitEvent((hipStream_t)stream, x->evDone, 0) == hipSuccess ? 0 : -3;
}

// One call per control tick of the pipelined form: `stream` waits for the gather begun last (this tick's windows), then the NEXT tick's
// gather is begun behind the last reader of its buffer -- the completion event of the control step launched last for h when the steps
// are tracked (ndp_track_steps), else everything `stream` holds so far.
int ndp_xchg_tick(ndp_xchg *x, ndp_handle *h, void *stream, const void *d_xr_next, size_t rows, void *d_gathered_next)
{
    int rc = ndp_xchg_end(x, stream);
    if (rc) return rc;
    void *ev = nullptr;
    if (h) {
        std::lock_guard<std::mutex> lk(h->mu);
        if (h->track_steps && h->step_seq && h->last_step_tracked) ev = (void *)h->stepDone[h->step_seq & 3];
    }
    return ndp_xchg_begin(x, d_xr_next, rows, d_gathered_next, ev ? nullptr : stream, ev);
}

const char *ndp_xchg_last_error(const ndp_xchg *x) { return x ? x->err.c_str() : "null exchange"; }

static void xchg_worker_stop(ndp_xchg *x)
{
    if (x->worker.joinable()) {
        x->stop.store(true, std::memory_order_release);
        x->worker.join();
        x->stop.store(false, std::memory_order_release);
    }
    x->async = false;
}

int ndp_xchg_destroy(ndp_xchg *x)
{
    if (!x) return -1;
    xchg_worker_stop(x);
    (void)hipSetDevice(x->device);
    if (x->cs) (void)hipStreamSynchronize(x->cs);
    if (x->comm) (void)g_rccl.destroy(x->comm);
    if (x->send) (void)hipFree(x->send);
    if (x->evReady) (void)hipEventDestroy(x->evReady);
    if (x->evDone) (void)hipEventDestroy(x->evDone);
    for (hipEvent_t e : x->evGather) if (e) (void)hipEventDestroy(e);
    if (x->cs) (void)hipStreamDestroy(x->cs);
    delete x;
    return 0;
}

int ndp_peer_stats(int device, const void *own_buf, unsigned long long *out4)
{
    if (!own_buf || !out4) return -1;
    if (hipSetDevice(device) != hipSuccess) return -2;
    if (hipDeviceSynchronize() != hipSuccess) return -3;
    return hipMemcpy(out4, (const unsigned long long *)own_buf + PEER_W_STAT, PEER_STAT_N * 8, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -3;
}

int ndp_destroy(ndp_handle *h)
{
    if (!h) return -1;
    (void)hipSetDevice(h->cfg.device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->ev_pending) (void)hipEventSynchronize(h->evLast);
    for (auto &e : h->events) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    if (h->evLast) (void)hipEventDestroy(h->evLast);
    if (h->aux) { (void)hipStreamSynchronize(h->aux); (void)hipStreamDestroy(h->aux); }
    for (hipEvent_t e : {h->evFork, h->evJoin, h->stepDone[0], h->stepDone[1], h->stepDone[2], h->stepDone[3]})
        if (e) (void)hipEventDestroy(e);
    h->pool.reset();
    void *ptrs[] = {h->dForceAB[0], h->dForceAB[1], h->dProto, h->dRingX, h->dTraj, h->dTables, h->dStamps, h->dRelay, h->dThr, h->sThr, h->dKC, h->dForce, h->dFrag,
                    h->dIn, h->dOut, h->sdbg, h->dQctr, h->dQids, h->dTickIndex, h->dTickThrust, h->dAct};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    for (auto &sl : h->slot) {
        if (sl.hIn) (void)hipHostFree(sl.hIn);
        if (sl.hOut) (void)hipHostFree(sl.hOut);
        if (sl.evOut) (void)hipEventDestroy(sl.evOut);
    }
    if (h->hIpm) {                          // (a copy into it may still be queued on a caller's stream)
        (void)hipDeviceSynchronize();
        (void)hipHostFree(h->hIpm);
    }
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return 0;
}

int ndp_create(const ndp_cfg *cfg, ndp_handle **out)
{
    if (!cfg || !out) { g_create_err = "ndp_create: null argument"; return -1; }
    *out = nullptr;
    if (cfg->batch < 1 || cfg->N < 2 || slots_for(cfg->N) > 5 || cfg->n_rti < 1 || cfg->qp_precision < 0 || cfg->qp_precision > 6 ||
        cfg->work_queue < 0 || cfg->work_queue > 2 || !(cfg->ts_nmpc > 0.0) || cfg->dt < cfg->ts_nmpc) {
        g_create_err = "ndp_create: need batch >= 1, 2 <= N <= 46, n_rti >= 1, qp_precision in 0..6, work_queue in 0..2, 0 < ts_nmpc <= dt";
        return -2;
    }
    if (cfg->qp_precision >= 5 && (cfg->N % 4 != 0 || cfg->N > 40)) {
        g_create_err = "ndp_create: the condensed study (qp_precision 5 / 6) tiles the 4N x 4N Hessian by 16: N must be a multiple of 4, at most 40";
        return -2;
    }
    if (cfg->ipm_refine > 0 && (slots_for(cfg->N) > 3 || cfg->qp_precision != 0)) {
        // (rounds 4-5 accepted the setting and ignored it -- a getter nobody called said so)
        g_create_err = "ndp_create: ipm_refine > 0 is not served for this shape: the refinement path (stiff sweeps + second solves while a STATE bound's "
                       "barrier term exceeds refine_gamma) lives in the three-slot fp64 kernels only (N <= 27, qp_precision 0); the five-slot kernels "
                       "(N >= 28) sit at the register limit without it.  Set ipm_refine = 0: strongly active state bounds at a tight tolerance then "
                       "end in status 4 (never a silent answer)";
        return -2;
    }
    {   // the reference list holds one point per control period and the window is every (dt / ts_nmpc)-th entry
        // (params/nmpc_params.py:40-43): the ratio has to be a whole number, else ring windows and direct windows disagree
        const double ratio = cfg->dt / cfg->ts_nmpc, nearest = (double)(long long)(ratio + 0.5);
        if (ratio - nearest > 1e-9 || nearest - ratio > 1e-9) {
            g_create_err = "ndp_create: dt must be a whole multiple of ts_nmpc (node spacing = every k-th entry of the reference list)";
            return -2;
        }
    }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= cfg->device) {
        g_create_err = std::string("ndp_create: no usable HIP device (") + hipGetErrorString(e) + ")";
        return -3;
    }
    ndp_handle *h = new (std::nothrow) ndp_handle;
    if (!h) return -4;
    h->cfg = *cfg;
    h->P = to_params(*cfg);
    h->list_step = (int)(cfg->dt / cfg->ts_nmpc + 0.5);      // params/nmpc_params.py:40-43: every 5th list entry is a node
    h->lds_per_wave = (lds_doubles(cfg->N) + 1) & ~1;   // keep 16-byte alignment per wave slice
    if (cfg->qp_precision >= 5) h->lds_per_wave += (cond_extra_doubles(cfg->N) + 1) & ~1;   // (the condensed study works behind the slice: one wave per workgroup)
    const size_t per_wave_bytes = (size_t)h->lds_per_wave * sizeof(double);
    h->waves = 4;
    while (h->waves > 1 && per_wave_bytes * h->waves > 160 * 1024) h->waves >>= 1;
    if (const char *e = getenv("NDP_DEV_WAVES")) {      // measurement switch: instances per workgroup (2: two workgroups per CU)
        const int w = atoi(e);
        if ((w == 1 || w == 2 || w == 4) && w <= h->waves) h->waves = w;
    }
    auto fail = [&](const char *what, hipError_t err) {
        g_create_err = std::string("ndp_create: ") + what + ": " + hipGetErrorString(err);
        ndp_destroy(h);
        return -5;
    };
    if ((e = hipSetDevice(cfg->device)) != hipSuccess) return fail("hipSetDevice", e);
    {
        hipDeviceProp_t prop;
        if ((e = hipGetDeviceProperties(&prop, cfg->device)) != hipSuccess) return fail("hipGetDeviceProperties", e);
        h->n_simd = 4 * prop.multiProcessorCount;
    }
    // work list: with one instance per SIMD nothing can be re-balanced; with two or more it is switched by what the steps do
    // (queue_policy) -- callers who know their workload set cfg.work_queue = 1 / 2.
    // The N = 40 / 2-iteration shape always takes the list: its producer kernel carries no interior-point code and does not
    // spill, which is worth 17 % even when nothing is listed (the in-place kernel of that shape uses 0.9 KB of scratch per lane)
    if (cfg->work_queue == 1 && !(queue_shape(h) && cfg->qp_mode == NDP_QP_AUTO)) {
        g_create_err = "ndp_create: work_queue = 1 needs qp_mode AUTO, qp_precision 0 and (N, n_rti) = (20, 1) or (40, 2)";
        delete h;
        return -2;
    }
    // reference shape, two or more instances per SIMD, cfg.work_queue = 0: in place to begin with, the list when the steps ask for it
    // (queue_policy)
    h->queue_auto = cfg->work_queue == 0 && queue_shape(h) && cfg->qp_mode == NDP_QP_AUTO && cfg->N == 20 && cfg->batch >= 2 * h->n_simd;
    h->use_queue = cfg->work_queue == 1 || (cfg->work_queue == 0 && queue_shape(h) && cfg->qp_mode == NDP_QP_AUTO && cfg->N == 40);
    if (h->queue_auto) {
        if ((e = hipHostMalloc((void **)&h->hIpm, 64, hipHostMallocDefault)) != hipSuccess) return fail("hipHostMalloc (work-list counter)", e);
        h->hIpm[0] = h->hIpm[1] = 0;
    }
    if ((e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)) != hipSuccess) return fail("hipStreamCreate", e);
    if ((e = hipEventCreateWithFlags(&h->evLast, hipEventDisableTiming)) != hipSuccess) return fail("hipEventCreate", e);
    const size_t B = cfg->batch;
#define ALLOC(p, n)                                                                      \
    if ((e = hipMalloc((void **)&(p), (n))) != hipSuccess) return fail("hipMalloc " #p, e)
    ALLOC(h->dForce, nfs(h) * 4); ALLOC(h->dFrag, FR_TOTAL * 4); ALLOC(h->dKC, KC_HOST * 8);
    ALLOC(h->dThr, B * 8 * 8); ALLOC(h->sThr, B * 11 * 8);
    ALLOC(h->dRelay, B * 4 * 8);
    ALLOC(h->dQctr, 256); ALLOC(h->dQids, B * 4);
    ALLOC(h->dAct, B * (size_t)act_pitch(cfg->N));
    (void)hipMemsetAsync(h->dAct, 0, B * (size_t)act_pitch(cfg->N), h->stream);
    (void)hipMemsetAsync(h->dRelay, 0, B * 4 * 8, h->stream);
    (void)hipMemsetAsync(h->dQctr, 0, 256, h->stream);
    {
        double kc[KC_HOST];
        fill_kc(h->P, kc);
        if ((e = hipMemcpyAsync(h->dKC, kc, sizeof(kc), hipMemcpyHostToDevice, h->stream)) != hipSuccess) return fail("hipMemcpy kc", e);
        ALLOC(h->dTables, TB_WORDS * 4);
        std::vector<int> tb(TB_WORDS);
        fill_tables(cfg->N, tb.data(), (cfg->qp_precision == 3 || cfg->qp_precision == 4) ? 1 : 0);   // the fp32 / bf16 sweeps keep a different column per lane
        if ((e = hipMemcpyAsync(h->dTables, tb.data(), TB_WORDS * 4, hipMemcpyHostToDevice, h->stream)) != hipSuccess) return fail("hipMemcpy tables", e);
        if ((e = hipStreamSynchronize(h->stream)) != hipSuccess) return fail("hipStreamSynchronize", e);
    }
    {   // input / output blocks and their views
        size_t o = 0;
        h->off_x0 = o; o += up256(B * NX * 8);
        h->off_xr = o; o += up256(nxs(h) * 8);
        h->off_ur = o; o += up256(nus(h) * 8);
        h->off_f = o; o += up256(nfs(h) * 8);        // (fp32 forces, or fp64 ones: ndp_step_ex_f64)
        h->off_other = o; o += up256(nxs(h) * 8);
        h->off_ego = o; o += up256(B * 2 * 8);
        h->in_bytes = o;
        o = 0;
        h->off_u0 = o; o += up256(B * NU * 8);
        h->off_st = o; o += up256(B * 4);
        h->off_it = o; o += up256(B * 4);
        h->out_bytes = o;
        // the persistent iterate lives right behind the small outputs: u0 | status | iterations | X | U come back to the host
        // in ONE copy when a caller asks for the iterate as well (ndp_step_ex).  All of it is HBM; the page-locked host
        // mirrors of the host-array entry points are allocated by their first call (ensure_slots).
        h->out_all = h->out_bytes + (nxs(h) + nus(h)) * 8;
        ALLOC(h->dIn, h->in_bytes);
        ALLOC(h->dOut, h->out_all);
        h->dX = (double *)(h->dOut + h->out_bytes); h->dU = h->dX + nxs(h);
        h->sx0 = (double *)(h->dIn + h->off_x0); h->sxr = (double *)(h->dIn + h->off_xr); h->sur = (double *)(h->dIn + h->off_ur);
        h->sf = (float *)(h->dIn + h->off_f); h->sother = (double *)(h->dIn + h->off_other); h->sego = (double *)(h->dIn + h->off_ego);
        h->su0 = (double *)(h->dOut + h->off_u0); h->dStatus = (int *)(h->dOut + h->off_st); h->dIters = (int *)(h->dOut + h->off_it);
        h->lastStatus = h->dStatus; h->lastIters = h->dIters;
        ALLOC(h->sdbg, (size_t)(lds_doubles(cfg->N) + DBG_EXTRA) * 8);
    }
#undef ALLOC
    (void)hipMemsetAsync(h->dX, 0, nxs(h) * 8, h->stream);
    (void)hipMemsetAsync(h->dU, 0, nus(h) * 8, h->stream);
    (void)hipMemsetAsync(h->dOut, 0, h->out_bytes, h->stream);
    (void)hipMemsetAsync(h->dForce, 0, nfs(h) * 4, h->stream);
    // allow the big dynamic-LDS launches
    const int lds_bytes = (int)(per_wave_bytes * h->waves);
    const void *fns[] = {(const void *)RTI_K(3, 4, false), (const void *)RTI_K(3, 2, false), (const void *)RTI_K(3, 1, false),
                         (const void *)RTI_K(5, 4, false), (const void *)RTI_K(5, 2, false), (const void *)RTI_K(5, 1, false),
                         (const void *)RTI_K(3, 4, true), (const void *)RTI_K(3, 2, true), (const void *)RTI_K(3, 1, true),
                         (const void *)RTI_K(3, 4, false, 20), (const void *)RTI_K(3, 4, true, 20),
                         (const void *)RTI_K(3, 2, false, 20), (const void *)RTI_K(3, 2, true, 20),
                         (const void *)RTI_K(3, 4, false, 20, 0, 1, 1), (const void *)RTI_K(3, 4, true, 20, 0, 1, 1), (const void *)RTI_K(3, 4, false, 20, 0, 1, 2),
                         (const void *)RTI_K(3, 4, false, 20, 0, 1, 3),
                         (const void *)RTI_K(3, 4, true, 20, 0, 1, 0, true), (const void *)RTI_K(3, 4, false, 20, 0, 1, 0, true),
                         (const void *)RTI_K(3, 4, true, 20, 0, 1, 1, true), (const void *)RTI_K(3, 4, false, 20, 0, 1, 1, true),
                         (const void *)RTI_K(5, 1, false, 0, 1), (const void *)RTI_K(5, 1, false, 0, 2),
                         (const void *)RTI_K(5, 1, false, 0, 3), (const void *)RTI_K(5, 1, false, 0, 4),
                         (const void *)RTI_K(5, 1, false, 0, 5), (const void *)RTI_K(5, 1, false, 0, 6),
                         (const void *)RTI_K(5, 2, false, 40, 3, 2), (const void *)RTI_K(5, 2, false, 40, 4, 2),
                         (const void *)RTI_K(5, 2, false, 40, 0, 2), (const void *)RTI_K(5, 2, false, 40, 0, 2, 1), (const void *)RTI_K(5, 2, false, 40, 0, 2, 2)};
    if ((e = hipFuncSetAttribute((const void *)mlp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)(FR_TOTAL * sizeof(float)))) != hipSuccess)
        return fail("hipFuncSetAttribute(mlp_kernel)", e);
    for (const void *fn : fns)
        if ((e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes)) != hipSuccess)
            return fail("hipFuncSetAttribute", e);
    hipLaunchKernelGGL(throttle_reset_kernel, dim3((cfg->batch + 255) / 256), dim3(256), 0, h->stream, h->dThr, 50.0, cfg->batch);
    if ((e = hipStreamSynchronize(h->stream)) != hipSuccess) return fail("hipStreamSynchronize", e);
    *out = h;
    return 0;
}

int ndp_set_mlp_weights(ndp_handle *h, const float *blob, size_t n)
{
    if (!h || !blob) return -1;
    std::lock_guard<std::mutex> lk(h->mu);
    if (n != NDP_MLP_NPARAM) { h->err = "ndp_set_mlp_weights: expected 17859 floats"; return -2; }
    NDP_HIP(h, hipSetDevice(h->cfg.device));
    std::vector<float> fr;
    make_fragments(blob, fr);
    NDP_HIP(h, hipMemcpyAsync(h->dFrag, fr.data(), FR_TOTAL * 4, hipMemcpyHostToDevice, h->stream));
    NDP_HIP(h, hipStreamSynchronize(h->stream));
    h->have_mlp = true;
    return 0;
}

// ---- enqueue helpers (no locking, no sync) ----
// defer: the caller may hand the pair to the launch itself (hipExtLaunchKernel's start / stop events: the dispatch packet's own
// timestamps, no event packets around the kernel -- an event pair recorded around a launch adds ~2.5 us of dispatch gap to what it
// measures); it records ev.a itself if it cannot.
static int begin_timing(ndp_handle *h, hipStream_t s, int kind, bool defer = false)
{
    h->timing_open = false;
    if (!h->timing || (h->launch_no[kind]++ % h->timing) != 0) return 0;
    h->timing_open = true;
    ndp_handle::Ev ev; ev.kind = kind;
    NDP_HIP(h, hipEventCreate(&ev.a)); NDP_HIP(h, hipEventCreate(&ev.b));
    if (!defer) NDP_HIP(h, hipEventRecord(ev.a, s));
    h->events.push_back(ev);
    return 0;
}
static int end_timing(ndp_handle *h, hipStream_t s)
{
    if (!h->timing_open) return 0;
    NDP_HIP(h, hipEventRecord(h->events.back().b, s));
    return 0;
}

// A *_device call enqueued on a caller's stream: remember it so that the getters (ndp_get_status, ndp_get_iterate, ...)
// wait for that work and not only for the library's own stream.  Streams being captured into a graph are skipped (an
// event recorded there belongs to the graph and cannot be waited on from the host).
static int note_stream(ndp_handle *h, hipStream_t s)
{
    if (s == h->stream) return 0;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return 0; }
    NDP_HIP(h, hipEventRecord(h->evLast, s));
    h->ev_pending = true;
    return 0;
}
static int wait_all(ndp_handle *h)
{
    NDP_HIP(h, hipStreamSynchronize(h->stream));
    if (h->ev_pending) {
        NDP_HIP(h, hipEventSynchronize(h->evLast));
        h->ev_pending = false;
    }
    if (h->track_pending) {            // tracked control steps on a caller's stream (ndp_xchg_tick_step): the last one's completion event
        if (h->stepDone[h->step_seq & 3]) NDP_HIP(h, hipEventSynchronize(h->stepDone[h->step_seq & 3]));
        h->track_pending = false;
    }
    return 0;
}

struct Neigh {                 // neighbour windows of a step (device pointers)
    const double *other = nullptr;
    int stride = NX;           // doubles per node: 10 or 6
    const int *index = nullptr;
    const double *ego_xy = nullptr;
    size_t pitch = 0;          // doubles between rows of `other`; 0 = dense, (N+1) stride
    size_t ego_pitch = 0;      // doubles between instances of ego_xy; 0 = dense, 2
};

static int launch_mlp(ndp_handle *h, const Neigh &nb, const double *d_ego, float *d_f, hipStream_t s, size_t ego_pitch = 0)
{
    if (!h->have_mlp) { h->err = "downwash requested but ndp_set_mlp_weights was never called"; return -6; }
    const int np1 = h->cfg.N + 1, rows = h->cfg.batch * np1;
    const int ntiles = (rows + 31) / 32;
    const int grid = (ntiles + 3) / 4;
    int rc = begin_timing(h, s, 1);
    if (rc) return rc;
    hipLaunchKernelGGL(mlp_kernel, dim3(grid), dim3(256), FR_TOTAL * sizeof(float), s, (const float *)h->dFrag, nb.other, d_ego, nb.ego_xy, d_f,
                       rows, np1, h->cfg.r_horiz * h->cfg.r_horiz, nb.stride, nb.index, peer_mapped(nb.other),
                       nb.pitch ? nb.pitch : (size_t)np1 * nb.stride, ego_pitch ? ego_pitch : (size_t)np1 * NX, nb.ego_pitch ? nb.ego_pitch : (size_t)2);
    NDP_HIP(h, hipGetLastError());
    return end_timing(h, s);
}

struct StepOut {               // where a step's status / iteration counts go and whether the new iterate is mirrored (device-accessible
    int *status = nullptr;     // pointers; null = the handle's HBM block / no mirror): the host-array step of small batches points
    int *iters = nullptr;      // them into a page-locked host block
    double *Xm = nullptr, *Um = nullptr;
    hipEvent_t done = nullptr; // marked by the step's last launch through its own dispatch packet (no event packet behind it), or null
    size_t xr_pitch = 0, ur_pitch = 0;   // doubles between the instances' reference windows; 0 = dense arrays (see BatchPtrs)
    double *cmd = nullptr;               // ndp_tick: the actuator command written by the control step itself (BatchPtrs::cmd) ...
    const double *kthr = nullptr;        // ... from k_throttle[B]
    double *thrust_keep = nullptr;
    bool f_f64 = false;                  // d_f holds doubles (ndp_step_ex_f64)
    const TickArgs *tick = nullptr;      // the launch is a whole control tick (rti_kernel<..., TICK>): list advance + estimator inside
};

// The automatic work-list rule (cfg.work_queue = 0, reference shape, at least two instances per SIMD).  The list re-balances interior-
// point solves over all SIMDs (+70 % at batch 4096 when a fifth of the instances iterate) but costs a step that lists nothing two more
// launches and a slower producer kernel: 8-15 % (batch 2048: 45.4 against 39.5 us; 16 384: 332 against 309 us).  So the handle starts in
// place and looks at what the steps do: the device keeps two monotonic counts -- instances that went into the interior-point loop (counted
// by the in-place kernel, added up by the list's reset launch) and control steps executed -- and a 16-byte copy of the pair is enqueued
// behind every QP_WINDOW-th launch.  Whenever a snapshot that covers QP_WINDOW more executed steps has LANDED (the host may be many
// launches ahead of the device), the fraction over those steps switches the list on at >= 4 %, off at <= 1.5 %.  Evaluated at launch
// time: a captured graph keeps the form it was captured in.
enum { QP_WINDOW = 8 };
static void queue_policy(ndp_handle *h, hipStream_t s)
{
    if (!h->queue_auto) return;
    const unsigned long long steps_now = ((volatile unsigned long long *)h->hIpm)[1], ipm_now = ((volatile unsigned long long *)h->hIpm)[0];
    if (steps_now >= h->steps_seen + QP_WINDOW && ipm_now >= h->ipm_seen) {
        const double frac = (double)(ipm_now - h->ipm_seen) / ((double)(steps_now - h->steps_seen) * (double)h->cfg.batch);
        if (frac >= 0.04) h->use_queue = true;
        else if (frac <= 0.015) h->use_queue = false;
        h->ipm_seen = ipm_now;
        h->steps_seen = steps_now;
    }
    if (++h->queue_launches < QP_WINDOW) return;
    h->queue_launches = 0;
    if (hipMemcpyAsync(h->hIpm, reinterpret_cast<unsigned long long *>(h->dQctr + 16), 16, hipMemcpyDeviceToHost, s) != hipSuccess) (void)hipGetLastError();
}

static int launch_rti(ndp_handle *h, const double *d_x0, const double *d_xr, const double *d_ur, const float *d_f,
                      double *d_u0, double *d_dbg, hipStream_t s, const Neigh *nb = nullptr, const StepOut *so = nullptr,
                      bool prefetched = false)
{
    int *d_status = so && so->status ? so->status : h->dStatus, *d_iters = so && so->iters ? so->iters : h->dIters;
    h->lastStatus = d_status; h->lastIters = d_iters;
    BatchPtrs bp{h->dKC, h->dTables, d_x0, d_xr, d_ur, d_f, h->dX, h->dU, d_u0, d_status, d_iters,
                 so ? so->Xm : nullptr, so ? so->Um : nullptr, d_dbg, h->dStamps,
                 so && so->xr_pitch ? so->xr_pitch : (size_t)(h->cfg.N + 1) * NX, so && so->ur_pitch ? so->ur_pitch : (size_t)h->cfg.N * NU, (size_t)NX,
                 so ? so->cmd : nullptr, so ? so->kthr : nullptr, so ? so->thrust_keep : nullptr, h->cfg.mass, so && so->f_f64 ? 1 : 0,
                 h->dAct};
    const bool fused = nb && nb->other;
    MlpArgs ma{fused ? h->dFrag : nullptr, fused ? nb->other : nullptr, fused ? nb->ego_xy : nullptr, h->dForce,
               h->cfg.r_horiz * h->cfg.r_horiz, fused ? nb->stride : NX, fused ? nb->index : nullptr, fused ? peer_mapped(nb->other) : 0,
               fused && nb->pitch ? nb->pitch : (size_t)(h->cfg.N + 1) * (fused ? nb->stride : NX), fused && nb->ego_pitch ? nb->ego_pitch : (size_t)2};
    QueueArgs qa{h->dQctr, h->dQids, reinterpret_cast<unsigned long long *>(h->dQctr + 16)};
    const int B = h->cfg.batch, W = h->waves;
    LateArgs la{prefetched ? h->dProto : nullptr, {h->dForceAB[0], h->dForceAB[1]}, h->prefetch_timeout_us,
                h->pf_groups_rti, h->pf_ntiles};
    KernArgs ka{h->P, bp, B, h->lds_per_wave, ma, qa, la, so && so->tick ? *so->tick : TickArgs{}};
    const bool tick1 = so && so->tick;   // (tick_enqueue hands a TickArgs over only for the shapes the TICK kernels exist for)
    const dim3 grid((B + W - 1) / W), block(64 * W);
    const size_t shm = (size_t)h->lds_per_wave * sizeof(double) * W;
    const int ns = slots_for(h->cfg.N);
    const bool q = h->use_queue && !d_dbg && !prefetched;      // (the late-force step is in place: its launch is the lean instantiation)
    // Tracked steps carry their completion event on a dispatch packet (hipExtLaunchKernel), which a stream capture cannot hold:
    // refuse instead of launching something the graph would silently drop the event of.
    h->last_step_tracked = false;
    if (h->track_steps) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cs) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusNone; }
        if (cs != hipStreamCaptureStatusNone) {
            h->err = "tracked control steps (ndp_track_steps) cannot be launched on a stream that is being captured: switch tracking off "
                     "and order the exchange through the stream (ndp_xchg_begin's after_stream)";
            return -15;
        }
    }
    int rc = begin_timing(h, s, 0, true);
    if (rc) return rc;
    // timed launch of a single-kernel step: the pair rides on the dispatch packet (otherwise recorded around the launches)
    const bool ext_timing = h->timing_open && !q && !h->cfg.qp_precision && !(so && so->done) && !h->track_steps;
    if (h->timing_open && !ext_timing) NDP_HIP(h, hipEventRecord(h->events.back().a, s));
    if (h->cfg.qp_precision) {      // BASELINE config 5 (unfused; run ndp_downwash first for a force)
        if (fused) { h->err = "qp_precision != 0 supports f / no disturbance only (run ndp_downwash first)"; return -12; }
        const size_t shm1 = (size_t)h->lds_per_wave * sizeof(double);
        const int pr = h->cfg.qp_precision;
        if (pr >= 3 && h->cfg.N == 40 && h->cfg.n_rti == 2 && W == 2) {   // config 5's own shape: compile-time horizon, 2 instances per workgroup
            if (pr == 3) hipLaunchKernelGGL(RTI_K(5, 2, false, 40, 3, 2), grid, block, shm, s, ka);
            else hipLaunchKernelGGL(RTI_K(5, 2, false, 40, 4, 2), grid, block, shm, s, ka);
        } else if (pr == 1) hipLaunchKernelGGL(RTI_K(5, 1, false, 0, 1), dim3(B), dim3(64), shm1, s, ka);   // any horizon: one wave per workgroup
        else if (pr == 2) hipLaunchKernelGGL(RTI_K(5, 1, false, 0, 2), dim3(B), dim3(64), shm1, s, ka);
        else if (pr == 3) hipLaunchKernelGGL(RTI_K(5, 1, false, 0, 3), dim3(B), dim3(64), shm1, s, ka);
        else if (pr == 4) hipLaunchKernelGGL(RTI_K(5, 1, false, 0, 4), dim3(B), dim3(64), shm1, s, ka);
        else if (pr == 5) hipLaunchKernelGGL(RTI_K(5, 1, false, 0, 5), dim3(B), dim3(64), shm1, s, ka);     // config 5, condensed: fp32 instruction
        else hipLaunchKernelGGL(RTI_K(5, 1, false, 0, 6), dim3(B), dim3(64), shm1, s, ka);                   // ... bf16 instruction
        NDP_HIP(h, hipGetLastError());
        if (so && so->done) NDP_HIP(h, hipEventRecord(so->done, s));
        else if (h->track_steps) {          // (a recorded event here: the precision studies are not on the exchange's fast path)
            NDP_HIP(h, hipEventRecord(h->stepDone[++h->step_seq & 3], s));
            h->last_step_tracked = true;
        }
        return end_timing(h, s);
    }
    // tracked steps: the LAST launch of the step carries the completion event (in-place kernel, or the work list's reset launch)
    hipEvent_t stop = nullptr;
    if (so && so->done) stop = so->done;
    else if (h->track_steps) { stop = h->stepDone[++h->step_seq & 3]; h->last_step_tracked = true; }
#define LAUNCH(...)                                                                                                  \
    do {                                                                                                             \
        if (ext_timing) {                                                                                            \
            hipExtLaunchKernelGGL(RTI_K(__VA_ARGS__), grid, block, (std::uint32_t)shm, s, h->events.back().a, h->events.back().b, 0, ka); \
            h->timing_open = false;                                                                                  \
        } else if (stop && !q) hipExtLaunchKernelGGL(RTI_K(__VA_ARGS__), grid, block, (std::uint32_t)shm, s, nullptr, stop, 0, ka); \
        else hipLaunchKernelGGL(RTI_K(__VA_ARGS__), grid, block, shm, s, ka);                                        \
    } while (0)
    if (q) {
        // work list: producer (every instance: one solve with the kept set's pins, done or defer), consumer (the deferred ones, from
        // scratch: active-set iterations, then the interior-point loop if need be; several RTI iterations: the automatic rule per
        // iteration), a one-wave launch that empties the list for the next step.  The consumer reads the fused producer's
        // force from dForce.
        KernArgs kc = ka;
        kc.bp.f = fused ? h->dForce : d_f;
        kc.ma.frag = nullptr; kc.ma.other = nullptr;
        // (active-set iterations off: a listed instance needs the interior-point loop, the consumer goes straight to it; on: the
        // producer lists every instance whose first solve -- with the kept set's pins -- did not settle, the consumer iterates on the set)
        if (h->cfg.n_rti == 1 && h->cfg.as_iter_max <= 0) kc.P.qp_mode = QP_IPM_ALWAYS;
        if (h->cfg.N == 20) {
            if (tick1) { if (fused) LAUNCH(3, 4, true, 20, 0, 1, 1, true); else LAUNCH(3, 4, false, 20, 0, 1, 1, true); }
            else if (fused) LAUNCH(3, 4, true, 20, 0, 1, 1); else LAUNCH(3, 4, false, 20, 0, 1, 1);
            NDP_HIP(h, hipGetLastError());
            hipLaunchKernelGGL(RTI_K(3, 4, false, 20, 0, 1, 2), grid, block, shm, s, kc);
        } else {
            LAUNCH(5, 2, false, 40, 0, 2, 1);
            NDP_HIP(h, hipGetLastError());
            hipLaunchKernelGGL(RTI_K(5, 2, false, 40, 0, 2, 2), grid, block, shm, s, kc);
        }
        NDP_HIP(h, hipGetLastError());
        if (stop) hipExtLaunchKernelGGL(queue_reset_kernel, dim3(1), dim3(64), 0, s, nullptr, stop, 0, h->dQctr, qa.ipm_total);
        else hipLaunchKernelGGL(queue_reset_kernel, dim3(1), dim3(64), 0, s, h->dQctr, qa.ipm_total);
        NDP_HIP(h, hipGetLastError());
        { const int rce = end_timing(h, s); queue_policy(h, s); return rce; }
    }
    if (h->cfg.N == 20 && h->cfg.n_rti == 1 && W == 4) {   // the reference configuration (params/nmpc_params.py:9, 1 RTI iteration): compile-time instantiation
        if (tick1) { if (fused) LAUNCH(3, 4, true, 20, 0, 1, 0, true); else LAUNCH(3, 4, false, 20, 0, 1, 0, true); }
        else if (fused) LAUNCH(3, 4, true, 20); else if (prefetched) LAUNCH(3, 4, false, 20, 0, 1, 3); else LAUNCH(3, 4, false, 20);
    } else if (h->cfg.N == 20 && h->cfg.n_rti == 1 && W == 2) {   // (NDP_DEV_WAVES = 2: the same program, two instances per workgroup)
        if (fused) LAUNCH(3, 2, true, 20); else LAUNCH(3, 2, false, 20);
    } else if (h->cfg.N == 40 && h->cfg.n_rti == 2 && W == 2 && !fused) {   // BASELINE config 5's shape, compile-time as well
        LAUNCH(5, 2, false, 40, 0, 2);
    } else if (fused) { if (W == 4) LAUNCH(3, 4, true); else if (W == 2) LAUNCH(3, 2, true); else LAUNCH(3, 1, true); }
    else if (ns <= 3) { if (W == 4) LAUNCH(3, 4, false); else if (W == 2) LAUNCH(3, 2, false); else LAUNCH(3, 1, false); }
    else { if (W == 4) LAUNCH(5, 4, false); else if (W == 2) LAUNCH(5, 2, false); else LAUNCH(5, 1, false); }
#undef LAUNCH
    NDP_HIP(h, hipGetLastError());
    { const int rce = end_timing(h, s); queue_policy(h, s); return rce; }
}

// downwash inside the RTI launch when one 32-row tile covers the horizon; otherwise mlp_kernel first
static bool can_fuse(const ndp_handle *h)
{
    return h->cfg.qp_precision == 0 && h->cfg.N + 1 <= 32 && slots_for(h->cfg.N) <= 3 &&
           (size_t)h->lds_per_wave * sizeof(double) * h->waves >= FR_TOTAL * sizeof(float);
}

// one control step on device pointers: [mlp_kernel ->] rti_kernel (no locking, no sync)
static int enqueue_step(ndp_handle *h, const double *d_x0, const double *d_xr, const double *d_ur, const float *d_f,
                        const Neigh &nb, double *d_u0, double *d_dbg, hipStream_t s, const StepOut *so = nullptr)
{
    if ((d_f || nb.other) && !h->cfg.use_fd) { h->err = "ndp_step: a disturbance force needs use_fd = 1 (NDP model)"; return -8; }
    if (nb.other) {
        if (d_f) { h->err = "ndp_step: pass either f or other, not both"; return -7; }
        if (nb.stride != 10 && nb.stride != 6) { h->err = "ndp_step: other_stride must be 10 or 6"; return -13; }
        if (!h->have_mlp) { h->err = "downwash requested but ndp_set_mlp_weights was never called"; return -6; }
        if (can_fuse(h)) return launch_rti(h, d_x0, d_xr, d_ur, nullptr, d_u0, d_dbg, s, &nb, so);
        int rc = launch_mlp(h, nb, d_xr, h->dForce, s, so ? so->xr_pitch : 0);
        if (rc) return rc;
        d_f = h->dForce;
    }
    return launch_rti(h, d_x0, d_xr, d_ur, d_f, d_u0, d_dbg, s, nullptr, so);
}

int ndp_reset_device(ndp_handle *h, const void *d_xr, const void *d_ur, void *stream)
{
    if (!h || !d_xr || !d_ur) return -1;
    std::lock_guard<std::mutex> lk(h->mu);
    hipStream_t s = stream ? (hipStream_t)stream : h->stream;
    NDP_HIP(h, hipSetDevice(h->cfg.device));
    NDP_HIP(h, hipMemcpyAsync(h->dX, d_xr, nxs(h) * 8, hipMemcpyDefault, s));
    NDP_HIP(h, hipMemcpyAsync(h->dU, d_ur, nus(h) * 8, hipMemcpyDefault, s));
    NDP_HIP(h, hipMemsetAsync(h->dAct, 0, act_bytes(h), s));      // a new iterate: the QPs start from an empty active set
    return note_stream(h, s);
}

int ndp_reset(ndp_handle *h, const double *xr, const double *ur)
{
    if (!h || !xr || !ur) return -1;
    std::lock_guard<std::mutex> lk(h->mu);
    NDP_HIP(h, hipSetDevice(h->cfg.device));
    int rc = wait_all(h);
    if (rc) return rc;
    NDP_HIP(h, hipMemcpyAsync(h->dX, xr, nxs(h) * 8, hipMemcpyDefault, h->stream));
    NDP_HIP(h, hipMemcpyAsync(h->dU, ur, nus(h) * 8, hipMemcpyDefault, h->stream));
    NDP_HIP(h, hipMemsetAsync(h->dAct, 0, act_bytes(h), h->stream));
    NDP_HIP(h, hipStreamSynchronize(h->stream));
    return 0;
}

int ndp_step_device_ex(ndp_handle *h, const void *d_x0, const void *d_xr, const void *d_ur, const void *d_f,
                       const void *d_other, int other_stride, const void *d_other_index, const void *d_ego_xy,
                       void *d_u0, void *stream)
{
    if (!h || !d_x0 || !d_xr || !d_ur || !d_u0) return -1;
    std::lock_guard<std::mutex> lk(h->mu);
    hipStream_t s = stream ? (hipStream_t)stream : h->stream;
    NDP_HIP(h, hipSetDevice(h->cfg.device));
    Neigh nb;
    nb.other = (const double *)d_other; nb.stride = other_stride; nb.index = (const int *)d_other_index; nb.ego_xy = (const double *)d_ego_xy;
    int rc = enqueue_step(h, (const double *)d_x0, (const double *)d_xr, (const double *)d_ur, (const float *)d_f, nb,
                          (double *)d_u0, nullptr, s);
    return rc ? rc : note_stream(h, s);
}

int ndp_step_device(ndp_handle *h, const void *d_x0, const void *d_xr, const void *d_ur, const void *d_f,
                    const void *d_other, const void *d_ego_xy, void *d_u0, void *stream)
{
    return ndp_step_device_ex(h, d_x0, d_xr, d_ur, d_f, d_other, NX, nullptr, d_ego_xy, d_u0, stream);
}

// ---- downwash one tick ahead (second stream) + the control step that consumes it
static int ensure_prefetch(ndp_handle *h)
{
    if (h->aux) return 0;
    if (!h->cfg.use_fd) { h->err = "downwash prefetch needs use_fd = 1 (NDP model)"; return -8; }
    if (h->use_queue && !h->queue_auto) { h->err = "downwash prefetch is not combined with the interior-point work list (set cfg.work_queue = 2)"; return -16; }
    if (h->cfg.qp_precision) { h->err = "downwash prefetch serves the fp64 product path only"; return -12; }
    if (h->cfg.N + 1 > 32) { h->err = "downwash prefetch needs N + 1 <= 32 (an instance's rows in at most two 32-row tiles)"; return -12; }
    if (!h->have_mlp) { h->err = "downwash requested but ndp_set_mlp_weights was never called"; return -6; }
    // (h->aux, created last, marks completion; members a failed earlier call did allocate are reused, not allocated again)
    for (int i = 0; i < 2; ++i) {
        if (!h->dForceAB[i]) NDP_HIP(h, hipMalloc((void **)&h->dForceAB[i], nfs(h) * 4));
        NDP_HIP(h, hipMemset(h->dForceAB[i], 0, nfs(h) * 4));
    }
    h->pf_ntiles = (unsigned)((h->cfg.batch * (h->cfg.N + 1) + 31) / 32);
    const size_t proto_words = (size_t)PF_EPOCH + 2 * (size_t)h->pf_ntiles;
    if (!h->dProto) NDP_HIP(h, hipMalloc((void **)&h->dProto, proto_words * 8));
    NDP_HIP(h, hipMemset(h->dProto, 0, proto_words * 8));
    {
        const unsigned grid_rti = (unsigned)((h->cfg.batch + h->waves - 1) / h->waves);
        h->pf_groups_rti = grid_rti < PF_GROUPS ? grid_rti : PF_GROUPS;
    }
    // its own hardware queue: the downwash launch has to RUN beside the control step (a stream that shares the caller's hardware
    // queue would execute behind it).  Streams of another priority level get queues of their own.
    int lo = 0, hi = 0;
    NDP_HIP(h, hipDeviceGetStreamPriorityRange(&lo, &hi));
    if (!h->evFork) NDP_HIP(h, hipEventCreateWithFlags(&h->evFork, hipEventDisableTiming));
    if (!h->evJoin) NDP_HIP(h, hipEventCreateWithFlags(&h->evJoin, hipEventDisableTiming));
    NDP_HIP(h, hipStreamCreateWithPriority(&h->aux, hipStreamNonBlocking, hi));
    return 0;
}

int ndp_downwash_prefetch_device(ndp_handle *h, const void *d_other, int other_stride, const void *d_other_index,
                                 const void *d_ego_ref, const void *d_ego_xy, void *after_stream, void *on_stream)
{
    if (!h || !d_other || !d_ego_ref) return -1;
    std::lock_guard<std::mutex> lk(h->mu);
    NDP_HIP(h, hipSetDevice(h->cfg.device));
    int rc = ensure_prefetch(h);
    if (rc) return rc;
    if (other_stride != 10 && other_stride != 6) { h->err = "ndp_downwash_prefetch_device: other_stride must be 10 or 6"; return -13; }
    hipStream_t a = on_stream ? (hipStream_t)on_stream : h->aux;
    if (after_stream) {      // the windows are produced on that stream (and, inside a capture, this is what brings the second stream in)
        NDP_HIP(h, hipEventRecord(h->evFork, (hipStream_t)after_stream));
        NDP_HIP(h, hipStreamWaitEvent(a, h->evFork, 0));
    }
    const int np1 = h->cfg.N + 1, rows = h->cfg.batch * np1;
    const int ntiles = (rows + 31) / 32;
    hipLaunchKernelGGL(prefetch_gate_kernel, dim3(1), dim3(64), 0, a, h->dProto, h->prefetch_timeout_us, h->pf_groups_rti);
    hipLaunchKernelGGL(mlp_stream_kernel, dim3((ntiles + 3) / 4), dim3(256), 0, a, (const float *)h->dFrag, (const double *)d_other,
                       (const double *)d_ego_ref, (const double *)d_ego_xy, h->dForceAB[0], h->dForceAB[1], rows, np1,
                       h->cfg.r_horiz * h->cfg.r_horiz, other_stride, (const int *)d_other_index, h->dProto, peer_mapped(d_other));
    hipLaunchKernelGGL(prefetch_done_kernel, dim3(1), dim3(64), 0, a, h->dProto);
    NDP_HIP(h, hipGetLastError());
    return 0;
}

int ndp_prefetch_join(ndp_handle *h, void *stream)
{
    if (!h) return -1;
    std::lock_guard<std::mutex> lk(h->mu);
    if (!h->aux) return 0;
    NDP_HIP(h, hipSetDevice(h->cfg.device));
    NDP_HIP(h, hipEventRecord(h->evJoin, h->aux));
    NDP_HIP(h, hipStreamWaitEvent(stream ? (hipStream_t)stream : h->stream, h->evJoin, 0));
    return 0;
}

int ndp_step_device_prefetched(ndp_handle *h, const void *d_x0, const void *d_xr, const void *d_ur, void *d_u0, void *stream)
{
    if (!h || !d_x0 || !d_xr || !d_ur || !d_u0) return -1;
    std::lock_guard<std::mutex> lk(h->mu);
    hipStream_t s = stream ? (hipStream_t)stream : h->stream;
    NDP_HIP(h, hipSetDevice(h->cfg.device));
    int rc = ensure_prefetch(h);
    if (rc) return rc;
    rc = launch_rti(h, (const double *)d_x0, (const double *)d_xr, (const double *)d_ur, nullptr, (double *)d_u0, nullptr, s,
                    nullptr, nullptr, true);
    return rc ? rc : note_stream(h, s);
}

int ndp_prefetch_stats(ndp_handle *h, unsigned long long *out4 /* [5] */)
{
    if (!h || !out4) return -1;
    std::lock_guard<std::mutex> lk(h->mu);
    out4[0] = out4[1] = out4[2] = out4[3] = out4[4] = 0;
    if (!h->aux) return 0;
    NDP_HIP(h, hipSetDevice(h->cfg.device));
    int rc = wait_all(h);
    if (rc) return rc;
    NDP_HIP(h, hipStreamSynchronize(h->aux));
    std::vector<unsigned long long> w(PF_EPOCH);
    NDP_HIP(h, hipMemcpy(w.data(), h->dProto, (size_t)PF_EPOCH * 8, hipMemcpyDeviceToHost));
    out4[0] = w[PF_CUR_M]; out4[1] = w[PF_RTI_C2] / h->pf_groups_rti;
    out4[2] = (unsigned)(w[PF_MISSED] & 0xffffffffu); out4[3] = (unsigned)(w[PF_GATE_TIMEOUT] & 0xffffffffu);
    out4[4] = (unsigned)(w[PF_SLOW] & 0xffffffffu);
    return 0;
}

void *ndp_device_force_slot(ndp_handle *h, int slot) { return h && (slot == 0 || slot == 1) ? h->dForceAB[slot] : nullptr; }

// ---- host-array step: ndp_step_begin (pack -> H2D -> kernel -> D2H, nothing waits) + ndp_step_end (wait, hand the results over)
static int ensure_slots(ndp_handle *h)
{
    if (h->slots_ready) return 0;
    for (int i = 0; i < 2; ++i) {
        ndp_handle::HostSlot &sl = h->slot[i];
        // (only what is missing: a call that failed part-way left the members it did allocate, and they are reused -- ndp_destroy frees them)
        if (!sl.hIn) NDP_HIP(h, hipHostMalloc((void **)&sl.hIn, h->in_bytes, hipHostMallocDefault));
        if (!sl.hOut) {
            NDP_HIP(h, hipHostMalloc((void **)&sl.hOut, h->out_all, hipHostMallocDefault));
            memset(sl.hOut, 0, h->out_bytes);
        }
        if (!sl.evOut) NDP_HIP(h, hipEventCreateWithFlags(&sl.evOut, hipEventDisableTiming));
    }
    // pack threads: NDP_PACK_THREADS, else half the cores this process may really use, at most 8; the caller packs too, so small blocks
    // need none.  (Rounds 3-5 counted the machine's hardware threads: in a container whose CPU quota is a fraction of the machine that
    // put seven spinning threads on two or three cores' worth of time -- the same ndp_step_begin / _end leg measured 4.5 M solves/s on one
    // box and 11.5 M on another.)
    int nt = 0;
    h->host_cores = usable_cores();
    if (const char *e = getenv("NDP_PACK_THREADS")) nt = atoi(e);
    else if (h->in_bytes > 2 * PACK_CHUNK) nt = (h->host_cores / 2 > 8 ? 8 : h->host_cores / 2) - 1;
    h->pack_threads = nt > 0 ? (nt > 64 ? 64 : nt) : 0;
    h->pool.reset(new (std::nothrow) PackPool(h->pack_threads));
    if (!h->pool) { h->err = "ensure_slots: out of memory"; return -4; }
    h->slots_ready = true;
    return 0;
}

static int step_begin_locked(ndp_handle *h, const double *x0, const double *xr, const double *ur, const float *f,
                             const double *other, const double *ego_xy, bool want_iter, double *dump, bool f_f64 = false)
{
    const size_t B = h->cfg.batch;
    hipStream_t s = h->stream;
    NDP_HIP(h, hipSetDevice(h->cfg.device));
    int rc = ensure_slots(h);
    if (rc) return rc;
    if (h->slots_busy == 2) { h->err = "ndp_step_begin: two steps are already in flight (call ndp_step_end first)"; return -14; }
    if (h->ev_pending && (rc = wait_all(h))) return rc;       // work a caller left on its own stream comes first
    ndp_handle::HostSlot &sl = h->slot[h->slot_head];
    // the slot's previous use is over: its results were handed out by ndp_step_end (busy is false), so the kernel that read
    // its input mirror and wrote its output mirror has completed and both are free to overwrite
    // the block of THIS step: the arrays that were given, one behind the other (256-byte aligned), so that one transfer moves
    // exactly what the kernel reads; neighbour windows as their 6 position / velocity columns
    struct Seg { size_t off; const void *src; size_t len; size_t rows, row_len, src_stride; };
    Seg segs[6];
    int ns = 0;
    size_t o = 0;
    auto up256 = [](size_t x) { return (x + 255) & ~(size_t)255; };
    auto add = [&](const void *src, size_t len, size_t rows = 0, size_t row_len = 0, size_t src_stride = 0) {
        segs[ns] = {o, src, len, rows, row_len, src_stride};
        o += up256(len);
        return segs[ns++].off;
    };
    const size_t rows_o = B * (size_t)(h->cfg.N + 1);
    const size_t o_x0 = add(x0, B * NX * 8), o_xr = add(xr, nxs(h) * 8), o_ur = add(ur, nus(h) * 8);
    const size_t o_f = f ? add(f, nfs(h) * (f_f64 ? 8 : 4)) : 0;
    const size_t o_other = other ? add(other, rows_o * 6 * 8, rows_o, 6 * 8, NX * 8) : 0;
    const size_t o_ego = ego_xy ? add(ego_xy, B * 2 * 8) : 0;
    const size_t used = o;          // <= in_bytes (the mirror holds every array at full width)
    std::vector<PackJob> jobs;
    for (int i = 0; i < ns; ++i) {
        if (segs[i].rows) {
            const size_t rows_per = PACK_CHUNK / segs[i].src_stride;
            for (size_t r = 0; r < segs[i].rows; r += rows_per) {
                const size_t n = segs[i].rows - r < rows_per ? segs[i].rows - r : rows_per;
                jobs.push_back({sl.hIn + segs[i].off + r * segs[i].row_len, (const unsigned char *)segs[i].src + r * segs[i].src_stride,
                                segs[i].row_len, n, segs[i].src_stride});
            }
            continue;
        }
        for (size_t c = 0; c < segs[i].len; c += PACK_CHUNK) {
            const size_t n = segs[i].len - c < PACK_CHUNK ? segs[i].len - c : PACK_CHUNK;
            jobs.push_back({sl.hIn + segs[i].off + c, (const unsigned char *)segs[i].src + c, n});
        }
    }
    const int nj = (int)jobs.size();
    PackPool &pool = *h->pool;
    const auto tp0 = std::chrono::steady_clock::now();
    pool.post(jobs.data(), nj);      // from here to pool.finish() nothing returns early: the workers read `jobs`
    for (int i = 0; i < nj; ++i) pool.wait_job(i);
    pool.finish();
    const auto tp1 = std::chrono::steady_clock::now();
    // Zero-copy: the kernel reads the input mirror itself over PCIe and writes u0 | status | iterations (| the new iterate, when
    // asked for) into the output mirror itself: page-locked host memory, device-accessible, no DMA operation.  (One transfer of
    // the block per step into an HBM copy on a stream of its own, beside the previous tick's kernel, was built and measured in
    // four sessions: 85-103 us per step at batch 1024 against 90-96 this way, 290-343 against 315-342 at 4096 -- between 8 %
    // better and 10 % worse, one tick at a time always 5 % worse.  PCIe moves ~40-48 GB/s here whoever issues the reads.  Not kept.)
    const unsigned char *ib = sl.hIn;
    (void)used;
    Neigh nb;
    nb.other = other ? (const double *)(ib + o_other) : nullptr;
    nb.stride = 6;
    nb.ego_xy = ego_xy ? (const double *)(ib + o_ego) : nullptr;
    StepOut so;
    so.status = (int *)(sl.hOut + h->off_st); so.iters = (int *)(sl.hOut + h->off_it);
    if (want_iter) { so.Xm = (double *)(sl.hOut + h->out_bytes); so.Um = so.Xm + nxs(h); }
    so.f_f64 = f_f64;
    // one packet less between the kernel's end and the host seeing it: the completion event rides on the step's last dispatch packet
    const bool ext_done = !dump && h->cfg.qp_precision == 0 && (!other || can_fuse(h)) && !getenv("NDP_HOST_EVENT_RECORD");
    if (ext_done) so.done = sl.evOut;
    rc = enqueue_step(h, (const double *)(ib + o_x0), (const double *)(ib + o_xr), (const double *)(ib + o_ur),
                      f ? (const float *)(ib + o_f) : nullptr, nb, (double *)(sl.hOut + h->off_u0), dump ? h->sdbg : nullptr, s, &so);
    if (rc) return rc;
    // (sl.evOut was marked by the step's own dispatch packet -- so.done -- unless the step ran the separate downwash launch path,
    // a debug dump or a precision study, where it is recorded behind the launches as usual)
    if (!ext_done) NDP_HIP(h, hipEventRecord(sl.evOut, s));
    sl.busy = true; sl.want_iter = want_iter; sl.dump = dump;
    h->host_us[0] = std::chrono::duration<double, std::micro>(tp1 - tp0).count();
    h->host_us[1] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tp1).count();
    h->slot_head ^= 1;
    ++h->slots_busy;
    return 0;
}

static int step_end_locked(ndp_handle *h, double *u0, double *X_out, double *U_out, int32_t *status_out, int32_t *iters_out)
{
    const size_t B = h->cfg.batch;
    if (h->slots_busy == 0) { h->err = "ndp_step_end: no step in flight (ndp_step_begin first)"; return -14; }
    if (h->tslot[h->slot_tail].busy) { h->err = "ndp_step_end: the oldest call in flight is a tick (ndp_tick_end it)"; return -14; }
    ndp_handle::HostSlot &sl = h->slot[h->slot_tail];
    NDP_HIP(h, hipSetDevice(h->cfg.device));
    if ((X_out || U_out) && !sl.want_iter) { h->err = "ndp_step_end: the iterate was not requested at ndp_step_begin (flags bit 0)"; return -15; }
    // the step is a few tens of microseconds from done: poll the event before blocking on it (a blocking wait adds its wake-up)
    const auto tw0 = std::chrono::steady_clock::now();
    hipError_t e = hipErrorNotReady;
    for (int spin = 0; spin < 4000 && e == hipErrorNotReady; ++spin) e = hipEventQuery(sl.evOut);
    if (e == hipErrorNotReady) e = hipEventSynchronize(sl.evOut);
    if (e == hipSuccess && sl.dump)
        e = hipMemcpy(sl.dump, h->sdbg, (size_t)(lds_doubles(h->cfg.N) + DBG_EXTRA) * 8, hipMemcpyDeviceToHost);
    sl.busy = false;
    h->slot_tail ^= 1;
    --h->slots_busy;
    NDP_HIP(h, e);
    const auto tw1 = std::chrono::steady_clock::now();
    const unsigned char *ho = sl.hOut;
    if (u0) memcpy(u0, ho + h->off_u0, B * NU * 8);
    const int32_t *st = (const int32_t *)(ho + h->off_st);
    if (status_out) memcpy(status_out, st, B * 4);
    if (iters_out) copy_ipm_iters(iters_out, reinterpret_cast<const int32_t *>(ho + h->off_it), B);
    if (X_out) memcpy(X_out, ho + h->out_bytes, nxs(h) * 8);
    if (U_out) memcpy(U_out, ho + h->out_bytes + nxs(h) * 8, nus(h) * 8);
    int w = 0;
    for (size_t i = 0; i < B; ++i) w = st[i] > w ? st[i] : w;
    h->host_us[2] = std::chrono::duration<double, std::micro>(tw1 - tw0).count();
    h->host_us[3] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tw1).count();
    return w;
}

int ndp_debug_host_info(ndp_handle *h, int32_t *out3)
{
    if (!h || !out3) return -1;
    std::lock_guard<std::mutex> lk(h->mu);
    out3[0] = (int32_t)std::thread::hardware_concurrency();
    out3[1] = h->slots_ready ? h->host_cores : usable_cores();
    out3[2] = h->slots_ready ? h->pack_threads : -1;
    return 0;
}

int ndp_debug_host_timing(ndp_handle *h, double *out4)
{
    if (!h || !out4) return -1;
    std::lock_guard<std::mutex> lk(h->mu);
    for (int i = 0; i < 4; ++i) out4[i] = h->host_us[i];
    return 0;
}

int ndp_step_begin(ndp_handle *h, const double *x0, const double *xr, const double *ur, const float *f,
                   const double *other, const double *ego_xy, int flags)
{
    if (!h || !x0 || !xr || !ur) return -1;
    std::lock_guard<std::mutex> lk(h->mu);
    return step_begin_locked(h, x0, xr, ur, f, other, ego_xy, (flags & 1) != 0, nullptr);
}

int ndp_step_end(ndp_handle *h, double *u0, double *X_out, double *U_out, int32_t *status_out, int32_t *iters_out)
{
    if (!h || !u0) return -1;
    std::lock_guard<std::mutex> lk(h->mu);
    return step_end_locked(h, u0, X_out, U_out, status_out, iters_out);
}

// The synchronous form: begin + end under ONE lock (a concurrent caller cannot slip a step in between).
static int step_host(ndp_handle *h, const double *x0, const double *xr, const double *ur, const float *f,
                     const double *other, const double *ego_xy, double *u0, double *X_out, double *U_out,
                     int32_t *status_out, int32_t *iters_out, double *dump, bool f_f64 = false)
{
    if (!h || !x0 || !xr || !ur || !u0) return -1;
    std::lock_guard<std::mutex> lk(h->mu);
    if (h->slots_busy) { h->err = "ndp_step: steps begun with ndp_step_begin are still in flight (ndp_step_end them first)"; return -14; }
    int rc = step_begin_locked(h, x0, xr, ur, f, other, ego_xy, X_out || U_out, dump, f_f64);
    if (rc) return rc;
    return step_end_locked(h, u0, X_out, U_out, status_out, iters_out);
}

int ndp_step(ndp_handle *h, const double *x0, const double *xr, const double *ur, const float *f,
             const double *other, const double *ego_xy, double *u0)
{
    return step_host(h, x0, xr, ur, f, other, ego_xy, u0, nullptr, nullptr, nullptr, nullptr, nullptr);
}

int ndp_step_ex(ndp_handle *h, const double *x0, const double *xr, const double *ur, const float *f,
                const double *other, const double *ego_xy, double *u0, double *X_out, double *U_out,
                int32_t *status_out, int32_t *iters_out)
{
    return step_host(h, x0, xr, ur, f, other, ego_xy, u0, X_out, U_out, status_out, iters_out, nullptr);
}

int ndp_step_ex_f64(ndp_handle *h, const double *x0, const double *xr, const double *ur, const double *f, double *u0,
                    double *X_out, double *U_out, int32_t *status_out, int32_t *iters_out)
{
    return step_host(h, x0, xr, ur, reinterpret_cast<const float *>(f), nullptr, nullptr, u0, X_out, U_out, status_out, iters_out, nullptr, f != nullptr);
}

int ndp_step_debug(ndp_handle *h, const double *x0, const double *xr, const double *ur, const float *f,
                   const double *other, const double *ego_xy, double *u0, double *lds_dump)
{
    if (h && h->cfg.batch != 1) { h->err = "ndp_step_debug: batch must be 1"; return -9; }
    return step_host(h, x0, xr, ur, f, other, ego_xy, u0, nullptr, nullptr, nullptr, nullptr, lds_dump);
}

int ndp_downwash_device(ndp_handle *h, const void *d_other, const void *d_ego_ref, const void *d_ego_xy,
                        void *d_f_out, void *stream)
{
    if (!h || !d_other || !d_ego_ref || !d_f_out) return -1;
    std::lock_guard<std::mutex> lk(h->mu);
    hipStream_t s = stream ? (hipStream_t)stream : h->stream;
    NDP_HIP(h, hipSetDevice(h->cfg.device));
    Neigh nb;
    nb.other = (const double *)d_other; nb.ego_xy = (const double *)d_ego_xy;
    int rc = launch_mlp(h, nb, (const double *)d_ego_ref, (float *)d_f_out, s);
    return rc ? rc : note_stream(h, s);
}


// measurement hook: the LDS-free downwash kernel (see mlp_stream_kernel) on a caller's stream
int ndp_debug_downwash_stream_device(ndp_handle *h, const void *d_other, const void *d_ego_ref, const void *d_ego_xy,
                                     void *d_f_out, void *stream)
{
    if (!h || !d_other || !d_ego_ref || !d_f_out) return -1;
    std::lock_guard<std::mutex> lk(h->mu);
    hipStream_t s = stream ? (hipStream_t)stream : h->stream;
    NDP_HIP(h, hipSetDevice(h->cfg.device));
    if (!h->have_mlp) { h->err = "downwash requested but ndp_set_mlp_weights was never called"; return -6; }
    const int np1 = h->cfg.N + 1, rows = h->cfg.batch * np1;
    const int ntiles = (rows + 31) / 32;
    hipLaunchKernelGGL(mlp_stream_kernel, dim3((ntiles + 3) / 4), dim3(256), 0, s, (const float *)h->dFrag, (const double *)d_other,
                       (const double *)d_ego_ref, (const double *)d_ego_xy, (float *)d_f_out, (float *)d_f_out, rows, np1,
                       h->cfg.r_horiz * h->cfg.r_horiz, NX, (const int *)nullptr, (unsigned long long *)nullptr, peer_mapped(d_other));
    NDP_HIP(h, hipGetLastError());
    return note_stream(h, s);
}

int ndp_downwash(ndp_handle *h, const double *other, const double *ego_ref, const double *ego_xy, float *f_out)
{
    if (!h || !other || !ego_ref || !f_out) return -1;
    std::lock_guard<std::mutex> lk(h->mu);
    hipStream_t s = h->stream;
    const size_t B = h->cfg.batch;
    NDP_HIP(h, hipSetDevice(h->cfg.device));
    NDP_HIP(h, hipMemcpyAsync(h->sother, other, nxs(h) * 8, hipMemcpyHostToDevice, s));
    NDP_HIP(h, hipMemcpyAsync(h->sxr, ego_ref, nxs(h) * 8, hipMemcpyHostToDevice, s));
    if (ego_xy) NDP_HIP(h, hipMemcpyAsync(h->sego, ego_xy, B * 2 * 8, hipMemcpyHostToDevice, s));
    Neigh nb;
    nb.other = h->sother; nb.ego_xy = ego_xy ? h->sego : nullptr;
    int rc = launch_mlp(h, nb, h->sxr, h->dForce, s);
    if (rc) return rc;
    NDP_HIP(h, hipMemcpyAsync(f_out, h->dForce, nfs(h) * 4, hipMemcpyDeviceToHost, s));
    NDP_HIP(h, hipStreamSynchronize(s));
    return 0;
}

int ndp_get_iterate(ndp_handle *h, double *X, double *U)
{
    if (!h) return -1;
    std::lock_guard<std::mutex> lk(h->mu);
    NDP_HIP(h, hipSetDevice(h->cfg.device));
    int rc = wait_all(h);
    if (rc) return rc;
    if (X) NDP_HIP(h, hipMemcpy(X, h->dX, nxs(h) * 8, hipMemcpyDefault));
    if (U) NDP_HIP(h, hipMemcpy(U, h->dU, nus(h) * 8, hipMemcpyDefault));
    return 0;
}

int ndp_set_iterate(ndp_handle *h, const double *X, const double *U)
{
    if (!h) return -1;
    std::lock_guard<std::mutex> lk(h->mu);
    NDP_HIP(h, hipSetDevice(h->cfg.device));
    int rc = wait_all(h);
    if (rc) return rc;
    if (X) NDP_HIP(h, hipMemcpy(h->dX, X, nxs(h) * 8, hipMemcpyDefault));
    if (U) NDP_HIP(h, hipMemcpy(h->dU, U, nus(h) * 8, hipMemcpyDefault));
    NDP_HIP(h, hipMemset(h->dAct, 0, act_bytes(h)));
    return 0;
}

int ndp_get_active_set(ndp_handle *h, int32_t *sweeps, int8_t *act)
{
    if (!h) return -1;
    std::lock_guard<std::mutex> lk(h->mu);
    NDP_HIP(h, hipSetDevice(h->cfg.device));
    int rc = wait_all(h);
    if (rc) return rc;
    const size_t B = (size_t)h->cfg.batch;
    if (sweeps) {
        NDP_HIP(h, hipMemcpy(sweeps, h->lastIters, B * 4, hipMemcpyDefault));
        for (size_t i = 0; i < B; ++i) sweeps[i] = (int32_t)((uint32_t)sweeps[i] >> ITERS_SWEEP_SHIFT);
    }
    if (act) NDP_HIP(h, hipMemcpy(act, h->dAct, act_bytes(h), hipMemcpyDefault));
    return 0;
}

int ndp_set_active_set(ndp_handle *h, const int8_t *act)
{
    if (!h || !act) return -1;
    std::lock_guard<std::mutex> lk(h->mu);
    NDP_HIP(h, hipSetDevice(h->cfg.device));
    int rc = wait_all(h);
    if (rc) return rc;
    const size_t n = act_bytes(h);
    for (size_t i = 0; i < n; ++i)
        if (act[i] < -1 || act[i] > 1) { h->err = "ndp_set_active_set: entries are -1, 0 or +1"; return -1; }
    NDP_HIP(h, hipMemcpy(h->dAct, act, n, hipMemcpyDefault));
    return 0;
}

int ndp_get_status(ndp_handle *h, int32_t *status, int32_t *ipm_iters)
{
    if (!h) return -1;
    std::lock_guard<std::mutex> lk(h->mu);
    NDP_HIP(h, hipSetDevice(h->cfg.device));
    int rc = wait_all(h);
    if (rc) return rc;
    if (status) NDP_HIP(h, hipMemcpy(status, h->lastStatus, (size_t)h->cfg.batch * 4, hipMemcpyDefault));
    if (ipm_iters) {
        NDP_HIP(h, hipMemcpy(ipm_iters, h->lastIters, (size_t)h->cfg.batch * 4, hipMemcpyDefault));
        copy_ipm_iters(ipm_iters, ipm_iters, (size_t)h->cfg.batch);
    }
    return 0;
}

// ---- f3: hover-throttle estimator + actuator command (reference constants: params/estimator_params.py:13-18)
static ThrCfg thr_cfg(const ndp_handle *h)
{
    const double ts = 0.02, tau = 0.05;
    ThrCfg c;
    c.a1 = (2.0 * tau - ts) / (2.0 * tau + ts);
    c.a2 = 2.0 / (2.0 * tau + ts);
    c.hm = 1.0 / h->cfg.mass;
    c.g = h->cfg.gravity;
    c.R = 1.225; c.Q0 = 0.1; c.Q1 = 0.1;
    c.mass = h->cfg.mass;
    return c;
}

int ndp_throttle_reset(ndp_handle *h)
{
    if (!h) return -1;
    std::lock_guard<std::mutex> lk(h->mu);
    NDP_HIP(h, hipSetDevice(h->cfg.device));
    hipLaunchKernelGGL(throttle_reset_kernel, dim3((h->cfg.batch + 255) / 256), dim3(256), 0, h->stream, h->dThr, 50.0, h->cfg.batch);
    NDP_HIP(h, hipGetLastError());
    NDP_HIP(h, hipStreamSynchronize(h->stream));
    return 0;
}

// The host-pointer forms below hold the handle's lock from the first staging copy to the read-back: they share the
// staging area sThr (and sx0 / sxr / sur / su0 of the step), which a concurrent call must not overwrite in between.
static int launch_throttle(ndp_handle *h, const double *d_vz, const double *d_throttle, double *d_k, hipStream_t s)
{
    hipLaunchKernelGGL(throttle_kernel, dim3((h->cfg.batch + 255) / 256), dim3(256), 0, s, thr_cfg(h), h->dThr, d_vz, d_throttle, d_k, h->cfg.batch);
    NDP_HIP(h, hipGetLastError());
    return 0;
}

int ndp_throttle_update_device(ndp_handle *h, const void *d_vz, const void *d_throttle, void *d_k, void *stream)
{
    if (!h || !d_vz || !d_throttle || !d_k) return -1;
    std::lock_guard<std::mutex> lk(h->mu);
    hipStream_t s = stream ? (hipStream_t)stream : h->stream;
    NDP_HIP(h, hipSetDevice(h->cfg.device));
    int rc = launch_throttle(h, (const double *)d_vz, (const double *)d_throttle, (double *)d_k, s);
    return rc ? rc : note_stream(h, s);
}

int ndp_throttle_update(ndp_handle *h, const double *vz, const double *throttle, double *k)
{
    if (!h || !vz || !throttle || !k) return -1;
    const size_t B = h->cfg.batch;
    std::lock_guard<std::mutex> lk(h->mu);
    NDP_HIP(h, hipSetDevice(h->cfg.device));
    NDP_HIP(h, hipMemcpyAsync(h->sThr, vz, B * 8, hipMemcpyHostToDevice, h->stream));
    NDP_HIP(h, hipMemcpyAsync(h->sThr + B, throttle, B * 8, hipMemcpyHostToDevice, h->stream));
    int rc = launch_throttle(h, h->sThr, h->sThr + B, h->sThr + 2 * B, h->stream);
    if (rc) return rc;
    NDP_HIP(h, hipMemcpyAsync(k, h->sThr + 2 * B, B * 8, hipMemcpyDeviceToHost, h->stream));
    NDP_HIP(h, hipStreamSynchronize(h->stream));
    return 0;
}

static int launch_actuator(ndp_handle *h, const double *d_u0, const double *d_k, double *d_cmd, hipStream_t s)
{
    hipLaunchKernelGGL(actuator_kernel, dim3((h->cfg.batch + 255) / 256), dim3(256), 0, s, d_u0, d_k, d_cmd, h->cfg.mass, h->cfg.batch);
    NDP_HIP(h, hipGetLastError());
    return 0;
}

int ndp_actuator_cmd_device(ndp_handle *h, const void *d_u0, const void *d_k, void *d_cmd, void *stream)
{
    if (!h || !d_u0 || !d_k || !d_cmd) return -1;
    std::lock_guard<std::mutex> lk(h->mu);
    hipStream_t s = stream ? (hipStream_t)stream : h->stream;
    NDP_HIP(h, hipSetDevice(h->cfg.device));
    int rc = launch_actuator(h, (const double *)d_u0, (const double *)d_k, (double *)d_cmd, s);
    return rc ? rc : note_stream(h, s);
}

int ndp_actuator_cmd(ndp_handle *h, const double *u0, const double *k, double *cmd)
{
    if (!h || !u0 || !k || !cmd) return -1;
    const size_t B = h->cfg.batch;
    std::lock_guard<std::mutex> lk(h->mu);
    NDP_HIP(h, hipSetDevice(h->cfg.device));
    NDP_HIP(h, hipMemcpyAsync(h->sThr + 2 * B, k, B * 8, hipMemcpyHostToDevice, h->stream));
    NDP_HIP(h, hipMemcpyAsync(h->sThr + 3 * B, u0, B * 32, hipMemcpyHostToDevice, h->stream));
    int rc = launch_actuator(h, h->sThr + 3 * B, h->sThr + 2 * B, h->sThr + 7 * B, h->stream);
    if (rc) return rc;
    NDP_HIP(h, hipMemcpyAsync(cmd, h->sThr + 7 * B, B * 32, hipMemcpyDeviceToHost, h->stream));
    NDP_HIP(h, hipStreamSynchronize(h->stream));
    return 0;
}

int ndp_throttle_get_state(ndp_handle *h, double *state)
{
    if (!h || !state) return -1;
    std::lock_guard<std::mutex> lk(h->mu);
    const size_t B = h->cfg.batch;
    NDP_HIP(h, hipSetDevice(h->cfg.device));
    int rc = wait_all(h);
    if (rc) return rc;
    std::vector<double> soa(B * 8);
    NDP_HIP(h, hipMemcpy(soa.data(), h->dThr, B * 64, hipMemcpyDeviceToHost));
    for (size_t v = 0; v < B; ++v)
        for (int i = 0; i < 8; ++i) state[v * 8 + i] = soa[(size_t)i * B + v];
    return 0;
}

// ---- f2: follower reference relay
int ndp_relay_reset(ndp_handle *h)
{
    if (!h) return -1;
    std::lock_guard<std::mutex> lk(h->mu);
    NDP_HIP(h, hipSetDevice(h->cfg.device));
    NDP_HIP(h, hipMemsetAsync(h->dRelay, 0, (size_t)h->cfg.batch * 32, h->stream));
    NDP_HIP(h, hipStreamSynchronize(h->stream));
    return 0;
}

int ndp_relay_formation(ndp_handle *h, const double *form, double *offset_out)
{
    if (!h || !form) return -1;
    std::lock_guard<std::mutex> lk(h->mu);
    const size_t B = h->cfg.batch;
    NDP_HIP(h, hipSetDevice(h->cfg.device));
    NDP_HIP(h, hipMemcpyAsync(h->sThr, form, B * 24, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(relay_formation_kernel, dim3((B + 255) / 256), dim3(256), 0, h->stream, 0.8, h->dRelay, (const double *)h->sThr, (int)B);
    NDP_HIP(h, hipGetLastError());
    if (offset_out) {
        std::vector<double> st(B * 4);
        NDP_HIP(h, hipMemcpyAsync(st.data(), h->dRelay, B * 32, hipMemcpyDeviceToHost, h->stream));
        NDP_HIP(h, hipStreamSynchronize(h->stream));
        for (size_t v = 0; v < B; ++v)
            for (int a = 0; a < 3; ++a) offset_out[v * 3 + a] = st[v * 4 + a];
    } else {
        NDP_HIP(h, hipStreamSynchronize(h->stream));
    }
    return 0;
}

static int launch_relay_reference(ndp_handle *h, const double *d_xr_lead, double *d_xr_out, hipStream_t s)
{
    const int np1 = h->cfg.N + 1, rows = h->cfg.batch * np1;
    const size_t pieces = (size_t)rows * 5, per_block = 256 * RELAY_UNROLL;
    hipLaunchKernelGGL(relay_reference_kernel, dim3((unsigned)((pieces + per_block - 1) / per_block)), dim3(256), 0, s, (const double *)h->dRelay, d_xr_lead, d_xr_out, rows, np1);
    NDP_HIP(h, hipGetLastError());
    return 0;
}

int ndp_relay_reference_device(ndp_handle *h, const void *d_xr_lead, void *d_xr_out, void *stream)
{
    if (!h || !d_xr_lead || !d_xr_out) return -1;
    std::lock_guard<std::mutex> lk(h->mu);
    hipStream_t s = stream ? (hipStream_t)stream : h->stream;
    NDP_HIP(h, hipSetDevice(h->cfg.device));
    int rc = launch_relay_reference(h, (const double *)d_xr_lead, (double *)d_xr_out, s);
    return rc ? rc : note_stream(h, s);
}

int ndp_relay_reference(ndp_handle *h, const double *xr_lead, double *xr_out)
{
    if (!h || !xr_lead || !xr_out) return -1;
    std::lock_guard<std::mutex> lk(h->mu);
    NDP_HIP(h, hipSetDevice(h->cfg.device));
    NDP_HIP(h, hipMemcpyAsync(h->sother, xr_lead, nxs(h) * 8, hipMemcpyHostToDevice, h->stream));
    int rc = launch_relay_reference(h, h->sother, h->sxr, h->stream);
    if (rc) return rc;
    NDP_HIP(h, hipMemcpyAsync(xr_out, h->sxr, nxs(h) * 8, hipMemcpyDeviceToHost, h->stream));
    NDP_HIP(h, hipStreamSynchronize(h->stream));
    return 0;
}

// ---- f1: reference window generation
int ndp_ref_set_trajectory(ndp_handle *h, int n_seg, const double *coeff_x, const double *coeff_y, const double *coeff_z,
                           const double *coeff_yaw, const double *time_cum, const double *time_seg, const double *final_pt)
{
    if (!h || n_seg < 1 || !coeff_x || !coeff_y || !coeff_z || !coeff_yaw || !time_cum || !time_seg || !final_pt) return -1;
    std::lock_guard<std::mutex> lk(h->mu);
    const size_t B = h->cfg.batch, S = (size_t)n_seg;
    const size_t n_coeff = B * S * 28, n_cum = B * (S + 1), n_seg_t = B * S, total = n_coeff + n_cum + n_seg_t + B * 3;
    std::vector<double> host(total);
    for (size_t b = 0; b < B; ++b)
        for (size_t s = 0; s < S; ++s) {
            double *d = host.data() + (b * S + s) * 28;          // interleave the four message arrays per segment
            for (int i = 0; i < 8; ++i) {
                d[i] = coeff_x[(b * S + s) * 8 + i];
                d[8 + i] = coeff_y[(b * S + s) * 8 + i];
                d[16 + i] = coeff_z[(b * S + s) * 8 + i];
            }
            for (int i = 0; i < 4; ++i) d[24 + i] = coeff_yaw[(b * S + s) * 4 + i];
        }
    memcpy(host.data() + n_coeff, time_cum, n_cum * 8);
    memcpy(host.data() + n_coeff + n_cum, time_seg, n_seg_t * 8);
    memcpy(host.data() + n_coeff + n_cum + n_seg_t, final_pt, B * 3 * 8);
    NDP_HIP(h, hipSetDevice(h->cfg.device));
    int rc = wait_all(h);
    if (rc) return rc;
    if (h->dTraj) { (void)hipFree(h->dTraj); h->dTraj = nullptr; }
    // (+ the one-launch tick's segment cache, [B][SEGC_PER] doubles, empty = NaNs: tick_early; + the segment hints, int[B]: ref_point;
    //  + the cache's second copy behind them: tick_cache_store)
    const size_t hint_doubles = (B * 4 + 7) / 8;
    NDP_HIP(h, hipMalloc((void **)&h->dTraj, (total + 2 * B * SEGC_PER + hint_doubles) * 8));
    NDP_HIP(h, hipMemcpy(h->dTraj, host.data(), total * 8, hipMemcpyHostToDevice));
    NDP_HIP(h, hipMemset(h->dTraj + total, 0xFF, B * SEGC_PER * 8));
    NDP_HIP(h, hipMemset(h->dTraj + total + B * SEGC_PER, 0, hint_doubles * 8));
    NDP_HIP(h, hipMemset(h->dTraj + total + B * SEGC_PER + hint_doubles, 0xFF, B * SEGC_PER * 8));
    h->segc_par = 0;
    h->traj_seg = n_seg;
    return 0;
}

// enqueue helper (no locking): windows at node-0 times d_t[b] (or 0 when null) + toff
static int launch_ref_window(ndp_handle *h, const double *d_t, double toff, double *d_xr, double *d_ur, hipStream_t s)
{
    if (!h->dTraj) { h->err = "ndp_ref_window: ndp_ref_set_trajectory was never called"; return -11; }
    const size_t B = h->cfg.batch, S = (size_t)h->traj_seg;
    const double *coeff = h->dTraj, *cum = coeff + B * S * 28, *seg = cum + B * (S + 1), *fpt = seg + B * S;
    RefCfg cf{h->cfg.batch, h->cfg.N, h->traj_seg, h->cfg.dt, h->cfg.mass, h->cfg.gravity, toff};
    const int rows = h->cfg.batch * (h->cfg.N + 1);
    hipLaunchKernelGGL(ref_window_kernel, dim3((rows + REF_ROWS - 1) / REF_ROWS), dim3(REF_ROWS), 0, s, cf, coeff, cum, seg, fpt,
                       d_t, d_xr, d_ur);
    NDP_HIP(h, hipGetLastError());
    return 0;
}

int ndp_ref_window_device(ndp_handle *h, const void *d_t, void *d_xr, void *d_ur, void *stream)
{
    if (!h || !d_t || !d_xr || !d_ur) return -1;
    std::lock_guard<std::mutex> lk(h->mu);
    hipStream_t s = stream ? (hipStream_t)stream : h->stream;
    NDP_HIP(h, hipSetDevice(h->cfg.device));
    int rc = launch_ref_window(h, (const double *)d_t, 0.0, (double *)d_xr, (double *)d_ur, s);
    return rc ? rc : note_stream(h, s);
}

int ndp_ref_window(ndp_handle *h, const double *t, double *xr, double *ur)
{
    if (!h || !t || !xr || !ur) return -1;
    std::lock_guard<std::mutex> lk(h->mu);
    NDP_HIP(h, hipSetDevice(h->cfg.device));
    NDP_HIP(h, hipMemcpyAsync(h->sThr, t, (size_t)h->cfg.batch * 8, hipMemcpyHostToDevice, h->stream));
    int rc = launch_ref_window(h, h->sThr, 0.0, h->sxr, h->sur, h->stream);
    if (rc) return rc;
    NDP_HIP(h, hipMemcpyAsync(xr, h->sxr, nxs(h) * 8, hipMemcpyDeviceToHost, h->stream));
    NDP_HIP(h, hipMemcpyAsync(ur, h->sur, nus(h) * 8, hipMemcpyDeviceToHost, h->stream));
    NDP_HIP(h, hipStreamSynchronize(h->stream));
    return 0;
}

// ---- f1, the reference's sliding list (ref_list_* kernels; layout: RingGeom)
static RingGeom ring_geom(const ndp_handle *h) { return RingGeom{h->list_step, h->cfg.N + 1}; }
static int list_ring(const ndp_handle *h) { return ring_geom(h).ring(); }

static int list_alloc(ndp_handle *h)
{
    if (h->dRingX) return 0;
    const RingGeom rg = ring_geom(h);
    NDP_HIP(h, hipMalloc((void **)&h->dRingX, (size_t)h->cfg.batch * (rg.px() + rg.pu()) * 8));
    h->dRingU = h->dRingX + (size_t)h->cfg.batch * rg.px();
    return 0;
}

static RefCfg ref_cfg(const ndp_handle *h, double toff)
{
    return RefCfg{h->cfg.batch, h->cfg.N, h->traj_seg, h->cfg.dt, h->cfg.mass, h->cfg.gravity, toff};
}

// points at (d_t ? d_t[b] : 0) + toff + i * ts_nmpc, i = 0 .. npts-1, become list entries j0 + i (dup0: point 0 also entry j0 - 1)
static int launch_list_fill(ndp_handle *h, const double *d_t, double toff, int npts, unsigned long long j0, int dup0, hipStream_t s)
{
    if (!h->dTraj) { h->err = "ndp_ref_list: ndp_ref_set_trajectory was never called"; return -11; }
    const size_t B = h->cfg.batch, S = (size_t)h->traj_seg;
    const double *coeff = h->dTraj, *cum = coeff + B * S * 28, *seg = cum + B * (S + 1), *fpt = seg + B * S;
    const int n = h->cfg.batch * npts;
    const int bs = npts == 1 ? 64 : 256;        // one point per vehicle (the per-tick advance): small blocks spread over the CUs
    hipLaunchKernelGGL(ref_list_fill_kernel, dim3((n + bs - 1) / bs), dim3(bs), 0, s, ref_cfg(h, toff), coeff, cum, seg, fpt, d_t,
                       h->cfg.ts_nmpc, npts, j0, ring_geom(h), dup0, h->dRingX, h->dRingU);
    NDP_HIP(h, hipGetLastError());
    return 0;
}

int ndp_ref_list_reset(ndp_handle *h)
{
    if (!h) return -1;
    std::lock_guard<std::mutex> lk(h->mu);
    NDP_HIP(h, hipSetDevice(h->cfg.device));
    int rc = list_alloc(h);
    if (rc) return rc;
    if ((rc = wait_all(h))) return rc;
    h->list_n = 0;
    // entries 1 .. ring-1 = the points at i * ts_nmpc, i = 0 .. ring-2; the first one duplicated as entry 0 (:62-76)
    rc = launch_list_fill(h, nullptr, 0.0, list_ring(h) - 1, 1, 1, h->stream);
    if (rc) return rc;
    NDP_HIP(h, hipStreamSynchronize(h->stream));
    return 0;
}

int ndp_ref_list_fix_pt(ndp_handle *h, const double *x_odom, int quirk_b1)
{
    if (!h || !x_odom) return -1;
    std::lock_guard<std::mutex> lk(h->mu);
    NDP_HIP(h, hipSetDevice(h->cfg.device));
    int rc = list_alloc(h);
    if (rc) return rc;
    if ((rc = wait_all(h))) return rc;
    h->list_n = 0;
    NDP_HIP(h, hipMemcpyAsync(h->sThr, x_odom, (size_t)h->cfg.batch * 80, hipMemcpyHostToDevice, h->stream));
    const RingGeom rg = ring_geom(h);
    const size_t n = (size_t)h->cfg.batch * rg.step * 2 * rg.np1;
    hipLaunchKernelGGL(ref_list_fix_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, (const double *)h->sThr,
                       quirk_b1 ? h->cfg.mass * h->cfg.gravity : h->cfg.gravity, h->cfg.batch, rg, h->dRingX, h->dRingU);
    NDP_HIP(h, hipGetLastError());
    NDP_HIP(h, hipStreamSynchronize(h->stream));
    return 0;
}

// pop the oldest entry, append the point at trajectory time t + T_horizon (get_nmpc_pts, :79-93)
static int list_advance(ndp_handle *h, const double *d_t, hipStream_t s)
{
    if (!h->dRingX) { h->err = "ndp_ref_list_advance: no list (ndp_ref_list_reset / ndp_ref_list_fix_pt first)"; return -11; }
    const int rc = launch_list_fill(h, d_t, h->cfg.N * h->cfg.dt, 1, h->list_n + (unsigned long long)list_ring(h), 0, s);
    if (rc) return rc;                                             // nothing was launched (e.g. no trajectory): the list stays as it is
    ++h->list_n;
    return 0;
}

int ndp_ref_list_advance_device(ndp_handle *h, const void *d_t, void *stream)
{
    if (!h || !d_t) return -1;
    std::lock_guard<std::mutex> lk(h->mu);
    hipStream_t s = stream ? (hipStream_t)stream : h->stream;
    NDP_HIP(h, hipSetDevice(h->cfg.device));
    int rc = list_advance(h, (const double *)d_t, s);
    return rc ? rc : note_stream(h, s);
}

static int launch_list_window(ndp_handle *h, double *d_xr, double *d_ur, hipStream_t s)
{
    if (!h->dRingX) { h->err = "ndp_ref_list_window: no list (ndp_ref_list_reset / ndp_ref_list_fix_pt first)"; return -11; }
    const size_t n = (size_t)h->cfg.batch * (5 * (h->cfg.N + 1) + 2 * h->cfg.N), per_block = 256 * WIN_UNROLL;
    hipLaunchKernelGGL(ref_list_window_kernel, dim3((unsigned)((n + per_block - 1) / per_block)), dim3(256), 0, s, (const double *)h->dRingX,
                       (const double *)h->dRingU, ring_geom(h), h->list_n, h->cfg.batch, d_xr, d_ur);
    NDP_HIP(h, hipGetLastError());
    return 0;
}

int ndp_ref_list_window_device(ndp_handle *h, void *d_xr, void *d_ur, void *stream)
{
    if (!h || !d_xr || !d_ur) return -1;
    std::lock_guard<std::mutex> lk(h->mu);
    hipStream_t s = stream ? (hipStream_t)stream : h->stream;
    NDP_HIP(h, hipSetDevice(h->cfg.device));
    int rc = launch_list_window(h, (double *)d_xr, (double *)d_ur, s);
    return rc ? rc : note_stream(h, s);
}

// t == NULL: only read the current window (get_nmpc_ref_from_long_list); else advance first (get_nmpc_pts)
int ndp_ref_list_window(ndp_handle *h, const double *t, double *xr, double *ur)
{
    if (!h || !xr || !ur) return -1;
    std::lock_guard<std::mutex> lk(h->mu);
    NDP_HIP(h, hipSetDevice(h->cfg.device));
    int rc = 0;
    if (t) {
        NDP_HIP(h, hipMemcpyAsync(h->sThr, t, (size_t)h->cfg.batch * 8, hipMemcpyHostToDevice, h->stream));
        if ((rc = list_advance(h, h->sThr, h->stream))) return rc;
    }
    if ((rc = launch_list_window(h, h->sxr, h->sur, h->stream))) return rc;
    NDP_HIP(h, hipMemcpyAsync(xr, h->sxr, nxs(h) * 8, hipMemcpyDeviceToHost, h->stream));
    NDP_HIP(h, hipMemcpyAsync(ur, h->sur, nus(h) * 8, hipMemcpyDeviceToHost, h->stream));
    NDP_HIP(h, hipStreamSynchronize(h->stream));
    return 0;
}

// ---- the node's control tick, end to end on the device (nmpc_node.py:211-231; kernels: rti_kernel<..., TICK>, or tick_pre_kernel + rti_kernel)
static int ensure_tick(ndp_handle *h)
{
    if (h->dTickThrust) return 0;
    NDP_HIP(h, hipMalloc((void **)&h->dTickThrust, (size_t)h->cfg.batch * 8));
    NDP_HIP(h, hipMemsetAsync(h->dTickThrust, 0, (size_t)h->cfg.batch * 8, h->stream));     // AttitudeTarget(): thrust 0 (nmpc_node.py:104)
    NDP_HIP(h, hipStreamSynchronize(h->stream));
    return 0;
}

int ndp_tick_config(ndp_handle *h, const int32_t *other_index, int gate_on_odometry)
{
    if (!h) return -1;
    std::lock_guard<std::mutex> lk(h->mu);
    NDP_HIP(h, hipSetDevice(h->cfg.device));
    int rc = wait_all(h);
    if (rc) return rc;
    if ((rc = ensure_tick(h))) return rc;
    const size_t B = h->cfg.batch;
    bool any = false;
    if (other_index)
        for (size_t i = 0; i < B; ++i) {
            if (other_index[i] >= (int32_t)B) { h->err = "ndp_tick_config: other_index names an instance outside the handle"; return -2; }
            any = any || other_index[i] >= 0;
        }
    if (any && !h->cfg.use_fd) { h->err = "ndp_tick_config: neighbours (downwash) need use_fd = 1 (NDP model)"; return -8; }
    if (any && !h->have_mlp) { h->err = "ndp_tick_config: neighbours given but ndp_set_mlp_weights was never called"; return -6; }
    if (any) {
        if (!h->dTickIndex) NDP_HIP(h, hipMalloc((void **)&h->dTickIndex, B * 4));
        NDP_HIP(h, hipMemcpy(h->dTickIndex, other_index, B * 4, hipMemcpyHostToDevice));
    } else if (h->dTickIndex) {
        (void)hipFree(h->dTickIndex);
        h->dTickIndex = nullptr;
    }
    h->tick_gate = gate_on_odometry != 0;
    h->tick_remote = nullptr;
    return 0;
}

// The control tick with neighbours on OTHER ranks (nmpc_node.py:116-133,229-230 -> ndp_nmpc_leader_node.py:40,60-76: every vehicle
// publishes its window every tick, the leader consumes its neighbour's): the neighbour rows come from the caller's exchange buffer, and
// a tick is three enqueues with the exchange between the first two and the last:
//   ndp_tick_advance_device    list advance (+ estimator): this rank's window of the tick is complete, node N included
//   ndp_tick_window_pv_device  that window's position / velocity columns [B][N+1][6] -> the exchange's send buffer   ... exchange ...
//   ndp_tick_step_device       the control step (gate + network + RTI + actuator command), neighbour rows from the gathered windows
// Same arithmetic as the one-launch tick with the neighbour in the same handle (bit-equal: tests/test_tick.py).
int ndp_tick_config_remote(ndp_handle *h, const void *d_windows, int stride, int64_t rows, const int32_t *other_index, int gate_on_odometry)
{
    if (!h || !d_windows || !other_index) return -1;
    std::lock_guard<std::mutex> lk(h->mu);
    NDP_HIP(h, hipSetDevice(h->cfg.device));
    int rc = wait_all(h);
    if (rc) return rc;
    if ((rc = ensure_tick(h))) return rc;
    if (stride != 6 && stride != NX) { h->err = "ndp_tick_config_remote: stride must be 6 (position / velocity columns) or 10"; return -13; }
    if (!h->cfg.use_fd) { h->err = "ndp_tick_config_remote: neighbours (downwash) need use_fd = 1 (NDP model)"; return -8; }
    if (!h->have_mlp) { h->err = "ndp_tick_config_remote: ndp_set_mlp_weights was never called"; return -6; }
    if (!can_fuse(h)) { h->err = "ndp_tick_config_remote: serves the shapes whose downwash is fused into the control step (N + 1 <= 32, fp64)"; return -12; }
    const size_t B = h->cfg.batch;
    for (size_t i = 0; i < B; ++i)
        if ((int64_t)other_index[i] >= rows) { h->err = "ndp_tick_config_remote: other_index names a row outside the window buffer"; return -2; }
    if (!h->dTickIndex) NDP_HIP(h, hipMalloc((void **)&h->dTickIndex, B * 4));
    NDP_HIP(h, hipMemcpy(h->dTickIndex, other_index, B * 4, hipMemcpyHostToDevice));
    h->tick_gate = gate_on_odometry != 0;
    h->tick_remote = (const double *)d_windows;
    h->tick_remote_stride = stride;
    return 0;
}

// nmpc_ctl.reset(*ref_pub.get_nmpc_ref_from_long_list()) (nmpc_node.py:92,151-152): the iterate := the list's current window
int ndp_tick_reset(ndp_handle *h)
{
    if (!h) return -1;
    std::lock_guard<std::mutex> lk(h->mu);
    NDP_HIP(h, hipSetDevice(h->cfg.device));
    if (h->slots_busy) { h->err = "ndp_tick_reset: ticks are still in flight (ndp_tick_end them first)"; return -14; }
    int rc = wait_all(h);
    if (rc) return rc;
    if ((rc = launch_list_window(h, h->dX, h->dU, h->stream))) return rc;
    NDP_HIP(h, hipMemsetAsync(h->dAct, 0, act_bytes(h), h->stream));      // reset(): the QPs start from an empty active set
    NDP_HIP(h, hipStreamSynchronize(h->stream));
    return 0;
}

enum { TICK_ESTIMATE = NDP_TICK_ESTIMATE, TICK_WANT_U0 = NDP_TICK_WANT_U0, TICK_T_UNIFORM = NDP_TICK_T_UNIFORM };

// One tick's launches on `s`.  Every pointer is device-accessible (HBM or page-locked host memory): x_odom[B][10]; t[B] or null (the
// list is not advanced: hover at a fixed point, or a vehicle between two trajectories); vz[B] or null (column 5 of x_odom);
// throttle[B] or null (the thrust this handle commanded last tick); cmd[B][4]; u0_copy[B][4] or null.
// adv: the list is advanced; its times are t[B] (device-accessible), or t_all for every vehicle when t is null.
static int tick_enqueue(ndp_handle *h, hipStream_t s, const double *x_odom, bool adv, const double *t, double t_all, const double *vz,
                        const double *throttle, int flags, double *cmd, double *u0_copy, StepOut so)
{
    int rc = ensure_tick(h);
    if (rc) return rc;
    if (!h->dRingX) { h->err = "ndp_tick: no reference list (ndp_ref_list_fix_pt, or ndp_ref_set_trajectory + ndp_ref_list_reset, first)"; return -11; }
    if (adv && !h->dTraj) { h->err = "ndp_tick: a trajectory time was given but ndp_ref_set_trajectory was never called"; return -11; }
    if (h->tick_remote) { h->err = "ndp_tick: neighbours come from an exchange buffer (ndp_tick_config_remote): a tick is ndp_tick_advance_device, the exchange, ndp_tick_step_device"; return -17; }
    const int B = h->cfg.batch;
    const RingGeom rg = ring_geom(h);
    const bool est = (flags & TICK_ESTIMATE) != 0;
    // ONE launch per tick (rti_kernel<..., TICK>: list advance and estimator inside the control step's waves) for the reference
    // configuration's compile-time kernels; any other shape: tick_pre_kernel in front of the control step.  NDP_TICK_FORM=pre forces
    // the two-launch form (A/B measurements).
    static const bool force_pre = [] { const char *e = getenv("NDP_TICK_FORM"); return e && !strcmp(e, "pre"); }();
    // (a neighbour's window node N is made INSIDE the one-launch kernel's fused downwash: without the fused form -- can_fuse -- the
    // two-launch form serves)
    const bool one_launch = !force_pre && h->cfg.N == 20 && h->cfg.n_rti == 1 && h->waves == 4 && h->cfg.qp_precision == 0 &&
                            (!h->dTickIndex || can_fuse(h));
    // the list position and the cache's copies move on only when the tick's launches have been accepted (below)
    const unsigned long long n_after = h->list_n + (adv ? 1ull : 0ull);
    TickArgs ta{};
    if (one_launch && (adv || est)) {
        if (adv) {
            const size_t Bs = (size_t)B, S = (size_t)h->traj_seg;
            ta.coeff = h->dTraj; ta.tcum = ta.coeff + Bs * S * 28; ta.tseg = ta.tcum + Bs * (S + 1); ta.fpt = ta.tseg + Bs * S;
            {   // the segment cache's two copies (tick_cache_store): [B][SEGC_PER] behind final_pt, the other behind the segment hints
                double *c0 = const_cast<double *>(ta.fpt + Bs * 3), *c1 = c0 + Bs * SEGC_PER + (Bs * 4 + 7) / 8;
                ta.segc = h->segc_par ? c1 : c0;
                ta.segc_wr = h->segc_par ? c0 : c1;
            }
            ta.n_seg = h->traj_seg;
        }
        ta.t = t; ta.t_all = t_all; ta.advance = adv ? 1 : 0;
        ta.toff = h->cfg.N * h->cfg.dt; ta.mass = h->cfg.mass; ta.g = h->cfg.gravity;
        ta.j_new = h->list_n + (unsigned long long)rg.ring();
        ta.new_slot = rg.slot(ta.j_new);
        ta.rg = rg; ta.rx = h->dRingX; ta.ru = h->dRingU;
        ta.thr = thr_cfg(h); ta.st = h->dThr;
        ta.vz = vz ? vz : x_odom + 5; ta.vz_pitch = vz ? 1 : NX;
        ta.throttle = throttle ? throttle : h->dTickThrust;
        ta.est = est ? 1 : 0;
        so.tick = &ta;
    } else if (adv || est) {
        TickPre a{};
        a.cf = ref_cfg(h, h->cfg.N * h->cfg.dt);
        if (adv) {
            const size_t Bs = (size_t)B, S = (size_t)h->traj_seg;
            a.coeff = h->dTraj; a.tcum = a.coeff + Bs * S * 28; a.tseg = a.tcum + Bs * (S + 1); a.fpt = a.tseg + Bs * S;
            a.seg_hint = reinterpret_cast<int *>(const_cast<double *>(a.fpt + Bs * 3 + Bs * SEGC_PER));
        }
        a.t = t; a.t_all = t_all; a.advance = adv ? 1 : 0;
        a.j_new = h->list_n + (unsigned long long)rg.ring();
        a.rg = rg; a.rx = h->dRingX; a.ru = h->dRingU;
        a.thr = thr_cfg(h); a.st = h->dThr;
        a.vz = vz ? vz : x_odom + 5; a.vz_pitch = vz ? 1 : NX;
        a.throttle = throttle ? throttle : h->dTickThrust;
        a.est = est ? 1 : 0;
        hipLaunchKernelGGL(tick_pre_kernel, dim3((B + 63) / 64), dim3(64), 0, s, a);
        NDP_HIP(h, hipGetLastError());
    }
    const size_t slot = rg.slot(n_after);
    Neigh nb;
    if (h->dTickIndex) {
        nb.other = h->dRingX + slot * 10; nb.stride = NX; nb.index = h->dTickIndex; nb.pitch = rg.px();
        if (h->tick_gate) { nb.ego_xy = x_odom; nb.ego_pitch = NX; }
    }
    so.xr_pitch = rg.px(); so.ur_pitch = rg.pu();
    // nmpc_u_2_att_tgt is the control step's own last store (RtiIo::cmd): no third launch.  k_throttle = row 1 of the estimator's state
    // (k_throttle_init until the estimator has run)
    so.cmd = cmd; so.kthr = h->dThr + (size_t)B; so.thrust_keep = h->dTickThrust;
    rc = enqueue_step(h, x_odom, h->dRingX + slot * 10, h->dRingU + slot * 4, nullptr, nb, u0_copy ? u0_copy : h->su0, nullptr, s, &so);
    if (rc) return rc;               // (refused: the list stays where it was -- an entry the pre-launch may have written lies beyond every window)
    h->list_n = n_after;
    if (adv && so.tick) h->segc_par ^= 1;
    return 0;
}

int ndp_tick_advance_device(ndp_handle *h, const void *d_x_odom, const void *d_t, const void *d_vz, const void *d_throttle, int flags, void *stream)
{
    if (!h || !d_x_odom) return -1;
    std::lock_guard<std::mutex> lk(h->mu);
    hipStream_t s = stream ? (hipStream_t)stream : h->stream;
    NDP_HIP(h, hipSetDevice(h->cfg.device));
    int rc = ensure_tick(h);
    if (rc) return rc;
    if (!h->dRingX) { h->err = "ndp_tick_advance: no reference list (ndp_ref_list_fix_pt, or ndp_ref_set_trajectory + ndp_ref_list_reset, first)"; return -11; }
    const bool adv = d_t != nullptr, est = (flags & TICK_ESTIMATE) != 0, uni = adv && (flags & TICK_T_UNIFORM);
    if (adv && !h->dTraj) { h->err = "ndp_tick_advance: a trajectory time was given but ndp_ref_set_trajectory was never called"; return -11; }
    if (adv || est) {
        const int B = h->cfg.batch;
        const RingGeom rg = ring_geom(h);
        TickPre a{};
        a.cf = ref_cfg(h, h->cfg.N * h->cfg.dt);
        if (adv) {
            const size_t Bs = (size_t)B, S = (size_t)h->traj_seg;
            a.coeff = h->dTraj; a.tcum = a.coeff + Bs * S * 28; a.tseg = a.tcum + Bs * (S + 1); a.fpt = a.tseg + Bs * S;
            a.seg_hint = reinterpret_cast<int *>(const_cast<double *>(a.fpt + Bs * 3 + Bs * SEGC_PER));
        }
        a.t = uni ? nullptr : (const double *)d_t; a.t_all = uni ? *(const double *)d_t : 0.0; a.advance = adv ? 1 : 0;
        a.j_new = h->list_n + (unsigned long long)rg.ring();
        a.rg = rg; a.rx = h->dRingX; a.ru = h->dRingU;
        a.thr = thr_cfg(h); a.st = h->dThr;
        a.vz = d_vz ? (const double *)d_vz : (const double *)d_x_odom + 5; a.vz_pitch = d_vz ? 1 : NX;
        a.throttle = d_throttle ? (const double *)d_throttle : h->dTickThrust;
        a.est = est ? 1 : 0;
        hipLaunchKernelGGL(tick_pre_kernel, dim3((B + 63) / 64), dim3(64), 0, s, a);
        NDP_HIP(h, hipGetLastError());
        if (adv) ++h->list_n;
    }
    return note_stream(h, s);
}

int ndp_tick_window_pv_device(ndp_handle *h, void *d_pv, void *stream)
{
    if (!h || !d_pv) return -1;
    std::lock_guard<std::mutex> lk(h->mu);
    hipStream_t s = stream ? (hipStream_t)stream : h->stream;
    NDP_HIP(h, hipSetDevice(h->cfg.device));
    if (!h->dRingX) { h->err = "ndp_tick_window_pv: no reference list"; return -11; }
    const RingGeom rg = ring_geom(h);
    const size_t B = h->cfg.batch, n = B * (size_t)(h->cfg.N + 1) * 3;
    hipLaunchKernelGGL(pack_pv_list_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, h->dRingX + rg.slot(h->list_n) * 10, rg.px(),
                       h->cfg.N + 1, (double *)d_pv, B);
    NDP_HIP(h, hipGetLastError());
    return note_stream(h, s);
}

// Stage 2 and the exchange in one call, on the tick's own stream: this tick's window columns packed out of the list into the
// exchange's send buffer, then ncclAllGather into d_gathered ([world * B][N+1][6]) -- both on `stream`, behind the list advance and in
// front of ndp_tick_step_device by stream order alone (no event operation, no second stream: the tick's chain is serial anyway).
int ndp_xchg_tick_windows(ndp_xchg *x, ndp_handle *h, void *d_gathered, void *stream)
{
    if (!x || !h || !d_gathered) return -1;
    std::lock_guard<std::mutex> lk(h->mu);
    if (x->device != h->cfg.device) { h->err = "ndp_xchg_tick_windows: the exchange and the handle live on different devices"; return -1; }
    hipStream_t s = stream ? (hipStream_t)stream : h->stream;
    NDP_HIP(h, hipSetDevice(h->cfg.device));
    if (!h->dRingX) { h->err = "ndp_xchg_tick_windows: no reference list"; return -11; }
    const RingGeom rg = ring_geom(h);
    const size_t B = h->cfg.batch, rows = B * (size_t)(h->cfg.N + 1), n = rows * 3;
    if (x->send_doubles < rows * 6) {
        if (x->send) { (void)hipStreamSynchronize(x->cs); (void)hipStreamSynchronize(s); (void)hipFree(x->send); x->send = nullptr; }
        NDP_HIP(h, hipMalloc((void **)&x->send, rows * 6 * sizeof(double)));
        x->send_doubles = rows * 6;
    }
    hipLaunchKernelGGL(pack_pv_list_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, h->dRingX + rg.slot(h->list_n) * 10, rg.px(),
                       h->cfg.N + 1, x->send, B);
    NDP_HIP(h, hipGetLastError());
    const int r = g_rccl.allgather(x->send, d_gathered, rows * 6, /* ncclFloat64 */ 8, x->comm, s);
    if (r != 0) { x->err = g_rccl.errstr ? g_rccl.errstr(r) : "ncclAllGather failed"; h->err = "ndp_xchg_tick_windows: " + x->err; return -22; }
    return note_stream(h, s);
}

// stage 3 on `s` (h->mu held): the control step of the window at list position `pos`, neighbour rows out of `windows`
static int tick_step_enqueue(ndp_handle *h, hipStream_t s, const double *x_odom, double *cmd, double *u0, const double *windows, unsigned long long pos)
{
    if (!h->tick_remote) { h->err = "ndp_tick_step: ndp_tick_config_remote first (neighbours in the same handle: ndp_tick_device)"; return -17; }
    if (!h->dRingX) { h->err = "ndp_tick_step: no reference list"; return -11; }
    const RingGeom rg = ring_geom(h);
    const size_t slot = rg.slot(pos), B = h->cfg.batch;
    Neigh nb;
    nb.other = windows; nb.stride = h->tick_remote_stride; nb.index = h->dTickIndex;
    if (h->tick_gate) { nb.ego_xy = x_odom; nb.ego_pitch = NX; }
    StepOut so;
    so.xr_pitch = rg.px(); so.ur_pitch = rg.pu();
    so.cmd = cmd; so.kthr = h->dThr + B; so.thrust_keep = h->dTickThrust;
    return enqueue_step(h, x_odom, h->dRingX + slot * 10, h->dRingU + slot * 4, nullptr, nb, u0 ? u0 : h->su0, nullptr, s, &so);
}

int ndp_tick_step_device(ndp_handle *h, const void *d_x_odom, void *d_cmd, void *d_u0, void *stream)
{
    if (!h || !d_x_odom || !d_cmd) return -1;
    std::lock_guard<std::mutex> lk(h->mu);
    hipStream_t s = stream ? (hipStream_t)stream : h->stream;
    NDP_HIP(h, hipSetDevice(h->cfg.device));
    int rc = tick_step_enqueue(h, s, (const double *)d_x_odom, (double *)d_cmd, (double *)d_u0, h->tick_remote, h->list_n);
    return rc ? rc : note_stream(h, s);
}

// ---- the remote tick with the exchange ONE CONTROL PERIOD AHEAD.  A vehicle's window is a function of time alone (the list advance
// reads the trajectory, not the odometry): the list advance of tick i+1, its window columns and their all-gather run on the exchange's
// own stream BESIDE the control step of tick i, into the other one of two gather buffers.  Per control period
//     ndp_xchg_tick_step(tick i: estimator, wait for gather i on the device, control step)   then   ndp_xchg_tick_begin(tick i+1)
// (one begin in front of the first step).  What orders what:
//   gather i+1 writes the buffer step i-1 read      -> the exchange stream waits for that step's completion event (ndp_track_steps: it
//                                                      rides on the step's dispatch packet; untracked: for everything its stream holds)
//   advance i+1 writes list entries                  -> of another phase row than window i's (RingGeom: entries per node spacing >= 2,
//                                                      refused otherwise), so it may run beside step i
//   step i reads window i and gather buffer i        -> its stream waits for its begin's event, recorded behind advance i, pack, gather
//   the estimator reads the thrust step i-1 commanded -> same stream as the steps, in front of step i
// the exchange stream's launches of one begin; returns 0 or the error code (the caller's thread or the exchange's own)
static int xchg_job_run(ndp_xchg *x, const ndp_xchg::Job &j)
{
    if (j.wait_ev && hipStreamWaitEvent(x->cs, j.wait_ev, 0) != hipSuccess) return -3;
    if (j.adv) hipLaunchKernelGGL(tick_pre_kernel, dim3((unsigned)((j.B + 63) / 64)), dim3(64), 0, x->cs, j.a);
    else {
        const size_t n = j.rows * 3;
        hipLaunchKernelGGL(pack_pv_list_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, x->cs, j.pack_base, j.pack_pitch, j.np1, x->send, j.B);
    }
    if (hipGetLastError() != hipSuccess) return -3;
    const int r = g_rccl.allgather(x->send, j.gathered, j.rows * 6, /* ncclFloat64 */ 8, x->comm, x->cs);
    if (r != 0) { x->err = g_rccl.errstr ? g_rccl.errstr(r) : "ncclAllGather failed"; return -22; }
    return hipEventRecord(x->evGather[j.p], x->cs) == hipSuccess ? 0 : -3;
}

static void xchg_worker(ndp_xchg *x)
{
    (void)hipSetDevice(x->device);
    int idle = 0;
    for (;;) {
        const unsigned want = x->done.load(std::memory_order_relaxed) + 1;
        if ((int)(x->posted.load(std::memory_order_acquire) - want) >= 0) {
            const int rc = xchg_job_run(x, x->job[want % 3]);
            if (rc) x->async_rc.store(rc, std::memory_order_relaxed);
            x->done.store(want, std::memory_order_release);
            idle = 0;
        } else if (x->stop.load(std::memory_order_acquire)) break;
        else if (++idle < 200000) __builtin_ia32_pause();                     // (~ a millisecond of spinning behind the last job, then naps)
        else std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
}

// on: begins are described by the caller and LAUNCHED by a thread of the exchange's own (see ndp_xchg::Job); off: launched by the caller
int ndp_xchg_tick_async(ndp_xchg *x, int on)
{
    if (!x) return -1;
    if (x->ahead != 0) { x->err = "ndp_xchg_tick_async: gathers are ahead of the control steps (step on them first)"; return -14; }
    if (on && !x->worker.joinable()) {
        x->stop.store(false);
        try { x->worker = std::thread(xchg_worker, x); } catch (...) { x->err = "ndp_xchg_tick_async: no thread"; return -4; }
        x->async = true;
    } else if (!on) xchg_worker_stop(x);
    return 0;
}

int ndp_xchg_tick_begin(ndp_xchg *x, ndp_handle *h, const void *d_t, int flags, void *d_gathered)
{
    if (!x || !h || !d_gathered) return -1;
    std::lock_guard<std::mutex> lk(h->mu);
    if (x->device != h->cfg.device) { h->err = "ndp_xchg_tick_begin: the exchange and the handle live on different devices"; return -1; }
    NDP_HIP(h, hipSetDevice(h->cfg.device));
    int rc = ensure_tick(h);
    if (rc) return rc;
    if (!h->dRingX) { h->err = "ndp_xchg_tick_begin: no reference list"; return -11; }
    if (d_t && !h->dTraj) { h->err = "ndp_xchg_tick_begin: a trajectory time was given but ndp_ref_set_trajectory was never called"; return -11; }
    if (x->ahead >= 2) { h->err = "ndp_xchg_tick_begin: two gathers are already ahead of the control steps (ndp_xchg_tick_step first)"; return -14; }
    if ((rc = x->async_rc.load(std::memory_order_relaxed))) { h->err = "ndp_xchg_tick_begin: an earlier begin failed on the exchange's thread: " + x->err; return rc; }
    const RingGeom rg = ring_geom(h);
    if (d_t && rg.step < 2) { h->err = "ndp_xchg_tick_begin: the list's entries are one node spacing apart -- the advance would overwrite the window a control step may be reading (use the serial form: ndp_tick_advance_device, ndp_xchg_tick_windows, ndp_tick_step_device)"; return -17; }
    const size_t B = h->cfg.batch, rows = B * (size_t)(h->cfg.N + 1);
    if (x->send_doubles < rows * 6) {
        while (x->done.load(std::memory_order_acquire) != x->posted.load(std::memory_order_relaxed)) __builtin_ia32_pause();
        if (x->send) { (void)hipStreamSynchronize(x->cs); (void)hipFree(x->send); x->send = nullptr; }
        NDP_HIP(h, hipMalloc((void **)&x->send, rows * 6 * sizeof(double)));
        x->send_doubles = rows * 6;
    }
    const unsigned n_job = x->posted.load(std::memory_order_relaxed) + 1;
    const int p = (int)(n_job % 3u);
    ndp_xchg::Job &j = x->job[p];                  // (free: at most two are ahead, and a step waits for its begin's launches)
    j = ndp_xchg::Job{};
    j.p = p; j.B = B; j.rows = rows; j.np1 = h->cfg.N + 1; j.gathered = d_gathered;
    // the gather overwrites a buffer: behind the control step that read it last
    const ndp_xchg::Reader *rd = nullptr;
    for (const ndp_xchg::Reader &r : x->readers) if (r.ptr == d_gathered) rd = &r;
    for (int q = 0; q < 3; ++q)                    // (a begin that is still ahead of its step names the same buffer: the caller cycles too few)
        if (x->buf[q] == d_gathered && (int)(x->job_n[q] - x->steps) > 0) { h->err = "ndp_xchg_tick_begin: this gather buffer holds a tick that has not been stepped on yet"; return -14; }
    if (rd) {
        const bool precise = h->track_steps && rd->seq && h->step_seq - rd->seq < 4u;
        if (precise) {
            j.wait_ev = h->stepDone[rd->seq & 3];
        } else {
            NDP_HIP(h, hipEventRecord(x->evReady, rd->stream));
            j.wait_ev = x->evReady;
        }
    } else if (n_job == 1) {            // the first gather: behind whatever made the list (ndp_ref_list_reset / ndp_tick_reset on the handle's stream)
        NDP_HIP(h, hipEventRecord(x->evReady, h->stream));
        j.wait_ev = x->evReady;
    }
    if (d_t) {
        const bool uni = (flags & TICK_T_UNIFORM) != 0;
        TickPre &a = j.a;
        a.cf = ref_cfg(h, h->cfg.N * h->cfg.dt);
        const size_t Bs = B, S = (size_t)h->traj_seg;
        a.coeff = h->dTraj; a.tcum = a.coeff + Bs * S * 28; a.tseg = a.tcum + Bs * (S + 1); a.fpt = a.tseg + Bs * S;
        a.seg_hint = reinterpret_cast<int *>(const_cast<double *>(a.fpt + Bs * 3 + Bs * SEGC_PER));
        a.t = uni ? nullptr : (const double *)d_t; a.t_all = uni ? *(const double *)d_t : 0.0; a.advance = 1;
        a.j_new = h->list_n + (unsigned long long)rg.ring();
        a.rg = rg; a.rx = h->dRingX; a.ru = h->dRingU;
        a.thr = thr_cfg(h); a.st = h->dThr;
        a.vz = h->dTickThrust; a.vz_pitch = 1; a.throttle = h->dTickThrust; a.est = 0;      // (no estimator here: it belongs to the step's side)
        a.pv = x->send; a.pv_slot = rg.slot(h->list_n + 1);                                 // ... and the advanced window's columns in the same launch
        j.adv = true;
        ++h->list_n;
    } else {
        j.pack_base = h->dRingX + rg.slot(h->list_n) * 10; j.pack_pitch = rg.px();
    }
    x->job_n[p] = n_job;
    x->buf[p] = d_gathered;
    if (x->async) x->posted.store(n_job, std::memory_order_release);          // the exchange's thread takes it from here
    else {
        rc = xchg_job_run(x, j);
        x->posted.store(n_job, std::memory_order_relaxed);
        x->done.store(n_job, std::memory_order_relaxed);
        if (rc) { h->err = "ndp_xchg_tick_begin: " + (rc == -22 ? x->err : std::string("a HIP call on the exchange's stream failed")); return rc; }
    }
    x->win_n[p] = h->list_n;
    ++x->ahead;
    return 0;
}

int ndp_xchg_tick_step(ndp_xchg *x, ndp_handle *h, const void *d_x_odom, const void *d_vz, const void *d_throttle, int flags,
                       void *d_cmd, void *d_u0, const void *d_gathered, void *stream)
{
    if (!x || !h || !d_x_odom || !d_cmd || !d_gathered) return -1;
    std::lock_guard<std::mutex> lk(h->mu);
    hipStream_t s = stream ? (hipStream_t)stream : h->stream;
    NDP_HIP(h, hipSetDevice(h->cfg.device));
    if (x->ahead < 1) { h->err = "ndp_xchg_tick_step: no gather was begun for this tick (ndp_xchg_tick_begin first)"; return -14; }
    int rc = ensure_tick(h);
    if (rc) return rc;
    const size_t B = h->cfg.batch;
    const unsigned k = x->steps + 1;               // this step consumes begin k
    const int p = (int)(k % 3u);
    if (x->buf[p] != d_gathered || x->job_n[p] != k) { h->err = "ndp_xchg_tick_step: this tick's gather was begun into another buffer"; return -14; }
    if (flags & TICK_ESTIMATE) {
        TickPre a{};
        a.cf = ref_cfg(h, h->cfg.N * h->cfg.dt);
        a.advance = 0;
        a.rg = ring_geom(h); a.rx = h->dRingX; a.ru = h->dRingU;
        a.thr = thr_cfg(h); a.st = h->dThr;
        a.vz = d_vz ? (const double *)d_vz : (const double *)d_x_odom + 5; a.vz_pitch = d_vz ? 1 : NX;
        a.throttle = d_throttle ? (const double *)d_throttle : h->dTickThrust;
        a.est = 1;
        hipLaunchKernelGGL(tick_pre_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, s, a);
        NDP_HIP(h, hipGetLastError());
    }
    // (asynchronous begins: the event must have been RECORDED by the exchange's thread before this stream is told to wait for it)
    while ((int)(x->done.load(std::memory_order_acquire) - k) < 0) __builtin_ia32_pause();
    if ((rc = x->async_rc.load(std::memory_order_relaxed))) { h->err = "ndp_xchg_tick_step: this tick's begin failed on the exchange's thread: " + x->err; return rc; }
    NDP_HIP(h, hipStreamWaitEvent(s, x->evGather[p], 0));
    rc = tick_step_enqueue(h, s, (const double *)d_x_odom, (double *)d_cmd, (double *)d_u0, (const double *)d_gathered, x->win_n[p]);
    if (rc) return rc;
    ndp_xchg::Reader *slot = nullptr;              // this buffer's entry, else the one not touched for longest
    for (ndp_xchg::Reader &r : x->readers) if (r.ptr == d_gathered) slot = &r;
    if (!slot) { slot = &x->readers[0]; for (ndp_xchg::Reader &r : x->readers) if (r.age < slot->age) slot = &r; }
    slot->ptr = d_gathered; slot->stream = s; slot->age = k;
    slot->seq = (h->track_steps && h->last_step_tracked) ? h->step_seq : 0u;
    x->steps = k;
    --x->ahead;
    if (slot->seq) { h->track_pending = true; return 0; }     // (the getters wait for the step's own completion event: no second one)
    return note_stream(h, s);
}

int ndp_tick_device(ndp_handle *h, const void *d_x_odom, const void *d_t, const void *d_vz, const void *d_throttle, int flags,
                    void *d_cmd, void *d_u0, void *stream)
{
    if (!h || !d_x_odom || !d_cmd) return -1;
    std::lock_guard<std::mutex> lk(h->mu);
    hipStream_t s = stream ? (hipStream_t)stream : h->stream;
    NDP_HIP(h, hipSetDevice(h->cfg.device));
    const bool uni = d_t && (flags & TICK_T_UNIFORM);       // (then d_t is HOST memory: one double, read here)
    int rc = tick_enqueue(h, s, (const double *)d_x_odom, d_t != nullptr, uni ? nullptr : (const double *)d_t, uni ? *(const double *)d_t : 0.0,
                          (const double *)d_vz, (const double *)d_throttle, flags, (double *)d_cmd, (double *)d_u0, StepOut());
    return rc ? rc : note_stream(h, s);
}

// host arrays: the inputs of a tick are packed into a slot's page-locked input mirror -- x_odom | t | vz | throttle, 80 + 24 bytes per
// vehicle at most -- which the tick's kernels read over PCIe themselves; cmd | status | iterations (| u0) are written into the
// slot's page-locked output mirror by the kernels (the same two slots, the same zero-copy scheme as ndp_step_begin / _end)
static int tick_begin_locked(ndp_handle *h, const double *x_odom, const double *t, const double *vz, const double *throttle, int flags)
{
    const size_t B = h->cfg.batch;
    NDP_HIP(h, hipSetDevice(h->cfg.device));
    int rc = ensure_slots(h);
    if (rc) return rc;
    if (h->slots_busy == 2) { h->err = "ndp_tick_begin: two ticks are already in flight (call ndp_tick_end first)"; return -14; }
    if (h->ev_pending && (rc = wait_all(h))) return rc;
    ndp_handle::HostSlot &sl = h->slot[h->slot_head];
    const auto tp0 = std::chrono::steady_clock::now();
    unsigned char *ib = sl.hIn;
    const size_t o_t = up256(B * NX * 8), o_vz = o_t + up256(B * 8), o_th = o_vz + up256(B * 8);    // (<= in_bytes: the mirror holds a whole step's inputs)
    memcpy(ib, x_odom, B * NX * 8);
    const bool uni = t && (flags & TICK_T_UNIFORM);         // one time for every vehicle: it travels in the kernel arguments
    if (t && !uni) memcpy(ib + o_t, t, B * 8);
    if (vz) memcpy(ib + o_vz, vz, B * 8);
    if (throttle) memcpy(ib + o_th, throttle, B * 8);
    const auto tp1 = std::chrono::steady_clock::now();
    StepOut so;
    so.status = (int *)(sl.hOut + h->off_st); so.iters = (int *)(sl.hOut + h->off_it);
    so.done = sl.evOut;
    const bool want_u0 = (flags & TICK_WANT_U0) != 0;
    rc = tick_enqueue(h, h->stream, (const double *)ib, t != nullptr, t && !uni ? (const double *)(ib + o_t) : nullptr, uni ? t[0] : 0.0,
                      vz ? (const double *)(ib + o_vz) : nullptr,
                      throttle ? (const double *)(ib + o_th) : nullptr, flags, (double *)(sl.hOut + h->off_u0),
                      want_u0 ? (double *)(sl.hOut + h->out_bytes) : nullptr, so);
    if (rc) return rc;
    sl.busy = true; sl.want_iter = false; sl.dump = nullptr;
    h->tslot[h->slot_head].busy = true; h->tslot[h->slot_head].want_u0 = want_u0;
    h->host_us[0] = std::chrono::duration<double, std::micro>(tp1 - tp0).count();
    h->host_us[1] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tp1).count();
    h->slot_head ^= 1;
    ++h->slots_busy;
    return 0;
}

static int tick_end_locked(ndp_handle *h, double *cmd, double *u0, int32_t *status_out, int32_t *iters_out)
{
    const size_t B = h->cfg.batch;
    if (h->slots_busy == 0 || !h->tslot[h->slot_tail].busy) { h->err = "ndp_tick_end: no tick in flight (ndp_tick_begin first)"; return -14; }
    ndp_handle::HostSlot &sl = h->slot[h->slot_tail];
    if (u0 && !h->tslot[h->slot_tail].want_u0) { h->err = "ndp_tick_end: u0 was not requested at ndp_tick_begin (flags bit 1)"; return -15; }
    NDP_HIP(h, hipSetDevice(h->cfg.device));
    const auto tw0 = std::chrono::steady_clock::now();
    hipError_t e = hipErrorNotReady;
    for (int spin = 0; spin < 4000 && e == hipErrorNotReady; ++spin) e = hipEventQuery(sl.evOut);
    if (e == hipErrorNotReady) e = hipEventSynchronize(sl.evOut);
    sl.busy = false;
    h->tslot[h->slot_tail].busy = false;
    h->slot_tail ^= 1;
    --h->slots_busy;
    NDP_HIP(h, e);
    const auto tw1 = std::chrono::steady_clock::now();
    const unsigned char *ho = sl.hOut;
    memcpy(cmd, ho + h->off_u0, B * NU * 8);
    if (u0) memcpy(u0, ho + h->out_bytes, B * NU * 8);
    const int32_t *st = (const int32_t *)(ho + h->off_st);
    if (status_out) memcpy(status_out, st, B * 4);
    if (iters_out) copy_ipm_iters(iters_out, reinterpret_cast<const int32_t *>(ho + h->off_it), B);
    int w = 0;
    for (size_t i = 0; i < B; ++i) w = st[i] > w ? st[i] : w;
    h->host_us[2] = std::chrono::duration<double, std::micro>(tw1 - tw0).count();
    h->host_us[3] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tw1).count();
    return w;
}

int ndp_tick_begin(ndp_handle *h, const double *x_odom, const double *t, const double *vz, const double *throttle, int flags)
{
    if (!h || !x_odom) return -1;
    std::lock_guard<std::mutex> lk(h->mu);
    return tick_begin_locked(h, x_odom, t, vz, throttle, flags);
}

int ndp_tick_end(ndp_handle *h, double *cmd, double *u0, int32_t *status_out, int32_t *ipm_iters_out)
{
    if (!h || !cmd) return -1;
    std::lock_guard<std::mutex> lk(h->mu);
    return tick_end_locked(h, cmd, u0, status_out, ipm_iters_out);
}

int ndp_tick(ndp_handle *h, const double *x_odom, const double *t, const double *vz, const double *throttle, int flags,
             double *cmd, double *u0, int32_t *status_out, int32_t *ipm_iters_out)
{
    if (!h || !x_odom || !cmd) return -1;
    std::lock_guard<std::mutex> lk(h->mu);
    if (h->slots_busy) { h->err = "ndp_tick: steps / ticks begun earlier are still in flight (end them first)"; return -14; }
    int rc = tick_begin_locked(h, x_odom, t, vz, throttle, flags | (u0 ? TICK_WANT_U0 : 0));
    if (rc) return rc;
    return tick_end_locked(h, cmd, u0, status_out, ipm_iters_out);
}

// ---- f4: plant step
static int launch_plant(ndp_handle *h, double *d_x, const double *d_u, const double *d_f, double dt, int substeps, hipStream_t s)
{
    hipLaunchKernelGGL(plant_kernel, dim3((h->cfg.batch + 255) / 256), dim3(256), 0, s, d_x, d_u, d_f, dt / substeps, substeps,
                       1.0 / h->cfg.mass, h->cfg.gravity, h->cfg.batch);
    NDP_HIP(h, hipGetLastError());
    return 0;
}

int ndp_plant_step_device(ndp_handle *h, void *d_x, const void *d_u, const void *d_f, double dt, int substeps, void *stream)
{
    if (!h || !d_x || !d_u || substeps < 1) return -1;
    std::lock_guard<std::mutex> lk(h->mu);
    hipStream_t s = stream ? (hipStream_t)stream : h->stream;
    NDP_HIP(h, hipSetDevice(h->cfg.device));
    int rc = launch_plant(h, (double *)d_x, (const double *)d_u, (const double *)d_f, dt, substeps, s);
    return rc ? rc : note_stream(h, s);
}

int ndp_plant_step(ndp_handle *h, double *x, const double *u, const double *f, double dt, int substeps)
{
    if (!h || !x || !u || substeps < 1) return -1;
    const size_t B = h->cfg.batch;
    std::lock_guard<std::mutex> lk(h->mu);
    NDP_HIP(h, hipSetDevice(h->cfg.device));
    NDP_HIP(h, hipMemcpyAsync(h->sx0, x, B * 80, hipMemcpyHostToDevice, h->stream));
    NDP_HIP(h, hipMemcpyAsync(h->su0, u, B * 32, hipMemcpyHostToDevice, h->stream));
    if (f) NDP_HIP(h, hipMemcpyAsync(h->sThr, f, B * 24, hipMemcpyHostToDevice, h->stream));
    int rc = launch_plant(h, h->sx0, h->su0, f ? h->sThr : nullptr, dt, substeps, h->stream);
    if (rc) return rc;
    NDP_HIP(h, hipMemcpyAsync(x, h->sx0, B * 80, hipMemcpyDeviceToHost, h->stream));
    NDP_HIP(h, hipStreamSynchronize(h->stream));
    return 0;
}

// ---- f4: closed-loop rollout, everything enqueued back to back on one stream, nothing returns to the host in between
int ndp_rollout_device(ndp_handle *h, int ticks, double t0, double dt_tick, int substeps, void *d_x, void *d_log, void *stream)
{
    if (!h || ticks < 1 || substeps < 1 || !d_x) return -1;
    std::lock_guard<std::mutex> lk(h->mu);
    hipStream_t s = stream ? (hipStream_t)stream : h->stream;
    NDP_HIP(h, hipSetDevice(h->cfg.device));
    if (h->cfg.use_fd) { h->err = "ndp_rollout_device: the rollout drives the NMPC model (use_fd = 0)"; return -8; }
    const size_t B = h->cfg.batch;
    double *x = (double *)d_x, *log = (double *)d_log;
    int rc = launch_ref_window(h, nullptr, t0, h->sxr, h->sur, s);     // reset(xr, ur) at the first tick's reference
    if (rc) return rc;
    NDP_HIP(h, hipMemcpyAsync(h->dX, h->sxr, nxs(h) * 8, hipMemcpyDefault, s));
    NDP_HIP(h, hipMemcpyAsync(h->dU, h->sur, nus(h) * 8, hipMemcpyDefault, s));
    for (int k = 0; k < ticks; ++k) {
        if (k > 0 && (rc = launch_ref_window(h, nullptr, t0 + k * dt_tick, h->sxr, h->sur, s))) return rc;
        if ((rc = launch_rti(h, x, h->sxr, h->sur, nullptr, h->su0, nullptr, s))) return rc;
        if ((rc = launch_plant(h, x, h->su0, nullptr, dt_tick, substeps, s))) return rc;
        if (log) NDP_HIP(h, hipMemcpyAsync(log + (size_t)k * B * NX, x, B * NX * 8, hipMemcpyDeviceToDevice, s));
    }
    return note_stream(h, s);
}

// test/profiling hook: every instance writes its phase stamps (shader clock) to [B][NDP_NSTAMP] doubles
int ndp_debug_stamps(ndp_handle *h, int enable, double *out)
{
    if (!h) return -1;
    std::lock_guard<std::mutex> lk(h->mu);
    NDP_HIP(h, hipSetDevice(h->cfg.device));
    NDP_HIP(h, hipStreamSynchronize(h->stream));
    NDP_HIP(h, hipDeviceSynchronize());
    const size_t bytes = (size_t)h->cfg.batch * NDP_NSTAMP * 8;
    if (out && h->dStamps) NDP_HIP(h, hipMemcpy(out, h->dStamps, bytes, hipMemcpyDeviceToHost));
    if (enable && !h->dStamps) {
        NDP_HIP(h, hipMalloc((void **)&h->dStamps, bytes));
        NDP_HIP(h, hipMemset(h->dStamps, 0, bytes));
    } else if (!enable && h->dStamps) {
        (void)hipFree(h->dStamps);
        h->dStamps = nullptr;
    }
    return 0;
}

void *ndp_device_iterate_x(ndp_handle *h) { return h ? h->dX : nullptr; }
void *ndp_device_iterate_u(ndp_handle *h) { return h ? h->dU : nullptr; }
void *ndp_device_force(ndp_handle *h) { return h ? h->dForce : nullptr; }
int ndp_work_queue_enabled(ndp_handle *h) { return h ? (int)h->use_queue : -1; }
int ndp_refine_active(ndp_handle *h) { return h ? (int)(h->cfg.ipm_refine > 0 && slots_for(h->cfg.N) <= 3 && h->cfg.qp_precision == 0) : -1; }

int ndp_synchronize(ndp_handle *h)
{
    if (!h) return -1;
    std::lock_guard<std::mutex> lk(h->mu);
    NDP_HIP(h, hipSetDevice(h->cfg.device));
    return wait_all(h);
}

int ndp_timing_enable(ndp_handle *h, int on)
{
    if (!h) return -1;
    std::lock_guard<std::mutex> lk(h->mu);
    for (auto &e : h->events) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    h->events.clear();
    h->timing = on > 0 ? on : 0;
    h->launch_no[0] = h->launch_no[1] = 0;
    return 0;
}

int ndp_timing_read(ndp_handle *h, const char *name, double *total_ms, int64_t *launches)
{
    if (!h || !name) return -1;
    std::lock_guard<std::mutex> lk(h->mu);
    const int kind = (name[0] == 'm') ? 1 : 0;
    double tot = 0.0;
    int64_t n = 0;
    for (auto &e : h->events) {
        if (e.kind != kind) continue;
        NDP_HIP(h, hipEventSynchronize(e.b));
        float ms = 0.0f;
        NDP_HIP(h, hipEventElapsedTime(&ms, e.a, e.b));
        tot += ms;
        ++n;
    }
    if (total_ms) *total_ms = tot;
    if (launches) *launches = n;
    return n ? 0 : -10;
}

}  // extern "C"
