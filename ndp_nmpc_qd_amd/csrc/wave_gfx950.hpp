// wave_gfx950.hpp -- the wave backend of rti_wave.hpp on CDNA4 (gfx950): one HIP thread = one lane.
// Per-lane values are plain registers; cross-lane traffic is v_readlane / ds_bpermute; the matrix
// instruction is v_mfma_f64_16x16x4_f64; the LDS slice is addressed through an address_space(3)
// pointer so every access is a ds_read/ds_write (no flat instructions).
#pragma once
#include <hip/hip_runtime.h>

#define NDP_D __device__ __forceinline__
#define NDP_HD __host__ __device__ inline

namespace ndp {

struct RtiParams;
struct LdsMap;

struct WaveGfx950 {
    using vd = double;
    using vi = int;
    using vb = bool;
    struct vd4 { double r[4]; };
    // matrix values of the Riccati sweeps (RtiWave::md): this backend's instruction is v_mfma_f64_16x16x4_f64
    using md = double;
    using md4 = vd4;
    static constexpr bool packed_k = false, delta_ok = true, has_mma4 = true;
    static NDP_D md to_m(vd a) { return a; }
    static NDP_D vd to_d(md a) { return a; }
    static NDP_D md4 mzero4() { return zero4(); }
    static NDP_D md mavg(md a, md b) { return (a + b) * 0.5; }
    static NDP_D md msel(vb p, md a, md b) { return p ? a : b; }
    static NDP_D vi lcol(vi lane) { return lane & 15; }        // the matrix column a lane holds (RtiWave::lane_preds)
    typedef __attribute__((address_space(3))) double *lds_ptr;
    typedef double d4_t __attribute__((ext_vector_type(4)));

    // BASELINE config 5's study: one QP in condensed form on the fp32 / bf16 matrix instructions (cond_qp.hpp)
    template <int MODE, class ME, class CE>
    static __device__ bool cond_solve(const RtiParams &P, const LdsMap &m, lds_ptr lds, int N, ME m_entry, CE c_entry);

    static NDP_D vi lane() { return (int)(threadIdx.x & 63u); }
    // the lane id as a value the compiler cannot trace back to the thread id: index arithmetic built on it stays WHERE it is written
    // instead of being hoisted to the kernel's start and parked in registers / scratch for the whole program (the constraint
    // slots of the interior-point loop, 50 integers per lane: build_slots)
    static NDP_D vi lane_here() { int l = (int)(threadIdx.x & 63u); asm volatile("" : "+v"(l)); return l; }
    static NDP_D vd sel(vb p, vd a, vd b) { return p ? a : b; }
    static NDP_D vi sel(vb p, vi a, vi b) { return p ? a : b; }
    static NDP_D vd vmin(vd a, vd b) { return fmin(a, b); }
    static NDP_D vd vmax(vd a, vd b) { return fmax(a, b); }
    static NDP_D vd vabs(vd a) { return fabs(a); }
    static NDP_D vi div3(vi a) { return a / 3; }
    static NDP_D vi div6(vi a) { return a / 6; }

    // LDS (one wave owns its slice; DS operations of one wave execute in order)
    static NDP_D vd ld(lds_ptr lds, vi off) { return lds[off]; }
    // two adjacent doubles at an even offset: one 16-byte DS read
    typedef double d2_t __attribute__((ext_vector_type(2)));
    static NDP_D void ld2(lds_ptr lds, vi off, vd &a, vd &b)
    {
        const d2_t v = *(const __attribute__((address_space(3))) d2_t *)__builtin_assume_aligned(lds + off, 16);
        a = v[0]; b = v[1];
    }
    static NDP_D vd ldp(lds_ptr lds, vi off, vb p) { return p ? lds[off] : 0.0; }
    // predicated read WITHOUT control flow: lanes outside the predicate read the slice's first word and get 0.  (ldp is an exec-masked
    // block, into which the compiler sinks the value's first use together with the wait for the load: a chain of them is a chain of
    // LDS round trips.)
    static NDP_D vd ldz(lds_ptr lds, vi off, vb p) { const vd v = lds[p ? off : 0]; return p ? v : 0.0; }
    static NDP_D void stp(lds_ptr lds, vi off, vd v, vb p) { if (p) lds[off] = v; }
    static NDP_D void st(lds_ptr lds, vi off, vd v) { lds[off] = v; }
    static NDP_D void pin() { __builtin_amdgcn_sched_barrier(0); }   // instruction-scheduling fence only
    static NDP_D void keep(vd a) { asm volatile("" ::"v"(a)); }         // extends a's live range to this point (no code)
    static NDP_D void sync()
    {   // lanes exchange data through LDS: forbid the compiler to move LDS accesses across this point
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }

    // global memory
    static NDP_D vd gld(const double *g, vi off, vb p) { return p ? g[off] : 0.0; }
    static NDP_D vd gldf(const float *g, vi off, vb p) { return p ? (double)g[off] : 0.0; }
    static NDP_D vd gldu(const double *g, vi off) { return g[off]; }
    static NDP_D vd gldfu(const float *g, vi off) { return (double)g[off]; }
    static NDP_D vi gldi(const int *g, vi off) { return g[off]; }
    // The step's parameter block for code that runs RARELY: the same bytes through a pointer into the kernel's argument segment that has
    // passed an empty asm, so that the fetches stay where they are written (constant-address-space scalar loads, scalar-cache hits)
    // instead of being hoisted to wherever the compiler first sees a path to them.  The block is the FIRST member of the kernel's one
    // argument (ndp_hip.hip: KernArgs::P, asserted there).  (Taking the parameter's address instead would make the compiler copy the
    // whole argument to scratch memory.)
    template <class T>
    static NDP_D const __attribute__((address_space(4))) T *late_params(const T &)
    {
        typedef const __attribute__((address_space(4))) T *kptr;
        kptr base = (kptr)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(base));
        return base;
    }
    static NDP_D vi gld_i8(const signed char *g, vi off) { return (int)g[off]; }
    static NDP_D vi d2i(vd a) { return (int)a; }
    static NDP_D vd i2d(vi a) { return (double)a; }
    static NDP_D void gst_i8(signed char *g, vi off, vi v, vb p) { if (p) g[off] = (signed char)v; }
    static NDP_D vi imin(vi a, vi b) { return a < b ? a : b; }
    static NDP_D void gst(double *g, vi off, vd v, vb p) { if (p) g[off] = v; }
    // element i of the concatenation [a (n elements) | b]: one store through a per-lane pointer instead of two predicated ones
    static NDP_D void gst2(double *a, double *b, vi i, int n, vd v) { (i < n ? a + i : b + (i - n))[0] = v; }
    static NDP_D void gsti(int *g, int v) { if (g && lane() == 0) *g = v; }
    // ndp_tick: element idx (0..3) of the actuator command made of u_0's element v (RtiIo::cmd): body rates pass through, thrust =
    // c * mass / k_throttle, 0 if k == 0 (nmpc_u_2_att_tgt, nmpc_node.py:281) -- the product rounded by itself, then an IEEE divide,
    // as Python evaluates it (and as actuator_kernel does)
    static NDP_D void cmd_store(double *cmd, double *keep, double k, double mass, vi idx, vd v, vb p)
    {
        if (p) {
            if (idx == 3) {
                double cm;
                {
#pragma clang fp contract(off)
                    cm = v * mass;
                }
                v = k != 0.0 ? cm / k : 0.0;
                *keep = v;
            }
            cmd[idx] = v;
        }
    }
    // late-force protocol (rti_wave.hpp: RtiIo::f_late): wait until both words have reached `want` (agent-scope acquire), at most
    // timeout_us.  The words differ from instance to instance (per-tile epochs of the downwash launch): 1024 waves reading ONE
    // word at agent scope serialise at the memory side (the XCDs' L2s are not coherent with each other: such loads bypass them) --
    // measured 7 us per wave.
    static NDP_D bool wait_ge(const unsigned long long *flag, const unsigned long long *flag2, unsigned long long want, unsigned timeout_us)
    {
        auto both = [&](int order) {
            const unsigned long long a = __hip_atomic_load(flag, order, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long b = __hip_atomic_load(flag2, order, __HIP_MEMORY_SCOPE_AGENT);
            return a >= want && b >= want;
        };
        if (both(__ATOMIC_ACQUIRE)) return true;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();        // 100 MHz
        while (!both(__ATOMIC_RELAXED)) {
            __builtin_amdgcn_s_sleep(8);                                        // ~0.25 us between polls
            if (__builtin_amdgcn_s_memrealtime() - t0 > 100ull * timeout_us) return false;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        return true;
    }
    // a float written by another launch that is still running elsewhere on the device: read past this XCD's L2 (agent scope)
    static NDP_D vd gldf_fresh(const float *g, vi off)
    {
        const unsigned bits = __hip_atomic_load(reinterpret_cast<const unsigned *>(g) + off, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return (double)__uint_as_float(bits);
    }
    static NDP_D void count(int *ctr) { if (lane() == 0) atomicAdd(ctr, 1); }
    static NDP_D void count64(unsigned long long *ctr) { if (lane() == 0) __hip_atomic_fetch_add(ctr, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    // "this wave has its force values".  Three levels, because an agent-scope atomic on ONE address costs 30-60 ns and they
    // serialise (the eight XCDs' L2s are not coherent: such atomics execute at the memory side) -- one per wave made the launch
    // 36 us instead of 18, one per workgroup (256) still 35: (1) the waves of a workgroup count themselves in an LDS word; (2) the
    // one that counts last there adds the workgroup to its GROUP's counter (workgroup index mod 8: eight addresses, 32 atomics each
    // at batch 1024) with a relaxed returning atomic whose result is not looked at before the end of the step; (3) the workgroup
    // that completes its group for this launch adds one to the count of completed groups (late_publish, nothing returned).  All
    // counters only ever grow: launch t has been read completely when the completed-groups word has reached t * groups, and a
    // launch's own number is (completed groups) / groups + 1 -- stable while it runs.  Relaxed is enough for the counting: the
    // loads being protected have completed (their values were used) before the first atomic is issued.
    typedef unsigned late_t;
    static NDP_D late_t late_none() { return 0xffffffffu; }
    static NDP_D late_t late_count(unsigned *cnt, void *group, unsigned group_size)
    {
        late_t prev = 0xffffffffu;
        if (lane() == 0) {
            typedef __attribute__((address_space(3))) unsigned *lds_u32;
            // group: LDS offset of the workgroup's counter + 1, or null
            const bool last_of_group = !group || __hip_atomic_fetch_add((lds_u32)((unsigned)(size_t)group - 1u), 1u, __ATOMIC_RELAXED,
                                                                       __HIP_MEMORY_SCOPE_WORKGROUP) == group_size - 1;
            if (last_of_group) prev = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return prev;
    }
    static NDP_D void late_publish(late_t prev, unsigned gsize, unsigned long long *done_groups)
    {
        if (lane() == 0 && prev != 0xffffffffu && (prev + 1u) % gsize == 0u)
            __hip_atomic_fetch_add(done_groups, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }

    // 1/a: v_rcp_f64 seed + two Newton steps (a full IEEE divide is ~100 dependent cycles on gfx950)
    static NDP_D vd rcp(vd a)
    {
        double r = __builtin_amdgcn_rcp(a);
        r = __builtin_fma(__builtin_fma(-a, r, 1.0), r, r);
        r = __builtin_fma(__builtin_fma(-a, r, 1.0), r, r);
        return r;
    }
    // precision study: x rounded to fp32 (mode 1) or bf16 (mode 2), round-to-nearest-even, returned as a double
    static NDP_D vd round_op(vd x, int mode)
    {
        float f = (float)x;
        if (mode == 2) {
            unsigned u = __float_as_uint(f);
            u = (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u;
            f = __uint_as_float(u);
        }
        return (double)f;
    }
    // the pieces of rcp / quad_sum, for callers that interleave them with matrix instructions
    static NDP_D vd rcp_seed(vd a) { return __builtin_amdgcn_rcp(a); }
    static NDP_D vd fma(vd a, vd b, vd c) { return __builtin_fma(a, b, c); }
    static NDP_D vd quad_swap1(vd a) { return dpp_quad<0xB1>(a); }   // lane ^ 1
    static NDP_D vd quad_swap2(vd a) { return dpp_quad<0x4E>(a); }   // lane ^ 2
    // x + csum1(x), then + csum2 of that: the sum over the four lanes (same g) that hold columns 4q..4q+3 of one matrix row
    static NDP_D vd csum1(vd a) { return dpp_quad<0xB1>(a); }
    static NDP_D vd csum2(vd a) { return dpp_quad<0x4E>(a); }
    // sum over the 4 lanes of each aligned quad, result in all 4 (DPP quad_perm, no LDS)
    static NDP_D vd quad_sum(vd a)
    {
        a = a + dpp_quad<0xB1>(a);   // quad_perm [1,0,3,2]
        a = a + dpp_quad<0x4E>(a);   // quad_perm [2,3,0,1]
        return a;
    }
    template <int CTRL>
    static NDP_D vd dpp_quad(vd a)
    {
        const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(a), CTRL, 0xF, 0xF, true);
        const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(a), CTRL, 0xF, 0xF, true);
        return __hiloint2double(hi, lo);
    }
    static NDP_D vd clock() { return (double)__builtin_amdgcn_s_memtime(); }   // shader-clock ticks (debug stamps)
    // clock read that is ordered after `dep` has been produced (debug only): the asm consumes dep, fences scheduling
    static NDP_D vd clock_after(vd dep)
    {
        unsigned long long t;
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 0\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : "v"(dep) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        return (double)t;
    }

    // cross-lane
    static NDP_D double readlane(vd a, int l)
    {
        const int lo = __builtin_amdgcn_readlane(__double2loint(a), l);
        const int hi = __builtin_amdgcn_readlane(__double2hiint(a), l);
        return __hiloint2double(hi, lo);
    }
    static NDP_D double uniform(vd a)
    {
        const int lo = __builtin_amdgcn_readfirstlane(__double2loint(a));
        const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(a));
        return __hiloint2double(hi, lo);
    }
    // wave reductions: four DPP steps (lane ^ 1, lane ^ 2, row_ror 4 and 8: every lane of a 16-lane row then holds the row's
    // result), the four rows through v_readlane -- ~25 instructions; the xor-butterfly over ds_bpermute took ~650 cycles each
    // and the interior-point loop does four per iteration
    template <class Op>
    static NDP_D double wave_reduce(vd a, Op op)
    {
        a = op(a, dpp_quad<0xB1>(a));
        a = op(a, dpp_quad<0x4E>(a));
        a = op(a, dpp_quad<0x124>(a));
        a = op(a, dpp_quad<0x128>(a));
        const double r0 = readlane(a, 0), r1 = readlane(a, 16), r2 = readlane(a, 32), r3 = readlane(a, 48);
        return uniform(op(op(r0, r1), op(r2, r3)));      // (the same in every lane: told to the compiler -- the result then lives in scalar registers)
    }
    static NDP_D double wave_min(vd a) { return wave_reduce(a, [](double x, double y) { return fmin(x, y); }); }
    static NDP_D double wave_max(vd a) { return wave_reduce(a, [](double x, double y) { return fmax(x, y); }); }
    static NDP_D double wave_sum(vd a) { return wave_reduce(a, [](double x, double y) { return x + y; }); }
    static NDP_D bool all(vb p) { return __all((int)p) != 0; }
    static NDP_D bool any(vb p) { return __any((int)p) != 0; }
    // lane-wise and / or WITHOUT C++'s short circuit: `a && b` on per-lane bools is control flow, and the compiler sinks b's operands --
    // LDS loads included -- into the branch: every test then waits for its own load (seen in the step's tail: eight branches, eight waits)
    static NDP_D vb band(vb a, vb b) { return (vb)((int)a & (int)b); }
    static NDP_D vb bor(vb a, vb b) { return (vb)((int)a | (int)b); }

    // v_mfma_f64_16x16x4_f64: A lane l = A[l&15][l>>4], B lane l = B[l>>4][l&15], D reg r lane l = D[(l>>4)+4r][l&15]
    static NDP_D vd4 zero4() { vd4 z; z.r[0] = z.r[1] = z.r[2] = z.r[3] = 0.0; return z; }
    static NDP_D vd4 mfma(vd a, vd b, const vd4 &c)
    {
        d4_t acc = {c.r[0], c.r[1], c.r[2], c.r[3]};
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        vd4 d;
        d.r[0] = acc[0]; d.r[1] = acc[1]; d.r[2] = acc[2]; d.r[3] = acc[3];
        return d;
    }
    // v_mfma_f64_4x4x4_4b_f64: four independent 4x4x4 products, one per block b = (l >> 2) & 3 -- block b's A[i][k] in lane
    // i + 4b + 16k, B[k][j] in lane j + 4b + 16k, D[i][j] in lane j + 4b + 16i (scripts/ubench/mfma_f64_4x4x4.hip;
    // tests/test_gpu_parity.py::test_mfma_register_maps).  Read as ONE product it is (16x4 A, rows i + 4b) x (4x4 B shared by
    // the blocks) or (4x4 A shared) x (4x16 B, columns j + 4b) with the operands exactly where the 16x16x4 form keeps them:
    // a quarter of the work in a quarter of the time (18 cycles issue, 44 dependent, against 64) wherever one side of a product
    // is only four wide -- the gains, adj(Lam) T, and every matrix-vector product of the forward / second-solve sweeps.
    static NDP_D md mfma4(md a, md b, md c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }
    // lane 4Q of every 16-lane row to all lanes of the row (DPP row_newbcast): column 0 of block Q of a mfma4 result
    // becomes the B operand "vector element k = l >> 4, the same in every column" of the next one
    // rotation inside every 16-lane row (DPP row_ror:4n): lane l takes the value of lane (l - 4n) mod 16 of its row, i.e. the four
    // lanes of block b take block b - n's (pinned on the device: tests/test_gpu_parity.py::test_mfma_register_maps)
    template <int n>
    static NDP_D md rowror4(md a)
    {
        const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(a), 0x120 + 4 * n, 0xF, 0xF, true);
        const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(a), 0x120 + 4 * n, 0xF, 0xF, true);
        return __hiloint2double(hi, lo);
    }
    template <int Q>
    static NDP_D md rowb(md a)
    {
        // bound_ctrl = true: every lane has a valid source under row_newbcast, and with it the compiler knows the destination's
        // previous value is dead -- without it each v_mov_b32_dpp was preceded by a v_mov_b32 dst, 0 (160 per forward sweep)
        const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(a), 0x150 + 4 * Q, 0xF, 0xF, true);
        const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(a), 0x150 + 4 * Q, 0xF, 0xF, true);
        return __hiloint2double(hi, lo);
    }
};

// ---------------------------------------------------------------------------------------------------------------------
// BASELINE config 5 ("fp32 vs bf16 MFMA on the QP"): the same wave program with the Riccati sweeps on the fp32 matrix
// instruction.  v_mfma_f32_16x16x4_f32: A lane l = A[l&15][l>>4], B lane l = B[l>>4][l&15] like the f64 form, but accumulator
// register r of lane l holds D[4 (l>>4) + r][l&15] (f64: D[(l>>4) + 4r][l&15]).  Renumbering rows and columns by
// i -> (i >> 2) + 4 (i & 3) turns that into the f64 picture (register r <-> rows g + 4r, contraction step c <-> indices
// 4c..4c+3), so the program only has to be told which column a lane holds (lcol) and how to sum over a matrix row's four
// lanes (stride 4 inside the 16-lane row: DPP row_ror 4 and 8 instead of the quad swaps).  Everything outside the sweeps
// stays fp64; operands are converted on the way in and out.
struct WaveGfx950F32 : WaveGfx950 {
    using md = float;
    struct md4 { float r[4]; };
    typedef float f4_t __attribute__((ext_vector_type(4)));
    static constexpr bool packed_k = false, delta_ok = false, has_mma4 = false;
    static NDP_D md to_m(vd a) { return (float)a; }
    static NDP_D vd to_d(md a) { return (double)a; }
    static NDP_D md4 mzero4() { md4 z; z.r[0] = z.r[1] = z.r[2] = z.r[3] = 0.0f; return z; }
    static NDP_D md mavg(md a, md b) { return (a + b) * 0.5f; }
    static NDP_D md msel(vb p, md a, md b) { return p ? a : b; }
    static NDP_D vi lcol(vi lane) { const int jt = lane & 15; return (jt >> 2) + 4 * (jt & 3); }
    static NDP_D vd csum1(vd a) { return dpp_quad<0x124>(a); }   // row_ror:4
    static NDP_D vd csum2(vd a) { return dpp_quad<0x128>(a); }   // row_ror:8
    static NDP_D md4 mfma(md a, md b, const md4 &c)
    {
        f4_t acc = {c.r[0], c.r[1], c.r[2], c.r[3]};
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
        md4 d;
        d.r[0] = acc[0]; d.r[1] = acc[1]; d.r[2] = acc[2]; d.r[3] = acc[3];
        return d;
    }
};

// ... and on the bf16-input instruction with fp32 accumulators, v_mfma_f32_16x16x16_bf16: lane l supplies four bf16 values
// A[l&15][4 (l>>4) + i] / B[4 (l>>4) + i][l&15], i = 0..3 -- under the renumbering above element i is exactly the operand of
// contraction step i of the fp32 form, so up to four steps on one accumulator become ONE instruction (packed_k).
struct WaveGfx950BF16 : WaveGfx950F32 {
    typedef short s4_t __attribute__((ext_vector_type(4)));
    static constexpr bool packed_k = true;
    static NDP_D short bf16_bits(float f)
    {
        unsigned u = __float_as_uint(f);
        u = (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;          // round to nearest even (finite inputs)
        return (short)u;
    }
    static NDP_D md4 mfma_k(const md *a, const md *b, int n, const md4 &c)
    {
        s4_t av = {0, 0, 0, 0}, bv = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < n) { av[i] = bf16_bits(a[i]); bv[i] = bf16_bits(b[i]); }
        f4_t acc = {c.r[0], c.r[1], c.r[2], c.r[3]};
        acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(av, bv, acc, 0, 0, 0);
        md4 d;
        d.r[0] = acc[0]; d.r[1] = acc[1]; d.r[2] = acc[2]; d.r[3] = acc[3];
        return d;
    }
};

}  // namespace ndp
