// wave_emu.hpp -- host lock-step emulation of a 64-lane wavefront (TEST INFRASTRUCTURE ONLY).
//
// Lets tests/ run the exact text of ndp_nmpc_qd_amd/csrc/rti_wave.hpp on the CPU: every per-lane
// value is an array of 64 elements, every operation is applied to all lanes, and the matrix
// instruction follows the gfx950 register maps documented for v_mfma_f64_16x16x4_f64
// (A: lane l holds A[l&15][l>>4]; B: lane l holds B[l>>4][l&15]; C/D: register r of lane l holds
// D[(l>>4)+4r][l&15]).  Never linked into the product library.
#pragma once
#include <cmath>
#include <cstring>

#define NDP_D inline
#define NDP_HD inline

namespace emu {

struct vb { bool v[64]; };
struct vi {
    int v[64];
    vi() {}
    vi(int s) { for (int l = 0; l < 64; ++l) v[l] = s; }
};
struct vd {
    double v[64];
    vd() {}
    vd(double s) { for (int l = 0; l < 64; ++l) v[l] = s; }
};
struct vd4 { vd r[4]; };

#define EMU_BIN(T, R, op) \
    inline R operator op(const T &a, const T &b) { R o; for (int l = 0; l < 64; ++l) o.v[l] = a.v[l] op b.v[l]; return o; }
EMU_BIN(vd, vd, +) EMU_BIN(vd, vd, -) EMU_BIN(vd, vd, *) EMU_BIN(vd, vd, /)
EMU_BIN(vd, vb, <) EMU_BIN(vd, vb, >) EMU_BIN(vd, vb, <=) EMU_BIN(vd, vb, >=) EMU_BIN(vd, vb, ==)
EMU_BIN(vi, vi, +) EMU_BIN(vi, vi, -) EMU_BIN(vi, vi, *) EMU_BIN(vi, vi, &) EMU_BIN(vi, vi, >>)
EMU_BIN(vi, vb, <) EMU_BIN(vi, vb, >) EMU_BIN(vi, vb, <=) EMU_BIN(vi, vb, >=) EMU_BIN(vi, vb, ==)
EMU_BIN(vb, vb, &&) EMU_BIN(vb, vb, ||)
#undef EMU_BIN
// mixed scalar forms (the implicit constructors make the right-hand conversions; these cover scalar-on-the-left)
inline vd operator+(double a, const vd &b) { return vd(a) + b; }
inline vd operator-(double a, const vd &b) { return vd(a) - b; }
inline vd operator*(double a, const vd &b) { return vd(a) * b; }
inline vd operator/(double a, const vd &b) { return vd(a) / b; }
inline vd operator+(const vd &a, double b) { return a + vd(b); }
inline vd operator-(const vd &a, double b) { return a - vd(b); }
inline vd operator*(const vd &a, double b) { return a * vd(b); }
inline vd operator/(const vd &a, double b) { return a / vd(b); }
inline vb operator<(const vd &a, double b) { return a < vd(b); }
inline vb operator>(const vd &a, double b) { return a > vd(b); }
inline vi operator+(const vi &a, int b) { return a + vi(b); }
inline vi operator-(const vi &a, int b) { return a - vi(b); }
inline vi operator*(const vi &a, int b) { return a * vi(b); }
inline vi operator&(const vi &a, int b) { return a & vi(b); }
inline vi operator>>(const vi &a, int b) { return a >> vi(b); }
inline vi operator+(int a, const vi &b) { return vi(a) + b; }
inline vi operator*(int a, const vi &b) { return vi(a) * b; }
inline vb operator<(const vi &a, int b) { return a < vi(b); }
inline vb operator>(const vi &a, int b) { return a > vi(b); }
inline vb operator<=(const vi &a, int b) { return a <= vi(b); }
inline vb operator>=(const vi &a, int b) { return a >= vi(b); }
inline vb operator==(const vi &a, int b) { return a == vi(b); }
inline vd operator-(const vd &a) { vd o; for (int l = 0; l < 64; ++l) o.v[l] = -a.v[l]; return o; }
inline vb operator!(const vb &a) { vb o; for (int l = 0; l < 64; ++l) o.v[l] = !a.v[l]; return o; }

struct Stats { long mfma = 0, lds_ld = 0, lds_st = 0, readlane = 0, mfma4 = 0; };
inline Stats &stats() { static thread_local Stats s; return s; }

struct Wave {
    using vd = emu::vd;
    using vi = emu::vi;
    using vb = emu::vb;
    using vd4 = emu::vd4;
    using lds_ptr = double *;
    // matrix values of the Riccati sweeps (RtiWave::md): the f64 instruction
    using md = emu::vd;
    using md4 = emu::vd4;
    static constexpr bool packed_k = false, delta_ok = true, has_mma4 = true;
    static vd to_m(const vd &a) { return a; }
    static vd to_d(const vd &a) { return a; }
    static vd4 mzero4() { return zero4(); }
    static vd mavg(const vd &a, const vd &b) { return (a + b) * 0.5; }
    static vd msel(const vb &p, const vd &a, const vd &b) { return sel(p, a, b); }
    static vi lcol(const vi &lane) { return lane & 15; }
    static vd csum1(const vd &a) { return quad_swap1(a); }
    static vd csum2(const vd &a) { return quad_swap2(a); }

    static int &lds_limit() { static thread_local int n = 0; return n; }
    static void chk(int i) { if (i < 0 || i >= lds_limit()) __builtin_trap(); }

    static vi lane() { vi o; for (int l = 0; l < 64; ++l) o.v[l] = l; return o; }
    static vi lane_here() { return lane(); }
    static vd sel(const vb &p, const vd &a, const vd &b) { vd o; for (int l = 0; l < 64; ++l) o.v[l] = p.v[l] ? a.v[l] : b.v[l]; return o; }
    static vi sel(const vb &p, const vi &a, const vi &b) { vi o; for (int l = 0; l < 64; ++l) o.v[l] = p.v[l] ? a.v[l] : b.v[l]; return o; }
    static vd vmin(const vd &a, const vd &b) { vd o; for (int l = 0; l < 64; ++l) o.v[l] = std::fmin(a.v[l], b.v[l]); return o; }
    static vd vmax(const vd &a, const vd &b) { vd o; for (int l = 0; l < 64; ++l) o.v[l] = std::fmax(a.v[l], b.v[l]); return o; }
    static vd vabs(const vd &a) { vd o; for (int l = 0; l < 64; ++l) o.v[l] = std::fabs(a.v[l]); return o; }
    static vi div3(const vi &a) { vi o; for (int l = 0; l < 64; ++l) o.v[l] = a.v[l] / 3; return o; }
    static vi div6(const vi &a) { vi o; for (int l = 0; l < 64; ++l) o.v[l] = a.v[l] / 6; return o; }

    // LDS
    static vd ld(const double *lds, const vi &off) { vd o; stats().lds_ld++; for (int l = 0; l < 64; ++l) { chk(off.v[l]); o.v[l] = lds[off.v[l]]; } return o; }
    static void ld2(const double *lds, const vi &off, vd &a, vd &b)
    {
        stats().lds_ld++;
        for (int l = 0; l < 64; ++l) { if (off.v[l] & 1) __builtin_trap(); chk(off.v[l]); chk(off.v[l] + 1); a.v[l] = lds[off.v[l]]; b.v[l] = lds[off.v[l] + 1]; }
    }
    static vd ldp(const double *lds, const vi &off, const vb &p) { vd o; stats().lds_ld++; for (int l = 0; l < 64; ++l) { if (p.v[l]) { chk(off.v[l]); o.v[l] = lds[off.v[l]]; } else o.v[l] = 0.0; } return o; }
    static vd ldz(const double *lds, const vi &off, const vb &p) { return ldp(lds, off, p); }
    static void stp(double *lds, const vi &off, const vd &val, const vb &p) { stats().lds_st++; for (int l = 0; l < 64; ++l) if (p.v[l]) { chk(off.v[l]); lds[off.v[l]] = val.v[l]; } }
    static void st(double *lds, const vi &off, const vd &val) { stats().lds_st++; for (int l = 0; l < 64; ++l) { chk(off.v[l]); lds[off.v[l]] = val.v[l]; } }
    static void pin() {}
    static void keep(const vd &) {}
    static void sync() {}

    // global memory
    static vd gld(const double *g, const vi &off, const vb &p) { vd o; for (int l = 0; l < 64; ++l) o.v[l] = p.v[l] ? g[off.v[l]] : 0.0; return o; }
    static vd gldf(const float *g, const vi &off, const vb &p) { vd o; for (int l = 0; l < 64; ++l) o.v[l] = p.v[l] ? (double)g[off.v[l]] : 0.0; return o; }
    static void gst(double *g, const vi &off, const vd &val, const vb &p) { for (int l = 0; l < 64; ++l) if (p.v[l]) g[off.v[l]] = val.v[l]; }
    static void cmd_store(double *cmd, double *keep, double k, double mass, const vi &idx, const vd &val, const vb &p)
    {
        for (int l = 0; l < 64; ++l)
            if (p.v[l]) {
                double v = val.v[l];
                if (idx.v[l] == 3) { v = k != 0.0 ? v * mass / k : 0.0; *keep = v; }
                cmd[idx.v[l]] = v;
            }
    }
    static vd gldu(const double *g, const vi &off) { vd o; for (int l = 0; l < 64; ++l) o.v[l] = g[off.v[l]]; return o; }
    static vd gldfu(const float *g, const vi &off) { vd o; for (int l = 0; l < 64; ++l) o.v[l] = (double)g[off.v[l]]; return o; }
    static vi gldi(const int *g, const vi &off) { vi o; for (int l = 0; l < 64; ++l) o.v[l] = g[off.v[l]]; return o; }
    template <class T> static const T *late_params(const T &ref) { return &ref; }
    static vi d2i(const vd &a) { vi o; for (int l = 0; l < 64; ++l) o.v[l] = (int)a.v[l]; return o; }
    static vd i2d(const vi &a) { vd o; for (int l = 0; l < 64; ++l) o.v[l] = (double)a.v[l]; return o; }
    static vi gld_i8(const signed char *g, const vi &off) { vi o; for (int l = 0; l < 64; ++l) o.v[l] = (int)g[off.v[l]]; return o; }
    static void gst_i8(signed char *g, const vi &off, const vi &val, const vb &p) { for (int l = 0; l < 64; ++l) if (p.v[l]) g[off.v[l]] = (signed char)val.v[l]; }
    static vi imin(const vi &a, const vi &b) { vi o; for (int l = 0; l < 64; ++l) o.v[l] = a.v[l] < b.v[l] ? a.v[l] : b.v[l]; return o; }
    static void gst2(double *a, double *b, const vi &i, int n, const vd &val) { for (int l = 0; l < 64; ++l) (i.v[l] < n ? a + i.v[l] : b + (i.v[l] - n))[0] = val.v[l]; }
    static void gsti(int *g, int val) { if (g) *g = val; }
    static bool wait_ge(const unsigned long long *flag, const unsigned long long *flag2, unsigned long long want, unsigned) { return *flag >= want && *flag2 >= want; }
    static vd gldf_fresh(const float *g, const vi &off) { return gldfu(g, off); }
    static void count(int *ctr) { ++*ctr; }
    static void count64(unsigned long long *ctr) { ++*ctr; }
    typedef unsigned late_t;
    static late_t late_none() { return 0xffffffffu; }
    static late_t late_count(unsigned *cnt, void *, unsigned) { return cnt ? (*cnt)++ : 0xffffffffu; }
    static void late_publish(late_t prev, unsigned gsize, unsigned long long *done_groups)
    {
        if (done_groups && prev != 0xffffffffu && (prev + 1u) % gsize == 0u) ++*done_groups;
    }

    static vd clock() { return vd(0.0); }
    static vd clock_after(const vd &) { return vd(0.0); }
    static vd rcp(const vd &a) { vd o; for (int l = 0; l < 64; ++l) o.v[l] = 1.0 / a.v[l]; return o; }
    static vd round_op(const vd &x, int mode)
    {
        vd o;
        for (int l = 0; l < 64; ++l) {
            float f = (float)x.v[l];
            if (mode == 2) {
                unsigned u; std::memcpy(&u, &f, 4);
                u = (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u;
                std::memcpy(&f, &u, 4);
            }
            o.v[l] = (double)f;
        }
        return o;
    }
    static vd rcp_seed(const vd &a) { return rcp(a); }
    static vd fma(const vd &a, const vd &b, const vd &c) { vd o; for (int l = 0; l < 64; ++l) o.v[l] = std::fma(a.v[l], b.v[l], c.v[l]); return o; }
    static vd quad_swap1(const vd &a) { vd o; for (int l = 0; l < 64; ++l) o.v[l] = a.v[l ^ 1]; return o; }
    static vd quad_swap2(const vd &a) { vd o; for (int l = 0; l < 64; ++l) o.v[l] = a.v[l ^ 2]; return o; }
    static vd quad_sum(const vd &a)
    {   // same association as the device: (x + x^1) + (x + x^1)^2
        vd t, o;
        for (int l = 0; l < 64; ++l) t.v[l] = a.v[l] + a.v[l ^ 1];
        for (int l = 0; l < 64; ++l) o.v[l] = t.v[l] + t.v[l ^ 2];
        return o;
    }

    // cross-lane
    static double readlane(const vd &a, int l) { stats().readlane++; return a.v[l]; }
    static double wave_min(const vd &a) { double m = a.v[0]; for (int l = 1; l < 64; ++l) m = std::fmin(m, a.v[l]); return m; }
    static double wave_max(const vd &a) { double m = a.v[0]; for (int l = 1; l < 64; ++l) m = std::fmax(m, a.v[l]); return m; }
    static double wave_sum(const vd &a)
    {   // xor-butterfly order, as the device reduction
        double t[64];
        std::memcpy(t, a.v, sizeof(t));
        for (int s = 32; s >= 1; s >>= 1) { double u[64]; for (int l = 0; l < 64; ++l) u[l] = t[l] + t[l ^ s]; std::memcpy(t, u, sizeof(t)); }
        return t[0];
    }
    static bool all(const vb &p) { for (int l = 0; l < 64; ++l) if (!p.v[l]) return false; return true; }
    static bool any(const vb &p) { for (int l = 0; l < 64; ++l) if (p.v[l]) return true; return false; }
    static vb band(const vb &a, const vb &b) { return a && b; }
    static vb bor(const vb &a, const vb &b) { return a || b; }

    // v_mfma_f64_16x16x4_f64: D = A(16x4) * B(4x16) + C
    static vd4 zero4() { vd4 z; for (int r = 0; r < 4; ++r) z.r[r] = vd(0.0); return z; }
    static vd4 mfma(const vd &a, const vd &b, const vd4 &c)
    {
        stats().mfma++;
        vd4 d;
        for (int r = 0; r < 4; ++r)
            for (int l = 0; l < 64; ++l) {
                const int row = (l >> 4) + 4 * r, col = l & 15;
                double acc = c.r[r].v[l];
                for (int k = 0; k < 4; ++k) acc = std::fma(a.v[row + 16 * k], b.v[col + 16 * k], acc);
                d.r[r].v[l] = acc;
            }
        return d;
    }
    // v_mfma_f64_4x4x4_4b_f64: block b = (l >> 2) & 3; A[i][k] lane i + 4b + 16k, B[k][j] lane j + 4b + 16k, D[i][j] lane j + 4b + 16i
    static vd mfma4(const vd &a, const vd &b, const vd &c)
    {
        stats().mfma4++;
        vd d;
        for (int l = 0; l < 64; ++l) {
            const int j = l & 3, blk = (l >> 2) & 3, i = l >> 4;
            double acc = c.v[l];
            for (int k = 0; k < 4; ++k) acc = std::fma(a.v[i + 4 * blk + 16 * k], b.v[j + 4 * blk + 16 * k], acc);
            d.v[l] = acc;
        }
        return d;
    }
    template <int n>
    static vd rowror4(const vd &a) { vd o; for (int l = 0; l < 64; ++l) o.v[l] = a.v[(l & 48) | ((l - 4 * n) & 15)]; return o; }   // DPP row_ror:4n
    template <int Q>
    static vd rowb(const vd &a) { vd o; for (int l = 0; l < 64; ++l) o.v[l] = a.v[(l & 48) + 4 * Q]; return o; }   // DPP row_newbcast:4Q
};

// The fp32 / bf16 backends of BASELINE config 5 (wave_gfx950.hpp: WaveGfx950F32, WaveGfx950BF16) emulated: matrix values are
// floats carried in doubles, accumulator register r of lane l holds D[4 (l>>4) + r][l&15], every product-accumulate step is
// one fmaf (v_mfma_f32_16x16x4_f32 is bitwise an fmaf chain), the row sums run over lanes 4 apart.
inline float f32(double x) { return (float)x; }
inline float bf16r(float f)
{
    unsigned u; std::memcpy(&u, &f, 4);
    u = (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u;
    std::memcpy(&f, &u, 4);
    return f;
}

struct Wave32 : Wave {
    static constexpr bool packed_k = false, delta_ok = false, has_mma4 = false;
    static vd to_m(const vd &a) { vd o; for (int l = 0; l < 64; ++l) o.v[l] = (double)f32(a.v[l]); return o; }
    static vd mavg(const vd &a, const vd &b) { vd o; for (int l = 0; l < 64; ++l) o.v[l] = (double)((f32(a.v[l]) + f32(b.v[l])) * 0.5f); return o; }
    static vi lcol(const vi &lane) { vi o; for (int l = 0; l < 64; ++l) { const int jt = lane.v[l] & 15; o.v[l] = (jt >> 2) + 4 * (jt & 3); } return o; }
    static vd csum1(const vd &a) { vd o; for (int l = 0; l < 64; ++l) o.v[l] = a.v[(l & 48) | ((l + 12) & 15)]; return o; }   // row_ror:4
    static vd csum2(const vd &a) { vd o; for (int l = 0; l < 64; ++l) o.v[l] = a.v[(l & 48) | ((l + 8) & 15)]; return o; }    // row_ror:8
    static vd4 mfma(const vd &a, const vd &b, const vd4 &c)
    {
        stats().mfma++;
        vd4 d;
        for (int r = 0; r < 4; ++r)
            for (int l = 0; l < 64; ++l) {
                const int row = 4 * (l >> 4) + r, col = l & 15;
                float acc = f32(c.r[r].v[l]);
                for (int k = 0; k < 4; ++k) acc = std::fmaf(f32(a.v[row + 16 * k]), f32(b.v[col + 16 * k]), acc);
                d.r[r].v[l] = (double)acc;
            }
        return d;
    }
};

struct WaveBF16 : Wave32 {
    static constexpr bool packed_k = true;
    // up to four contraction steps in one instruction: bf16 operands, products exact in fp32, fp32 accumulation
    static vd4 mfma_k(const vd *a, const vd *b, int n, const vd4 &c)
    {
        stats().mfma++;
        vd4 d;
        for (int r = 0; r < 4; ++r)
            for (int l = 0; l < 64; ++l) {
                const int row = 4 * (l >> 4) + r, col = l & 15;
                float acc = f32(c.r[r].v[l]);
                for (int i = 0; i < n; ++i)
                    for (int k = 0; k < 4; ++k)
                        acc = std::fmaf(bf16r(f32(a[i].v[row + 16 * k])), bf16r(f32(b[i].v[col + 16 * k])), acc);
                d.r[r].v[l] = (double)acc;
            }
        return d;
    }
};

}  // namespace emu
