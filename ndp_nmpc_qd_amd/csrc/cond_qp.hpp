// cond_qp.hpp -- BASELINE config 5's study: the QP of one RTI iteration solved in CONDENSED form on the fp32 / bf16 matrix instructions.
//
// The reference does not condense (qp_solver_cond_N = N, nmpc_body_rate_ctl.py:79: HPIPM's Riccati recursion IS its QP solve, and
// the fp64 Riccati sweeps of rti_wave.hpp are the product path).  BASELINE.json's configs[4] and the north star nevertheless name a
// precision / throughput study -- "fp32 vs bf16 MFMA on the condensed QP", "MFMA for the batched condensed-QP GEMMs", "the condensed
// KKT factorisation staged in LDS"  -- and this file is that study, kept as a mode (ndp_cfg.qp_precision 5 / 6), not
// offered as a product path: DESIGN section 8 has the numbers.
//
// One wave per instance, after the step's linearisation (stage blocks MB_k = [A b B], cost blocks CB_k in the wave's LDS slice):
//   * prediction matrices in homogeneous form: z~_k = Z~_k [U ; 1], Z~_k (16 x (4N+1)): rows 0..9 = [Gamma_k | g_k] (state response to
//     the inputs / free response), row 10 = e_last (the constant 1), rows 12..15 = the selector of u_k.  Z_{k+1} = (M~_k Z~_k)[0..9]:
//     one 16x16 block times a 16 x (4N+1) panel, tile by tile on the matrix instruction (A, B, b in one product).
//   * condensed Hessian AND gradient in one accumulation:  H~ = sum_k Z~_k' C~_k Z~_k  ((4N+1) x (4N+1), lower tiles of 16 x 16, fp32
//     accumulators resident in LDS): W = C~_k Z~_k (matrix instruction), H~(I, J) += Z~_k(:, I)' W(:, J) (matrix instruction).  Its
//     leading 4N x 4N block is H = R + Gamma' Q Gamma, its last row the gradient.  Tiles whose columns are still zero at stage k are skipped.
//   * MODE 1: every product on v_mfma_f32_16x16x4_f32; MODE 2: on v_mfma_f32_16x16x16_bf16 (operands rounded to bf16, fp32 accumulate).
//   * Cholesky H = L L' in LDS in fp32, blocked by the same tiles (diagonal tile in registers, panel by substitution, trailing update on
//     v_mfma_f32_16x16x4_f32), two substitutions, then the state step by an fp64 rollout of the linearised dynamics.
// The caller (RtiWave::run, COND != 0) then applies the fp64 inside-the-box test to the result: inside -> it is the step; otherwise
// (or when the factorisation fails: H not positive definite in fp32) the fp64 Riccati path solves the QP as if nothing had happened --
// every block it needs is untouched (the condensed solve works in an extra LDS area behind the wave's slice).
#pragma once

namespace ndp {

typedef __attribute__((address_space(3))) float *cq_lds;
typedef float cq_f4 __attribute__((ext_vector_type(4)));
typedef short cq_s4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) cq_f4 *cq_lds4;

NDP_HD int cond_ntiles(int N) { return (4 * N + 1 + 15) / 16; }
// doubles of LDS behind the wave's slice: Z (10 x NCP) | W tile (256) | right-hand side (NCP) | lower tiles of H~ (256 each), as floats
NDP_HD int cond_extra_doubles(int N)
{
    const int nt = cond_ntiles(N), ncp = nt * 16;
    return (10 * ncp + 256 + ncp + nt * (nt + 1) / 2 * 256 + 1) / 2 + 2;
}

__device__ __forceinline__ short cq_bf16(float f)
{
    unsigned u = __float_as_uint(f);
    u = (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;          // round to nearest even (finite inputs)
    return (short)u;
}

// one 16 x 16 x 16 product-accumulate: lane (i = l & 15, g = l >> 4) supplies a[s] = A[i][4 s + g], b[s] = B[4 s + g][l & 15];
// accumulator register r of lane l = D[4 g + r][l & 15]
template <int MODE>
__device__ __forceinline__ cq_f4 cq_mma(const float a[4], const float b[4], cq_f4 c)
{
    if constexpr (MODE == 2) {
        cq_s4 av, bv;
#pragma unroll
        for (int s = 0; s < 4; ++s) { av[s] = cq_bf16(a[s]); bv[s] = cq_bf16(b[s]); }
        return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(av, bv, c, 0, 0, 0);
    } else {
#pragma unroll
        for (int s = 0; s < 4; ++s) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[s], c, 0, 0, 0);
        return c;
    }
}

__device__ __forceinline__ float cq_readlane(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

// element (row i, column c) of a 16 x 16 tile kept in accumulator order (lane * 4 + register)
__device__ __forceinline__ int cq_eo(int i, int c) { return (((i >> 2) << 4) + c) * 4 + (i & 3); }

// m_entry(r, c) / c_entry(r, c): LDS offsets (doubles) of element (r, c) of M~_0 / C~_0 (RtiWave::m_entry / c_entry); stage k: + k * stride.
// Returns false when H is not positive definite in fp32 (nothing the caller needs has been written then).
template <int MODE, class ME, class CE>
__device__ bool WaveGfx950::cond_solve(const RtiParams &P, const LdsMap &m, lds_ptr lds, int N, ME m_entry, CE c_entry)
{
    const int lane = (int)(threadIdx.x & 63u), g = lane >> 4, j = lane & 15;
    const int n = 4 * N, NCOL = n + 1, NT = (NCOL + 15) >> 4, NCP = NT * 16, NTH = n >> 4;     // (n is a multiple of 16: checked by the host)
    cq_lds Zb = (cq_lds)(lds + m.total), Wt = Zb + 10 * NCP, bv = Wt + 256, Hh = bv + NCP;
    auto tile = [&](int I, int J) { return Hh + (I * (I + 1) / 2 + J) * 256; };
    int mo[4], co[4];
    bool cx[4];                                   // operand entries that exist in the terminal block (no input part there)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        mo[s] = m_entry(j, 4 * s + g);
        co[s] = c_entry(j, 4 * s + g);
        cx[s] = j < 12 && 4 * s + g < 12;
    }
    for (int i = lane; i < 10 * NCP; i += 64) Zb[i] = 0.0f;
    for (int i = lane; i < NT * (NT + 1) / 2 * 256; i += 64) Hh[i] = 0.0f;
    sync();
    if (lane < 10) Zb[lane * NCP + (NCOL - 1)] = (float)lds[m.ZX + lane];          // Z_0 = [0 | dx_0]
    sync();
    // B operand (and, transposed, A operand) of column tile T of Z~_k: lane (g, j) holds Z~_k[4 s + g][16 T + j]
    auto zop = [&](int k, int T, float z[4]) {
        const int col = 16 * T + j;
        z[0] = Zb[g * NCP + col];
        z[1] = Zb[(4 + g) * NCP + col];
        z[2] = g < 2 ? Zb[(8 + g) * NCP + col] : ((g == 2 && col == NCOL - 1) ? 1.0f : 0.0f);
        z[3] = (k < N && col == 4 * k + g) ? 1.0f : 0.0f;
    };
    for (int k = 0; k <= N; ++k) {
        // columns that can be non-zero at stage k: the inputs of stages 0..k (k < N), and the constant's column
        const int TU = k < N ? (4 * k + 3) >> 4 : NTH - 1;
        auto active = [&](int T) { return T <= TU || T == NT - 1; };
        for (int J = 0; J < NT; ++J) {
            if (!active(J)) continue;
            float bz[4], ca[4];
            zop(k, J, bz);
#pragma unroll
            for (int s = 0; s < 4; ++s) ca[s] = (k == N && !cx[s]) ? 0.0f : (float)lds[co[s] + k * int(CB_STRIDE)];
            cq_f4 w = {0.0f, 0.0f, 0.0f, 0.0f};
            w = cq_mma<MODE>(ca, bz, w);                                           // W(:, J) = C~_k Z~_k(:, J)
            sync();
#pragma unroll
            for (int r = 0; r < 4; ++r) Wt[(4 * g + r) * 16 + j] = w[r];
            sync();
            float bw[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) bw[s] = Wt[(4 * s + g) * 16 + j];
            // H~(I, J) += Z~_k(:, I)' W(:, J) for the active I >= J: the rows J..TU and the constant's row NT - 1.  Two tiles per round --
            // one wave has nothing else to hide an LDS round trip or a dependent matrix instruction under
            const int nI = (TU >= J ? TU - J + 1 : 0) + ((NT - 1 > TU && NT - 1 >= J) ? 1 : 0);
            auto Iof = [&](int q) { const int I = J + q; return (TU >= J && I <= TU) ? I : NT - 1; };
            for (int q = 0; q < nI; q += 2) {
                const int I0 = Iof(q), I1 = q + 1 < nI ? Iof(q + 1) : I0;
                float a0[4], a1[4];
                zop(k, I0, a0);
                zop(k, I1, a1);
                cq_lds4 t0 = (cq_lds4)(tile(I0, J) + lane * 4), t1 = (cq_lds4)(tile(I1, J) + lane * 4);
                cq_f4 c0 = *t0, c1 = *t1;
                c0 = cq_mma<MODE>(a0, bw, c0);
                c1 = cq_mma<MODE>(a1, bw, c1);
                *t0 = c0;
                if (q + 1 < nI) *t1 = c1;
            }
        }
        if (k < N) {
            for (int J = 0; J < NT; ++J) {
                if (!active(J)) continue;
                float bz[4], ma[4];
                zop(k, J, bz);
#pragma unroll
                for (int s = 0; s < 4; ++s) ma[s] = (float)lds[mo[s] + k * int(MB_STRIDE)];
                cq_f4 d = {0.0f, 0.0f, 0.0f, 0.0f};
                d = cq_mma<MODE>(ma, bz, d);                                       // Z_{k+1}(:, J) = (M~_k Z~_k(:, J))[0..9]
                sync();
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (4 * g + r < 10) Zb[(4 * g + r) * NCP + 16 * J + j] = d[r];
                sync();
            }
        }
    }
    sync();
    // right-hand side -h: the gradient is row n of H~ = row 0 of the last tile row
    for (int c = lane; c < n; c += 64) bv[c] = -tile(NT - 1, c >> 4)[(c & 15) * 4];
    sync();
    // ---- H = L L' (fp32), blocked by tiles
    for (int p = 0; p < NTH; ++p) {
        float a[16];
        {
            cq_lds t = tile(p, p);
#pragma unroll
            for (int c = 0; c < 16; ++c) a[c] = t[cq_eo(j, c)];                    // lane j (every 16-lane row alike) holds row j
        }
        bool pd = true;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const float d = cq_readlane(a[c], c);
            pd = pd && d > 0.0f;
            const float inv = 1.0f / sqrtf(d);
            const float lc = a[c] * inv;                                           // lane j > c: L[j][c]; lane c: sqrt(d)
            a[c] = lc;
#pragma unroll
            for (int c2 = c + 1; c2 < 16; ++c2) a[c2] -= lc * cq_readlane(lc, c2);
        }
        if (!pd) return false;
        if (lane < 16) {
            cq_lds t = tile(p, p);
#pragma unroll
            for (int c = 0; c < 16; ++c) t[cq_eo(lane, c)] = a[c];
        }
        float rd[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) rd[c] = 1.0f / cq_readlane(a[c], c);
        const int nr = (NTH - 1 - p) * 16;
        for (int base = 0; base < nr; base += 64) {                                // panel: X L_pp' = H_Ip, one row per lane
            const int rr = base + lane;
            const bool valid = rr < nr;
            const int rc = valid ? rr : 0;
            cq_lds t = tile(p + 1 + (rc >> 4), p);
            const int i = rc & 15;
            float x[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) x[c] = t[cq_eo(i, c)];
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                float s = x[c];
#pragma unroll
                for (int q = 0; q < c; ++q) s -= x[q] * cq_readlane(a[q], c);      // L_pp[c][q]: lane c's register q
                x[c] = s * rd[c];
            }
            if (valid) {
#pragma unroll
                for (int c = 0; c < 16; ++c) t[cq_eo(i, c)] = x[c];
            }
        }
        sync();
        for (int J = p + 1; J < NTH; ++J) {                                        // trailing update H_IJ -= L_Ip L_Jp'
            float bl[4];
            {
                cq_lds t = tile(J, p);
#pragma unroll
                for (int s = 0; s < 4; ++s) bl[s] = t[cq_eo(j, 4 * s + g)];
            }
            for (int I = J; I < NTH; ++I) {
                float al[4];
                cq_lds t = tile(I, p);
#pragma unroll
                for (int s = 0; s < 4; ++s) al[s] = -t[cq_eo(j, 4 * s + g)];
                cq_lds4 acc = (cq_lds4)(tile(I, J) + lane * 4);
                *acc = cq_mma<1>(al, bl, *acc);
            }
        }
        sync();
    }
    // ---- L y = -h, L' U = y (rows lane, lane + 64, lane + 128 of the right-hand side belong to this lane)
    auto Lel = [&](int r, int c) { return tile(r >> 4, c >> 4)[cq_eo(r & 15, c & 15)]; };
    for (int c = 0; c < n; ++c) {
        const float yc = bv[c] / Lel(c, c);
        sync();
        if (lane == 0) bv[c] = yc;
        for (int r = lane; r < n; r += 64)
            if (r > c) bv[r] -= Lel(r, c) * yc;
        sync();
    }
    for (int c = n - 1; c >= 0; --c) {
        const float uc = bv[c] / Lel(c, c);
        sync();
        if (lane == 0) bv[c] = uc;
        for (int r = lane; r < c; r += 64) bv[r] -= Lel(c, r) * uc;
        sync();
    }
    for (int e = lane; e < n; e += 64) lds[m.ZU + e] = (double)bv[e];
    sync();
    // ---- the state step: fp64 rollout of dx+ = A dx + B du + b with the step's own blocks
    int me[16];
    const int ir = lane < 10 ? lane : 9;
#pragma unroll
    for (int c = 0; c < 16; ++c) me[c] = m_entry(ir, c);
    for (int k = 0; k < N; ++k) {
        double mv[16], zv[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) {                 // every read of the stage first: one wait
            mv[c] = c == 11 ? 0.0 : lds[me[c] + k * int(MB_STRIDE)];
            zv[c] = c < 10 ? lds[m.ZX + k * int(NX) + c] : (c == 10 ? 1.0 : (c == 11 ? 0.0 : lds[m.ZU + k * int(NU) + (c - 12)]));
        }
        double acc = 0.0;
#pragma unroll
        for (int c = 0; c < 16; ++c) acc += mv[c] * zv[c];
        sync();
        if (lane < 10) lds[m.ZX + (k + 1) * int(NX) + lane] = acc;
        sync();
    }
    (void)P;
    return true;
}

}  // namespace ndp
