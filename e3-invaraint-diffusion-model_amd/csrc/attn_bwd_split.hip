// Backward of the fused relative-key attention on the bf16 matrix cores (bf16x3 split arithmetic, fp32
// accumulation) -- the same two launches, layouts, LDS buffers and partial-block scheme as attn_bwd.hip
// (fp32 MFMA; kept for the exact / fp32-grade modes), with every 32x32x64 product rebuilt from
// v_mfma_f32_32x32x16_bf16 on operands split in registers into bf16 hi + lo:  12 MFMA slots of 32 cycles
// instead of 32 slots of 64 cycles per product (launch A: 96 x 32 instead of 256 x 64 cycles per
// (query tile, key tile)).
//
// Operand sources, launch A (lane = (qi = lane & 31, half)):
//   S^T = K Q^T, T^T = E Q^T, dP^T = V dO^T : rows split from global (8 consecutive head dims per 16-wide k block)
//   dQ^T += K^T dS^T, += E^T dT^T           : B = the accumulator-layout tile (registers 8 st .. 8 st + 7 =
//                                             rows rho(st, half, j)), A = the rows gathered with the same
//                                             permutation, two head dims (2c, 2c+1) per lane -- the PV scheme
//                                             of the forward kernel
//   dE block += dT^T Q                      : A = dT^T[x][query] read from the LDS tile with the inverse skew,
//                                             B = Q[query][d]; k = the 32 queries of the tile in natural order
// Launch B streams the materialised P / dS tiles (fp32) with k = the queries in natural order.
#include "e3d_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int D = 64;
constexpr int RING_LD = 34, RING_F = 64 * RING_LD;  // T^T ring (forward recompute)
constexpr int X_LD = 33, X_F = 32 * X_LD;           // 32x32 tile transpose / inverse-skew buffer
constexpr int WAVE_LDS_F = RING_F + X_F;

struct Frag {   // one MFMA operand: bf16 hi and lo terms of 8 fp32 values
    bf16x8 hi, lo;
};

__device__ __forceinline__ Frag split8(const float (&x)[8]) {
    Frag f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 p = (__bf16)x[j];
        f.hi[j] = p;
        f.lo[j] = (__bf16)(x[j] - (float)p);
    }
    return f;
}

// acc += a . b from the three significant cross terms, smallest first
__device__ __forceinline__ f32x16 mfma3(const Frag& a, const Frag& b, f32x16 acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi, b.lo, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.lo, b.hi, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi, b.hi, acc, 0, 0, 0);
    return acc;
}

// one row's 64 head-dim values -> 4 k-blocks (lane takes floats 16 kb + 8 half .. + 7); also returns them
__device__ __forceinline__ void load_row(float (&x)[4][8], const float* row_ptr, int half) {
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        const f32x4 lo = *reinterpret_cast<const f32x4*>(row_ptr + 16 * kb + 8 * half);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(row_ptr + 16 * kb + 8 * half + 4);
        x[kb][0] = lo[0]; x[kb][1] = lo[1]; x[kb][2] = lo[2]; x[kb][3] = lo[3];
        x[kb][4] = hi[0]; x[kb][5] = hi[1]; x[kb][6] = hi[2]; x[kb][7] = hi[3];
    }
}
__device__ __forceinline__ void load_row_split(Frag (&f)[4], const float* row_ptr, int half) {
    float x[4][8];
    load_row(x, row_ptr, half);
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) f[kb] = split8(x[kb]);
}

// tile[i][j] = X_i . Y_j (i on accumulator rows, j on lanes)
__device__ __forceinline__ f32x16 dot_tile(const Frag (&x)[4], const Frag (&y)[4]) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) acc = mfma3(x[kb], y[kb], acc);
    return acc;
}

// o^T[d, j] += sum_i Z[row_i][d] * w[i][j]:  w is an accumulator-layout tile (rows i in registers, columns j on
// lanes); lane (c, half) feeds Z[row0 + rho(st, half, j)][2c .. 2c+1] (rows clamped to [row_min, row_max]).
__device__ __forceinline__ void acc_times_rows(f32x16& o0, f32x16& o1, const float* z_base, int64_t z_rs, int row0,
                                               int row_min, int row_max, const f32x16& w, int c, int half) {
#pragma unroll
    for (int st = 0; st < 2; ++st) {
        float wv[8], z0[8], z1[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int row = min(max(row0 + mfma32_row(8 * st + j, half), row_min), row_max);
            const float2 zz = *reinterpret_cast<const float2*>(z_base + (int64_t)row * z_rs + 2 * c);
            wv[j] = w[8 * st + j];
            z0[j] = zz.x;
            z1[j] = zz.y;
        }
        const Frag wb = split8(wv);
        o0 = mfma3(split8(z0), wb, o0);
        o1 = mfma3(split8(z1), wb, o1);
    }
}

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// store o^T (d = 2*row + {0,1}, column = token on the lane) as token-major rows of 64 floats
__device__ __forceinline__ void store_rows64(const f32x16& o0, const f32x16& o1, float* row_ptr, int half) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        f32x4 lo, hi;
        lo[0] = o0[4 * g + 0]; lo[1] = o1[4 * g + 0]; lo[2] = o0[4 * g + 1]; lo[3] = o1[4 * g + 1];
        hi[0] = o0[4 * g + 2]; hi[1] = o1[4 * g + 2]; hi[2] = o0[4 * g + 3]; hi[3] = o1[4 * g + 3];
        *reinterpret_cast<f32x4*>(row_ptr + 16 * g + 8 * half) = lo;
        *reinterpret_cast<f32x4*>(row_ptr + 16 * g + 8 * half + 4) = hi;
    }
}

template <bool RELKEY, bool DROP>
__global__ __launch_bounds__(256) void attn_bwd_dq_split_kernel(
    const float* __restrict__ q, int64_t q_bs, int64_t q_rs, const float* __restrict__ k, int64_t k_bs, int64_t k_rs,
    const float* __restrict__ v, int64_t v_bs, int64_t v_rs, const float* __restrict__ dist_emb, int P,
    const float* __restrict__ key_mask, const float* __restrict__ dout, const float* __restrict__ outp,
    const float* __restrict__ lse, float* __restrict__ dq, int64_t dq_bs, int64_t dq_rs, float* __restrict__ Pm,
    float* __restrict__ dSm, float* __restrict__ dE_part, int nh, int Lq, int Lk, int q_tiles, int n_units,
    E3dDrop drop_in) {
    const E3dDrop drop = e3d_drop_resolve(drop_in);   // + the device-side epoch (graph replays: e3d_common.h)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int qi = lane & 31, half = lane >> 5;
    const int unit = xcd_remap(blockIdx.x, gridDim.x) * (blockDim.x >> 6) + wid;
    if (unit >= n_units) return;
    const int qt = unit % q_tiles, bh = unit / q_tiles, h = bh % nh, b = bh / nh;
    float* ring = smem + wid * WAVE_LDS_F;
    float* X = ring + RING_F;

    const int q0 = qt * 32;
    const int lq = min(q0 + qi, Lq - 1);
    const bool q_ok = q0 + qi < Lq;
    const int HD = nh * D;
    Frag qf[4], dof[4];
    load_row_split(qf, q + b * q_bs + (int64_t)lq * q_rs + h * D, half);
    float delta;
    {
        float dox[4][8], ox[4][8];
        load_row(dox, dout + ((int64_t)b * Lq + lq) * HD + h * D, half);
        load_row(ox, outp + ((int64_t)b * Lq + lq) * HD + h * D, half);
        float part = 0.f;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
            for (int j = 0; j < 8; ++j) part = fmaf(ox[kb][j], dox[kb][j], part);
            dof[kb] = split8(dox[kb]);
        }
        delta = part + __shfl_xor(part, 32, 64);
    }
    const float lse_l = lse[((int64_t)b * nh + h) * Lq + lq];

    const float* kb_ = k + b * k_bs + h * D;
    const float* vb = v + b * v_bs + h * D;
    const float* qb = q + b * q_bs + h * D;
    const float* mb = key_mask ? key_mask + (int64_t)b * Lk : nullptr;
    float* Pbh = Pm + (((int64_t)b * nh + h) * Lq) * Lk;
    float* dSbh = dSm + (((int64_t)b * nh + h) * Lq) * Lk;

    f32x16 dq0, dq1, elo0, elo1, ehi0, ehi1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { dq0[r] = 0.f; dq1[r] = 0.f; elo0[r] = 0.f; elo1[r] = 0.f; ehi0[r] = 0.f; ehi1[r] = 0.f; }

    const int k_tiles = (Lk + 31) >> 5;
    float* part_base = RELKEY ? dE_part + (int64_t)unit * (k_tiles + 1) * 32 * D : nullptr;
    int rot = 0;
    if (RELKEY) {
        const int e = min(max(q0 + 1 + P - 1 + qi, 0), 2 * P - 2);
        Frag ef[4];
        load_row_split(ef, dist_emb + (int64_t)e * D, half);
        const f32x16 t = dot_tile(ef, qf);
#pragma unroll
        for (int r = 0; r < 16; ++r) ring[(32 + mfma32_row(r, half)) * RING_LD + qi] = t[r];
    }

    for (int kt = 0; kt < k_tiles; ++kt) {
        const int r0 = kt * 32;
        const int e_lo = q0 - r0 - 31 + P - 1;
        f32x16 s;
        {
            Frag kf[4];
            load_row_split(kf, kb_ + (int64_t)min(r0 + qi, Lk - 1) * k_rs, half);
            s = dot_tile(kf, qf);
        }
        if (RELKEY) {
            const int e = min(max(e_lo + qi, 0), 2 * P - 2);
            Frag ef[4];
            load_row_split(ef, dist_emb + (int64_t)e * D, half);
            const f32x16 t = dot_tile(ef, qf);
#pragma unroll
            for (int r = 0; r < 16; ++r) ring[((mfma32_row(r, half) + rot) & 63) * RING_LD + qi] = t[r];
            wave_lds_sync();
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int x = qi - mfma32_row(r, half) + 31;
                s[r] += ring[((x + rot) & 63) * RING_LD + qi];
            }
            __builtin_amdgcn_wave_barrier();
            rot ^= 32;
        }
        // probabilities (exact: the forward's log-sum-exp is given)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = r0 + mfma32_row(r, half);
            float pr = 0.f;
            if (key < Lk && q_ok) {
                const float bias = mb ? (1.0f - mb[key]) * -10000.0f : 0.f;
                pr = expf(s[r] * 0.125f + bias - lse_l);
            }
            s[r] = pr;
        }
        // dP^T = V dO^T ; dS^T = P^T (dP^T - delta) / 8
        f32x16 ds;
        {
            Frag vf[4];
            load_row_split(vf, vb + (int64_t)min(r0 + qi, Lk - 1) * v_rs, half);
            ds = dot_tile(vf, dof);
        }
        if (DROP) {
            // forward: O = (P o m) V with m in {0, 1/(1-p)}  =>  dP = (V dO^T) o m, dS = P (dP - delta) / 8 with
            // delta = rowsum(dO o O) unchanged; launch B needs P o m (dV = (P o m)^T dO), materialised below
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float m[4];
                e3d_drop_mult4(drop, e3d_attn_drop_idx4(bh, Lq, Lk, q0 + qi, r0 + 8 * g + 4 * half), m);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    ds[4 * g + j] = s[4 * g + j] * (ds[4 * g + j] * m[j] - delta) * 0.125f;
                    s[4 * g + j] *= m[j];
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) ds[r] = s[r] * (ds[r] - delta) * 0.125f;
        }

        // materialise P and dS (query-major) through the LDS transpose buffer
        wave_lds_sync();  // previous tile's readers of X are done
#pragma unroll
        for (int r = 0; r < 16; ++r) X[mfma32_row(r, half) * X_LD + qi] = s[r];
        wave_lds_sync();
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int ll = 2 * i + half;
            const float val = X[qi * X_LD + ll];  // lane = key column qi
            if (q0 + ll < Lq && r0 + qi < Lk) Pbh[(int64_t)(q0 + ll) * Lk + r0 + qi] = val;
        }
        wave_lds_sync();
#pragma unroll
        for (int r = 0; r < 16; ++r) X[mfma32_row(r, half) * X_LD + qi] = ds[r];
        wave_lds_sync();
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int ll = 2 * i + half;
            const float val = X[qi * X_LD + ll];
            if (q0 + ll < Lq && r0 + qi < Lk) dSbh[(int64_t)(q0 + ll) * Lk + r0 + qi] = val;
        }

        // dQ^T += K^T dS^T
        acc_times_rows(dq0, dq1, kb_, k_rs, r0, 0, Lk - 1, ds, qi, half);

        if (RELKEY) {
            // inverse skew: dT^T[x][l] = dS^T[l - x + 31][l] for window offset x in [0,63]
            f32x16 dt_lo, dt_hi;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int x = mfma32_row(r, half);
                const int rr_lo = qi - x + 31, rr_hi = qi - x - 1;
                dt_lo[r] = (rr_lo >= 0 && rr_lo < 32) ? X[rr_lo * X_LD + qi] : 0.f;
                dt_hi[r] = (rr_hi >= 0 && rr_hi < 32) ? X[rr_hi * X_LD + qi] : 0.f;
            }
            // dQ^T += E^T dT^T (both live 32-row blocks of E)
            acc_times_rows(dq0, dq1, dist_emb, D, e_lo, 0, 2 * P - 2, dt_lo, qi, half);
            acc_times_rows(dq0, dq1, dist_emb, D, e_lo + 32, 0, 2 * P - 2, dt_hi, qi, half);
            // dE blocks += dT^T Q : A = dT^T (row = window offset x = qi, k = query ll), B = Q[ll][d] (d = qi, 32 + qi)
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                float a_lo[8], a_hi[8], b0[8], b1[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int ll = 16 * st + 8 * half + j;  // k slot -> local query index
                    const int rr_lo = ll - qi + 31, rr_hi = ll - qi - 1;
                    a_lo[j] = (rr_lo >= 0 && rr_lo < 32) ? X[rr_lo * X_LD + ll] : 0.f;
                    a_hi[j] = (rr_hi >= 0 && rr_hi < 32) ? X[rr_hi * X_LD + ll] : 0.f;
                    const float* qrow = qb + (int64_t)min(q0 + ll, Lq - 1) * q_rs;
                    b0[j] = qrow[qi];
                    b1[j] = qrow[32 + qi];
                }
                const Frag fa_lo = split8(a_lo), fa_hi = split8(a_hi), fb0 = split8(b0), fb1 = split8(b1);
                elo0 = mfma3(fa_lo, fb0, elo0);
                elo1 = mfma3(fa_lo, fb1, elo1);
                ehi0 = mfma3(fa_hi, fb0, ehi0);
                ehi1 = mfma3(fa_hi, fb1, ehi1);
            }
            // the upper block is complete: block kt covers E rows q0 - 32 kt + P .. + 31
            float* blk = part_base + (int64_t)kt * 32 * D;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                blk[mfma32_row(r, half) * D + qi] = ehi0[r];
                blk[mfma32_row(r, half) * D + 32 + qi] = ehi1[r];
            }
            ehi0 = elo0; ehi1 = elo1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { elo0[r] = 0.f; elo1[r] = 0.f; }
        }
    }
    if (RELKEY) {
        float* blk = part_base + (int64_t)k_tiles * 32 * D;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            blk[mfma32_row(r, half) * D + qi] = ehi0[r];
            blk[mfma32_row(r, half) * D + 32 + qi] = ehi1[r];
        }
    }
    if (q_ok) store_rows64(dq0, dq1, dq + b * dq_bs + (int64_t)(q0 + qi) * dq_rs + h * D, half);
}

__global__ __launch_bounds__(256) void attn_bwd_dkv_split_kernel(
    const float* __restrict__ q, int64_t q_bs, int64_t q_rs, const float* __restrict__ dout,
    const float* __restrict__ Pm, const float* __restrict__ dSm, float* __restrict__ dk, int64_t dk_bs, int64_t dk_rs,
    float* __restrict__ dv, int64_t dv_bs, int64_t dv_rs, int nh, int Lq, int Lk, int k_tiles, int n_units) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int c = lane & 31, half = lane >> 5;
    const int unit = xcd_remap(blockIdx.x, gridDim.x) * (blockDim.x >> 6) + wid;
    if (unit >= n_units) return;
    const int kt = unit % k_tiles, bh = unit / k_tiles, h = bh % nh, b = bh / nh;
    const int r0 = kt * 32;
    const bool key_ok = r0 + c < Lk;
    const int HD = nh * D;
    const float* Pbh = Pm + (((int64_t)b * nh + h) * Lq) * Lk + min(r0 + c, Lk - 1);
    const float* dSbh = dSm + (((int64_t)b * nh + h) * Lq) * Lk + min(r0 + c, Lk - 1);
    const float* dob = dout + (int64_t)b * Lq * HD + h * D + 2 * c;
    const float* qb = q + b * q_bs + h * D + 2 * c;

    f32x16 dv0, dv1, dk0, dk1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { dv0[r] = 0.f; dv1[r] = 0.f; dk0[r] = 0.f; dk1[r] = 0.f; }
    const int q_tiles = (Lq + 31) >> 5;
    for (int qt = 0; qt < q_tiles; ++qt) {
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            float pb[8], sb[8], do0[8], do1[8], q0v[8], q1v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int l = qt * 32 + 16 * st + 8 * half + j;   // k slot -> query, natural order
                const bool ok = l < Lq && key_ok;
                const int lc = min(l, Lq - 1);
                const float pv = Pbh[(int64_t)lc * Lk], sv = dSbh[(int64_t)lc * Lk];
                pb[j] = ok ? pv : 0.f;
                sb[j] = ok ? sv : 0.f;
                const float2 a_do = *reinterpret_cast<const float2*>(dob + (int64_t)lc * HD);
                const float2 a_q = *reinterpret_cast<const float2*>(qb + (int64_t)lc * q_rs);
                do0[j] = a_do.x; do1[j] = a_do.y;
                q0v[j] = a_q.x; q1v[j] = a_q.y;
            }
            const Frag fp = split8(pb), fs = split8(sb);
            dv0 = mfma3(split8(do0), fp, dv0);
            dv1 = mfma3(split8(do1), fp, dv1);
            dk0 = mfma3(split8(q0v), fs, dk0);
            dk1 = mfma3(split8(q1v), fs, dk1);
        }
    }
    if (key_ok) {
        store_rows64(dv0, dv1, dv + b * dv_bs + (int64_t)(r0 + c) * dv_rs + h * D, half);
        store_rows64(dk0, dk1, dk + b * dk_bs + (int64_t)(r0 + c) * dk_rs + h * D, half);
    }
}

}  // namespace

// Launches A and B of the backward in bf16x3 arithmetic (arguments validated by e3d_relkey_attn_bwd_ex, which
// also runs launch C afterwards).
int e3d_attn_bwd_split_launch(const float* q, int64_t q_bs, int64_t q_rs, const float* k, int64_t k_bs, int64_t k_rs,
                              const float* v, int64_t v_bs, int64_t v_rs, const float* dist_emb, int P,
                              const float* key_mask, const float* out, const float* lse, const float* dout, float* dq,
                              int64_t dq_bs, int64_t dq_rs, float* dk, int64_t dk_bs, int64_t dk_rs, float* dv,
                              int64_t dv_bs, int64_t dv_rs, float* Pm, float* dSm, float* part, int B, int nh, int Lq,
                              int Lk, E3dDrop drop, bool dropping, hipStream_t s) {
    const int q_tiles = (Lq + 31) / 32, k_tiles = (Lk + 31) / 32;
    const int wpb = 4;
    {
        const int n_units = B * nh * q_tiles;
        const int n_blocks = (n_units + wpb - 1) / wpb;
        const size_t lds = (size_t)wpb * WAVE_LDS_F * sizeof(float);
#define E3D_BWD_DQ(RK, DR)                                                                                          \
    hipLaunchKernelGGL((attn_bwd_dq_split_kernel<RK, DR>), dim3(n_blocks), dim3(64 * wpb), lds, s, q, q_bs, q_rs, k, \
                       k_bs, k_rs, v, v_bs, v_rs, dist_emb, P, key_mask, dout, out, lse, dq, dq_bs, dq_rs, Pm, dSm,  \
                       part, nh, Lq, Lk, q_tiles, n_units, drop)
        if (dist_emb) {
            if (dropping) E3D_BWD_DQ(true, true);
            else E3D_BWD_DQ(true, false);
        } else {
            if (dropping) E3D_BWD_DQ(false, true);
            else E3D_BWD_DQ(false, false);
        }
#undef E3D_BWD_DQ
        int rc = e3d_launch_status("e3d_relkey_attn_bwd (dq, bf16x3)");
        if (rc) return rc;
    }
    {
        const int n_units = B * nh * k_tiles;
        hipLaunchKernelGGL(attn_bwd_dkv_split_kernel, dim3((n_units + wpb - 1) / wpb), dim3(64 * wpb), 0, s, q, q_bs, q_rs,
                           dout, Pm, dSm, dk, dk_bs, dk_rs, dv, dv_bs, dv_rs, nh, Lq, Lk, k_tiles, n_units);
        return e3d_launch_status("e3d_relkey_attn_bwd (dkv, bf16x3)");
    }
}
