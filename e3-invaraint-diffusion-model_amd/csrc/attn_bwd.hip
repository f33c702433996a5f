// Backward of the fused relative-key attention (training configs), fp32 MFMA, head dim 64.
//
// Two launches + one small reduction, all deterministic except the final dist_emb atomics:
//   A. attn_bwd_dq_kernel   one wavefront per (b, head, 32-query tile), same transposed layout as the
//      forward (query on the lane): recomputes P^T = exp(S^T - lse) incl. the relative-key term,
//      dP^T = V dO^T, dS^T = P^T (dP^T - delta) / sqrt(d); accumulates dQ^T (+= K^T dS^T and, through the
//      inverse skew, += E^T dT^T); accumulates the dist_emb gradient blocks dE = dT^T Q in registers
//      (each 32-row block of E is touched by exactly two consecutive key tiles) and writes them ONCE
//      per wave to a partial buffer; materialises P and dS [B,nh,Lq,Lk] (query-major, key
//      contiguous) for launch B through an LDS transpose.
//   B. attn_bwd_dkv_kernel  one wavefront per (b, head, 32-key tile), key on the lane:
//      dV^T += dO^T P, dK^T += Q^T dS streaming the materialised tiles (coalesced 128-byte rows).
//   C. dist_emb_reduce_kernel  sums the per-wave dE partial blocks into dE[2P-1, 64].
#include "e3d_common.h"

namespace {

constexpr int D = 64;
constexpr int RING_LD = 34, RING_F = 64 * RING_LD;  // T^T ring (forward recompute)
constexpr int X_LD = 33, X_F = 32 * X_LD;           // 32x32 tile transpose / inverse-skew buffer
constexpr int WAVE_LDS_F = RING_F + X_F;

__device__ __forceinline__ void load_frag8(f32x4 (&f)[8], const float* row_ptr, int half) {
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = *reinterpret_cast<const f32x4*>(row_ptr + 8 * j + 4 * half);
}

// tile[i][j] = X_i . Y_j  with i on the accumulator rows and j on the lanes (lane supplies row lane&31 of both)
__device__ __forceinline__ f32x16 mfma_tile(const f32x4 (&a)[8], const f32x4 (&b)[8]) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j][i], b[j][i], acc, 0, 0, 0);
    return acc;
}

// o^T[d, j] += sum_i Z[row_i][d] * w[i][j]:  w is an accumulator-layout tile (rows i in registers,
// columns j on lanes); lane (c, half) feeds Z[row(st,half)][2c .. 2c+1].  rows clamped to row_max.
__device__ __forceinline__ void acc_times_rows(f32x16& o0, f32x16& o1, const float* z_base, int64_t z_rs, int row0,
                                               int row_min, int row_max, const f32x16& w, int c, int half) {
#pragma unroll
    for (int st = 0; st < 16; ++st) {
        const int row = min(max(row0 + mfma32_row(st, half), row_min), row_max);
        const float2 zz = *reinterpret_cast<const float2*>(z_base + (int64_t)row * z_rs + 2 * c);
        o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(zz.x, w[st], o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(zz.y, w[st], o1, 0, 0, 0);
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
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(
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
    f32x4 qf[8], dof[8];
    load_frag8(qf, q + b * q_bs + (int64_t)lq * q_rs + h * D, half);
    load_frag8(dof, dout + ((int64_t)b * Lq + lq) * HD + h * D, half);
    float delta;
    {
        f32x4 of[8];
        load_frag8(of, outp + ((int64_t)b * Lq + lq) * HD + h * D, half);
        float part = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) part = fmaf(of[j][i], dof[j][i], part);
        delta = part + __shfl_xor(part, 32, 64);
    }
    const float lse_l = lse[((int64_t)b * nh + h) * Lq + lq];

    const float* kb = k + b * k_bs + h * D;
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
        f32x4 ef[8];
        load_frag8(ef, dist_emb + (int64_t)e * D, half);
        const f32x16 t = mfma_tile(ef, qf);
#pragma unroll
        for (int r = 0; r < 16; ++r) ring[(32 + mfma32_row(r, half)) * RING_LD + qi] = t[r];
    }

    for (int kt = 0; kt < k_tiles; ++kt) {
        const int r0 = kt * 32;
        const int e_lo = q0 - r0 - 31 + P - 1;
        f32x16 s;
        {
            f32x4 kf[8];
            load_frag8(kf, kb + (int64_t)min(r0 + qi, Lk - 1) * k_rs, half);
            s = mfma_tile(kf, qf);
        }
        if (RELKEY) {
            const int e = min(max(e_lo + qi, 0), 2 * P - 2);
            f32x4 ef[8];
            load_frag8(ef, dist_emb + (int64_t)e * D, half);
            const f32x16 t = mfma_tile(ef, qf);
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
            f32x4 vf[8];
            load_frag8(vf, vb + (int64_t)min(r0 + qi, Lk - 1) * v_rs, half);
            ds = mfma_tile(vf, dof);
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
        acc_times_rows(dq0, dq1, kb, k_rs, r0, 0, Lk - 1, ds, qi, half);

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
            // dE blocks += dT^T Q : A = dT^T (row = window offset on lane&31, k = query), B = Q rows
#pragma unroll
            for (int st = 0; st < 16; ++st) {
                const int ll = 2 * st + half;  // k slot -> local query index
                const int rr_lo = ll - qi + 31, rr_hi = ll - qi - 1;
                const float a_lo = (rr_lo >= 0 && rr_lo < 32) ? X[rr_lo * X_LD + ll] : 0.f;
                const float a_hi = (rr_hi >= 0 && rr_hi < 32) ? X[rr_hi * X_LD + ll] : 0.f;
                const float* qrow = qb + (int64_t)min(q0 + ll, Lq - 1) * q_rs;
                const float b0 = qrow[qi], b1 = qrow[32 + qi];
                elo0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_lo, b0, elo0, 0, 0, 0);
                elo1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_lo, b1, elo1, 0, 0, 0);
                ehi0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_hi, b0, ehi0, 0, 0, 0);
                ehi1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_hi, b1, ehi1, 0, 0, 0);
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

__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(
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
    const float* Pbh = Pm + (((int64_t)b * nh + h) * Lq) * Lk + r0 + c;
    const float* dSbh = dSm + (((int64_t)b * nh + h) * Lq) * Lk + r0 + c;
    const float* dob = dout + (int64_t)b * Lq * HD + h * D + 2 * c;
    const float* qb = q + b * q_bs + h * D + 2 * c;

    f32x16 dv0, dv1, dk0, dk1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { dv0[r] = 0.f; dv1[r] = 0.f; dk0[r] = 0.f; dk1[r] = 0.f; }
    const int q_tiles = (Lq + 31) >> 5;
    for (int qt = 0; qt < q_tiles; ++qt) {
#pragma unroll
        for (int st = 0; st < 16; ++st) {
            const int l = qt * 32 + mfma32_row(st, half);
            const bool ok = l < Lq && key_ok;
            const int lc = min(l, Lq - 1);
            const float pb = ok ? Pbh[(int64_t)lc * Lk] : 0.f;
            const float sb = ok ? dSbh[(int64_t)lc * Lk] : 0.f;
            const float2 a_do = *reinterpret_cast<const float2*>(dob + (int64_t)lc * HD);
            const float2 a_q = *reinterpret_cast<const float2*>(qb + (int64_t)lc * q_rs);
            dv0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_do.x, pb, dv0, 0, 0, 0);
            dv1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_do.y, pb, dv1, 0, 0, 0);
            dk0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_q.x, sb, dk0, 0, 0, 0);
            dk1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_q.y, sb, dk1, 0, 0, 0);
        }
    }
    if (key_ok) {
        store_rows64(dv0, dv1, dv + b * dv_bs + (int64_t)(r0 + c) * dv_rs + h * D, half);
        store_rows64(dk0, dk1, dk + b * dk_bs + (int64_t)(r0 + c) * dk_rs + h * D, half);
    }
}

// dE[e, d] += sum over this block's units of the partial block row that covers E row e
__global__ __launch_bounds__(64) void dist_emb_reduce_kernel(const float* __restrict__ part, float* __restrict__ dE, int P,
                                                             int q_tiles, int k_tiles, int n_units, int units_per_block) {
    const int e = blockIdx.x, d = threadIdx.x;
    const int u0 = blockIdx.y * units_per_block, u1 = min(n_units, u0 + units_per_block);
    float acc = 0.f;
    // 8 units per round, their loads issued together (a unit whose window does not cover row e loads nothing); summed in
    // unit order.  (One dependent load per unit, as this loop was first written, took 40 us at B=32, L=128.)
    for (int u = u0; u < u1; u += 8) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int uu = u + i;
            const int q0 = (uu % q_tiles) * 32;
            const int t = q0 + P - e + 31;
            const int j = t >> 5;
            const int row = e - (q0 - 32 * j + P);
            v[i] = (uu < u1 && t >= 0 && j <= k_tiles) ? part[(((int64_t)uu * (k_tiles + 1) + j) * 32 + row) * D + d] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) acc += v[i];
    }
    atomicAdd(dE + (int64_t)e * D + d, acc);
}

}  // namespace

extern "C" int64_t e3d_relkey_attn_bwd_workspace_floats(int B, int nh, int Lq, int Lk, int relkey) {
    const int64_t q_tiles = (Lq + 31) / 32, k_tiles = (Lk + 31) / 32;
    const int64_t pm = (int64_t)B * nh * Lq * Lk;
    const int64_t part = relkey ? (int64_t)B * nh * q_tiles * (k_tiles + 1) * 32 * D : 0;
    // [P | dS] of the two-launch kernels; the fused kernel parks the distance-table planes in the same region
    const int64_t head = relkey ? (e3d_attn_bwd_coop_scratch_bytes(Lk) + 3) / 4 : 0;
    // (+ one liveness word per (b, head, query tile) of the fused kernel: after the chunk sums)
    return (2 * pm > head ? 2 * pm : head) + part + (relkey ? e3d_attn_bwd_coop_de_floats(Lq, Lk) + (int64_t)B * nh * q_tiles : 0);
}

extern "C" int e3d_relkey_attn_bwd_ex(const float* q, int64_t q_bs, int64_t q_rs, const float* k, int64_t k_bs,
                                      int64_t k_rs, const float* v, int64_t v_bs, int64_t v_rs, const float* dist_emb,
                                      int P, const float* key_mask, const float* out, const float* lse,
                                      const float* dout, float* dq, int64_t dq_bs, int64_t dq_rs, float* dk,
                                      int64_t dk_bs, int64_t dk_rs, float* dv, int64_t dv_bs, int64_t dv_rs,
                                      float* d_dist_emb, float* workspace, int B, int nh, int Lq, int Lk, int terms,
                                      float drop_p, uint64_t drop_seed, void* stream) {
    E3D_REQUIRE(terms == 0 || terms == 3 || terms == 6, "attn_bwd: terms must be 0 (fp32 MFMA), 3 or 6 (got %d)", terms);
    E3D_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "attn_bwd: drop_p=%g outside [0, 1)", (double)drop_p);
    const E3dDrop drop = e3d_drop_make(drop_p, drop_seed);
    E3D_REQUIRE(q && k && v && out && lse && dout && dq && dk && dv && workspace, "attn_bwd: null pointer");
    E3D_REQUIRE(B > 0 && nh > 0 && Lq > 0 && Lk > 0, "attn_bwd: bad shape");
    E3D_REQUIRE(q_rs % 4 == 0 && k_rs % 4 == 0 && v_rs % 4 == 0 && dq_rs % 4 == 0 && dk_rs % 4 == 0 && dv_rs % 4 == 0 &&
                    q_bs % 4 == 0 && k_bs % 4 == 0 && v_bs % 4 == 0 && dq_bs % 4 == 0 && dk_bs % 4 == 0 && dv_bs % 4 == 0,
                "attn_bwd: strides must keep 16B alignment");
    if (dist_emb) {
        E3D_REQUIRE(Lq == Lk && Lq <= P && d_dist_emb, "attn_bwd: relative_key needs Lq == Lk <= P and d_dist_emb");
    }
    hipStream_t s = (hipStream_t)stream;
    const int q_tiles = (Lq + 31) / 32, k_tiles = (Lk + 31) / 32;
    const int64_t pm = (int64_t)B * nh * Lq * Lk;
    const int64_t head = dist_emb ? (e3d_attn_bwd_coop_scratch_bytes(Lk) + 3) / 4 : 0;
    float* Pm = workspace;
    float* dSm = workspace + pm;
    float* part = workspace + (2 * pm > head ? 2 * pm : head);
    const int wpb = 4;
    if (terms == 3 && e3d_attn_bwd_coop_supported(Lq, Lk, drop_p > 0.f)) {
        // fused recomputing kernel (attn_bwd_coop.hip): one launch, P / dS never reach HBM
        E3D_REQUIRE(((uintptr_t)workspace % 16) == 0, "attn_bwd: workspace must be 16-byte aligned");
        const int rc = e3d_attn_bwd_coop_launch(q, q_bs, q_rs, k, k_bs, k_rs, v, v_bs, v_rs, dist_emb, P, key_mask, out, lse, dout,
                                                dq, dq_bs, dq_rs, dk, dk_bs, dk_rs, dv, dv_bs, dv_rs, d_dist_emb, workspace, part,
                                                B, nh, Lq, Lk, drop, drop_p > 0.f, s);
        return rc;     // (incl. the dE reduction: streaming, deterministic)
    } else if (terms == 3) {   // bf16x3 arithmetic (attn_bwd_split.hip); 0 and 6 keep the fp32 MFMA kernels below
        const int rc = e3d_attn_bwd_split_launch(q, q_bs, q_rs, k, k_bs, k_rs, v, v_bs, v_rs, dist_emb, P, key_mask, out, lse,
                                                 dout, dq, dq_bs, dq_rs, dk, dk_bs, dk_rs, dv, dv_bs, dv_rs, Pm, dSm, part, B,
                                                 nh, Lq, Lk, drop, drop_p > 0.f, s);
        if (rc) return rc;
    } else {
    {
        const int n_units = B * nh * q_tiles;
        const int n_blocks = (n_units + wpb - 1) / wpb;
        const size_t lds = (size_t)wpb * WAVE_LDS_F * sizeof(float);
#define E3D_BWD_DQ(RK, DR)                                                                                              \
    hipLaunchKernelGGL((attn_bwd_dq_kernel<RK, DR>), dim3(n_blocks), dim3(64 * wpb), lds, s, q, q_bs, q_rs, k, k_bs, k_rs, v, \
                       v_bs, v_rs, dist_emb, P, key_mask, dout, out, lse, dq, dq_bs, dq_rs, Pm, dSm, part, nh, Lq, Lk,      \
                       q_tiles, n_units, drop)
        if (dist_emb) {
            if (drop_p > 0.f) E3D_BWD_DQ(true, true);
            else E3D_BWD_DQ(true, false);
        } else {
            if (drop_p > 0.f) E3D_BWD_DQ(false, true);
            else E3D_BWD_DQ(false, false);
        }
#undef E3D_BWD_DQ
        int rc = e3d_launch_status("e3d_relkey_attn_bwd (dq)");
        if (rc) return rc;
    }
    {
        const int n_units = B * nh * k_tiles;
        hipLaunchKernelGGL(attn_bwd_dkv_kernel, dim3((n_units + wpb - 1) / wpb), dim3(64 * wpb), 0, s, q, q_bs, q_rs, dout, Pm,
                           dSm, dk, dk_bs, dk_rs, dv, dv_bs, dv_rs, nh, Lq, Lk, k_tiles, n_units);
        int rc = e3d_launch_status("e3d_relkey_attn_bwd (dkv)");
        if (rc) return rc;
    }
    }
    if (dist_emb) {
        hipError_t e = e3d_zero_async(d_dist_emb, (size_t)(2 * P - 1) * D, s);
        E3D_REQUIRE(e == hipSuccess, "attn_bwd: memset failed: %s", hipGetErrorString(e));
        const int n_units = B * nh * q_tiles, upb = 64;
        hipLaunchKernelGGL(dist_emb_reduce_kernel, dim3(2 * P - 1, (n_units + upb - 1) / upb), dim3(64), 0, s, part,
                           d_dist_emb, P, q_tiles, k_tiles, n_units, upb);
        return e3d_launch_status("e3d_relkey_attn_bwd (dE)");
    }
    return 0;
}

extern "C" int e3d_relkey_attn_bwd(const float* q, int64_t q_bs, int64_t q_rs, const float* k, int64_t k_bs,
                                   int64_t k_rs, const float* v, int64_t v_bs, int64_t v_rs, const float* dist_emb, int P,
                                   const float* key_mask, const float* out, const float* lse, const float* dout,
                                   float* dq, int64_t dq_bs, int64_t dq_rs, float* dk, int64_t dk_bs, int64_t dk_rs,
                                   float* dv, int64_t dv_bs, int64_t dv_rs, float* d_dist_emb, float* workspace, int B,
                                   int nh, int Lq, int Lk, void* stream) {
    return e3d_relkey_attn_bwd_ex(q, q_bs, q_rs, k, k_bs, k_rs, v, v_bs, v_rs, dist_emb, P, key_mask, out, lse, dout, dq,
                                  dq_bs, dq_rs, dk, dk_bs, dk_rs, dv, dv_bs, dv_rs, d_dist_emb, workspace, B, nh, Lq, Lk, 0,
                                  0.f, 0, stream);
}

extern "C" int e3d_relkey_attn_bwd_drop(const float* q, int64_t q_bs, int64_t q_rs, const float* k, int64_t k_bs,
                                        int64_t k_rs, const float* v, int64_t v_bs, int64_t v_rs, const float* dist_emb,
                                        int P, const float* key_mask, const float* out, const float* lse,
                                        const float* dout, float* dq, int64_t dq_bs, int64_t dq_rs, float* dk,
                                        int64_t dk_bs, int64_t dk_rs, float* dv, int64_t dv_bs, int64_t dv_rs,
                                        float* d_dist_emb, float* workspace, int B, int nh, int Lq, int Lk, float drop_p,
                                        uint64_t drop_seed, void* stream) {
    return e3d_relkey_attn_bwd_ex(q, q_bs, q_rs, k, k_bs, k_rs, v, v_bs, v_rs, dist_emb, P, key_mask, out, lse, dout, dq,
                                  dq_bs, dq_rs, dk, dk_bs, dk_rs, dv, dv_bs, dv_rs, d_dist_emb, workspace, B, nh, Lq, Lk, 0,
                                  drop_p, drop_seed, stream);
}
