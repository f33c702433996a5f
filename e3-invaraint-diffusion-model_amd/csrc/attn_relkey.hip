// Fused relative-key attention forward, head dim 64, fp32 on the gfx950 matrix cores.
//
// One wavefront owns a 32-query tile of one (batch, head) and sweeps the keys in tiles of 32
// with an online softmax.  Everything is computed TRANSPOSED so that the query sits on the MFMA
// lane (column) and the key on the accumulator rows:
//     S^T[r,l]  = K[r,:] . Q[l,:]            A = K tile,       B = Q^T   (32 MFMA 32x32x2)
//     T^T[e,l]  = E[e,:] . Q[l,:]            A = dist_emb rows, B = Q^T   (32 MFMA, rel-key only)
//     O^T[d,l] += V^T[d,r] P^T[r,l]          A = V^T tile,     B = P^T   (32 MFMA)
// With the query on the lane, the row max / row sum of the softmax are per-lane register
// reductions plus one cross-half shuffle, the P accumulator is already the B operand of the PV
// product (no LDS, no lane movement), and the O rescale is a per-lane scalar.
//
// Operand traffic: K, E and V fragments are read straight from global memory (L2-resident: one
// (b,h) is 2 x 64 KB at L=256; dist_emb is shared by every workgroup) in the exact lane layout
// the MFMA wants -- 16 B/lane for K and E, 8 B/lane for V -- so no LDS staging and no barriers.
//
// relative_key skew: score (l,r) needs T[l - r + P - 1][l].  For the key tile at r0 the wave needs
// E rows e_lo .. e_lo+62, e_lo = q0 - r0 - 31 + P - 1; consecutive key tiles shift that window by
// -32, so each step computes ONE new 32-row block of T^T and re-uses the previous one.  The two
// live blocks sit in a per-wave 64x34-float LDS ring; lane (query qi, row rr) reads ring row
// qi - rr + 31 (stride 34 keeps the diagonal read conflict-free).
#include "e3d_common.h"

namespace {

constexpr int D = 64;          // head dim
constexpr int RING_LD = 34;    // floats per ring row
constexpr int RING_F = 64 * RING_LD;

__device__ __forceinline__ void load_frag8(f32x4 (&f)[8], const float* row_ptr, int half) {
    // lane (row, half) takes k = 8j + 4*half + i  (j = 0..7, i = 0..3): 8 x 16 B
#pragma unroll
    for (int j = 0; j < 8; ++j)
        f[j] = *reinterpret_cast<const f32x4*>(row_ptr + 8 * j + 4 * half);
}

__device__ __forceinline__ f32x16 mfma_tile(const f32x4 (&a)[8], const f32x4 (&b)[8]) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j][i], b[j][i], acc, 0, 0, 0);
    return acc;
}

template <bool RELKEY>
__global__ __launch_bounds__(256) void attn_fwd_kernel(
    const float* __restrict__ q, int64_t q_bs, int64_t q_rs, const float* __restrict__ k,
    int64_t k_bs, int64_t k_rs, const float* __restrict__ v, int64_t v_bs, int64_t v_rs,
    const float* __restrict__ dist_emb, int P, const float* __restrict__ key_mask,
    float* __restrict__ out, float* __restrict__ lse, int nh, int Lq, int Lk, int q_tiles,
    int n_units) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int qi = lane & 31, half = lane >> 5;
    // unit = (b, head, query tile); consecutive units share K/V, keep them on one XCD
    const int waves_per_block = blockDim.x >> 6;
    const int n_blocks = gridDim.x;
    const int unit = xcd_remap(blockIdx.x, n_blocks) * waves_per_block + wid;
    if (unit >= n_units) return;  // whole wave exits; no barriers in this kernel
    const int qt = unit % q_tiles;
    const int bh = unit / q_tiles;
    const int h = bh % nh, b = bh / nh;
    float* ring = smem + wid * RING_F;

    const int q0 = qt * 32;
    const int lq = min(q0 + qi, Lq - 1);
    f32x4 qf[8];
    load_frag8(qf, q + b * q_bs + (int64_t)lq * q_rs + h * D, half);

    const float* kb = k + b * k_bs + h * D;
    const float* vb = v + b * v_bs + h * D;
    const float* mb = key_mask ? key_mask + (int64_t)b * Lk : nullptr;

    f32x16 o0, o1;  // O^T rows d = 2*row + {0,1}
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;

    int rot = 0;  // ring row of window offset x is (x + rot) & 63
    if (RELKEY) {
        // prime the ring with the block that the first key tile uses as its UPPER half:
        // rows e_lo(0)+32 .. e_lo(0)+63, e_lo(0) = q0 - 31 + P - 1
        const int e = min(max(q0 + 1 + P - 1 + qi, 0), 2 * P - 2);
        f32x4 ef[8];
        load_frag8(ef, dist_emb + (int64_t)e * D, half);
        const f32x16 t = mfma_tile(ef, qf);
#pragma unroll
        for (int r = 0; r < 16; ++r)
            ring[(32 + mfma32_row(r, half)) * RING_LD + qi] = t[r];
    }

    const int k_tiles = (Lk + 31) >> 5;
    for (int kt = 0; kt < k_tiles; ++kt) {
        const int r0 = kt * 32;
        // ---- S^T = K Q^T
        f32x4 kf[8];
        load_frag8(kf, kb + (int64_t)min(r0 + qi, Lk - 1) * k_rs, half);
        f32x16 s = mfma_tile(kf, qf);

        if (RELKEY) {
            // new LOWER block: rows e_lo .. e_lo+31 -> ring rows (0 + rot) .. (31 + rot)
            const int e_lo = q0 - r0 - 31 + P - 1;
            const int e = min(max(e_lo + qi, 0), 2 * P - 2);
            f32x4 ef[8];
            load_frag8(ef, dist_emb + (int64_t)e * D, half);
            const f32x16 t = mfma_tile(ef, qf);
#pragma unroll
            for (int r = 0; r < 16; ++r)
                ring[((mfma32_row(r, half) + rot) & 63) * RING_LD + qi] = t[r];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int x = qi - mfma32_row(r, half) + 31;  // 0..62
                s[r] += ring[((x + rot) & 63) * RING_LD + qi];
            }
            __builtin_amdgcn_wave_barrier();
            rot ^= 32;  // this tile's lower block is the next tile's upper block
        }

        // ---- scale, mask, online softmax (query on the lane; rows = keys)
        float tmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = r0 + mfma32_row(r, half);
            float bias;
            if (key < Lk)
                bias = mb ? (1.0f - mb[key]) * -10000.0f : 0.f;
            else
                bias = -INFINITY;
            s[r] = s[r] * 0.125f + bias;
            tmax = fmaxf(tmax, s[r]);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m_run, tmax);
        const float alpha = expf(m_run - m_new);  // first tile: exp(-inf) = 0
        m_run = m_new;
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[r] = expf(s[r] - m_new);
            psum += s[r];
        }
        l_run = l_run * alpha + psum;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }

        // ---- O^T += V^T P^T : lane (di, half) feeds V[r0 + row(s,half)][2*di + {0,1}]
#pragma unroll
        for (int st = 0; st < 16; ++st) {
            const int key = min(r0 + mfma32_row(st, half), Lk - 1);
            const float2 vv = *reinterpret_cast<const float2*>(vb + (int64_t)key * v_rs + 2 * qi);
            o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(vv.x, s[st], o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(vv.y, s[st], o1, 0, 0, 0);
        }
    }

    // ---- finish: both halves hold partial row sums of the same query
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (q0 + qi < Lq) {
        float* orow = out + ((int64_t)b * Lq + q0 + qi) * (nh * D) + h * D;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            // regs 4g..4g+3 -> rows 8g + 4*half + {0..3} -> d = 16g + 8*half + {0..7}
            f32x4 lo, hi;
            lo[0] = o0[4 * g + 0] * inv; lo[1] = o1[4 * g + 0] * inv;
            lo[2] = o0[4 * g + 1] * inv; lo[3] = o1[4 * g + 1] * inv;
            hi[0] = o0[4 * g + 2] * inv; hi[1] = o1[4 * g + 2] * inv;
            hi[2] = o0[4 * g + 3] * inv; hi[3] = o1[4 * g + 3] * inv;
            *reinterpret_cast<f32x4*>(orow + 16 * g + 8 * half) = lo;
            *reinterpret_cast<f32x4*>(orow + 16 * g + 8 * half + 4) = hi;
        }
        if (lse && half == 0) lse[((int64_t)b * nh + h) * Lq + q0 + qi] = m_run + logf(l_tot);
    }
}

}  // namespace

extern "C" int e3d_relkey_attn_fwd(const float* q, int64_t q_bs, int64_t q_rs, const float* k,
                                   int64_t k_bs, int64_t k_rs, const float* v, int64_t v_bs,
                                   int64_t v_rs, const float* dist_emb, int P,
                                   const float* key_mask, float* out, float* lse, int B, int nh,
                                   int Lq, int Lk, void* stream) {
    E3D_REQUIRE(q && k && v && out, "attn: null pointer");
    E3D_REQUIRE(B > 0 && nh > 0 && Lq > 0 && Lk > 0, "attn: bad shape B=%d nh=%d Lq=%d Lk=%d", B, nh, Lq, Lk);
    E3D_REQUIRE(q_rs % 4 == 0 && k_rs % 4 == 0 && v_rs % 2 == 0 && q_bs % 4 == 0 && k_bs % 4 == 0 && v_bs % 2 == 0,
                "attn: strides must keep 16B (q,k) / 8B (v) alignment");
    E3D_REQUIRE(((uintptr_t)q % 16) == 0 && ((uintptr_t)k % 16) == 0 && ((uintptr_t)v % 16) == 0 &&
                    ((uintptr_t)out % 16) == 0, "attn: pointers must be 16B aligned");
    if (dist_emb) {
        // the distance_embedding lookup l - r + P - 1 must stay inside [0, 2P-2]
        E3D_REQUIRE(Lq == Lk && Lq <= P, "attn: relative_key needs Lq == Lk <= P (Lq=%d Lk=%d P=%d)", Lq, Lk, P);
        E3D_REQUIRE(((uintptr_t)dist_emb % 16) == 0, "attn: dist_emb must be 16B aligned");
    }
    const int q_tiles = (Lq + 31) / 32;
    const int64_t n_units64 = (int64_t)B * nh * q_tiles;
    E3D_REQUIRE(n_units64 < (1ll << 30), "attn: too many tiles");
    const int n_units = (int)n_units64;
    const int wpb = 4;
    const int n_blocks = (n_units + wpb - 1) / wpb;
    const size_t lds = dist_emb ? (size_t)wpb * RING_F * sizeof(float) : 0;
    hipStream_t s = (hipStream_t)stream;
    if (dist_emb)
        hipLaunchKernelGGL(attn_fwd_kernel<true>, dim3(n_blocks), dim3(64 * wpb), lds, s, q, q_bs, q_rs, k, k_bs,
                           k_rs, v, v_bs, v_rs, dist_emb, P, key_mask, out, lse, nh, Lq, Lk, q_tiles, n_units);
    else
        hipLaunchKernelGGL(attn_fwd_kernel<false>, dim3(n_blocks), dim3(64 * wpb), lds, s, q, q_bs, q_rs, k, k_bs,
                           k_rs, v, v_bs, v_rs, dist_emb, P, key_mask, out, lse, nh, Lq, Lk, q_tiles, n_units);
    return e3d_launch_status("e3d_relkey_attn_fwd");
}
