// Fused relative-key attention forward on the bf16 matrix cores with SPLIT fp32 operands and fp32
// accumulation -- same algorithm, layouts and LDS ring as attn_relkey.hip (query on the MFMA lane,
// key on the accumulator rows, online softmax in fp32 registers), but every product
//     S^T = K Q^T,   T^T = E Q^T (rel-key),   O^T += V^T P^T
// runs as 3 (TERMS=3: operands split in 2 bf16 terms) or 6 (TERMS=6: 3 terms, fp32-grade) cross
// products of v_mfma_f32_32x32x16_bf16: 36 / 72 MFMA issue slots of 32 cycles per 32x32 score tile
// instead of 96 slots of 64 cycles for the exact fp32 MFMA kernel.
//
// Operand maps (32x32x16: lane (r = lane&31, h = lane>>5) holds k = 8h + j, j = 0..7):
//   K / E / Q rows: 8 consecutive head-dim floats (2 x 16 B loads) per 16-wide k block;
//   P^T as the B operand of PV: accumulator registers 8s..8s+7 of the score tile, i.e. key row
//     rho(s,h,j) = (j&3) + 8(2s + (j>>2)) + 4h, so V^T is gathered with the same permutation
//     (8 x 8-byte loads of V[r0 + rho][2c .. 2c+1] per step; the lane owns head dims 2c, 2c+1).
#include "e3d_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
// element type of the split terms: __bf16 (bf16x3 / bf16x6) or _Float16 (f16x3, NS = 2 only; include/e3d_hip.h)
template <typename E> struct AV;
template <> struct AV<__bf16> { typedef bf16x8 x8; };
template <> struct AV<_Float16> { typedef f16x8 x8; };
template <typename X> struct Elem;
template <> struct Elem<bf16x8> { typedef __bf16 type; };
template <> struct Elem<f16x8> { typedef _Float16 type; };
__device__ __forceinline__ f32x16 mma16(const bf16x8 a, const bf16x8 b, const f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mma16(const f16x8 a, const f16x8 b, const f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

constexpr int D = 64;
constexpr int RING_LD = 34;
constexpr int RING_F = 64 * RING_LD;
constexpr int STG_LD = 68;             // staging rows: 64 floats + 16 B pad (conflict-free b128 fragment reads)
constexpr int STG_F = 32 * STG_LD;
constexpr int WAVE_LDS_F = RING_F + STG_F + 32;  // + per-tile key bias row

template <int NS, typename X8>
__device__ __forceinline__ void split8(const float (&x)[8], X8 (&parts)[NS]) {
    typedef typename Elem<X8>::type E;
    float r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = x[j];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const E p = (E)r[j];
            parts[s][j] = p;
            r[j] -= (float)p;
        }
}

template <int NS>
__device__ __forceinline__ void split8(const float (&x)[8], f16x8 (&parts)[NS]) {
    static_assert(NS == 2, "fp16 terms exist as the two-term split only");
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        e3d_f16x2 h, l;
        e3d_split2_f16(x[j], x[j + 1], h, l);
        parts[0][j] = h[0]; parts[0][j + 1] = h[1];
        parts[1][j] = l[0]; parts[1][j + 1] = l[1];
    }
}

// one row's 64 head-dim values -> 4 k-blocks x NS parts (lane takes floats 16kb + 8h .. +7)
template <int NS, typename X8>
__device__ __forceinline__ void load_row_split(X8 (&f)[4][NS], const float* row_ptr, int half) {
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        const f32x4 lo = *reinterpret_cast<const f32x4*>(row_ptr + 16 * kb + 8 * half);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(row_ptr + 16 * kb + 8 * half + 4);
        const float x[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        split8<NS>(x, f[kb]);
    }
}

// A 32-row x 64-float operand tile (rows row_lo .. row_lo+31 clamped to [row_min,row_max], row stride rs)
// -> MFMA fragments.  Fragment-shaped global loads (16 B per lane from 32 different rows) are
// address-coalescer bound, so the wave loads the tile in full 256-byte rows (4 rows per
// instruction), parks it in its private LDS staging buffer and reads the fragments back.
struct TileRegs { f32x4 v[8]; };

// ``base`` is wave-uniform (scalar registers); lane offsets stay 32-bit (one (b,h) slab is < 2^31 elements)
__device__ __forceinline__ void tile_load(TileRegs& t, const float* base, int rs, int row_lo, int row_min,
                                          int row_max, int lane) {
    if (row_lo >= row_min && row_lo + 31 <= row_max) {  // wave-uniform: one lane offset + scalar row steps
        const unsigned off = (unsigned)((row_lo + (lane >> 4)) * rs + 4 * (lane & 15));
#pragma unroll
        for (int i = 0; i < 8; ++i) t.v[i] = *reinterpret_cast<const f32x4*>(base + (off + (unsigned)(4 * i * rs)));
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = min(max(row_lo + 4 * i + (lane >> 4), row_min), row_max);
            t.v[i] = *reinterpret_cast<const f32x4*>(base + (unsigned)(row * rs + 4 * (lane & 15)));
        }
    }
}

template <int NS, typename X8>
__device__ __forceinline__ void tile_to_frags(X8 (&f)[4][NS], const TileRegs& t, float* stg, int lane) {
    __builtin_amdgcn_wave_barrier();  // earlier readers of the staging buffer are done (in-order LDS)
#pragma unroll
    for (int i = 0; i < 8; ++i)
        *reinterpret_cast<f32x4*>(stg + (4 * i + (lane >> 4)) * STG_LD + 4 * (lane & 15)) = t.v[i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const float* row = stg + (lane & 31) * STG_LD + 8 * (lane >> 5);
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        const f32x4 lo = *reinterpret_cast<const f32x4*>(row + 16 * kb);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(row + 16 * kb + 4);
        const float x[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        split8<NS>(x, f[kb]);
    }
}

// exp(x) for x <= 0 via the hardware exp2 (v_exp_f32): relative error <= ~2e-6 for x in [-20, 0],
// results below 2^-126 flush to zero (harmless in a softmax numerator).
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }

struct VRegs { float2 v[16]; };

// V rows in PV-operand order: entry 8*st + j is key row rho(st,half,j), head dims 2c, 2c+1
__device__ __forceinline__ void v_load(VRegs& t, const float* vb, int v_rs, int r0, int Lk, int c, int half) {
    if (r0 + 32 <= Lk) {  // wave-uniform fast path
        const unsigned off = (unsigned)((r0 + 4 * half) * v_rs + 2 * c);
#pragma unroll
        for (int i = 0; i < 16; ++i)
            t.v[i] = *reinterpret_cast<const float2*>(vb + (off + (unsigned)(((i & 3) + 8 * (i >> 2)) * v_rs)));
    } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int key = min(r0 + mfma32_row(i, half), Lk - 1);
            t.v[i] = *reinterpret_cast<const float2*>(vb + (unsigned)(key * v_rs + 2 * c));
        }
    }
}

// acc += sum over the significant cross terms of a (A operand parts) x b (B operand parts)
template <int NS, typename X8>
__device__ __forceinline__ f32x16 mfma_terms(const X8 (&a)[NS], const X8 (&b)[NS], f32x16 acc) {
    if (NS == 3) {
        acc = mma16(a[1], b[1], acc);
        acc = mma16(a[0], b[NS - 1], acc);
        acc = mma16(a[NS - 1], b[0], acc);
    }
    acc = mma16(a[0], b[1], acc);
    acc = mma16(a[1], b[0], acc);
    acc = mma16(a[0], b[0], acc);
    return acc;
}

// tile[i][j] = X_i . Y_j (i on accumulator rows, j on lanes)
template <int NS, typename X8>
__device__ __forceinline__ f32x16 dot_tile(const X8 (&x)[4][NS], const X8 (&y)[4][NS]) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) acc = mfma_terms<NS>(x[kb], y[kb], acc);
    return acc;
}

template <int NS, bool RELKEY, bool DROP, typename E>
__global__ __launch_bounds__(256, 2) void attn_fwd_split_kernel(
    const float* __restrict__ q, int64_t q_bs, int64_t q_rs, const float* __restrict__ k, int64_t k_bs,
    int64_t k_rs, const float* __restrict__ v, int64_t v_bs, int64_t v_rs, const float* __restrict__ dist_emb,
    int P, const float* __restrict__ key_mask, float* __restrict__ out, float* __restrict__ lse, int nh, int Lq,
    int Lk, int q_tiles, int n_units, int skip_padded_tiles, E3dBounds bnd, E3dDrop drop_in) {
    const E3dDrop drop = e3d_drop_resolve(drop_in);   // + the device-side epoch (graph replays: e3d_common.h)
    typedef typename AV<E>::x8 bf16x8;   // (name kept from the bf16 form: 8 split terms of type E)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int qi = lane & 31, half = lane >> 5;
    const int unit = xcd_remap(blockIdx.x, gridDim.x) * (blockDim.x >> 6) + wid;
    if (unit >= n_units) return;
    const int qt = unit % q_tiles, bh = unit / q_tiles, h = bh % nh, b = bh / nh;
    float* ring = smem + wid * WAVE_LDS_F;
    float* stg = ring + RING_F;
    float* kbias = stg + STG_F;

    const int q0 = qt * 32;
    const int lq = min(q0 + qi, Lq - 1);
    bf16x8 qf[4][NS];
    load_row_split<NS>(qf, q + b * q_bs + (int64_t)lq * q_rs + h * D, half);

    const float* kb_ = k + b * k_bs + h * D;
    const float* vb = v + b * v_bs + h * D;
    const float* mb = key_mask ? key_mask + (int64_t)b * Lk : nullptr;

    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;

    int rot = 0;
    if (RELKEY) {
        TileRegs ereg;
        tile_load(ereg, dist_emb, D, q0 + 1 + P - 1, 0, 2 * P - 2, lane);
        bf16x8 ef[4][NS];
        tile_to_frags<NS>(ef, ereg, stg, lane);
        const f32x16 t = dot_tile<NS>(ef, qf);
#pragma unroll
        for (int r = 0; r < 16; ++r) ring[(32 + mfma32_row(r, half)) * RING_LD + qi] = t[r];
    }

    // Trailing key tiles that are padding in every position contribute exp(s - 10000 - m) == 0.0f exactly
    // (fp32 underflow) AS LONG AS the scores of a row spread over less than ~9900 -- the reference's mask is additive
    // (structure_model/model.py:226-231), so with |q|, |k| in the thousands a padded key does reach the softmax, and
    // skipping it would be wrong (found at 0.47 max-norm in the weights-x4 regime of the margin test).  The sweep stops
    // after the tile of the last valid key only when the caller's element bounds prove the spread small
    // (e3d_mask_skip_is_exact); the result is then bit-identical to the full sweep.  An all-padding item (no valid key)
    // keeps the full sweep, as the reference then softmaxes the uniformly shifted scores.
    int k_tiles = (Lk + 31) >> 5;
    if (mb && skip_padded_tiles && e3d_mask_skip_is_exact(bnd)) {
        int last = -1;
        for (int base = 0; base < Lk; base += 64) {
            const int key = base + lane;
            const bool valid = key < Lk && mb[key] != 0.f;
            const unsigned long long bits = __ballot(valid);
            if (bits) last = base + 63 - __builtin_clzll(bits);
        }
        if (last >= 0) k_tiles = (last >> 5) + 1;
    }
    for (int kt = 0; kt < k_tiles; ++kt) {
        const int r0 = kt * 32;
        TileRegs kreg, ereg;
        VRegs vreg;
        tile_load(kreg, kb_, (int)k_rs, r0, 0, Lk - 1, lane);
        if (RELKEY) tile_load(ereg, dist_emb, D, q0 - r0 - 31 + P - 1, 0, 2 * P - 2, lane);
        {   // additive key bias of this tile: one coalesced load, shared through LDS
            const int key = r0 + qi;
            float bias = -INFINITY;
            if (key < Lk) bias = mb ? (1.0f - mb[key]) * -10000.0f : 0.f;
            if (half == 0) kbias[qi] = bias;
        }
        f32x16 s;
        {
            bf16x8 kf[4][NS];
            tile_to_frags<NS>(kf, kreg, stg, lane);
            s = dot_tile<NS>(kf, qf);
        }
        if (RELKEY) {
            bf16x8 ef[4][NS];
            tile_to_frags<NS>(ef, ereg, stg, lane);
            const f32x16 t = dot_tile<NS>(ef, qf);
#pragma unroll
            for (int r = 0; r < 16; ++r) ring[((mfma32_row(r, half) + rot) & 63) * RING_LD + qi] = t[r];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int x = qi - mfma32_row(r, half) + 31;
                s[r] += ring[((x + rot) & 63) * RING_LD + qi];
            }
            __builtin_amdgcn_wave_barrier();
            rot ^= 32;
        }

        v_load(vreg, vb, (int)v_rs, r0, Lk, qi, half);  // in flight under the softmax arithmetic
        float tmax = -INFINITY;
#pragma unroll
        for (int g = 0; g < 4; ++g) {  // rows 8g + 4*half + {0..3}: one 16-byte LDS read per group
            const f32x4 bv = *reinterpret_cast<const f32x4*>(kbias + 8 * g + 4 * half);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s[4 * g + j] = s[4 * g + j] * 0.125f + bv[j];
                tmax = fmaxf(tmax, s[4 * g + j]);
            }
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m_run, tmax);
        const float alpha = fast_exp(m_run - m_new);
        m_run = m_new;
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[r] = fast_exp(s[r] - m_new);
            psum += s[r];
        }
        l_run = l_run * alpha + psum;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
        if (DROP) {   // dropout on the probabilities: the row sum above stays un-dropped (softmax first, then dropout)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float m[4];
                e3d_drop_mult4(drop, e3d_attn_drop_idx4(bh, Lq, Lk, q0 + qi, r0 + 8 * g + 4 * half), m);
#pragma unroll
                for (int j = 0; j < 4; ++j) s[4 * g + j] *= m[j];
            }
        }

        // O^T += V^T P^T, two 16-key steps; P^T registers 8st..8st+7 are the B operand
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            float pv[8], v0[8], v1[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                pv[j] = s[8 * st + j];
                v0[j] = vreg.v[8 * st + j].x;
                v1[j] = vreg.v[8 * st + j].y;
            }
            bf16x8 pb[NS], a0[NS], a1[NS];
            split8<NS>(pv, pb);
            split8<NS>(v0, a0);
            split8<NS>(v1, a1);
            o0 = mfma_terms<NS>(a0, pb, o0);
            o1 = mfma_terms<NS>(a1, pb, o1);
        }
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (q0 + qi < Lq) {
        float* orow = out + ((int64_t)b * Lq + q0 + qi) * (nh * D) + h * D;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
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

int g_skip_padded = 1;

template <int NS, bool DROP = false, typename E = __bf16>
int launch(const float* q, int64_t q_bs, int64_t q_rs, const float* k, int64_t k_bs, int64_t k_rs, const float* v,
           int64_t v_bs, int64_t v_rs, const float* dist_emb, int P, const float* key_mask, float* out, float* lse,
           int B, int nh, int Lq, int Lk, E3dBounds bnd, hipStream_t s, E3dDrop drop = E3dDrop{0, 0, 1.f}) {
    const int q_tiles = (Lq + 31) / 32;
    const int n_units = B * nh * q_tiles;
    const int wpb = 4;
    const int n_blocks = (n_units + wpb - 1) / wpb;
    const size_t lds = (size_t)wpb * WAVE_LDS_F * sizeof(float);
    if (dist_emb)
        hipLaunchKernelGGL((attn_fwd_split_kernel<NS, true, DROP, E>), dim3(n_blocks), dim3(64 * wpb), lds, s, q, q_bs, q_rs, k,
                           k_bs, k_rs, v, v_bs, v_rs, dist_emb, P, key_mask, out, lse, nh, Lq, Lk, q_tiles, n_units, g_skip_padded, bnd, drop);
    else
        hipLaunchKernelGGL((attn_fwd_split_kernel<NS, false, DROP, E>), dim3(n_blocks), dim3(64 * wpb), lds, s, q, q_bs, q_rs, k,
                           k_bs, k_rs, v, v_bs, v_rs, dist_emb, P, key_mask, out, lse, nh, Lq, Lk, q_tiles, n_units, g_skip_padded, bnd, drop);
    return e3d_launch_status("e3d_relkey_attn_fwd_split");
}

}  // namespace

extern "C" int e3d_attn_skip_padded_tiles(int enable) {
    const int prev = g_skip_padded;
    g_skip_padded = enable ? 1 : 0;
    return prev;
}

extern "C" int e3d_relkey_attn_fwd_split_ex(const float* q, int64_t q_bs, int64_t q_rs, const float* k, int64_t k_bs,
                                            int64_t k_rs, const float* v, int64_t v_bs, int64_t v_rs,
                                            const float* dist_emb, int P, const float* key_mask, float* out,
                                            float* lse, int B, int nh, int Lq, int Lk, int terms, float drop_p,
                                            uint64_t drop_seed, void* e_scratch, int e_scratch_ready, const float* q_absmax,
                                            const float* k_absmax, float* e_absmax, void* stream) {
    const E3dBounds bnd{q_absmax, k_absmax, dist_emb ? e_absmax : nullptr};
    E3D_REQUIRE(!dist_emb || !q_absmax || e_absmax, "attn_split: rel-key attention with element bounds needs e_absmax too");
    E3D_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "attn_split: drop_p=%g outside [0, 1)", (double)drop_p);
    E3D_REQUIRE(q && k && v && out, "attn_split: null pointer");
    E3D_REQUIRE(B > 0 && nh > 0 && Lq > 0 && Lk > 0, "attn_split: bad shape B=%d nh=%d Lq=%d Lk=%d", B, nh, Lq, Lk);
    E3D_REQUIRE(q_rs % 4 == 0 && k_rs % 4 == 0 && v_rs % 2 == 0 && q_bs % 4 == 0 && k_bs % 4 == 0 && v_bs % 2 == 0,
                "attn_split: strides must keep 16B (q,k) / 8B (v) alignment");
    E3D_REQUIRE(((uintptr_t)q % 16) == 0 && ((uintptr_t)k % 16) == 0 && ((uintptr_t)v % 16) == 0 &&
                    ((uintptr_t)out % 16) == 0, "attn_split: pointers must be 16B aligned");
    E3D_REQUIRE(terms == 3 || terms == 6 || terms == E3D_TERMS_F16X3, "attn_split: terms must be 3, 6 or 19 (got %d)", terms);
    const int f16 = terms == E3D_TERMS_F16X3;
    E3D_REQUIRE((int64_t)Lk * k_rs < (1ll << 30) && (int64_t)Lk * v_rs < (1ll << 30),
                "attn_split: one batch item's K/V slab must stay below 2^30 elements (32-bit lane offsets)");
    if (dist_emb) {
        E3D_REQUIRE(Lq == Lk && Lq <= P, "attn_split: relative_key needs Lq == Lk <= P (Lq=%d Lk=%d P=%d)", Lq, Lk, P);
        E3D_REQUIRE(((uintptr_t)dist_emb % 16) == 0, "attn_split: dist_emb must be 16B aligned");
    }
    E3D_REQUIRE((int64_t)B * nh * ((Lq + 31) / 32) < (1ll << 30), "attn_split: too many tiles");
    hipStream_t s = (hipStream_t)stream;
    if (dist_emb && e_scratch && !e_scratch_ready) {
        // whichever kernel serves this call, a scratch handed in is valid afterwards (callers cache it)
        const int rc = e3d_attn_fill_planes(dist_emb, P, Lk, e_scratch, f16, e_absmax, s);   // (also raises *e_absmax)
        if (rc) return rc;
        e_scratch_ready = 1;
    }
    static int coop = -1;   // E3D_ATTN_COOP=0: per-wave kernel below for every shape (A/B experiments)
    if (coop < 0) {
        const char* e = getenv("E3D_ATTN_COOP");
        coop = e ? atoi(e) : 1;
    }
    const bool dropping = drop_p > 0.f;
    const E3dDrop d = e3d_drop_make(drop_p, drop_seed);
    // two-wave groups (q_tiles % 4 != 0) measured slower than the per-wave kernel: too little sharing per barrier
    // the cooperative kernel reads the distance table as fragment-order bf16 planes from a CALLER-provided scratch
    // (e3d_attn_scratch_bytes); without one the per-wave kernel below serves the call -- the library never allocates.
    // With dropout (training) the cooperative kernel exists in bf16x3.
    // (its staging goes through buffer descriptors of one key tile's rows: 32-bit byte offsets and record counts, i.e. an
    //  item's K / V rows must span less than 2 GiB -- Lk x row stride x 4 bytes; the per-wave kernel serves anything wider)
    const bool span_ok = (int64_t)Lk * (k_rs > v_rs ? k_rs : v_rs) * 4 < ((int64_t)1 << 31) && k_rs > 0 && v_rs > 0 &&
                         k_rs < ((int64_t)1 << 24) && v_rs < ((int64_t)1 << 24);     // (a tile's 32 rows in a 32-bit offset)
    if ((terms == 3 || (f16 && !dropping)) && coop && span_ok && v_rs % 4 == 0 && v_bs % 4 == 0 && ((Lq + 31) / 32) % 4 == 0 &&
        (!dist_emb || e_scratch))
        return e3d_attn_coop_launch(q, q_bs, q_rs, k, k_bs, k_rs, v, v_bs, v_rs, dist_emb, P, key_mask, out, lse, B, nh,
                                    Lq, Lk, g_skip_padded, bnd, e_scratch, e_scratch_ready, f16, d, dropping, s);
    if (dropping) {   // training with attention-probability dropout, other shapes / arithmetics: per-wave kernel
        if (terms == 3)
            return launch<2, true>(q, q_bs, q_rs, k, k_bs, k_rs, v, v_bs, v_rs, dist_emb, P, key_mask, out, lse, B, nh, Lq, Lk, bnd, s, d);
        return launch<3, true>(q, q_bs, q_rs, k, k_bs, k_rs, v, v_bs, v_rs, dist_emb, P, key_mask, out, lse, B, nh, Lq, Lk, bnd, s, d);
    }
    if (f16)
        return launch<2, false, _Float16>(q, q_bs, q_rs, k, k_bs, k_rs, v, v_bs, v_rs, dist_emb, P, key_mask, out, lse, B, nh,
                                          Lq, Lk, bnd, s);
    if (terms == 3)
        return launch<2>(q, q_bs, q_rs, k, k_bs, k_rs, v, v_bs, v_rs, dist_emb, P, key_mask, out, lse, B, nh, Lq, Lk, bnd, s);
    return launch<3>(q, q_bs, q_rs, k, k_bs, k_rs, v, v_bs, v_rs, dist_emb, P, key_mask, out, lse, B, nh, Lq, Lk, bnd, s);
}

extern "C" int e3d_relkey_attn_fwd_split(const float* q, int64_t q_bs, int64_t q_rs, const float* k, int64_t k_bs,
                                         int64_t k_rs, const float* v, int64_t v_bs, int64_t v_rs,
                                         const float* dist_emb, int P, const float* key_mask, float* out, float* lse,
                                         int B, int nh, int Lq, int Lk, int terms, void* stream) {
    return e3d_relkey_attn_fwd_split_ex(q, q_bs, q_rs, k, k_bs, k_rs, v, v_bs, v_rs, dist_emb, P, key_mask, out, lse, B,
                                        nh, Lq, Lk, terms, 0.f, 0, nullptr, 0, nullptr, nullptr, nullptr, stream);
}

extern "C" int e3d_relkey_attn_fwd_split_drop(const float* q, int64_t q_bs, int64_t q_rs, const float* k, int64_t k_bs,
                                              int64_t k_rs, const float* v, int64_t v_bs, int64_t v_rs,
                                              const float* dist_emb, int P, const float* key_mask, float* out,
                                              float* lse, int B, int nh, int Lq, int Lk, int terms, float drop_p,
                                              uint64_t drop_seed, void* stream) {
    return e3d_relkey_attn_fwd_split_ex(q, q_bs, q_rs, k, k_bs, k_rs, v, v_bs, v_rs, dist_emb, P, key_mask, out, lse, B,
                                        nh, Lq, Lk, terms, drop_p, drop_seed, nullptr, 0, nullptr, nullptr, nullptr, stream);
}
