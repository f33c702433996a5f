// Fused relative-key attention forward, workgroup-cooperative form (bf16x3 split arithmetic).
//
// Same mathematics and accumulator layouts as attn_relkey_split.hip (query on the MFMA lane, key on
// the accumulator rows, online softmax in fp32, rel-key skew through a per-wave LDS ring), but the
// W waves of a workgroup are W consecutive 32-query tiles of ONE (batch, head):
//   * every key tile's K and V are fetched from HBM/L2 once per workgroup (not once per wave), split
//     into bf16 hi/lo planes once, and parked in a double-buffered LDS image that all waves read
//     their MFMA fragments from; the loads of tile t+1 are in flight while tile t is computed;
//   * V is stored TRANSPOSED ([d][key], keys in the PV-operand order) so that the A operand of
//     O^T += V^T P^T is one ds_read_b128 per fragment instead of 8 scalar gathers;
//   * the distance embedding E is split into bf16 planes by a tiny pre-pass (it is a parameter:
//     (2P-1) x 64 values) and each wave reads its 32-row block as MFMA fragments straight from
//     those planes (L2-resident, 130 KB), so E costs neither LDS nor VALU in the hot loop;
//   * the output tile is transposed through the wave's (then idle) ring and leaves in full 256-byte
//     rows.
// In-kernel stamps of the per-wave kernel showed each wave spending ~12k cycles per key tile for
// ~1.2k cycles of MFMA work, the rest being its private load -> split -> LDS -> fragment chain.
#include <type_traits>

#include "e3d_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int D = 64;
constexpr int RING_LD = 34;
constexpr int RING_F = 64 * RING_LD;          // 2176 floats: also holds the 32 x 68 output tile
constexpr int OUT_LD = 68;
constexpr int K_ROW_B = 144;                  // 64 bf16 + 16 B pad: conflict-free b128 fragment reads
constexpr int V_ROW_B = 80;                   // 32 bf16 + 16 B pad
constexpr int K_PLANE_B = 32 * K_ROW_B, V_PLANE_B = 64 * V_ROW_B;
constexpr int KV_BUF_B = 2 * K_PLANE_B + 2 * V_PLANE_B + 128;   // + 32 floats of key bias

__device__ __forceinline__ void split4x2(const f32x4 v, bf16x4& hi, bf16x4& lo) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const __bf16 p = (__bf16)v[j];
        hi[j] = p;
        lo[j] = (__bf16)(v[j] - (float)p);
    }
}

__device__ __forceinline__ void split8x2(const float (&x)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 p = (__bf16)x[j];
        hi[j] = p;
        lo[j] = (__bf16)(x[j] - (float)p);
    }
}

// acc += a.b from the three significant cross terms, smallest first ([0] = hi, [1] = lo)
__device__ __forceinline__ f32x16 mfma3(const bf16x8 (&a)[2], const bf16x8 (&b)[2], f32x16 acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
    return acc;
}

__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }

// position of key kappa (0..31) inside a V^T row: bits 2 and 3 swapped, so that the 8 keys
// rho(st, half, j) = (j&3) + 8(2 st + (j>>2)) + 4 half of one PV operand sit in 16 contiguous bytes
__device__ __forceinline__ int v_pos(int kappa) {
    return (kappa & 0x13) | ((kappa & 4) << 1) | ((kappa & 8) >> 1);
}

// Distance table -> bf16 hi/lo planes in MFMA-FRAGMENT order.  Every 32-row block of E the kernel touches starts
// at a row e0 = P + 32 m (q0 and r0 are multiples of 32), m = -J0 .. J0 - 1 with J0 = ceil(L / 32), so the
// fragments of block j = m + J0 are laid out as [j][plane][kb][lane] x 16 bytes: a wave's E load is then 8
// fully coalesced 1-KB reads instead of 8 reads of 64 scattered 16-byte pieces (measured: 17 % of the kernel).
// Rows outside [0, 2P-2] (never paired with a valid (query, key)) are clamped.
__global__ __launch_bounds__(256) void e_fragments_kernel(const float* __restrict__ e, bf16x8* __restrict__ frag, int P,
                                                          int J0, int n_items) {
    const int i = blockIdx.x * 256 + threadIdx.x;   // item = ((j * 2 + plane) * 4 + kb) * 64 + lane
    if (i >= n_items) return;
    const int lane = i & 63, kb = (i >> 6) & 3, plane = (i >> 8) & 1, j = i >> 9;
    const int row = min(max(P + 32 * (j - J0) + (lane & 31), 0), 2 * P - 2);
    const float* src = e + row * D + 16 * kb + 8 * (lane >> 5);
    bf16x8 out;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const float v = src[t];
        const __bf16 hi = (__bf16)v;
        out[t] = plane ? (__bf16)(v - (float)hi) : hi;
    }
    frag[i] = out;
}

#ifdef E3D_ATTN_STAMPS   // lab builds only (tools/lab_attn_stamps.py): phase time stamps of one wave
__device__ long long e3d_attn_stamps[16][8];
#define ASTAMP(slot)                                                                                     \
    do {                                                                                                 \
        if (blockIdx.x == 1000 && wid == 1 && lane == 0 && kt < 16) e3d_attn_stamps[kt][slot] = __builtin_readcyclecounter(); \
    } while (0)
#else
#define ASTAMP(slot) do {} while (0)
#endif

template <int W, bool RELKEY>
__global__ __launch_bounds__(W * 64, 2) void attn_coop_kernel(
    const float* __restrict__ q, int64_t q_bs, int64_t q_rs, const float* __restrict__ k, int64_t k_bs,
    int64_t k_rs, const float* __restrict__ v, int64_t v_bs, int64_t v_rs, const bf16x8* __restrict__ e_frag,
    int P, const float* __restrict__ key_mask, float* __restrict__ out, float* __restrict__ lse, int nh, int Lq, int Lk,
    int groups_per_bh, int skip_padded_tiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int NT = W * 64, NI = 512 / NT;   // float4 staging items per thread, for K and for V
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int qi = lane & 31, half = lane >> 5;
    const int blk = xcd_remap(blockIdx.x, gridDim.x);
    const int grp = blk % groups_per_bh, bh = blk / groups_per_bh, h = bh % nh, b = bh / nh;
    const int q0 = (grp * W + wid) * 32;
    float* ring = reinterpret_cast<float*>(smem_raw + 2 * KV_BUF_B) + wid * RING_F;

    // Q fragments (B operand of S^T and T^T): row = query, 8 head-dim values per 16-wide k block
    bf16x8 qf[4][2];
    {
        const float* qrow = q + b * q_bs + (int64_t)min(q0 + qi, Lq - 1) * q_rs + h * D + 8 * half;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            const f32x4 lo4 = *reinterpret_cast<const f32x4*>(qrow + 16 * kb);
            const f32x4 hi4 = *reinterpret_cast<const f32x4*>(qrow + 16 * kb + 4);
            const float x[8] = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
            split8x2(x, qf[kb][0], qf[kb][1]);
        }
    }

    const float* kb_ = k + b * k_bs + h * D;
    const float* vb_ = v + b * v_bs + h * D;
    const float* mb = key_mask ? key_mask + (int64_t)b * Lk : nullptr;

    // Trailing all-padding key tiles contribute exp(-10000 - m) == 0.0f exactly: stop after the tile
    // of the last valid key (bit-identical; an all-padding item keeps the full sweep).  Depends on
    // the batch item only, so every wave of the workgroup runs the same number of barriers.
    int k_tiles = (Lk + 31) >> 5;
    if (mb && skip_padded_tiles) {
        int last = -1;
        for (int base = 0; base < Lk; base += 64) {
            const int key = base + lane;
            const bool valid = key < Lk && mb[key] != 0.f;
            const unsigned long long bits = __ballot(valid);
            if (bits) last = base + 63 - __builtin_clzll(bits);
        }
        if (last >= 0) k_tiles = (last >> 5) + 1;
    }

    // cooperative staging: item i of this thread = row (f >> 4), head dims 4 (f & 15) .. +3, f = tid + NT i
    // Two register sets when they are small (W >= 4: one or two float4 per operand): tile t+2 is loaded while
    // tile t is computed and tile t+1 (loaded a tile earlier) is split and stored -- the global latency
    // (stamps: 2-3k cycles under load) then never shows.  Narrow workgroups keep one set (distance 1).
    constexpr bool DEEP = NI <= 2;
    constexpr int NSET = DEEP ? 2 : 1;
    f32x4 sk[NSET][NI], sv[NSET][NI];
    float smask[NSET];
    auto stage_load = [&](auto set_tag, int r0) {
        constexpr int SET = decltype(set_tag)::value;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int f = tid + NT * i;
            const int key = min(r0 + (f >> 4), Lk - 1);
            sk[SET][i] = *reinterpret_cast<const f32x4*>(kb_ + (unsigned)(key * (int)k_rs + 4 * (f & 15)));
            sv[SET][i] = *reinterpret_cast<const f32x4*>(vb_ + (unsigned)(key * (int)v_rs + 4 * (f & 15)));
        }
        // key-mask value of key r0 + (tid & 31): loaded by every thread (no branch around a load: hipcc then
        // counts vmcnt exactly), used by the first 32
        smask[SET] = mb ? mb[min(r0 + (tid & 31), Lk - 1)] : 1.0f;
    };
    auto stage_store = [&](auto set_tag, int r0, unsigned char* buf) {
        constexpr int SET = decltype(set_tag)::value;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int f = tid + NT * i, row = f >> 4, c4 = f & 15;
            bf16x4 hi, lo;
            split4x2(sk[SET][i], hi, lo);
            *reinterpret_cast<bf16x4*>(buf + row * K_ROW_B + 8 * c4) = hi;
            *reinterpret_cast<bf16x4*>(buf + K_PLANE_B + row * K_ROW_B + 8 * c4) = lo;
            split4x2(sv[SET][i], hi, lo);
            unsigned char* vt = buf + 2 * K_PLANE_B + (4 * c4) * V_ROW_B + 2 * v_pos(row);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                *reinterpret_cast<__bf16*>(vt + e * V_ROW_B) = hi[e];
                *reinterpret_cast<__bf16*>(vt + V_PLANE_B + e * V_ROW_B) = lo[e];
            }
        }
        if (tid < 32)
            reinterpret_cast<float*>(buf + 2 * K_PLANE_B + 2 * V_PLANE_B)[tid] =
                r0 + tid < Lk ? (1.0f - smask[SET]) * -10000.0f : -INFINITY;
    };
    using Set0 = std::integral_constant<int, 0>;
    using Set1 = std::integral_constant<int, DEEP ? 1 : 0>;

    // rel-key: E fragments of block j (distance-table rows P + 32 (j - J0) .. + 31) from the fragment-order planes
    bf16x8 ef[4][2];
    const int J0 = (Lk + 31) >> 5, qt = q0 >> 5;
    auto e_load = [&](int j) {
        const bf16x8* pj = e_frag + (size_t)j * 512 + lane;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            ef[kb][0] = pj[kb * 64];
            ef[kb][1] = pj[256 + kb * 64];
        }
    };
    auto dot_q = [&](const bf16x8 (&x)[4][2]) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) acc = mfma3(x[kb], qf[kb], acc);
        return acc;
    };

    stage_load(Set0{}, 0);
    if (RELKEY) {
        e_load(qt + J0);          // rows q0 + P ..: the block "before" key tile 0 fills the upper half of the ring
        const f32x16 t = dot_q(ef);
#pragma unroll
        for (int r = 0; r < 16; ++r) ring[(32 + mfma32_row(r, half)) * RING_LD + qi] = t[r];
        e_load(qt - 1 + J0);      // key tile 0: rows q0 - 32 + P ..
    }
    stage_store(Set0{}, 0, smem_raw);
    if (DEEP) stage_load(Set1{}, (k_tiles > 1 ? 1 : 0) * 32);
    __syncthreads();

    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;
    // Ring addressing without per-tile integer math.  Tiles alternate PAR = 0 / 1 (ring rotation 0 / 32 rows, LDS
    // buffer 0 / 1).  With rho = mfma32_row(r, half) = rp(r) + 4 half, rp(r) = (r & 3) + 8 (r >> 2) <= 27:
    //   T^T row rho is written to ring row rho + rot:      ring_w[(rot + rp) * RING_LD]           (immediates)
    //   score row rho reads window row x = qi - rho + 31 (0..62) at ring row x ^ rot:
    //       rot = 0:  ring_e[(27 - rp) * RING_LD]                                                 (immediates)
    //       rot = 32: ring[rd_odd[r]]                                           (16 precomputed lane offsets)
    float* const ring_w = ring + 4 * half * RING_LD + qi;
    const float* const ring_e = ring + (qi + 31 - 4 * half - 27) * RING_LD + qi;
    int rd_odd[16];
    if (RELKEY) {
#pragma unroll
        for (int r = 0; r < 16; ++r) rd_odd[r] = (((qi - mfma32_row(r, half) + 31) ^ 32) * RING_LD + qi);
    }

    auto tile = [&](auto par_tag, int kt) {
        constexpr int PAR = decltype(par_tag)::value;
        const unsigned char* buf = smem_raw + PAR * KV_BUF_B;
        const bool more = kt + 1 < k_tiles;
        const int kt_next = more ? kt + 1 : kt;   // last tile: harmless re-load (no branch around the loads:
        ASTAMP(0);                                //  hipcc then counts vmcnt exactly instead of draining to 0)
        ASTAMP(1);

        f32x16 s;
        {   // S^T = K Q^T
            bf16x8 kf[4][2];
            const unsigned char* kr = buf + qi * K_ROW_B + 16 * half;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                kf[kb][0] = *reinterpret_cast<const bf16x8*>(kr + 32 * kb);
                kf[kb][1] = *reinterpret_cast<const bf16x8*>(kr + K_PLANE_B + 32 * kb);
            }
            s = dot_q(kf);
        }
        ASTAMP(2);
        if (RELKEY) {
            const f32x16 t = dot_q(ef);
            e_load(qt - kt_next - 1 + J0);   // next tile's block, in flight under the softmax
#pragma unroll
            for (int r = 0; r < 16; ++r) ring_w[(32 * PAR + (r & 3) + 8 * (r >> 2)) * RING_LD] = t[r];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int r = 0; r < 16; ++r)
                s[r] += PAR ? ring[rd_odd[r]] : ring_e[(27 - ((r & 3) + 8 * (r >> 2))) * RING_LD];
            __builtin_amdgcn_wave_barrier();
        }
        // K/V of the next tile: issued only now -- the T MFMAs above wait for the E fragments with an in-order
        // vmcnt, so any younger load in flight at that point would be waited for as well (stamps: ~1500 cycles)
        // (two sets: tile t+2 into the set whose tile-t data was stored a tile ago; clamped re-loads at the end)
        const int kt_ld = DEEP ? (kt + 2 < k_tiles ? kt + 2 : k_tiles - 1) : kt_next;
        if (PAR == 0) stage_load(Set0{}, kt_ld * 32);
        else stage_load(Set1{}, kt_ld * 32);
        ASTAMP(3);

        const float* kbias = reinterpret_cast<const float*>(buf + 2 * K_PLANE_B + 2 * V_PLANE_B);
        float tmax = -INFINITY;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
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
        ASTAMP(4);

        // O^T += V^T P^T: two 16-key steps; P^T registers 8 st .. 8 st + 7 are the B operand, the A
        // operand rows are head dims qi (o0) and 32 + qi (o1) of the transposed V image
        const unsigned char* vr = buf + 2 * K_PLANE_B + qi * V_ROW_B + 16 * half;
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            float pv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) pv[j] = s[8 * st + j];
            bf16x8 pb[2], a0[2], a1[2];
            split8x2(pv, pb[0], pb[1]);
            a0[0] = *reinterpret_cast<const bf16x8*>(vr + 32 * st);
            a0[1] = *reinterpret_cast<const bf16x8*>(vr + V_PLANE_B + 32 * st);
            a1[0] = *reinterpret_cast<const bf16x8*>(vr + 32 * V_ROW_B + 32 * st);
            a1[1] = *reinterpret_cast<const bf16x8*>(vr + V_PLANE_B + 32 * V_ROW_B + 32 * st);
            o0 = mfma3(a0, pb, o0);
            o1 = mfma3(a1, pb, o1);
        }

        ASTAMP(5);
        if (more) {   // tile t+1: from the other set when there are two, from the one just loaded otherwise
            if (DEEP == (PAR == 0)) stage_store(Set1{}, kt_next * 32, smem_raw + (PAR ^ 1) * KV_BUF_B);
            else stage_store(Set0{}, kt_next * 32, smem_raw + (PAR ^ 1) * KV_BUF_B);
        }
        ASTAMP(6);
        __syncthreads();
        ASTAMP(7);
    };
    for (int kt = 0; kt < k_tiles; kt += 2) {
        tile(std::integral_constant<int, 0>{}, kt);
        if (kt + 1 < k_tiles) tile(std::integral_constant<int, 1>{}, kt + 1);
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    // transpose the 64 x 32 O^T tile through the ring: out rows leave as 256-byte segments
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        f32x4 a, c;
#pragma unroll
        for (int j = 0; j < 4; ++j) { a[j] = o0[4 * g + j] * inv; c[j] = o1[4 * g + j] * inv; }
        *reinterpret_cast<f32x4*>(ring + qi * OUT_LD + 8 * g + 4 * half) = a;
        *reinterpret_cast<f32x4*>(ring + qi * OUT_LD + 32 + 8 * g + 4 * half) = c;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float* obase = out + ((int64_t)b * Lq + q0) * (nh * D) + h * D;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = 4 * i + (lane >> 4);
        if (q0 + row < Lq)
            *reinterpret_cast<f32x4*>(obase + (int64_t)row * (nh * D) + 4 * (lane & 15)) =
                *reinterpret_cast<const f32x4*>(ring + row * OUT_LD + 4 * (lane & 15));
    }
    if (lse && half == 0 && q0 + qi < Lq) lse[((int64_t)b * nh + h) * Lq + q0 + qi] = m_run + logf(l_tot);
}

template <int W, bool RELKEY>
int launch_w(const float* q, int64_t q_bs, int64_t q_rs, const float* k, int64_t k_bs, int64_t k_rs, const float* v,
             int64_t v_bs, int64_t v_rs, const bf16x8* e_frag, int P, const float* key_mask,
             float* out, float* lse, int B, int nh, int Lq, int Lk, int q_tiles, int skip, hipStream_t s) {
    const size_t lds = 2 * KV_BUF_B + (size_t)W * RING_F * sizeof(float);
    static std::atomic<uint64_t> lds_ok{0};
    e3d_allow_lds(lds_ok, attn_coop_kernel<W, RELKEY>, lds);
    const int groups = q_tiles / W;
    hipLaunchKernelGGL((attn_coop_kernel<W, RELKEY>), dim3(B * nh * groups), dim3(W * 64), lds, s, q, q_bs, q_rs, k, k_bs,
                       k_rs, v, v_bs, v_rs, e_frag, P, key_mask, out, lse, nh, Lq, Lk, groups, skip);
    return e3d_launch_status("e3d_relkey_attn_fwd_split (cooperative)");
}

template <bool RELKEY>
int launch_any(int W, const float* q, int64_t q_bs, int64_t q_rs, const float* k, int64_t k_bs, int64_t k_rs,
               const float* v, int64_t v_bs, int64_t v_rs, const bf16x8* e_frag, int P,
               const float* key_mask, float* out, float* lse, int B, int nh, int Lq, int Lk, int q_tiles, int skip,
               hipStream_t s) {
#define E3D_COOP_CASE(w)                                                                                          \
    case w:                                                                                                       \
        return launch_w<w, RELKEY>(q, q_bs, q_rs, k, k_bs, k_rs, v, v_bs, v_rs, e_frag, P, key_mask, out, lse, B, nh, \
                                   Lq, Lk, q_tiles, skip, s)
    switch (W) {
        E3D_COOP_CASE(8);
        E3D_COOP_CASE(4);
        E3D_COOP_CASE(2);
        E3D_COOP_CASE(1);
    }
#undef E3D_COOP_CASE
    return -1;
}

}  // namespace

// bf16x3 attention, cooperative kernel.  Same contract as e3d_relkey_attn_fwd_split with terms = 3
// (arguments already validated there); v rows must be 16-byte aligned (v_rs % 4 == 0).
int e3d_attn_coop_launch(const float* q, int64_t q_bs, int64_t q_rs, const float* k, int64_t k_bs, int64_t k_rs,
                         const float* v, int64_t v_bs, int64_t v_rs, const float* dist_emb, int P,
                         const float* key_mask, float* out, float* lse, int B, int nh, int Lq, int Lk, int skip,
                         void* e_scratch, int e_ready, hipStream_t s) {
    const int q_tiles = (Lq + 31) / 32;
    static int w_max = 0;   // E3D_ATTN_W caps the waves per workgroup (experiments)
    if (!w_max) {
        const char* e = getenv("E3D_ATTN_W");
        w_max = e ? atoi(e) : 8;
    }
    int W = q_tiles % 8 == 0 ? 8 : (q_tiles % 4 == 0 ? 4 : (q_tiles % 2 == 0 ? 2 : 1));
    while (W > w_max && W > 1) W >>= 1;
    if (!dist_emb)
        return launch_any<false>(W, q, q_bs, q_rs, k, k_bs, k_rs, v, v_bs, v_rs, nullptr, P, key_mask, out, lse, B, nh, Lq,
                                 Lk, q_tiles, skip, s);
    // distance table -> fragment-order bf16 planes in the caller's scratch (e3d_attn_scratch_bytes(Lk) bytes)
    if (!e_scratch) {
        e3d_set_error("attn_coop: rel-key attention needs the caller's scratch for the distance-table planes");
        return -1;
    }
    const int J0 = (Lk + 31) / 32, n_items = 2 * J0 * 512;
    bf16x8* planes = reinterpret_cast<bf16x8*>(e_scratch);
    if (!e_ready)   // e_ready: the caller kept the planes of this (dist_emb, Lk) from an earlier call
        hipLaunchKernelGGL(e_fragments_kernel, dim3((n_items + 255) / 256), dim3(256), 0, s, dist_emb, planes, P, J0, n_items);
    return launch_any<true>(W, q, q_bs, q_rs, k, k_bs, k_rs, v, v_bs, v_rs, planes, P, key_mask, out, lse, B, nh, Lq, Lk,
                            q_tiles, skip, s);
}

#ifdef E3D_ATTN_STAMPS
extern "C" int e3d_debug_read_attn_stamps(long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(e3d_attn_stamps), sizeof(long long) * 16 * 8);
}
#endif

// the pre-pass alone: fragment-order planes of dist_emb for key length Lk into ``scratch``
int e3d_attn_fill_planes(const float* dist_emb, int P, int Lk, void* scratch, hipStream_t s) {
    const int J0 = (Lk + 31) / 32, n_items = 2 * J0 * 512;
    hipLaunchKernelGGL(e_fragments_kernel, dim3((n_items + 255) / 256), dim3(256), 0, s, dist_emb,
                       reinterpret_cast<bf16x8*>(scratch), P, J0, n_items);
    return e3d_launch_status("e3d_relkey_attn_fwd_split (distance-table planes)");
}

extern "C" int64_t e3d_attn_scratch_bytes(int Lk) { return (int64_t)2 * ((Lk + 31) / 32) * 512 * 16; }
