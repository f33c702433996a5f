// Fused relative-key attention forward, workgroup-cooperative form (bf16x3 split arithmetic).
//
// Same mathematics and accumulator layouts as attn_relkey_split.hip (query on the MFMA lane, key on
// the accumulator rows, online softmax in fp32, rel-key skew through a per-wave LDS ring), but the
// W waves of a workgroup are W consecutive 32-query tiles of ONE (batch, head):
//   * every key tile's K and V are fetched from HBM/L2 once per workgroup (not once per wave), split
//     into bf16 hi/lo planes once, and parked in a double-buffered LDS image that all waves read
//     their MFMA fragments from; the loads of tile t+1 are in flight while tile t is computed;
//   * V is parked row-major ([key][d], 192-byte rows) and the A operand of O^T += V^T P^T is fetched with
//     ds_read_b64_tr_b16, gfx950's transposing LDS read: a 16-lane group reads a 4-key x 16-dim block and
//     every lane receives its head dim of the 4 keys -- exactly the keys rho(8 st + j, half) the P^T registers
//     hold, so staging V costs two 8-byte LDS stores per thread and tile (a transposed image cost eight
//     2-byte scatters plus their address arithmetic);
//   * softmax in the exp2 domain: Q is pre-multiplied by log2(e) / sqrt(d) before it is split, so scores leave
//     the MFMAs ready for v_exp_f32; the running maximum is only raised when a tile exceeds it by more than
//     2^RESCALE_TAU (deferred rescale: O and l stay relative to a slightly stale maximum, which cancels in O / l),
//     so the 32 + 2 multiplies of the rescale are skipped on almost every tile;
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
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
// element type of the two split terms: __bf16 (bf16x3) or _Float16 (f16x3: 11 + 11 bits per operand, for operands
// inside the fp16 range -- Q is pre-scaled by 0.18, P <= 2^tau, K / V are projection outputs, E a parameter table)
template <typename E> struct AV;
template <> struct AV<__bf16> { typedef bf16x8 x8; typedef bf16x4 x4; };
template <> struct AV<_Float16> { typedef f16x8 x8; typedef f16x4 x4; };
template <typename X> struct Elem;
template <> struct Elem<bf16x8> { typedef __bf16 type; };
template <> struct Elem<bf16x4> { typedef __bf16 type; };
template <> struct Elem<f16x8> { typedef _Float16 type; };
template <> struct Elem<f16x4> { typedef _Float16 type; };
__device__ __forceinline__ f32x16 mma16(const bf16x8 a, const bf16x8 b, const f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mma16(const f16x8 a, const f16x8 b, const f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

constexpr int D = 64;
constexpr int RING_LD = 34;
constexpr int RING_F = 64 * RING_LD;          // 2176 floats: also holds the 32 x 68 output tile
constexpr int OUT_LD = 68;
constexpr int K_ROW_B = 144;                  // 64 bf16 + 16 B pad: conflict-free b128 fragment reads
constexpr int V_ROW_B = 192;                  // 64 bf16 + 64 B pad: 4 consecutive key rows x 64 B hit 64 distinct banks
constexpr int K_PLANE_B = 32 * K_ROW_B, V_PLANE_B = 32 * V_ROW_B;
constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
constexpr float Q_SCALE = 0.125f * LOG2E;     // 1 / sqrt(64), scores in log2 units
constexpr float MASK_BIAS = -10000.0f * LOG2E;
float g_rescale_tau = 8.0f;                   // raise the running maximum only past m + 8 (probabilities <= 2^8); e3d_attn_rescale_tau
constexpr int KV_BUF_B = 2 * K_PLANE_B + 2 * V_PLANE_B + 128;   // + 32 floats of key bias

template <typename X4>
__device__ __forceinline__ void split4x2(const f32x4 v, X4& hi, X4& lo) {
    typedef typename Elem<X4>::type E;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const E p = (E)v[j];
        hi[j] = p;
        lo[j] = (E)(v[j] - (float)p);
    }
}

template <typename X8>
__device__ __forceinline__ void split8x2(const float (&x)[8], X8& hi, X8& lo) {
    typedef typename Elem<X8>::type E;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const E p = (E)x[j];
        hi[j] = p;
        lo[j] = (E)(x[j] - (float)p);
    }
}

// fp16 terms: the 4-instruction pair split of e3d_common.h
__device__ __forceinline__ void split4x2(const f32x4 v, f16x4& hi, f16x4& lo) {
    e3d_f16x2 h0, l0, h1, l1;
    e3d_split2_f16(v[0], v[1], h0, l0);
    e3d_split2_f16(v[2], v[3], h1, l1);
    hi = f16x4{h0[0], h0[1], h1[0], h1[1]};
    lo = f16x4{l0[0], l0[1], l1[0], l1[1]};
}
__device__ __forceinline__ void split8x2(const float (&x)[8], f16x8& hi, f16x8& lo) {
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        e3d_f16x2 h, l;
        e3d_split2_f16(x[j], x[j + 1], h, l);
        hi[j] = h[0]; hi[j + 1] = h[1];
        lo[j] = l[0]; lo[j + 1] = l[1];
    }
}

// acc += a.b from the three significant cross terms, smallest first ([0] = hi, [1] = lo)
template <typename X8>
__device__ __forceinline__ f32x16 mfma3(const X8 (&a)[2], const X8 (&b)[2], f32x16 acc) {
    acc = mma16(a[0], b[1], acc);
    acc = mma16(a[1], b[0], acc);
    acc = mma16(a[0], b[0], acc);
    return acc;
}

typedef short short4v __attribute__((ext_vector_type(4)));
typedef short short8v __attribute__((ext_vector_type(8)));
// two transposing reads (4 keys x this lane's head dim each) -> one 8-key MFMA operand
template <typename X8>
__device__ __forceinline__ X8 tr_read8(const unsigned char* lo4, const unsigned char* hi4) {
    typedef __attribute__((address_space(3))) short4v* lds_p;
    const short4v a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(lo4));
    const short4v b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(hi4));
    const short8v c = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(X8, c);
}

// Distance table -> bf16 hi/lo planes in MFMA-FRAGMENT order.  Every 32-row block of E the kernel touches starts
// at a row e0 = P + 32 m (q0 and r0 are multiples of 32), m = -J0 .. J0 - 1 with J0 = ceil(L / 32), so the
// fragments of block j = m + J0 are laid out as [j][plane][kb][lane] x 16 bytes: a wave's E load is then 8
// fully coalesced 1-KB reads instead of 8 reads of 64 scattered 16-byte pieces (measured: 17 % of the kernel).
// Rows outside [0, 2P-2] (never paired with a valid (query, key)) are clamped.
template <typename E>
// ``e_absmax`` (or NULL): raised to the largest |element| of the rows read here -- every row the attention kernels can pair
// with a valid (query, key) -- so that the bound of the padded-tile skip costs no launch of its own (round 4).
__global__ __launch_bounds__(256) void e_fragments_kernel(const float* __restrict__ e, typename AV<E>::x8* __restrict__ frag,
                                                          int P, int J0, int n_items, float* __restrict__ e_absmax) {
    const int i = blockIdx.x * 256 + threadIdx.x;   // item = ((j * 2 + plane) * 4 + kb) * 64 + lane
    if (i >= n_items) return;                       // (n_items is a multiple of 512: whole workgroups)
    const int lane = i & 63, kb = (i >> 6) & 3, plane = (i >> 8) & 1, j = i >> 9;
    const int row = min(max(P + 32 * (j - J0) + (lane & 31), 0), 2 * P - 2);
    const float* src = e + row * D + 16 * kb + 8 * (lane >> 5);
    typename AV<E>::x8 out;
    unsigned amax = 0;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const float v = src[t];
        const E hi = (E)v;
        out[t] = plane ? (E)(v - (float)hi) : hi;
        e3d_absmax_accum(amax, v);
    }
    frag[i] = out;
    if (e_absmax && plane == 0) e3d_absmax_commit(amax, e_absmax, lane);   // (plane is uniform per wave: i >> 8)
}

// (the timing-only ablation switches of round 4 -- no loads / no ring / no MFMAs ... -- live in
// tools/lab/archive/attn_relkey_coop_r04_ablation_switches.hip.txt, built by tools/lab/attn_ablations.sh: nothing in this file
// computes a wrong result by a -D flag; their measurements: profiles/r04_attn_w4_vs_w8.log)
#ifdef E3D_ATTN_STAMPS   // lab builds only (tools/lab_attn_stamps.py): phase time stamps of one wave
__device__ long long e3d_attn_stamps[16][8];
#define ASTAMP(slot)                                                                                     \
    do {                                                                                                 \
        if (blockIdx.x == 1000 && wid == 1 && lane == 0 && kt < 16) e3d_attn_stamps[kt][slot] = __builtin_readcyclecounter(); \
    } while (0)
// kernel-level stamps of the same wave: slot 8 + i of row 15 is unused by the tile stamps when L <= 480
#define KSTAMP(i)                                                                                        \
    do {                                                                                                 \
        if (blockIdx.x == 1000 && wid == 1 && lane == 0) e3d_attn_stamps[15][i] = __builtin_readcyclecounter(); \
    } while (0)
#else
#define ASTAMP(slot) do {} while (0)
#define KSTAMP(i) do {} while (0)
#endif

// DROP (training): dropout multipliers on the probabilities, regenerated from (seed, element index) exactly as the
// per-wave kernel and the backward do (e3d_common.h); the row sum stays un-dropped (softmax first, then dropout).
template <int W, bool RELKEY, typename E, bool DROP = false>
__global__ __launch_bounds__(W * 64, 2) void attn_coop_kernel(
    const float* __restrict__ q, int64_t q_bs, int64_t q_rs, const float* __restrict__ k, int64_t k_bs,
    int64_t k_rs, const float* __restrict__ v, int64_t v_bs, int64_t v_rs, const typename AV<E>::x8* __restrict__ e_frag,
    int P, const float* __restrict__ key_mask, float* __restrict__ out, float* __restrict__ lse, int nh, int Lq, int Lk,
    int groups_per_bh, int skip_padded_tiles, E3dBounds bnd, float rescale_tau, E3dDrop drop_in) {
    const E3dDrop drop = e3d_drop_resolve(drop_in);   // + the device-side epoch (graph replays: e3d_common.h)
    typedef typename AV<E>::x8 bf16x8;   // (names kept from the bf16 form: 8 / 4 split terms of type E)
    typedef typename AV<E>::x4 bf16x4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int NT = W * 64, NI = 512 / NT;   // float4 staging items per thread, for K and for V
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int qi = lane & 31, half = lane >> 5;
    const int blk = xcd_remap(blockIdx.x, gridDim.x);
    const int grp = blk % groups_per_bh, bh = blk / groups_per_bh, h = bh % nh, b = bh / nh;
    const int q0 = (grp * W + wid) * 32;
    float* ring = reinterpret_cast<float*>(smem_raw + 2 * KV_BUF_B) + wid * RING_F;
    KSTAMP(0);   // kernel entry

    // ---- prologue, part 1: ISSUE every global load the prologue needs before consuming any of them.  Kernel-level
    // stamps showed the prologue at 35 % of a workgroup's life (22 k of 65 k cycles) with its loads in a dependent
    // chain -- Q -> split, then the key-mask scan (one load per 64 keys, each behind a ballot), then K/V tile 0, then
    // the first distance block -- i.e. ~6 exposed cold-miss round trips on a CU that has nothing else to run (one
    // workgroup per CU).  Issued together they cost one.
    const float* kb_ = k + b * k_bs + h * D;
    const float* vb_ = v + b * v_bs + h * D;
    const float* mb = key_mask ? key_mask + (int64_t)b * Lk : nullptr;
    f32x4 qraw[8];   // Q rows (B operand of S^T and T^T): row = query, 8 head-dim values per 16-wide k block
    {
        const float* qrow = q + b * q_bs + (int64_t)min(q0 + qi, Lq - 1) * q_rs + h * D + 8 * half;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            qraw[2 * kb] = *reinterpret_cast<const f32x4*>(qrow + 16 * kb);
            qraw[2 * kb + 1] = *reinterpret_cast<const f32x4*>(qrow + 16 * kb + 4);
        }
    }
    // key-mask values of keys 64 c + lane, c < 4 (Lk <= 256: every configuration of the reference); unconditional
    // loads from a valid address (no branch around a load), longer rows finish with the loop below
    const bool scan_mask = mb && skip_padded_tiles && e3d_mask_skip_is_exact(bnd);   // (block-uniform)
    float mv[4];
    {
        const float* mp = mb ? mb : kb_;
#pragma unroll
        for (int c = 0; c < 4; ++c) mv[c] = mp[min(64 * c + lane, Lk - 1)];
    }

    bf16x8 qf[4][2];
    int k_tiles = (Lk + 31) >> 5;

    // cooperative staging: item i of this thread = row (f >> 4), head dims 4 (f & 15) .. +3, f = tid + NT i
    // Two register sets when they are small (W >= 4: one or two float4 per operand): tile t+2 is loaded while
    // tile t is computed and tile t+1 (loaded a tile earlier) is split and stored -- the global latency
    // (stamps: 2-3k cycles under load) then never shows.  Narrow workgroups keep one set (distance 1).
    constexpr bool DEEP = NI <= 2;
    constexpr int NSET = DEEP ? 2 : 1;
    f32x4 sk[NSET][NI], sv[NSET][NI];
    float smask[NSET];
    // Staging loads through BUFFER descriptors (round 4): a tile's base (rows r0 ..) and its byte count are wave-uniform and
    // live in the descriptor's SGPRs, the per-item offset inside a tile never changes -- so a tile's four loads need no
    // vector address arithmetic at all (the flat form spent ~20 VALU instructions per tile on clamps and 64-bit adds), and
    // rows past the last key come back as zeros from the hardware's range check (the flat form re-read row Lk - 1; either
    // way those keys carry a bias of -inf).  The range check covers the vector offset only, hence base = tile, not item.
    int voff_k[NI], voff_v[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int f = tid + NT * i;
        voff_k[i] = ((f >> 4) * (int)k_rs + 4 * (f & 15)) * 4;
        voff_v[i] = ((f >> 4) * (int)v_rs + 4 * (f & 15)) * 4;
    }
    const int voff_m = (tid & 31) * 4;
    auto tile_rsrc = [&](const float* base, int64_t rs, int r0, int tail_floats) {
        const int rows = Lk - r0;                                                    // >= 1 (callers clamp the tile index)
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base + (int64_t)r0 * rs), 0,
                                                 (int)(((int64_t)(rows - 1) * rs + tail_floats) * 4), 0x00020000);
    };
    auto stage_load = [&](auto set_tag, int r0) {
        constexpr int SET = decltype(set_tag)::value;
        const auto rk = tile_rsrc(kb_, k_rs, r0, D), rv = tile_rsrc(vb_, v_rs, r0, D);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            sk[SET][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rk, voff_k[i], 0, 0));
            sv[SET][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rv, voff_v[i], 0, 0));
        }
        // key-mask value of key r0 + (tid & 31): loaded by every thread (no branch around a load: hipcc then
        // counts vmcnt exactly), used by the first 32; without a mask the K rows stand in (value unused)
        const auto rm = tile_rsrc(mb ? mb : kb_, 1, r0, 1);
        const float mval = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rm, voff_m, 0, 0));
        smask[SET] = mb ? mval : 1.0f;
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
            *reinterpret_cast<bf16x4*>(buf + 2 * K_PLANE_B + row * V_ROW_B + 8 * c4) = hi;
            *reinterpret_cast<bf16x4*>(buf + 2 * K_PLANE_B + V_PLANE_B + row * V_ROW_B + 8 * c4) = lo;
        }
        if (tid < 32)
            reinterpret_cast<float*>(buf + 2 * K_PLANE_B + 2 * V_PLANE_B)[tid] =
                r0 + tid < Lk ? (1.0f - smask[SET]) * MASK_BIAS : -INFINITY;
    };
    using Set0 = std::integral_constant<int, 0>;
    using Set1 = std::integral_constant<int, DEEP ? 1 : 0>;

    // rel-key: E fragments of block j (distance-table rows P + 32 (j - J0) .. + 31) from the fragment-order planes
    bf16x8 ef[4][2];
    const int J0 = (Lk + 31) >> 5, qt = q0 >> 5;
    // (buffer loads as well: block j is a wave-uniform scalar offset, the lane's offset and the k-block / plane offsets are
    //  constants: no vector address arithmetic per tile)
    const auto e_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16x8*>(e_frag ? e_frag : reinterpret_cast<const bf16x8*>(k)), 0,
                                                          2 * ((Lk + 31) >> 5) * 512 * 16, 0x00020000);
    const int voff_e = lane * 16;
    auto e_load = [&](int j) {
        const int so = __builtin_amdgcn_readfirstlane(j * 8192);   // (j comes from the wave index: uniform, but hipcc cannot prove it
                                                                   //  and would wrap every load in a waterfall loop)
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            ef[kb][0] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(e_rsrc, voff_e + kb * 1024, so, 0));
            ef[kb][1] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(e_rsrc, voff_e + kb * 1024, so + 4096, 0));
        }
    };
    auto dot_q_acc = [&](const bf16x8 (&x)[4][2], f32x16 acc) {
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) acc = mfma3(x[kb], qf[kb], acc);
        return acc;
    };
    auto dot_q = [&](const bf16x8 (&x)[4][2]) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        return dot_q_acc(x, acc);
    };

    stage_load(Set0{}, 0);
    if (RELKEY) e_load(qt + J0);  // rows q0 + P ..: the block "before" key tile 0 fills the upper half of the ring

    // ---- prologue, part 2: consume.  Trailing all-padding key tiles contribute exp(-10000 - m) == 0.0f exactly: stop
    // after the tile of the last valid key (bit-identical; an all-padding item keeps the full sweep).  Depends on
    // the batch item only, so every wave of the workgroup runs the same number of barriers.
    if (scan_mask) {
        int last = -1;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const bool valid = 64 * c + lane < Lk && mv[c] != 0.f;
            const unsigned long long bits = __ballot(valid);
            if (bits) last = 64 * c + 63 - __builtin_clzll(bits);
        }
        for (int base = 256; base < Lk; base += 64) {
            const int key = base + lane;
            const bool valid = key < Lk && mb[key] != 0.f;
            const unsigned long long bits = __ballot(valid);
            if (bits) last = base + 63 - __builtin_clzll(bits);
        }
        if (last >= 0) k_tiles = (last >> 5) + 1;
    }
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        const f32x4 lo4 = qraw[2 * kb], hi4 = qraw[2 * kb + 1];
        float x[8] = {lo4[0] * Q_SCALE, lo4[1] * Q_SCALE, lo4[2] * Q_SCALE, lo4[3] * Q_SCALE,
                      hi4[0] * Q_SCALE, hi4[1] * Q_SCALE, hi4[2] * Q_SCALE, hi4[3] * Q_SCALE};
        // Pin the rounded fp32 products before the split.  With fp16 terms hipcc may otherwise form the hi term by a
        // mixed-precision fused multiply-convert (ONE rounding of the exact product) but the residual from the
        // fp32-rounded product (TWO roundings): at a double-rounding tie the two disagree by one fp16 ulp and the
        // lo term no longer complements the hi term (found as a 2^-10 error on 1 of 49152 query elements).
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(x[j]));
        split8x2(x, qf[kb][0], qf[kb][1]);
    }
    if (RELKEY) {
        // the block "before" key tile 0 fills the upper half of the ring, the block of tile 0 the lower half (every
        // later tile's block is computed at the END of the tile before it, see below)
        f32x16 t = dot_q(ef);
        e_load(qt - 1 + J0);      // key tile 0: rows q0 - 32 + P ..
#pragma unroll
        for (int r = 0; r < 16; ++r) ring[(32 + mfma32_row(r, half)) * RING_LD + qi] = t[r];
        t = dot_q(ef);
        e_load(qt - (k_tiles > 1 ? 1 : 0) - 1 + J0);   // key tile 1
#pragma unroll
        for (int r = 0; r < 16; ++r) ring[mfma32_row(r, half) * RING_LD + qi] = t[r];
    }
    stage_store(Set0{}, 0, smem_raw);
    if (DEEP) stage_load(Set1{}, (k_tiles > 1 ? 1 : 0) * 32);
    __syncthreads();
    KSTAMP(1);   // prologue done: Q fragments, first K/V tile staged, first distance block in the ring

    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    float m_run = 0.f, l_run = 0.f;      // (m_run: placeholder until the first tile sets it)
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

        // The accumulator of S^T = K Q^T starts from everything that is ADDED to the scores -- the additive key bias, the
        // rel-key term (this tile's T^T block, written to the ring at the end of the previous tile or by the prologue, so the
        // write -> read round trip of the skew is off the critical path) and minus the running maximum -- so the scores
        // leave the MFMAs relative to the running maximum, ready for v_exp_f32: 16 + 16 additions in front of the MFMAs
        // instead of 16 + 16 + 16 behind them (round 4; the instruction counts are in profiles/r04_attn_coop_pmc_insts_*)
        const float* kbias = reinterpret_cast<const float*>(buf + 2 * K_PLANE_B + 2 * V_PLANE_B);
        f32x16 s;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(kbias + 8 * g + 4 * half);
#pragma unroll
            for (int j = 0; j < 4; ++j) s[4 * g + j] = bv[j] - m_run;     // scores are in log2 units / sqrt(d) (Q_SCALE)
        }
        if (RELKEY) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                s[r] += PAR ? ring[rd_odd[r]] : ring_e[(27 - ((r & 3) + 8 * (r >> 2))) * RING_LD];
        }
        {   // S^T = K Q^T
            bf16x8 kf[4][2];
            const unsigned char* kr = buf + qi * K_ROW_B + 16 * half;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                kf[kb][0] = *reinterpret_cast<const bf16x8*>(kr + 32 * kb);
                kf[kb][1] = *reinterpret_cast<const bf16x8*>(kr + K_PLANE_B + 32 * kb);
            }
            s = dot_q_acc(kf, s);
        }
        ASTAMP(2);
        // K/V of tile t+2 into the set whose tile-t data was stored a tile ago (clamped re-loads at the end)
        const int kt_ld = DEEP ? (kt + 2 < k_tiles ? kt + 2 : k_tiles - 1) : kt_next;
        if (PAR == 0) stage_load(Set0{}, kt_ld * 32);
        else stage_load(Set1{}, kt_ld * 32);
        ASTAMP(3);

        float tmax = -INFINITY;     // the tile's largest score RELATIVE to the running maximum
#pragma unroll
        for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, s[r]);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        // deferred rescale: only when some query of the tile outgrows its running maximum by more than 2^TAU -- and always
        // on the first tile, whose "running maximum" is the placeholder 0.  Wave-uniform branch; the lanes that do not need
        // it move by 0 (alpha = 1).
        const bool grow = kt == 0 || tmax > rescale_tau;
        if (__builtin_amdgcn_ballot_w64(grow) != 0ull) {
            const float delta = grow ? tmax : 0.f;
            const float alpha = kt == 0 ? 0.f : __builtin_amdgcn_exp2f(-delta);   // (first tile: O = l = 0 anyway)
            m_run += delta;
            l_run *= alpha;
#pragma unroll
            for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; s[r] -= delta; }
        }
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[r] = __builtin_amdgcn_exp2f(s[r]);
            psum += s[r];
        }
        l_run += psum;
        if (DROP) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float m[4];
                e3d_drop_mult4(drop, e3d_attn_drop_idx4(bh, Lq, Lk, min(q0 + qi, Lq - 1), kt * 32 + 8 * g + 4 * half), m);
#pragma unroll
                for (int j = 0; j < 4; ++j) s[4 * g + j] *= m[j];
            }
        }
        ASTAMP(4);

        // O^T += V^T P^T: two 16-key steps; P^T registers 8 st .. 8 st + 7 are the B operand (keys
        // rho(8 st + j, half) = 16 st + 8 (j >> 2) + 4 half + (j & 3)); the A operand -- head dims qi (o0) and 32 + qi
        // (o1) of those keys -- comes from the row-major V image through two transposing reads per fragment:
        // lane 4 q + p of a 16-lane group addresses key row q, dims 4 p .. 4 p + 3 of the group's 16 dims
        const unsigned char* vr = buf + 2 * K_PLANE_B + (4 * half + ((lane >> 2) & 3)) * V_ROW_B +
                                  2 * (16 * ((lane >> 4) & 1) + 4 * (lane & 3));
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            float pv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) pv[j] = s[8 * st + j];
            bf16x8 pb[2], a0[2], a1[2];
            split8x2(pv, pb[0], pb[1]);
            const unsigned char* v0 = vr + 16 * st * V_ROW_B;
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                a0[pl] = tr_read8<bf16x8>(v0 + pl * V_PLANE_B, v0 + pl * V_PLANE_B + 8 * V_ROW_B);
                a1[pl] = tr_read8<bf16x8>(v0 + pl * V_PLANE_B + 64, v0 + pl * V_PLANE_B + 8 * V_ROW_B + 64);
            }
            o0 = mfma3(a0, pb, o0);
            o1 = mfma3(a1, pb, o1);
        }

        ASTAMP(5);
        if (more) {   // tile t+1: from the other set when there are two, from the one just loaded otherwise
            if (DEEP == (PAR == 0)) stage_store(Set1{}, kt_next * 32, smem_raw + (PAR ^ 1) * KV_BUF_B);
            else stage_store(Set0{}, kt_next * 32, smem_raw + (PAR ^ 1) * KV_BUF_B);
        }
        ASTAMP(6);
        if (RELKEY && more) {
            // T^T block of the NEXT tile (needs only Q and the distance table): 12 MFMAs + the ring writes here, where
            // the faster waves of the workgroup would otherwise idle at the barrier; its successor's fragments then
            // have a whole tile to arrive.  Rows 32 (PAR ^ 1) .. + 31 hold the block of tile t - 1, which this tile has
            // finished reading.
            const f32x16 t = dot_q(ef);
            e_load(qt - (kt + 2 < k_tiles ? kt + 2 : k_tiles - 1) - 1 + J0);
#pragma unroll
            for (int r = 0; r < 16; ++r) ring_w[(32 * (PAR ^ 1) + (r & 3) + 8 * (r >> 2)) * RING_LD] = t[r];
        }
        __syncthreads();
        ASTAMP(7);
    };
    for (int kt = 0; kt < k_tiles; kt += 2) {
        tile(std::integral_constant<int, 0>{}, kt);
        if (kt + 1 < k_tiles) tile(std::integral_constant<int, 1>{}, kt + 1);
    }

    KSTAMP(2);   // key sweep done
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
    if (lse && half == 0 && q0 + qi < Lq) lse[((int64_t)b * nh + h) * Lq + q0 + qi] = m_run * LN2 + logf(l_tot);
    KSTAMP(3);   // output stores issued
#ifdef E3D_ATTN_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    KSTAMP(4);   // output stores retired
#endif
}

template <int W, bool RELKEY, typename E>
int launch_w(const float* q, int64_t q_bs, int64_t q_rs, const float* k, int64_t k_bs, int64_t k_rs, const float* v,
             int64_t v_bs, int64_t v_rs, const void* e_frag, int P, const float* key_mask,
             float* out, float* lse, int B, int nh, int Lq, int Lk, int q_tiles, int skip, E3dBounds bnd, E3dDrop drop,
             bool dropping, hipStream_t s) {
    const size_t lds = 2 * KV_BUF_B + (size_t)W * RING_F * sizeof(float);
    const int groups = q_tiles / W;
    if constexpr (std::is_same<E, __bf16>::value) {   // dropout exists in the training arithmetic (bf16x3) only
        if (dropping) {
            static std::atomic<uint64_t> lds_ok_d{0};
            e3d_allow_lds(lds_ok_d, attn_coop_kernel<W, RELKEY, E, true>, lds);
            hipLaunchKernelGGL((attn_coop_kernel<W, RELKEY, E, true>), dim3(B * nh * groups), dim3(W * 64), lds, s, q, q_bs, q_rs,
                               k, k_bs, k_rs, v, v_bs, v_rs, reinterpret_cast<const typename AV<E>::x8*>(e_frag), P, key_mask,
                               out, lse, nh, Lq, Lk, groups, skip, bnd, g_rescale_tau, drop);
            return e3d_launch_status("e3d_relkey_attn_fwd_split (cooperative, dropout)");
        }
    }
    static std::atomic<uint64_t> lds_ok{0};
    e3d_allow_lds(lds_ok, attn_coop_kernel<W, RELKEY, E>, lds);
    hipLaunchKernelGGL((attn_coop_kernel<W, RELKEY, E>), dim3(B * nh * groups), dim3(W * 64), lds, s, q, q_bs, q_rs, k, k_bs,
                       k_rs, v, v_bs, v_rs, reinterpret_cast<const typename AV<E>::x8*>(e_frag), P, key_mask, out, lse, nh, Lq, Lk, groups, skip, bnd, g_rescale_tau, drop);
    return e3d_launch_status("e3d_relkey_attn_fwd_split (cooperative)");
}

template <bool RELKEY, typename E>
int launch_any(int W, const float* q, int64_t q_bs, int64_t q_rs, const float* k, int64_t k_bs, int64_t k_rs,
               const float* v, int64_t v_bs, int64_t v_rs, const void* e_frag, int P,
               const float* key_mask, float* out, float* lse, int B, int nh, int Lq, int Lk, int q_tiles, int skip,
               E3dBounds bnd, E3dDrop drop, bool dropping, hipStream_t s) {
#define E3D_COOP_CASE(w)                                                                                          \
    case w:                                                                                                       \
        return launch_w<w, RELKEY, E>(q, q_bs, q_rs, k, k_bs, k_rs, v, v_bs, v_rs, e_frag, P, key_mask, out, lse, B, nh, \
                                   Lq, Lk, q_tiles, skip, bnd, drop, dropping, s)
    switch (W) {
        E3D_COOP_CASE(8);
        E3D_COOP_CASE(4);      // (two- and one-wave groups were lab forms: slower than the per-wave kernel, 20-88 B/lane of scratch)
    }
#undef E3D_COOP_CASE
    return -1;
}

}  // namespace

template <typename E>
static int coop_launch_t(const float* q, int64_t q_bs, int64_t q_rs, const float* k, int64_t k_bs, int64_t k_rs,
                         const float* v, int64_t v_bs, int64_t v_rs, const float* dist_emb, int P,
                         const float* key_mask, float* out, float* lse, int B, int nh, int Lq, int Lk, int skip,
                         E3dBounds bnd, void* e_scratch, int e_ready, E3dDrop drop, bool dropping, hipStream_t s) {
    const int q_tiles = (Lq + 31) / 32;
    // Waves per workgroup: 4 (the caller dispatches here for q_tiles % 4 == 0 only).  A 4-wave workgroup holds 78 KB of LDS, so
    // two of them share a CU -- one wave of each per SIMD, with barriers of their own: the two drift apart and one's matrix phases
    // fall beside the other's LDS / vector phases, and one's prologue (cold loads, first distance blocks) runs under the other's
    // key sweep.  The 8-wave form (one workgroup per CU, every K / V tile staged once per (b, head) instead of twice at L = 256)
    // keeps all eight waves in lockstep: measured 361 vs 338 us per launch inside the sampling step at B = 256, L = 256
    // (profiles/r04_attn_w4_vs_w8.log).  E3D_ATTN_W=8 selects it for A/B runs.
    static int w_pref = 0;
    if (!w_pref) {
        const char* e = getenv("E3D_ATTN_W");
        w_pref = e ? atoi(e) : 4;
    }
    const int W = (w_pref == 8 && q_tiles % 8 == 0) ? 8 : 4;
    if (!dist_emb)
        return launch_any<false, E>(W, q, q_bs, q_rs, k, k_bs, k_rs, v, v_bs, v_rs, nullptr, P, key_mask, out, lse, B, nh,
                                    Lq, Lk, q_tiles, skip, bnd, drop, dropping, s);
    // distance table -> fragment-order hi / lo planes in the caller's scratch (e3d_attn_scratch_bytes(Lk) bytes)
    if (!e_scratch) {
        e3d_set_error("attn_coop: rel-key attention needs the caller's scratch for the distance-table planes");
        return -1;
    }
    const int J0 = (Lk + 31) / 32, n_items = 2 * J0 * 512;
    if (!e_ready)   // e_ready: the caller kept the planes of this (dist_emb, Lk, terms) from an earlier call
        hipLaunchKernelGGL(e_fragments_kernel<E>, dim3((n_items + 255) / 256), dim3(256), 0, s, dist_emb,
                           reinterpret_cast<typename AV<E>::x8*>(e_scratch), P, J0, n_items, (float*)nullptr);
    return launch_any<true, E>(W, q, q_bs, q_rs, k, k_bs, k_rs, v, v_bs, v_rs, e_scratch, P, key_mask, out, lse, B, nh, Lq,
                               Lk, q_tiles, skip, bnd, drop, dropping, s);
}

// bf16x3 / f16x3 attention, cooperative kernel.  Same contract as e3d_relkey_attn_fwd_split with terms = 3 / 19
// (arguments already validated there); v rows must be 16-byte aligned (v_rs % 4 == 0).
int e3d_attn_coop_launch(const float* q, int64_t q_bs, int64_t q_rs, const float* k, int64_t k_bs, int64_t k_rs,
                         const float* v, int64_t v_bs, int64_t v_rs, const float* dist_emb, int P,
                         const float* key_mask, float* out, float* lse, int B, int nh, int Lq, int Lk, int skip,
                         E3dBounds bnd, void* e_scratch, int e_ready, int f16, E3dDrop drop, bool dropping, hipStream_t s) {
    if (f16)
        return coop_launch_t<_Float16>(q, q_bs, q_rs, k, k_bs, k_rs, v, v_bs, v_rs, dist_emb, P, key_mask, out, lse, B, nh, Lq,
                                       Lk, skip, bnd, e_scratch, e_ready, drop, dropping, s);
    return coop_launch_t<__bf16>(q, q_bs, q_rs, k, k_bs, k_rs, v, v_bs, v_rs, dist_emb, P, key_mask, out, lse, B, nh, Lq, Lk,
                                 skip, bnd, e_scratch, e_ready, drop, dropping, s);
}

// Diagnostic (tests): threshold of the deferred rescale in log2 units; 0 = raise the maximum on every new one (classic
// online softmax).  Returns the previous value; a negative argument only queries.
extern "C" float e3d_attn_rescale_tau(float tau) {
    const float prev = g_rescale_tau;
    if (tau >= 0.f) g_rescale_tau = tau;
    return prev;
}

#ifdef E3D_ATTN_STAMPS
extern "C" int e3d_debug_read_attn_stamps(long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(e3d_attn_stamps), sizeof(long long) * 16 * 8);
}
#endif

// the pre-pass alone: fragment-order planes of dist_emb for key length Lk into ``scratch``
int e3d_attn_fill_planes(const float* dist_emb, int P, int Lk, void* scratch, int f16, float* e_absmax, hipStream_t s) {
    const int J0 = (Lk + 31) / 32, n_items = 2 * J0 * 512;
    if (f16)
        hipLaunchKernelGGL(e_fragments_kernel<_Float16>, dim3((n_items + 255) / 256), dim3(256), 0, s, dist_emb,
                           reinterpret_cast<f16x8*>(scratch), P, J0, n_items, e_absmax);
    else
        hipLaunchKernelGGL(e_fragments_kernel<__bf16>, dim3((n_items + 255) / 256), dim3(256), 0, s, dist_emb,
                           reinterpret_cast<bf16x8*>(scratch), P, J0, n_items, e_absmax);
    return e3d_launch_status("e3d_relkey_attn_fwd_split (distance-table planes)");
}

extern "C" int64_t e3d_attn_scratch_bytes(int Lk) { return (int64_t)2 * ((Lk + 31) / 32) * 512 * 16; }
