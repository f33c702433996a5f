// Fused, recomputing backward of the relative-key attention (bf16x3 split arithmetic, fp32 accumulation): ONE launch
// per call computes dQ, dK, dV and the per-unit dE blocks from Q, K, V, dO, O and the forward's log-sum-exp -- the
// probabilities P and dS never leave the chip (attn_bwd_split.hip materialises both in HBM between its two launches
// and re-splits every operand row per (query tile, key tile) pair: 2600 VALU instructions per pair against 99 MFMAs).
//
// One workgroup of 4 waves per (batch item, head, phase), Lq, Lk <= 128 (every training configuration of the reference:
// max_seq_len = 128).  Operands are split into bf16 hi / lo planes ONCE per workgroup and parked in LDS images that
// serve both access patterns of the MFMA operands:
//   * "row fragments"  (lane = token, 8 consecutive head dims): products that contract over the head dim
//     (S = Q K^T, dP = dO V^T, T = Q E^T) -- one ds_read_b128 per fragment;
//   * "transposed fragments" (lane = head dim, 8 tokens in accumulator-row order rho): products that contract over
//     tokens (dV^T = dO^T P, dK^T = Q^T dS, dQ^T = K^T dS^T) -- two ds_read_b64_tr_b16 per fragment, the scheme of the
//     forward kernel's V operand.
// Image rows are 192 bytes (64 bf16 + pad) with the 16-byte chunk index XOR-ed by (row >> 2) & 3: both patterns are
// then conflict-free (rows r and r + 4 would otherwise share banks in the row-fragment reads).
//
// Phase A (wave = key tile, Q and dO images): S, P, dP, dS with the QUERY on the accumulator rows and the key on the
//   lanes -- P and dS are then B operands as they stand -- dV^T += dO^T P, dK^T += Q^T dS in registers; no cross-wave
//   sums.  The rel-key bias needs both 32-row blocks of T = Q E^T per pair here (the query tile changes every step).
// Phase B (wave = query tile, K and V images over the same LDS): S^T, P^T, dP^T, dS^T with the key on the rows
//   (the forward's orientation: one new T^T block per step through the ring), dQ^T += K^T dS^T, and through the
//   inverse skew dQ^T += E^T dT^T and the dE blocks (reduced afterwards by dist_emb_reduce_kernel, as before).
// S and dP are computed in both phases (2 x 24 of the ~170 MFMAs per pair): cheaper than any way of handing P / dS
// across waves (LDS is full, HBM is what this kernel exists to avoid), deterministic, no atomics.
// Softmax in the exp2 domain: P = exp2(s * log2(e) / 8 + bias2 - lse * log2(e)).
#include "e3d_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short short4v __attribute__((ext_vector_type(4)));
typedef short short8v __attribute__((ext_vector_type(8)));

constexpr int D = 64;
constexpr int MAX_TILES = 4;                       // 32-token tiles per operand: L <= 128
constexpr int ROW_B = 192;                         // image row: 64 bf16 + 64 B pad
constexpr int PLANE_B = MAX_TILES * 32 * ROW_B;    // one plane (hi or lo) of one operand: 24 KB
constexpr int IMG_B = 2 * PLANE_B;                 // hi + lo
constexpr int RA_LD = 65, RB_LD = 34, X_LD = 33;
constexpr int WAVE_F = 64 * RB_LD + 32 * X_LD;     // per-wave scratch floats: phase B ring + X (phase A ring 32 x 65 fits)
constexpr int ROWS_F = 3 * MAX_TILES * 32 + 32;    // lse2, delta, key bias (+ 32 words: query-tile liveness flags, round 4)
constexpr float LOG2E = 1.4426950408889634f;
constexpr float S_SCALE = 0.125f * LOG2E;
constexpr size_t LDS_BYTES = 2 * IMG_B + (ROWS_F + 4 * WAVE_F) * sizeof(float);

#ifdef BWD_STAMPS   // lab builds only (tools/lab/attn_bwd_stamps.py): s_memrealtime (100 MHz) stamps per workgroup and wave
__device__ long long bwd_stamps[2][512][4][8];
#define BSTAMP(slot)                                                                                           \
    do {                                                                                                       \
        if ((threadIdx.x & 63) == 0 && blockIdx.x < 512)                                                       \
            bwd_stamps[blockIdx.y][blockIdx.x][threadIdx.x >> 6][(slot)] = __builtin_amdgcn_s_memrealtime();   \
    } while (0)
#else
#define BSTAMP(slot) do {} while (0)
#endif

struct Frag { bf16x8 hi, lo; };

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
__device__ __forceinline__ f32x16 mfma3(const Frag& a, const Frag& b, f32x16 acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi, b.lo, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.lo, b.hi, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi, b.hi, acc, 0, 0, 0);
    return acc;
}
__device__ __forceinline__ f32x16 zero16() {
    f32x16 a;
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = 0.f;
    return a;
}
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- image access ------------------------------------------------------------------------------------------------
// byte offset of logical 16-byte chunk c (8 bf16: head dims 8c .. 8c+7) of image row ``row``
__device__ __forceinline__ int img_off(int row, int c) { return row * ROW_B + ((c ^ ((row >> 2) & 3)) << 4); }

// row fragments of tile t: lane (r = lane & 31, half) takes head dims 16 kb + 8 half .. + 7 of row 32 t + r
__device__ __forceinline__ void row_frags(Frag (&f)[4], const unsigned char* img, int t, int lane) {
    const int row = 32 * t + (lane & 31), half = lane >> 5;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        const int o = img_off(row, 2 * kb + half);
        f[kb].hi = *reinterpret_cast<const bf16x8*>(img + o);
        f[kb].lo = *reinterpret_cast<const bf16x8*>(img + PLANE_B + o);
    }
}

__device__ __forceinline__ bf16x8 tr_read8(const unsigned char* lo4, const unsigned char* hi4) {
    typedef __attribute__((address_space(3))) short4v* lds_p;
    const short4v a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(lo4));
    const short4v b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(hi4));
    const short8v c = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, c);
}
// transposed fragment of tile t, 16-token step st, head-dim half dh (dims 32 dh + (lane & 31)):
// A[i = head dim][k = token 32 t + rho(8 st + j, half)], j = 0..7
__device__ __forceinline__ Frag tr_frag(const unsigned char* img, int t, int st, int dh, int lane) {
    const int half = lane >> 5;
    const int row = 32 * t + 16 * st + 4 * half + ((lane >> 2) & 3);
    const int c = 4 * dh + 2 * ((lane >> 4) & 1) + ((lane & 3) >> 1), sub = 8 * (lane & 1);
    // (row >> 2) & 3 == half for the first 4 rows, half + 2 for the rows 8 further on (32 t + 16 st is a multiple of 16)
    const unsigned char* p0 = img + row * ROW_B + ((c ^ half) << 4) + sub;
    const unsigned char* p1 = img + (row + 8) * ROW_B + ((c ^ (half + 2)) << 4) + sub;
    Frag f;
    f.hi = tr_read8(p0, p1);
    f.lo = tr_read8(p0 + PLANE_B, p1 + PLANE_B);
    return f;
}

// cooperative staging of X[rows_valid][64] fp32 (row stride rs) into an image: 256 threads, float4 items
__device__ __forceinline__ void stage_image(unsigned char* img, const float* __restrict__ x, int64_t rs, int rows_valid,
                                            int tiles, int tid) {
    const int n_items = tiles * 32 * 16;
    for (int f = tid; f < n_items; f += 256) {
        const int row = f >> 4, c4 = f & 15;
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + (int64_t)min(row, rows_valid - 1) * rs + 4 * c4);
        bf16x4 hi, lo;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const __bf16 p = (__bf16)v[j];
            hi[j] = p;
            lo[j] = (__bf16)(v[j] - (float)p);
        }
        const int o = img_off(row, c4 >> 1) + 8 * (c4 & 1);
        *reinterpret_cast<bf16x4*>(img + o) = hi;
        *reinterpret_cast<bf16x4*>(img + PLANE_B + o) = lo;
    }
}

// one row's 64 values from global, split into the 4 row fragments of lane (row, half)
__device__ __forceinline__ void load_row_split(Frag (&f)[4], const float* row_ptr, int half) {
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        const f32x4 lo = *reinterpret_cast<const f32x4*>(row_ptr + 16 * kb + 8 * half);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(row_ptr + 16 * kb + 8 * half + 4);
        const float x[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        f[kb] = split8(x);
    }
}

__device__ __forceinline__ f32x16 dot_tile(const Frag (&a)[4], const Frag (&b)[4]) {
    f32x16 acc = zero16();
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) acc = mfma3(a[kb], b[kb], acc);
    return acc;
}

// two independent tiles at once, their MFMA chains interleaved: with ONE wave per SIMD (the LDS images leave room for one
// workgroup per CU) nothing else hides the latency of a dependent accumulator chain
__device__ __forceinline__ void dot_tile2(f32x16& c0, f32x16& c1, const Frag (&a0)[4], const Frag (&b0)[4], const Frag (&a1)[4],
                                          const Frag (&b1)[4]) {
    c0 = zero16();
    c1 = zero16();
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[kb].hi, b0[kb].lo, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[kb].hi, b1[kb].lo, c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[kb].lo, b0[kb].hi, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[kb].lo, b1[kb].hi, c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[kb].hi, b0[kb].hi, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[kb].hi, b1[kb].hi, c1, 0, 0, 0);
    }
}

// o^T[d][token] accumulators (o0: d = rho(r, half), o1: d = 32 + rho) -> token-major rows of 64 floats (lane = token)
__device__ __forceinline__ void store_rows_T(const f32x16& o0, const f32x16& o1, float* row_ptr, int half) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        f32x4 a, c;
#pragma unroll
        for (int j = 0; j < 4; ++j) { a[j] = o0[4 * g + j]; c[j] = o1[4 * g + j]; }
        *reinterpret_cast<f32x4*>(row_ptr + 8 * g + 4 * half) = a;
        *reinterpret_cast<f32x4*>(row_ptr + 32 + 8 * g + 4 * half) = c;
    }
}

// Distance table -> bf16 hi / lo planes, both fragment orders, for blocks j = m + J0 (rows P + 32 m .. + 31, m = -J0 .. J0-1):
//   row order  [j][plane][kb][lane]     x 16 B: lane (x = lane & 31, half) holds E[row x][16 kb + 8 half .. + 7]
//   tr order   [j][plane][st][dh][lane] x 16 B: lane (d = lane & 31, half) holds E[row rho(8 st + t, half)][32 dh + d], t = 0..7
__global__ __launch_bounds__(256) void e_planes_kernel(const float* __restrict__ e, bf16x8* __restrict__ row_order,
                                                       bf16x8* __restrict__ tr_order, int P, int J0, int n_items) {
    const int i = blockIdx.x * 256 + threadIdx.x;   // item = ((j * 2 + plane) * 4 + sub) * 64 + lane
    if (i >= n_items) return;
    const int lane = i & 63, sub = (i >> 6) & 3, plane = (i >> 8) & 1, j = i >> 9;
    const int half = lane >> 5, l31 = lane & 31;
    const int e0 = P + 32 * (j - J0);
    bf16x8 a, b;
    {
        const float* src = e + (int64_t)min(max(e0 + l31, 0), 2 * P - 2) * D + 16 * sub + 8 * half;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const float v = src[t];
            const __bf16 hi = (__bf16)v;
            a[t] = plane ? (__bf16)(v - (float)hi) : hi;
        }
    }
    {
        const int st = sub >> 1, dh = sub & 1;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int row = min(max(e0 + mfma32_row(8 * st + t, half), 0), 2 * P - 2);
            const float v = e[(int64_t)row * D + 32 * dh + l31];
            const __bf16 hi = (__bf16)v;
            b[t] = plane ? (__bf16)(v - (float)hi) : hi;
        }
    }
    row_order[i] = a;
    tr_order[i] = b;
}

// blockIdx.y = phase: 0 = dK / dV workgroups, 1 = dQ / dE workgroups.  The two halves share nothing but their inputs, so
// they run as separate workgroups of one launch (grid = B x heads x 2: 768 workgroups of ~half the length instead of
// 384 -- at one workgroup per CU a 384-workgroup grid leaves half the chip idle in its second round)
// DROP: the forward applied dropout multipliers m (0 or 1 / (1 - p), regenerated here from the seed) to the normalised
// probabilities: O = (P o m) V  =>  dP = (V dO^T) o m, dS = P (dP - delta) / 8 with delta = rowsum(dO o O) unchanged,
// dV = (P o m)^T dO.
template <bool RELKEY, bool DROP>
__global__ __launch_bounds__(256, 1) void attn_bwd_coop_kernel(
    const float* __restrict__ q, int64_t q_bs, int64_t q_rs, const float* __restrict__ k, int64_t k_bs, int64_t k_rs,
    const float* __restrict__ v, int64_t v_bs, int64_t v_rs, const bf16x8* __restrict__ e_row, const bf16x8* __restrict__ e_tr,
    int P, const float* __restrict__ key_mask, const float* __restrict__ dout, const float* __restrict__ outp,
    const float* __restrict__ lse, float* __restrict__ dq, int64_t dq_bs, int64_t dq_rs, float* __restrict__ dk,
    int64_t dk_bs, int64_t dk_rs, float* __restrict__ dv, int64_t dv_bs, int64_t dv_rs, float* __restrict__ dE_part, int* __restrict__ unit_live,
    int nh, int Lq, int Lk, E3dDrop drop_in) {
    const E3dDrop drop = e3d_drop_resolve(drop_in);   // + the device-side epoch (graph replays: e3d_common.h)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned char* img0 = smem_raw;               // phase A: Q      phase B: K
    unsigned char* img1 = smem_raw + IMG_B;       // phase A: dO     phase B: V
    float* lse2 = reinterpret_cast<float*>(smem_raw + 2 * IMG_B);
    float* delta = lse2 + MAX_TILES * 32;
    float* kbias = delta + MAX_TILES * 32;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    int* q_live = reinterpret_cast<int*>(kbias + MAX_TILES * 32);   // [q tile] != 0: some dO row of the tile is not all zeros
    float* scratch = kbias + MAX_TILES * 32 + 32 + wid * WAVE_F;
    const int bh = xcd_remap(blockIdx.x, gridDim.x), h = bh % nh, b = bh / nh;
    const int PHASE = blockIdx.y;
    const int q_tiles = (Lq + 31) >> 5, k_tiles = (Lk + 31) >> 5;
    const int HD = nh * D;
    const int J0 = k_tiles;                       // rel-key: Lq == Lk

    const float* qb = q + b * q_bs + h * D;
    const float* kb_ = k + b * k_bs + h * D;
    const float* vb = v + b * v_bs + h * D;
    const float* dob = dout + (int64_t)b * Lq * HD + h * D;
    const float* ob = outp + (int64_t)b * Lq * HD + h * D;

    BSTAMP(0);
    // ---- prologue: the phase's two images, per-row softmax constants ------------------------------------------------
    // Every global read of the prologue is ISSUED before the first one is used (round 4): with one workgroup per CU and one
    // wave per SIMD nothing else covers a load's latency, and the rolled staging loops this replaces paid it 24 times in a
    // row (8 trips x {image 0, image 1, delta}) -- more than the tile loops of a BioLiP-shaped ligand take.  Thread tid owns
    // the float4 items tid + 256 i (row = item / 16): the same mapping for the images and for delta, so phase 0 reads dO once.
    constexpr int NI = MAX_TILES * 2;              // float4 items per thread and array
    const float* src0 = PHASE == 0 ? qb : kb_;
    const float* src1 = PHASE == 0 ? dob : vb;
    const int64_t rs0 = PHASE == 0 ? q_rs : k_rs, rs1 = PHASE == 0 ? (int64_t)HD : v_rs;
    const int rows01 = PHASE == 0 ? Lq : Lk, tiles01 = PHASE == 0 ? q_tiles : k_tiles;
    f32x4 x0[NI], x1[NI], xd[NI], xo[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int f = tid + 256 * i, row = f >> 4, c4 = f & 15;
        if (i < 2 * tiles01) {                     // (workgroup-uniform)
            const int rc = min(row, rows01 - 1);
            x0[i] = *reinterpret_cast<const f32x4*>(src0 + (int64_t)rc * rs0 + 4 * c4);
            x1[i] = *reinterpret_cast<const f32x4*>(src1 + (int64_t)rc * rs1 + 4 * c4);
        }
        if (i < 2 * q_tiles) {
            const int rc = min(row, Lq - 1);
            xo[i] = *reinterpret_cast<const f32x4*>(ob + (int64_t)rc * HD + 4 * c4);
            if (PHASE == 1) xd[i] = *reinterpret_cast<const f32x4*>(dob + (int64_t)rc * HD + 4 * c4);
        }
    }
    float lse_in = 0.f, mask_in = 1.f;
    if (tid < MAX_TILES * 32) {
        if (tid < Lq) lse_in = lse[((int64_t)b * nh + h) * Lq + tid];
        if (tid < Lk && key_mask) mask_in = key_mask[(int64_t)b * Lk + tid];
    }
    // Dead query tiles (round 4).  A query row whose dO is ALL ZEROS contributes exact zeros to everything this kernel sums
    // (dP = dO V^T = 0, delta = 0, so dS = P (dP - delta) = 0; dV += dO^T P gets 0) and its own dQ row is 0: in training that is
    // every padded position -- the loss never sees them and, masked as keys, nothing valid depends on them -- i.e. 3 of the 4
    // query tiles of a BioLiP-shaped ligand (5-30 residues in a 128-row frame).  Found from the data itself while delta is
    // formed (no mask argument, nothing to prove: a tile with one nonzero dO element is simply alive); skipping is bit-exact.
    if (tid < 32) q_live[tid] = 0;
    __syncthreads();
#ifdef BWD_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    BSTAMP(1);
    auto to_image = [&](unsigned char* img, const f32x4& v, int row, int c4) {
        bf16x4 hi, lo;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const __bf16 p = (__bf16)v[j];
            hi[j] = p;
            lo[j] = (__bf16)(v[j] - (float)p);
        }
        const int o = img_off(row, c4 >> 1) + 8 * (c4 & 1);
        *reinterpret_cast<bf16x4*>(img + o) = hi;
        *reinterpret_cast<bf16x4*>(img + PLANE_B + o) = lo;
    };
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int f = tid + 256 * i, row = f >> 4, c4 = f & 15;
        if (i < 2 * tiles01) {
            to_image(img0, x0[i], row, c4);
            to_image(img1, x1[i], row, c4);
        }
        if (i < 2 * q_tiles) {                     // delta[q] = sum_d dO[q][d] O[q][d]: 16 consecutive threads per row
            const f32x4 a = PHASE == 0 ? x1[i] : xd[i], o = xo[i];
            if (row < Lq && (a[0] != 0.f || a[1] != 0.f || a[2] != 0.f || a[3] != 0.f)) q_live[row >> 5] = 1;
            float part = (a[0] * o[0] + a[1] * o[1]) + (a[2] * o[2] + a[3] * o[3]);
#pragma unroll
            for (int sh = 1; sh < 16; sh <<= 1) part += __shfl_xor(part, sh, 64);
            if (c4 == 0) delta[row] = part;
        }
    }
    if (tid < MAX_TILES * 32) {
        lse2[tid] = lse_in * LOG2E;
        kbias[tid] = (tid < Lk && key_mask) ? (1.0f - mask_in) * (-10000.0f * LOG2E) : 0.f;
    }
    __syncthreads();
    BSTAMP(2);

    // ================================================================================================ phase A: dK, dV
    if (PHASE == 0 && wid < k_tiles) {
        const int kt = wid, key = 32 * kt + l31;
        const bool key_ok = key < Lk;
        Frag kf[4], vf[4];
        load_row_split(kf, kb_ + (int64_t)min(key, Lk - 1) * k_rs, half);
        load_row_split(vf, vb + (int64_t)min(key, Lk - 1) * v_rs, half);
        const float bias_k = kbias[min(key, MAX_TILES * 32 - 1)];
        float* ring = scratch;                     // T[q][window x]: 32 x RA_LD
        f32x16 dk0 = zero16(), dk1 = zero16(), dv0 = zero16(), dv1 = zero16();
        BSTAMP(3);
        for (int qt = 0; qt < q_tiles; ++qt) {
            if (!q_live[qt]) continue;             // (workgroup-uniform) every dO row of this query tile is zero: exact zeros
            Frag qa[4], da[4];
            row_frags(qa, img0, qt, lane);
            row_frags(da, img1, qt, lane);
            f32x16 s, dp;                          // S[q = rho(r, half)][key = l31], dP[q][key]
            dot_tile2(s, dp, qa, kf, da, vf);
            if (RELKEY) {
                // both 32-row blocks of the window: rows e_lo + 32 blk + x, e_lo = P + 32 (qt - kt - 1)
                Frag e0[4], e1[4];
                const bf16x8* p0 = e_row + (size_t)min(max(qt - kt - 1 + J0, 0), 2 * J0 - 1) * 512 + lane;
                const bf16x8* p1 = e_row + (size_t)min(max(qt - kt + J0, 0), 2 * J0 - 1) * 512 + lane;
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) {
                    e0[kb].hi = p0[kb * 64]; e0[kb].lo = p0[256 + kb * 64];
                    e1[kb].hi = p1[kb * 64]; e1[kb].lo = p1[256 + kb * 64];
                }
                f32x16 t0, t1;                     // T[q][x = l31]
                dot_tile2(t0, t1, qa, e0, qa, e1);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    ring[mfma32_row(r, half) * RA_LD + l31] = t0[r];
                    ring[mfma32_row(r, half) * RA_LD + 32 + l31] = t1[r];
                }
                wave_lds_sync();
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ql = mfma32_row(r, half);
                    s[r] += ring[ql * RA_LD + ql - l31 + 31];
                }
                __builtin_amdgcn_wave_barrier();   // the next step's writes come after these reads
            }
            f32x16 ds;
            bool any_p = false;
#pragma unroll
            for (int g = 0; g < 4; ++g) {          // rows 8 g + 4 half + j: lse2 / delta as 16-byte LDS reads
                const f32x4 l4 = *reinterpret_cast<const f32x4*>(lse2 + 32 * qt + 8 * g + 4 * half);
                const f32x4 d4 = *reinterpret_cast<const f32x4*>(delta + 32 * qt + 8 * g + 4 * half);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int r = 4 * g + j;
                    const int qg = 32 * qt + 8 * g + 4 * half + j;
                    const bool ok = key_ok && qg < Lq;
                    const float p = ok ? __builtin_amdgcn_exp2f(fmaf(s[r], S_SCALE, bias_k) - l4[j]) : 0.f;
                    float mult = 1.0f;
                    if (DROP) {   // one hash per 4 consecutive keys of a query row: this lane's key is element key & 3
                        float m4[4];
                        e3d_drop_mult4(drop, e3d_attn_drop_idx4(bh, Lq, Lk, min(qg, Lq - 1), min(key, Lk - 1) & ~3), m4);
                        const int e = key & 3;
                        mult = e == 0 ? m4[0] : (e == 1 ? m4[1] : (e == 2 ? m4[2] : m4[3]));
                    }
                    s[r] = p * mult;
                    ds[r] = p * (dp[r] * mult - d4[j]) * 0.125f;
                    any_p = any_p || p != 0.f;
                }
            }
            // every probability of the pair is exactly 0.0f (an all-padding key tile whose exp underflowed, as the reference's
            // additive -10000 makes it whenever the scores are of ordinary size): P = dS = 0, nothing to add (wave-uniform)
            if (__builtin_amdgcn_ballot_w64(any_p) == 0ull) continue;
            // dV^T += dO^T P,  dK^T += Q^T dS: B = the accumulator tiles as they stand (k = query rows rho(8 st + j, half))
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                float pv[8], sv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) { pv[j] = s[8 * st + j]; sv[j] = ds[8 * st + j]; }
                const Frag pb = split8(pv), sb = split8(sv);
                dv0 = mfma3(tr_frag(img1, qt, st, 0, lane), pb, dv0);
                dv1 = mfma3(tr_frag(img1, qt, st, 1, lane), pb, dv1);
                dk0 = mfma3(tr_frag(img0, qt, st, 0, lane), sb, dk0);
                dk1 = mfma3(tr_frag(img0, qt, st, 1, lane), sb, dk1);
            }
        }
        BSTAMP(4);
        if (key_ok) {
            store_rows_T(dv0, dv1, dv + b * dv_bs + (int64_t)key * dv_rs + h * D, half);
            store_rows_T(dk0, dk1, dk + b * dk_bs + (int64_t)key * dk_rs + h * D, half);
        }
    }

#ifdef BWD_STAMPS
    if (PHASE == 0) { BSTAMP(5); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); BSTAMP(6); }
#endif

    // ================================================================================================ phase B: dQ, dE
    if (PHASE == 1 && wid < q_tiles) {
        const int qt = wid, q0 = 32 * qt, qrow = q0 + l31;
        const bool q_ok = qrow < Lq;
        Frag qf[4], dof[4];
        load_row_split(qf, qb + (int64_t)min(qrow, Lq - 1) * q_rs, half);
        load_row_split(dof, dob + (int64_t)min(qrow, Lq - 1) * HD, half);
        const float lse_q = lse2[min(qrow, MAX_TILES * 32 - 1)], delta_q = delta[min(qrow, MAX_TILES * 32 - 1)];
        float* ring = scratch;                     // T^T window: 64 x RB_LD (rows = window offset, cols = query)
        float* X = scratch + 64 * RB_LD;           // dS^T tile: 32 x X_LD
        f32x16 dq0 = zero16(), dq1 = zero16(), elo0 = zero16(), elo1 = zero16(), ehi0 = zero16(), ehi1 = zero16();
        Frag fq[2][2];                             // dE: B = Q[q0 + 16 st + 8 half + j][32 dh + l31] (natural k order), loop invariant
        if (RELKEY) {
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int dh = 0; dh < 2; ++dh) {
                    float x[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) x[j] = qb[(int64_t)min(q0 + 16 * st + 8 * half + j, Lq - 1) * q_rs + 32 * dh + l31];
                    fq[st][dh] = split8(x);
                }
        }
        float* part_base = RELKEY ? dE_part + ((int64_t)bh * q_tiles + qt) * (k_tiles + 1) * 32 * D : nullptr;
        BSTAMP(3);
        // (dE blocks of a dead query tile are neither written nor read: de_chunk_sum_kernel looks at unit_live)
        if (RELKEY && lane == 0) unit_live[bh * q_tiles + qt] = q_live[qt];
        if (!q_live[qt]) {                         // (wave-uniform) dO of this wave's query tile is all zeros: dQ = 0, dE blocks = 0
            if (q_ok) store_rows_T(dq0, dq1, dq + b * dq_bs + (int64_t)qrow * dq_rs + h * D, half);
#ifdef BWD_STAMPS
            BSTAMP(4); BSTAMP(5); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); BSTAMP(6);
#endif
            return;
        }
        auto e_rows = [&](Frag (&ef)[4], int j) {
            const bf16x8* pj = e_row + (size_t)min(max(j, 0), 2 * J0 - 1) * 512 + lane;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) { ef[kb].hi = pj[kb * 64]; ef[kb].lo = pj[256 + kb * 64]; }
        };
        int rot = 0;
        if (RELKEY) {   // the block "before" key tile 0 (rows q0 + P ..) fills the upper half of the ring
            Frag ef[4];
            e_rows(ef, qt + J0);
            const f32x16 t = dot_tile(ef, qf);
#pragma unroll
            for (int r = 0; r < 16; ++r) ring[(32 + mfma32_row(r, half)) * RB_LD + l31] = t[r];
        }
        for (int kt = 0; kt < k_tiles; ++kt) {
            const int r0 = 32 * kt;
            Frag ka[4], va[4];
            row_frags(ka, img0, kt, lane);
            row_frags(va, img1, kt, lane);
            f32x16 s, dp;                          // S^T[key = rho(r, half)][q = l31], dP^T[key][q]
            dot_tile2(s, dp, ka, qf, va, dof);
            if (RELKEY) {
                Frag ef[4];
                e_rows(ef, qt - kt - 1 + J0);      // rows e_lo .. e_lo + 31, e_lo = P + 32 (qt - kt - 1)
                const f32x16 t = dot_tile(ef, qf);
#pragma unroll
                for (int r = 0; r < 16; ++r) ring[((mfma32_row(r, half) + rot) & 63) * RB_LD + l31] = t[r];
                wave_lds_sync();
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int x = l31 - mfma32_row(r, half) + 31;
                    s[r] += ring[((x + rot) & 63) * RB_LD + l31];
                }
                __builtin_amdgcn_wave_barrier();
                rot ^= 32;
            }
            f32x16 ds;
            bool any_p = false;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(kbias + r0 + 8 * g + 4 * half);
                float m4[4] = {1.f, 1.f, 1.f, 1.f};
                if (DROP) e3d_drop_mult4(drop, e3d_attn_drop_idx4(bh, Lq, Lk, min(qrow, Lq - 1), r0 + 8 * g + 4 * half), m4);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int r = 4 * g + j;
                    const bool ok = q_ok && (r0 + 8 * g + 4 * half + j) < Lk;
                    const float p = ok ? __builtin_amdgcn_exp2f(fmaf(s[r], S_SCALE, b4[j]) - lse_q) : 0.f;
                    ds[r] = p * (dp[r] * m4[j] - delta_q) * 0.125f;
                    any_p = any_p || p != 0.f;
                }
            }
            // all 1024 probabilities of the pair exactly 0.0f (an all-padding key tile): dS = 0 -- no dQ term, no dE term; only
            // the block bookkeeping of the distance-table gradient moves on (wave-uniform)
            const bool dead_pair = __builtin_amdgcn_ballot_w64(any_p) == 0ull;
            // dQ^T += K^T dS^T
            if (!dead_pair) {
#pragma unroll
                for (int st = 0; st < 2; ++st) {
                    float sv[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) sv[j] = ds[8 * st + j];
                    const Frag sb = split8(sv);
                    dq0 = mfma3(tr_frag(img0, kt, st, 0, lane), sb, dq0);
                    dq1 = mfma3(tr_frag(img0, kt, st, 1, lane), sb, dq1);
                }
            }
            if (RELKEY && dead_pair) {
                // the upper block is complete (nothing was added to either block by this tile)
                float* blk = part_base + (int64_t)kt * 32 * D;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    blk[mfma32_row(r, half) * D + l31] = ehi0[r];
                    blk[mfma32_row(r, half) * D + 32 + l31] = ehi1[r];
                }
                ehi0 = elo0; ehi1 = elo1;
                elo0 = zero16(); elo1 = zero16();
            }
            if (RELKEY && !dead_pair) {
                wave_lds_sync();   // the previous tile's readers of X are done
#pragma unroll
                for (int r = 0; r < 16; ++r) X[mfma32_row(r, half) * X_LD + l31] = ds[r];
                wave_lds_sync();
                // inverse skew: dT^T[x][l] = dS^T[l - x + 31][l] for window offset x in [0, 63]
                f32x16 dt_lo, dt_hi;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int x = mfma32_row(r, half);
                    const int rr_lo = l31 - x + 31, rr_hi = l31 - x - 1;
                    dt_lo[r] = (rr_lo >= 0 && rr_lo < 32) ? X[rr_lo * X_LD + l31] : 0.f;
                    dt_hi[r] = (rr_hi >= 0 && rr_hi < 32) ? X[rr_hi * X_LD + l31] : 0.f;
                }
                // dQ^T += E^T dT^T over both live 32-row blocks of E (transposed-order planes of the pre-pass)
                const int j_lo = min(max(qt - kt - 1 + J0, 0), 2 * J0 - 1), j_hi = min(max(qt - kt + J0, 0), 2 * J0 - 1);
#pragma unroll
                for (int st = 0; st < 2; ++st) {
                    float lo8[8], hi8[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) { lo8[j] = dt_lo[8 * st + j]; hi8[j] = dt_hi[8 * st + j]; }
                    const Frag bl = split8(lo8), bhi = split8(hi8);
#pragma unroll
                    for (int dh = 0; dh < 2; ++dh) {
                        Frag al, ah;
                        const bf16x8* pl = e_tr + (size_t)j_lo * 512 + (2 * st + dh) * 64 + lane;
                        const bf16x8* ph = e_tr + (size_t)j_hi * 512 + (2 * st + dh) * 64 + lane;
                        al.hi = pl[0]; al.lo = pl[256];
                        ah.hi = ph[0]; ah.lo = ph[256];
                        if (dh == 0) { dq0 = mfma3(al, bl, dq0); dq0 = mfma3(ah, bhi, dq0); }
                        else { dq1 = mfma3(al, bl, dq1); dq1 = mfma3(ah, bhi, dq1); }
                    }
                }
                // dE blocks += dT^T Q: A = dT^T (row = window offset x = l31, k = query ll natural order), B = fq
#pragma unroll
                for (int st = 0; st < 2; ++st) {
                    float a_lo[8], a_hi[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int ll = 16 * st + 8 * half + j;
                        const int rr_lo = ll - l31 + 31, rr_hi = ll - l31 - 1;
                        a_lo[j] = (rr_lo >= 0 && rr_lo < 32) ? X[rr_lo * X_LD + ll] : 0.f;
                        a_hi[j] = (rr_hi >= 0 && rr_hi < 32) ? X[rr_hi * X_LD + ll] : 0.f;
                    }
                    const Frag fa_lo = split8(a_lo), fa_hi = split8(a_hi);
                    elo0 = mfma3(fa_lo, fq[st][0], elo0);
                    elo1 = mfma3(fa_lo, fq[st][1], elo1);
                    ehi0 = mfma3(fa_hi, fq[st][0], ehi0);
                    ehi1 = mfma3(fa_hi, fq[st][1], ehi1);
                }
                // the upper block is complete: block kt covers E rows q0 - 32 kt + P .. + 31
                float* blk = part_base + (int64_t)kt * 32 * D;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    blk[mfma32_row(r, half) * D + l31] = ehi0[r];
                    blk[mfma32_row(r, half) * D + 32 + l31] = ehi1[r];
                }
                ehi0 = elo0; ehi1 = elo1;
                elo0 = zero16(); elo1 = zero16();
            }
        }
        BSTAMP(4);
        if (RELKEY) {
            float* blk = part_base + (int64_t)k_tiles * 32 * D;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                blk[mfma32_row(r, half) * D + l31] = ehi0[r];
                blk[mfma32_row(r, half) * D + 32 + l31] = ehi1[r];
            }
        }
        BSTAMP(5);
        if (q_ok) store_rows_T(dq0, dq1, dq + b * dq_bs + (int64_t)qrow * dq_rs + h * D, half);
#ifdef BWD_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        BSTAMP(6);
#endif
    }
}

#ifdef BWD_STAMPS
extern "C" int e3d_debug_bwd_stamps(long long* t) {
    return (int)hipMemcpyFromSymbol(t, HIP_SYMBOL(bwd_stamps), sizeof(long long) * 2 * 512 * 4 * 8);
}
#endif

constexpr int DE_CHUNKS = 12;   // (b, head) chunks of the dE reduction's first stage

// ---- dE = sum of the per-unit partial blocks, streaming and deterministic (the two-launch path's reduce gathers one
// 256-byte row per unit and E row at a stride of 40 KB -- 36 us for 63 MB at B=32, L=128 -- and ends in float atomics).
// Stage 1: block (slab = (qt, j), chunk c) sums slab (qt, j) -- 32 x 64 floats, contiguous 8 KB per unit -- over the
// (b, head) units of chunk c, in unit order: every load is a coalesced 16-byte piece of an 8-KB run.
// Units whose query tile saw an all-zero dO (``unit_live`` = 0, written by the fused kernel) hold no blocks: skipped.
__global__ __launch_bounds__(256) void de_chunk_sum_kernel(const float* __restrict__ part, float* __restrict__ chunk_sums,
                                                           const int* __restrict__ unit_live, int slabs, int slabs_per_qt,
                                                           int n_bh, int bh_per_chunk) {
    const int slab = blockIdx.x, c = blockIdx.y, tid = threadIdx.x;
    const int b0 = c * bh_per_chunk, b1 = min(n_bh, b0 + bh_per_chunk);
    const int qt = slab / slabs_per_qt, q_tiles = slabs / slabs_per_qt;
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
    const float* p = part + ((int64_t)b0 * slabs + slab) * 2048 + 4 * tid;
    for (int bh = b0; bh < b1; ++bh, p += (int64_t)slabs * 2048) {
        if (unit_live[bh * q_tiles + qt] == 0) continue;     // (workgroup-uniform)
        a0 += *reinterpret_cast<const f32x4*>(p);
        a1 += *reinterpret_cast<const f32x4*>(p + 1024);
    }
    float* o = chunk_sums + ((int64_t)c * slabs + slab) * 2048 + 4 * tid;
    *reinterpret_cast<f32x4*>(o) = a0;
    *reinterpret_cast<f32x4*>(o + 1024) = a1;
}
// Stage 2: dE[e][d] = sum over chunks and over the (qt, j) slabs whose window holds row e (slab (qt, j) covers rows
// P + 32 (qt - j) .. + 31), in a fixed order; rows no slab covers (|distance| >= L) get 0.  No memset, no atomics.
__global__ __launch_bounds__(64) void de_final_sum_kernel(const float* __restrict__ chunk_sums, float* __restrict__ dE, int P,
                                                          int q_tiles, int k_tiles, int n_chunks) {
    const int e = blockIdx.x, d = threadIdx.x, slabs = q_tiles * (k_tiles + 1);
    float acc = 0.f;
    for (int qt = 0; qt < q_tiles; ++qt) {
        const int t = 32 * qt + P - e + 31;          // as dist_emb_reduce_kernel: block j = t >> 5, row = e - (32 (qt - j) + P)
        const int j = t >> 5;
        if (t < 0 || j > k_tiles) continue;
        const int row = e - (32 * (qt - j) + P);
        const float* p = chunk_sums + ((int64_t)(qt * (k_tiles + 1) + j) * 32 + row) * D + d;
        float v[DE_CHUNKS];       // all chunk loads of a slab in flight together, summed in chunk order
#pragma unroll
        for (int c = 0; c < DE_CHUNKS; ++c) v[c] = p[(int64_t)min(c, n_chunks - 1) * slabs * 2048];
#pragma unroll
        for (int c = 0; c < DE_CHUNKS; ++c)
            if (c < n_chunks) acc += v[c];
    }
    dE[(int64_t)e * D + d] = acc;
}

}  // namespace

// bytes of scratch the fused kernel needs for the two plane orders of the distance table (inside the caller's workspace)
int64_t e3d_attn_bwd_coop_scratch_bytes(int Lk) { return (int64_t)2 * 2 * ((Lk + 31) / 32) * 512 * 16; }

bool e3d_attn_bwd_coop_supported(int Lq, int Lk, bool dropping) {
    static int on = -1;   // E3D_ATTN_BWD_COOP=0: the two-launch kernels of attn_bwd_split.hip for every shape (A/B timing)
    if (on < 0) {
        const char* e = getenv("E3D_ATTN_BWD_COOP");
        on = e ? atoi(e) : 1;
    }
    (void)dropping;   // (dropout is regenerated inside the fused kernel)
    return on && Lq <= 32 * MAX_TILES && Lk <= 32 * MAX_TILES;
}

// chunk partial sums of the dE reduction: floats appended to the unit blocks in the caller's workspace
int64_t e3d_attn_bwd_coop_de_floats(int Lq, int Lk) {
    return (int64_t)DE_CHUNKS * ((Lq + 31) / 32) * ((Lk + 31) / 32 + 1) * 2048;
}

// The fused backward, bf16x3 (arguments validated by e3d_relkey_attn_bwd_ex; ``e_scratch``: 16-byte aligned,
// e3d_attn_bwd_coop_scratch_bytes(Lk) bytes, only read / written when dist_emb is given; ``part``: the unit blocks
// followed by e3d_attn_bwd_coop_de_floats(Lq, Lk) floats; writes every row of d_dist_emb).
int e3d_attn_bwd_coop_launch(const float* q, int64_t q_bs, int64_t q_rs, const float* k, int64_t k_bs, int64_t k_rs,
                             const float* v, int64_t v_bs, int64_t v_rs, const float* dist_emb, int P,
                             const float* key_mask, const float* out, const float* lse, const float* dout, float* dq,
                             int64_t dq_bs, int64_t dq_rs, float* dk, int64_t dk_bs, int64_t dk_rs, float* dv,
                             int64_t dv_bs, int64_t dv_rs, float* d_dist_emb, void* e_scratch, float* part, int B, int nh,
                             int Lq, int Lk, E3dDrop drop, bool dropping, hipStream_t s) {
    const int J0 = (Lk + 31) / 32, n_items = 2 * J0 * 512;
    bf16x8* e_row = reinterpret_cast<bf16x8*>(e_scratch);
    bf16x8* e_tr = e_row ? e_row + n_items : nullptr;
    if (dist_emb)
        hipLaunchKernelGGL(e_planes_kernel, dim3((n_items + 255) / 256), dim3(256), 0, s, dist_emb, e_row, e_tr, P, J0, n_items);
    // liveness words of the (b, head, query tile) units: after the chunk sums (e3d_relkey_attn_bwd_workspace_floats)
    int* unit_live = dist_emb ? reinterpret_cast<int*>(part + (int64_t)B * nh * ((Lq + 31) / 32) * ((Lk + 31) / 32 + 1) * 2048 +
                                                       e3d_attn_bwd_coop_de_floats(Lq, Lk))
                              : nullptr;
    static std::atomic<uint64_t> ok[4];
#define E3D_BWD_COOP(RK, DR)                                                                                               \
    do {                                                                                                                   \
        e3d_allow_lds(ok[2 * RK + DR], attn_bwd_coop_kernel<RK, DR>, LDS_BYTES);                                           \
        hipLaunchKernelGGL((attn_bwd_coop_kernel<RK, DR>), dim3(B * nh, 2), dim3(256), LDS_BYTES, s, q, q_bs, q_rs, k, k_bs, \
                           k_rs, v, v_bs, v_rs, e_row, e_tr, P, key_mask, dout, out, lse, dq, dq_bs, dq_rs, dk, dk_bs,      \
                           dk_rs, dv, dv_bs, dv_rs, part, unit_live, nh, Lq, Lk, drop);                                    \
    } while (0)
    if (dist_emb) {
        if (dropping) E3D_BWD_COOP(true, true);
        else E3D_BWD_COOP(true, false);
    } else {
        if (dropping) E3D_BWD_COOP(false, true);
        else E3D_BWD_COOP(false, false);
    }
#undef E3D_BWD_COOP
    if (dist_emb) {
        const int q_tiles = (Lq + 31) / 32, k_tiles = (Lk + 31) / 32, slabs = q_tiles * (k_tiles + 1), n_bh = B * nh;
        const int per = (n_bh + DE_CHUNKS - 1) / DE_CHUNKS, n_chunks = (n_bh + per - 1) / per;
        float* chunk_sums = part + (int64_t)n_bh * slabs * 2048;
        hipLaunchKernelGGL(de_chunk_sum_kernel, dim3(slabs, n_chunks), dim3(256), 0, s, part, chunk_sums, unit_live, slabs,
                           k_tiles + 1, n_bh, per);
        hipLaunchKernelGGL(de_final_sum_kernel, dim3(2 * P - 1), dim3(64), 0, s, chunk_sums, d_dist_emb, P, q_tiles, k_tiles,
                           n_chunks);
    }
    return e3d_launch_status("e3d_relkey_attn_bwd (fused, bf16x3)");
}
