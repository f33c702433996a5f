// fp32 GEMM through the bf16 matrix cores with SPLIT operands and fp32 accumulation:
//     out[M,N] = act(A[M,K] . B[N,K]^T + bias)         (A, B given K-contiguous or K-major)
// Each fp32 operand x is split on the fly into bf16 terms x = x0 + x1 (+ x2) (xi = rne_bf16 of the
// running residual) and the product is rebuilt from the significant cross terms:
//     TERMS = 1:  a0*b0                                         (plain bf16 products, ~2^-8: what the reference's train
//                                                                scripts select with set_float32_matmul_precision("medium"))
//     TERMS = 3:  a0*b0 + a0*b1 + a1*b0                         (error ~2^-17 per product)
//     TERMS = 6:  + a0*b2 + a2*b0 + a1*b1                       (error ~2^-24: fp32 grade)
// v_mfma_f32_32x32x16_bf16 runs at 16x the rate of the fp32 MFMA, so the 3-/6-term products cost
// 3/16 / 6/16 of the exact fp32 kernel (gemm_f32.hip).  Accumulation stays fp32 in the MFMA.
//
// Workgroup tile 256x128x32, 8 waves as 4(M) x 2(N), each wave 2x2 MFMA tiles of 32x32 (64
// accumulator registers).  Staging: global loads (fp32) -> split in registers -> ds_write_b64
// into per-term bf16 images of unpadded 64-byte rows whose 16-byte k-chunks are XOR-swizzled by the
// row quad (conflict-free ds_read_b128 fragment reads), double buffered for TERMS = 3.
//
// Operand layouts (the three GEMMs of a Linear layer share one kernel):
//     forward   y  = x W^T : A = x  [M,K] K-contiguous,  B = W [N,K] K-contiguous
//     dgrad     dx = dy W  : A = dy [M,K'] K-contiguous, B = W as [K'][N'] -> "K-major"
//     wgrad     dW = dy^T x: A = dy as [K'][M'] K-major,  B = x as [K'][N'] K-major
// A K-major operand is read with one dword per lane (lanes along the contiguous output index, so
// every k-row is a coalesced 256-byte segment) and transposed for free by the LDS write; when BOTH
// are K-major and come in whole quads (the weight gradients) StagerT reads float4s instead and the
// transposition moves to the fragment reads (ds_read_b64_tr_b16).
#include <type_traits>

#include "e3d_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// Element type of the split terms: __bf16 (8 significant bits per term, fp32 exponent range: bf16x3 / bf16x6) or
// _Float16 (11 bits per term: TWO terms carry 22 bits, so the three cross products a0 b0 + a0 b1 + a1 b0 are
// ~2^-21 accurate -- fp32 grade at the cost of bf16x3 -- for operands inside the fp16 range: |x| < 65504, and
// elements below 2^-14 keep an absolute error of 2^-25 instead of a relative one: "f16x3").
template <typename E> struct Vec;
template <> struct Vec<__bf16> { typedef bf16x8 x8; typedef bf16x4 x4; };
template <> struct Vec<_Float16> { typedef f16x8 x8; typedef f16x4 x4; };
__device__ __forceinline__ f32x16 mma16(const bf16x8 a, const bf16x8 b, const f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mma16(const f16x8 a, const f16x8 b, const f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// what the forward epilogues do besides bias + activation: ``absmax`` (may be null) is raised to the largest |out|
// (e3d_common.h), ``scale`` multiplies the accumulator before the bias -- the inverse of the power of two a caller
// scaled the weight by so that its fp16 terms sit in the normal range (include/e3d_hip.h, out_scale)
struct Epi { float* absmax; float scale; };

constexpr int BN = 128, BK = 32;
#ifdef GEMM_STAMPS   // lab builds only (tools/lab/gemm_stamps.py): s_memtime stamps of waves 0 and 4 of ONE workgroup of the general kernel
__device__ long long gemm_stamps[2][64][8];
#define GSTAMP(slot)                                                                                                    \
    do {                                                                                                                \
        if (stamp_on && kt < 64 && lane == 0) gemm_stamps[wid >> 2][kt][(slot)] = __builtin_readcyclecounter();         \
    } while (0)
#else
#define GSTAMP(slot) do {} while (0)
#endif
#ifndef E3D_GEMM_FRAG_PREFETCH
#define E3D_GEMM_FRAG_PREFETCH 1
#endif
// LDS rows are 64 bytes (32 bf16) unpadded, with the 16-byte k-chunk XOR-swizzled by (row >> 2) & 3: the 16 lanes
// of a ds_read_b128 group (4 row quads) then hit 16 distinct 4-bank slots.
constexpr int ROW_B = 64;
__device__ __forceinline__ int swz_off(int row, int chunk) { return row * ROW_B + ((chunk ^ ((row >> 2) & 3)) << 4); }

template <int NS, typename E>
__device__ __forceinline__ void split4(const f32x4 v, typename Vec<E>::x4 (&parts)[NS]) {
    f32x4 r = v;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const E p = (E)r[j];
            parts[s][j] = p;
            r[j] -= (float)p;
        }
    }
}

template <>
__device__ __forceinline__ void split4<2, _Float16>(const f32x4 v, f16x4 (&parts)[2]) {
    e3d_f16x2 h0, l0, h1, l1;
    e3d_split2_f16(v[0], v[1], h0, l0);
    e3d_split2_f16(v[2], v[3], h1, l1);
    parts[0] = f16x4{h0[0], h0[1], h1[0], h1[1]};
    parts[1] = f16x4{l0[0], l0[1], l1[0], l1[1]};
}

// One operand's staging: NV float4-equivalents (4 consecutive k of one row) per thread.
#ifndef GEMM_LAB_RAW   // timing-only lab builds (results are garbage): bit 1 = operand A, bit 2 = operand B of the K-contiguous
#define GEMM_LAB_RAW 0  // layouts staged WITHOUT the fp32 -> two-term split (what pre-split operand planes would cost the k-step)
#endif
template <int ROWS, bool KMAJ, int NT = 512, bool RAW = false>
struct Stager {
    static constexpr int NV = ROWS * 8 / NT;   // items per thread (512 threads: A 4, B 2)
    const float* src[NV];   // K-contiguous: the item's row base;  K-major: the operand base (uniform)
    int off[NV];
    int64_t ld;
    int kbase[NV];          // first k of the item inside a k-tile (K-major: wave-uniform, kept in SGPRs)
    int roff[NV];           // K-major: the lane's element offset along the contiguous index

    __device__ __forceinline__ void init(const float* base, int64_t ld_, int row0, int row_limit, int tid) {
        ld = ld_;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int f = tid + NT * i;
            int r, kg;
            if (KMAJ) { r = f % ROWS; kg = f / ROWS; }   // lanes along rows: coalesced k-rows
            else { r = f >> 3; kg = f & 7; }             // lanes along k: 128-byte row segments
            int gr = row0 + r;
            gr = gr < row_limit ? gr : row_limit - 1;    // clamp: rows past the edge are discarded later
            src[i] = KMAJ ? base : base + (int64_t)gr * ld + kg * 4;
            // K-major: kg = f / ROWS is the same for the 64 lanes of a wave (ROWS is a multiple of 64); telling the
            // compiler so keeps the k-dependent part of every address (k * ld) in scalar registers: the loads become
            // SGPR base + one 32-bit lane offset, without per-element 64-bit VALU arithmetic or divergent branches
            kbase[i] = KMAJ ? __builtin_amdgcn_readfirstlane(kg * 4) : kg * 4;
            roff[i] = gr;
            off[i] = swz_off(r, kg >> 1) + (kg & 1) * 8;
        }
    }
    __device__ __forceinline__ void load(f32x4 (&v)[NV], int k0, int K) const {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if (KMAJ) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k = min(k0 + kbase[i] + j, K - 1);        // wave-uniform; the K tail is zeroed at store time
                    v[i][j] = (src[i] + (int64_t)k * ld)[roff[i]];
                }
            } else {
                v[i] = *reinterpret_cast<const f32x4*>(src[i] + k0);
            }
        }
    }
    __device__ __forceinline__ void load_item(int i, f32x4& v, int k0, int K) const {
        if (KMAJ) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = min(k0 + kbase[i] + j, K - 1);            // wave-uniform; the K tail is zeroed at store time
                v[j] = (src[i] + (int64_t)k * ld)[roff[i]];
            }
        } else {
            v = *reinterpret_cast<const f32x4*>(src[i] + k0);
        }
    }
    // ``k0`` = first k of the k-tile the registers hold: K-major items zero their reduction tail (k >= K, token
    // counts that are not a multiple of 32) here, away from the loads, so that no load result is needed early
    template <int NS, typename E>
    __device__ __forceinline__ void store_item(int i, const f32x4& vin, unsigned char* img, int part_bytes, int k0, int K) const {
        f32x4 v = vin;
        if (KMAJ) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = k0 + kbase[i] + j < K ? v[j] : 0.f;
        }
        typename Vec<E>::x4 p[NS];
        if constexpr (RAW && NS == 2) {
            typedef float f32x2_ __attribute__((ext_vector_type(2)));
            p[0] = __builtin_bit_cast(typename Vec<E>::x4, f32x2_{v[0], v[1]});
            p[1] = __builtin_bit_cast(typename Vec<E>::x4, f32x2_{v[2], v[3]});
        } else {
            split4<NS, E>(v, p);
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) *reinterpret_cast<typename Vec<E>::x4*>(img + s * part_bytes + off[i]) = p[s];
    }
    // sum of the item's 4 values with the same reduction-tail masking as store_item (K-major items)
    __device__ __forceinline__ float masked_sum(int i, const f32x4& v, int k0, int K) const {
        float t[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] = (!KMAJ || k0 + kbase[i] + j < K) ? v[j] : 0.f;
        return (t[0] + t[1]) + (t[2] + t[3]);
    }
    template <int NS, typename E>
    __device__ __forceinline__ void store(const f32x4 (&v)[NV], unsigned char* img, int part_bytes, int k0, int K) const {
#pragma unroll
        for (int i = 0; i < NV; ++i) store_item<NS, E>(i, v[i], img, part_bytes, k0, K);
    }
};

// Transposing staging of a K-major operand (the weight-gradient layout, TR): an item is a float4 along the CONTIGUOUS
// output index (4 rows of the operand) for ONE k -- a quarter of the load instructions of the dword scheme above, each
// 16 bytes wide (a wave reads whole 1-KB / 512-byte k-rows).  The image is kept in the memory's own orientation,
// [k][row] bf16 (pitch = 2 ROWS bytes: the same bytes per plane as the [row][k] image), and the MFMA fragments -- 8 k
// values per lane -- are read with ds_read_b64_tr_b16, gfx950's transposing LDS read (two per fragment; k then runs in
// the order 4 half + (j & 3) + 8 (j >> 2) inside a 16-block: the same for both operands, so the contraction is intact).
// 32-byte units (16 rows of one k) are XOR-ed by 2 (k & 3): the 4 k-rows x 2 units a 32-lane half of a transposing read
// touches then fall into 8 distinct 8-bank windows, and a wave's 8-byte writes still cover whole k-rows.
// (host) the transposing stager's buffer descriptors count bytes in 32 bits: reduction length x row stride x 4 < 2 GiB
// (and a k-tile's 32 rows must fit a 32-bit vector offset whatever the reduction length)
static inline bool e3d_tr_span_ok(int64_t k_len, int64_t ld) {
    return k_len > 0 && ld > 0 && ld < ((int64_t)1 << 24) && k_len * ld * 4 < ((int64_t)1 << 31);
}

template <int ROWS, int NT>
struct StagerT {
    static constexpr int NV = ROWS * 8 / NT;
    static constexpr int PITCH = ROWS * 2, Q = ROWS / 4, KSTEP = NT / Q;   // KSTEP: k-rows one pass of the workgroup covers
    static_assert(KSTEP * NV == 32, "a k-tile is 32 k-rows");
    // Loads go through a BUFFER descriptor of the k-tile's rows (round 4): base = operand + k0 * ld and the byte count up to
    // the end of reduction row K - 1 are wave-uniform (SGPRs), a thread's offsets inside a k-tile never change -- so a load
    // needs no vector address arithmetic, and reduction rows >= K come back as ZEROS from the hardware's range check: the
    // per-k-step clamp (v_min + 64-bit multiply-add per item) and the tail masking (4 v_cndmask per item) of the flat form,
    // 48 of 254 instructions per k-step and wave in the ragged weight-gradient kernel, are gone.
    const float* ubase;     // operand base: wave-uniform (kernel argument / problem table)
    int64_t ld;
    int kk0, n4, lim;
    int voff[NV];           // byte offset of item i inside a k-tile: ((kk0 + KSTEP i) ld + first row) 4

    __device__ __forceinline__ void init(const float* base, int64_t ld_, int row0, int row_limit, int tid) {
        ld = ld_;
        ubase = base;
        lim = row_limit;
        n4 = tid % Q;
        kk0 = tid / Q;
        const int gr = min(row0 + 4 * n4, row_limit - 4);     // row_limit % 4 == 0 (checked by the launcher)
#pragma unroll
        for (int i = 0; i < NV; ++i) voff[i] = (int)(((int64_t)kk(i) * ld + gr) * 4);
    }
    __device__ __forceinline__ int kk(int i) const { return kk0 + KSTEP * i; }
    __device__ __forceinline__ int off(int i) const {
        const int k = kk(i);
        return k * PITCH + (((n4 >> 2) ^ (2 * (k & 3))) << 5) + (n4 & 3) * 8;
    }
    __device__ __forceinline__ __amdgpu_buffer_rsrc_t tile_rsrc(int k0, int K) const {
        // rows k0 .. K - 1 of the operand (callers keep k0 < K); the range check covers the vector offset only
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ubase + (int64_t)k0 * ld), 0,
                                                 (int)(((int64_t)(K - 1 - k0) * ld + lim) * 4), 0x00020000);
    }
    __device__ __forceinline__ void load_item(int i, f32x4& v, int k0, int K) const {
        v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(tile_rsrc(k0, K), voff[i], 0, 0));
    }
    __device__ __forceinline__ void load(f32x4 (&v)[NV], int k0, int K) const {
        const auto r = tile_rsrc(k0, K);
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff[i], 0, 0));
    }
    // (reduction rows past K inside a k-tile were loaded as zeros: nothing to mask per row.  What remains is the WHOLE tile past
    //  the end: the pipeline's last steps re-load the last k-tile under a logical k0 >= K -- a wave-uniform test)
    __device__ __forceinline__ f32x4 masked(int, const f32x4& v, int k0, int K) const { return k0 < K ? v : f32x4{0.f, 0.f, 0.f, 0.f}; }
    template <int NS, typename E>
    __device__ __forceinline__ void store_item(int i, const f32x4& v, unsigned char* img, int part_bytes, int, int) const {
        typename Vec<E>::x4 p[NS];
        split4<NS, E>(v, p);
        const int o = off(i);
#pragma unroll
        for (int s = 0; s < NS; ++s) *reinterpret_cast<typename Vec<E>::x4*>(img + s * part_bytes + o) = p[s];
    }
    template <int NS, typename E>
    __device__ __forceinline__ void store(const f32x4 (&v)[NV], unsigned char* img, int part_bytes, int k0, int K) const {
#pragma unroll
        for (int i = 0; i < NV; ++i) store_item<NS, E>(i, v[i], img, part_bytes, k0, K);
    }
    // (interface parity with Stager; the column sums of the TR form are kept per row quad by the caller)
    __device__ __forceinline__ float masked_sum(int, const f32x4&, int, int) const { return 0.f; }
};

// Transposed fragment of a [k][row] image (StagerT): rows r0 .. r0 + 31 (lane & 31), the 16-block ks of the k-tile.
// ``lane_off`` = tr_lane_off(PITCH, r0, lane): everything but the k-block.
__device__ __forceinline__ int tr_lane_off(int pitch, int r0, int lane) {
    const int q = (lane >> 2) & 3, g = (lane >> 4) & 1, half = lane >> 5;
    return (4 * half + q) * pitch + ((((r0 >> 4) + g) ^ (2 * q)) << 5) + 8 * (lane & 3);
}
template <typename X8>
__device__ __forceinline__ X8 tr_frag8(const unsigned char* img, int pitch, int lane_off, int ks) {
    typedef short short4v __attribute__((ext_vector_type(4)));
    typedef short short8v __attribute__((ext_vector_type(8)));
    typedef __attribute__((address_space(3))) short4v* lds_p;
    const unsigned char* p0 = img + 16 * ks * pitch + lane_off;
    const short4v a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p0));
    const short4v b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p0 + 8 * pitch));
    const short8v c = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(X8, c);
}

// WM = waves along M: 4 -> 256x128 tile, 512 threads; 2 -> 128x128 tile, 256 threads (used for grids of a few
// tiles only: it halves the padded rows of an M = 64 problem; at M = 4096 it measured 25 % slower than WM = 4).
// WN = waves along N (64 columns each).  WN = 1 (128 x 64 tiles of two waves, three workgroups per CU, twice the grid)
// was measured for the medium-M launches of training and trimmed sampling and is not dispatched: 45 us against 31 us
// for the 128 x 128 form at M = 4096, N = K = 768 (two waves per workgroup hide even less of the k-step chain).
// The workgroup body, shared by the plain kernel (one problem; block y = split-K slice) and the grouped
// weight-gradient kernel below (block -> (problem, tile), no split-K).  ``colsum_out`` (K-major x K-major layout only):
// the workgroups of tile column 0 also write out-row sums of A over the reduction index -- the bias gradient
// sum_m dz[m][n] of a linear layer, whose weight gradient dz^T . x this layout computes -- from the values they stage
// anyway.  ``accumulate``: out += instead of out = (a weight used twice in one backward pass).
template <int NS, int ACT, bool A_KMAJ, bool B_KMAJ, int WM, int WN, typename E, int TN = 2, bool TR = false>
__device__ __forceinline__ void gemm_split_body(unsigned char* smem_raw, const float* __restrict__ A, int64_t lda,
                                                const float* __restrict__ Bm, int64_t ldb,
                                                const float* __restrict__ bias, float* __restrict__ out, int64_t ldc,
                                                int M, int N, int K_total, int tiles_m, int tiles_n, int k_chunk,
                                                int bx, int by, int ny, float* __restrict__ colsum_out, int accumulate,
                                                float* __restrict__ absmax = nullptr, float out_scale = 1.0f) {
    constexpr int NBUF = NS <= 2 ? 2 : 1;
    constexpr int BM = WM * 64, BN = WN * 32 * TN, NT = WM * WN * 64;   // (BN shadows the file-scope 128)
    constexpr int A_BYTES = BM * ROW_B, B_BYTES = BN * ROW_B;
    constexpr int BUF_BYTES = NS * (A_BYTES + B_BYTES);  // per buffer: A parts, then B parts

    // split-K (weight gradients: few output tiles, long reduction over the tokens): block y reduces
    // k in [k_begin, K) and adds its partial tile atomically into the zero-initialised output
    const int k_begin = by * k_chunk;
    const int K = min(K_total, k_begin + k_chunk);
    const bool split = ny > 1;
    const int lid = bx;   // logical tile index (the callers apply xcd_remap)
    const int tm = lid / tiles_n, tn = lid % tiles_n;
    const int row0 = tm * BM, col0 = tn * BN;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid / WN, wc = wid % WN;
    const int l31 = lane & 31, half = lane >> 5;

    static_assert(!TR || (A_KMAJ && B_KMAJ), "the transposing staging exists for the K-major x K-major layout");
    typedef typename std::conditional<TR, StagerT<BM, NT>, Stager<BM, A_KMAJ, NT, (GEMM_LAB_RAW & 1) && !A_KMAJ && !B_KMAJ>>::type SA;
    typedef typename std::conditional<TR, StagerT<BN, NT>, Stager<BN, B_KMAJ, NT, (GEMM_LAB_RAW & 2) && !A_KMAJ && !B_KMAJ>>::type SB;
    SA sa;
    SB sb;
    sa.init(A, lda, row0, M, tid);
    sb.init(Bm, ldb, col0, N, tid);

    f32x16 acc[2][TN];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    f32x4 ra[SA::NV], rb[SB::NV];
    f32x4 ra1[SA::NV], rb1[SB::NV];   // second staging set (PIPE): odd k-tiles
    constexpr bool CS = A_KMAJ && B_KMAJ;          // row sums of A over k ride along (cheap: 3 adds per staged item)
    float cs[SA::NV];
    f32x4 cs4 = {0.f, 0.f, 0.f, 0.f};              // TR: an item is 4 ROWS of one k -- one sum per row of the thread's quad
#pragma unroll
    for (int i = 0; i < SA::NV; ++i) cs[i] = 0.f;
    auto cs_add = [&](int i, const f32x4& v, int k0) {
        if constexpr (TR) cs4 += sa.masked(i, v, k0, K);
        else cs[i] += sa.masked_sum(i, v, k0, K);
    };
    sa.load(ra, k_begin, K);
    sb.load(rb, k_begin, K);
    if (CS) {
#pragma unroll
        for (int i = 0; i < SA::NV; ++i) cs_add(i, ra[i], k_begin);
    }
    sa.template store<NS, E>(ra, smem_raw, A_BYTES, k_begin, K);
    sb.template store<NS, E>(rb, smem_raw + NS * A_BYTES, B_BYTES, k_begin, K);
    __syncthreads();

    const int nk = (K - k_begin + BK - 1) / BK;
    const int a_row = wr * 64 + l31, b_row = wc * (32 * TN) + l31;   // fragment rows of m / n tile 0 (tile 1: +32)
    int cur = 0;
    // Two buffers: interleaved staging as in the 256x256 kernel below -- each of the 4 MFMA groups of a k-step is
    // preceded by its share of the staging of tile t+1 (split + ds_write of a register item, then the re-issue of
    // that item's global load for tile t+2; the last steps re-stage the last tile into the buffer nobody reads).
    constexpr bool PIPE = NBUF == 2;
    constexpr int NA_ = SA::NV, NB_ = SB::NV;
    // DEEP: two staging register sets (below) for the K-contiguous layouts; the K-major layouts (4 dword loads per item)
    // measured 12 % SLOWER with twice the loads in flight (weight-gradient probe 1770 -> 1986 us) and keep one set
    constexpr bool DEEP = PIPE && ((!A_KMAJ && !B_KMAJ) || TR);   // (TR: float4 loads, as few as the K-contiguous layouts')
    int tr_a[2], tr_b[TN];                          // TR: per-lane offsets of the transposed fragments (m / n tile)
    if constexpr (TR) {
#pragma unroll
        for (int m = 0; m < 2; ++m) tr_a[m] = tr_lane_off(2 * BM, wr * 64 + 32 * m, lane);
#pragma unroll
        for (int n = 0; n < TN; ++n) tr_b[n] = tr_lane_off(2 * BN, wc * (32 * TN) + 32 * n, lane);
    }
    if (DEEP) {
        const int k1 = k_begin + (nk > 1 ? BK : 0), k2 = k_begin + (nk > 2 ? 2 : (nk > 1 ? 1 : 0)) * BK;
        sa.load(ra1, k1, K);
        sb.load(rb1, k1, K);
        sa.load(ra, k2, K);
        sb.load(rb, k2, K);
    } else if (PIPE) {
        const int k1 = k_begin + (nk > 1 ? BK : 0);
        sa.load(ra, k1, K);
        sb.load(rb, k1, K);
    }
    // Staging distance (DEEP): TWO register sets -- tile t lives in set t & 1; iteration kt stores tile kt + 1 from its set
    // into the other LDS buffer and re-issues that set's global loads for tile kt + 3, while the other set's loads (tile
    // kt + 2, issued an iteration ago) are still in flight.  With one set (rounds 1-2) a load had to arrive within ONE
    // k-step: at M = 4096 the k-step was its latency (the same launch with a third of the MFMAs, terms = 1, was only 10 %
    // faster), i.e. a CU never had more than ~32 KB in flight.
#ifdef GEMM_STAMPS
    const bool stamp_on = bx == GEMM_STAMPS && by == 0 && (wid & 3) == 0;
#endif
    auto k_step = [&](int kt, f32x4 (&xa)[NA_], f32x4 (&xb)[NB_]) {
        const bool more = kt + 1 < nk;
        GSTAMP(0);
        if (!PIPE && more) {
            sa.load(xa, k_begin + (kt + 1) * BK, K);
            sb.load(xb, k_begin + (kt + 1) * BK, K);
        }
        // the set just stored (tile kt + 1) is re-loaded with tile kt + 3 (DEEP: a load has TWO k-steps to arrive) / kt + 2
        constexpr int AHEAD = DEEP ? 3 : 2;
        const int k2 = k_begin + (kt + AHEAD < nk ? kt + AHEAD : nk - 1) * BK;
        unsigned char* nxt = smem_raw + (cur ^ 1) * BUF_BYTES;
        const unsigned char* ab = smem_raw + cur * BUF_BYTES;
        const unsigned char* bb = smem_raw + cur * BUF_BYTES + NS * A_BYTES;
        // form 3 (64x32 per wave: 6 fragment reads per 6 MFMAs and k16 block): the fragments of BOTH k16 blocks are
        // read before the first MFMA, so the second block's LDS latency hides under the first block's MFMAs
        constexpr bool PREF = PIPE && TN == 1 && E3D_GEMM_FRAG_PREFETCH;
        typename Vec<E>::x8 fa2[2][NS][2], fb2[2][NS][TN];
#pragma unroll
        for (int ks = 0; ks < (PREF ? 2 : 0); ++ks)
#pragma unroll
            for (int s = 0; s < NS; ++s) {
#pragma unroll
                for (int m = 0; m < 2; ++m)
                    fa2[ks][s][m] = *reinterpret_cast<const typename Vec<E>::x8*>(ab + s * A_BYTES + swz_off(a_row + 32 * m, 2 * ks + half));
#pragma unroll
                for (int n = 0; n < TN; ++n)
                    fb2[ks][s][n] = *reinterpret_cast<const typename Vec<E>::x8*>(bb + s * B_BYTES + swz_off(b_row + 32 * n, 2 * ks + half));
            }
#ifdef GEMM_STAMPS
        if (PREF) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); GSTAMP(1); }
#endif
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            typename Vec<E>::x8 fa[NS][2], fb[NS][TN];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    if constexpr (TR) fa[s][m] = tr_frag8<typename Vec<E>::x8>(ab + s * A_BYTES, 2 * BM, tr_a[m], ks);
                    else fa[s][m] = PREF ? fa2[ks][s][m]
                                         : *reinterpret_cast<const typename Vec<E>::x8*>(ab + s * A_BYTES + swz_off(a_row + 32 * m, 2 * ks + half));
                }
#pragma unroll
                for (int n = 0; n < TN; ++n) {
                    if constexpr (TR) fb[s][n] = tr_frag8<typename Vec<E>::x8>(bb + s * B_BYTES, 2 * BN, tr_b[n], ks);
                    else fb[s][n] = PREF ? fb2[ks][s][n]
                                         : *reinterpret_cast<const typename Vec<E>::x8*>(bb + s * B_BYTES + swz_off(b_row + 32 * n, 2 * ks + half));
                }
            }
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                if (PIPE) {
                    const int g = ks * 2 + m;
#pragma unroll
                    for (int i = 0; i < NA_; ++i)
                        if (i * 4 / NA_ == g) {
                            if (CS) cs_add(i, xa[i], k_begin + (kt + 1) * BK);
                            sa.template store_item<NS, E>(i, xa[i], nxt, A_BYTES, k_begin + (kt + 1) * BK, K);
                            sa.load_item(i, xa[i], k2, K);
                        }
#pragma unroll
                    for (int i = 0; i < NB_; ++i)
                        if (i * 4 / NB_ == g) {
                            sb.template store_item<NS, E>(i, xb[i], nxt + NS * A_BYTES, B_BYTES, k_begin + (kt + 1) * BK, K);
                            sb.load_item(i, xb[i], k2, K);
                        }
                }
#pragma unroll
                for (int n = 0; n < TN; ++n) {
                    // smallest terms first
                    if constexpr (NS == 3) {
                        acc[m][n] = mma16(fa[1][m], fb[1][n], acc[m][n]);
                        acc[m][n] = mma16(fa[0][m], fb[NS - 1][n], acc[m][n]);
                        acc[m][n] = mma16(fa[NS - 1][m], fb[0][n], acc[m][n]);
                    }
                    if constexpr (NS >= 2) {
                        acc[m][n] = mma16(fa[0][m], fb[1][n], acc[m][n]);
                        acc[m][n] = mma16(fa[1][m], fb[0][n], acc[m][n]);
                    }
                    acc[m][n] = mma16(fa[0][m], fb[0][n], acc[m][n]);
                }
                if (PIPE) __builtin_amdgcn_sched_barrier(0);   // keep the staging slices where they were put
            }
        }
        GSTAMP(2);
        if (PIPE) {
#ifdef GEMM_STAMPS
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // (what the barrier waits for anyway is told apart from the barrier)
            GSTAMP(3);
#endif
            __syncthreads();
            GSTAMP(4);
            cur ^= 1;
        } else {
            __syncthreads();  // everyone done reading the single buffer
            if (more) {
                if (CS) {
#pragma unroll
                    for (int i = 0; i < SA::NV; ++i) cs_add(i, xa[i], k_begin + (kt + 1) * BK);
                }
                sa.template store<NS, E>(xa, smem_raw, A_BYTES, k_begin + (kt + 1) * BK, K);
                sb.template store<NS, E>(xb, smem_raw + NS * A_BYTES, B_BYTES, k_begin + (kt + 1) * BK, K);
            }
            __syncthreads();
        }
        };
    for (int kt = 0; kt < nk; kt += 2) {
        if constexpr (DEEP) {
            k_step(kt, ra1, rb1);
            if (kt + 1 < nk) k_step(kt + 1, ra, rb);
        } else {
            k_step(kt, ra, rb);
            if (kt + 1 < nk) k_step(kt + 1, ra, rb);
        }
    }
    unsigned amax = 0;   // largest |out| written by this lane (absmax != nullptr: see e3d_absmax_accum)
#pragma unroll
    for (int n = 0; n < TN; ++n) {
        const int col = col0 + wc * (32 * TN) + n * 32 + l31;
        const float bv = (bias && col < N && by == 0) ? bias[col] : 0.f;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + wr * 64 + m * 32 + mfma32_row(r, half);
                float v = fmaf(acc[m][n][r], out_scale, bv);   // (out_scale = 1: acc + bv, bit for bit)
                if (ACT == E3D_ACT_GELU) v = gelu_erf(v);
                if (ACT == E3D_ACT_SILU) v = silu(v);
                if (row < M && col < N) {
                    float* o = out + (int64_t)row * ldc + col;
                    if (split) atomicAdd(o, v);
                    else *o = accumulate ? *o + v : v;
                    if (ACT == E3D_ACT_NONE && absmax) e3d_absmax_accum(amax, v);
                }
            }
        }
    }
    if (ACT == E3D_ACT_NONE && absmax) e3d_absmax_commit(amax, absmax, lane);
    if (CS && colsum_out && tn == 0) {
        // 8 threads (the 4-k groups of a k-tile) hold partial sums of each row: meet in LDS, summed in a fixed order.
        // (every wave is past its last fragment read: the k loop ends with a workgroup barrier)
        float* red = reinterpret_cast<float*>(smem_raw);
        if constexpr (TR) {   // the 8 threads that share a row quad (k-rows kk0, kk0 + 8, ...: groups of BM / 4 threads)
            static_assert(SA::KSTEP == 8, "8 partial sums per row, as in the dword scheme");
#pragma unroll
            for (int j = 0; j < 4; ++j) red[sa.kk0 * BM + 4 * sa.n4 + j] = cs4[j];
        } else {
#pragma unroll
            for (int i = 0; i < SA::NV; ++i) {
                const int f = tid + NT * i;
                red[(f / BM) * BM + f % BM] = cs[i];
            }
        }
        __syncthreads();
        if (tid < BM && row0 + tid < M) {
            float v = 0.f;
#pragma unroll
            for (int g = 0; g < 8; ++g) v += red[g * BM + tid];
            float* o = colsum_out + row0 + tid;
            if (split) atomicAdd(o, v);
            else *o = accumulate ? *o + v : v;
        }
    }
}

template <int NS, int ACT, bool A_KMAJ, bool B_KMAJ, int WM, int WN, typename E, int TN = 2, bool TR = false>
__global__ __launch_bounds__(WM * WN * 64) void gemm_split_kernel(const float* __restrict__ A, int64_t lda,
                                                         const float* __restrict__ Bm, int64_t ldb,
                                                         const float* __restrict__ bias,
                                                         float* __restrict__ out, int64_t ldc, int M,
                                                         int N, int K_total, int tiles_m, int tiles_n,
                                                         int k_chunk, float* __restrict__ absmax, float out_scale) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    gemm_split_body<NS, ACT, A_KMAJ, B_KMAJ, WM, WN, E, TN, TR>(smem_raw, A, lda, Bm, ldb, bias, out, ldc, M, N, K_total, tiles_m,
                                                        tiles_n, k_chunk, xcd_remap(blockIdx.x, tiles_m * tiles_n),
                                                        blockIdx.y, gridDim.y, nullptr, 0, absmax, out_scale);
}

// Weight gradients of up to 64 linear layers of ONE shape in one launch: problem p computes dW_p[N,K] = dz_p^T . x_p
// (dz_p [M,N], x_p [M,K]: both K-major in this kernel's terms, reduction over the M tokens) and db_p[N] = column sums
// of dz_p.  A layer alone has too few output tiles for the chip (18 of 256x128 for a 768x768 weight) and had to cut the
// reduction into split-K slices that meet through atomics on a zeroed output (82 TFLOP/s at M = 4096); a model's layers
// together fill it several times over with whole reductions (230-270 TFLOP/s measured on the same kernel body:
// tools/lab/wgrad_steady.py), without atomics or memsets: deterministic.
struct WgradGroup {
    const float* dz[64];
    const float* x[64];
    float* dw[64];
    float* db[64];
    unsigned long long accumulate;   // bit p: dW_p / db_p += (the weight already holds a gradient)
};

template <int NS, bool TR>
__global__ __launch_bounds__(512) void gemm_wgrad_grouped_kernel(const WgradGroup g, int64_t ldz, int64_t ldx, int N, int K,
                                                                 int M, int tiles_m, int tiles_n, int count) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tiles = tiles_m * tiles_n;
    const int lid = xcd_remap(blockIdx.x, tiles * count);
    const int p = lid / tiles;
    gemm_split_body<NS, E3D_ACT_NONE, true, true, 4, 2, __bf16, 2, TR>(smem_raw, g.dz[p], ldz, g.x[p], ldx, nullptr, g.dw[p], K,
                                                                        N, K, M, tiles_m, tiles_n, M, lid - p * tiles, 0, 1,
                                                                        g.db[p], (int)((g.accumulate >> p) & 1));
}

// The same launch for layers of DIFFERENT shapes and row strides (one token count): a model's weight gradients then need
// ceil(layers / 64) launches whose grids are whole multiples of the chip but for one tail, instead of one launch per
// (shape, stride) group with a partial last round each (structure model: 22 rounds of 256 workgroups for 17.3 rounds of
// tiles).  Problem p owns tiles [tile_start[p], tile_start[p + 1]); a workgroup finds its problem by bisection over the
// kernel-argument table (wave-uniform: scalar loads).
struct WgradRagged {
    const float* dz[64];
    const float* x[64];
    float* dw[64];
    float* db[64];                   // may be null per problem (no bias)
    int N[64], K[64], ldz[64], ldx[64];
    int tile_start[65];
    unsigned long long accumulate;
};
static_assert(sizeof(WgradRagged) <= 3400, "kernel arguments are limited to 4 KB");

template <int NS, bool TR>
__global__ __launch_bounds__(512) void gemm_wgrad_ragged_kernel(const WgradRagged g, int M, int count) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int lid = xcd_remap(blockIdx.x, g.tile_start[count]);
    int lo = 0, hi = count;          // the largest p with tile_start[p] <= lid
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (g.tile_start[mid] <= lid) lo = mid;
        else hi = mid;
    }
    const int p = lo, N = g.N[p], K = g.K[p];
    const int tiles_m = (N + 255) / 256, tiles_n = (K + 127) / 128;
    gemm_split_body<NS, E3D_ACT_NONE, true, true, 4, 2, __bf16, 2, TR>(smem_raw, g.dz[p], (int64_t)g.ldz[p], g.x[p], (int64_t)g.ldx[p],
                                                                        nullptr, g.dw[p], K, N, K, M, tiles_m, tiles_n, M,
                                                                        lid - g.tile_start[p], 0, 1, g.db[p],
                                                                        (int)((g.accumulate >> p) & 1));
}

// ---------------------------------------------------------------------------------------------
// Large-M forward variant: 256x256x32 workgroup tile, 8 waves as 2(M) x 4(N), each wave 128x64
// (4x2 MFMA tiles, 128 accumulator registers).  Versus the 256x128 kernel it moves 2/3 of the
// global->LDS bytes and 3/4 of the LDS fragment bytes per flop.  LDS rows are unpadded 64 bytes
// (so that two buffers of 2 x 512 rows fit) with the 16-byte k-chunk XOR-swizzled by (row >> 2) & 3:
// the 16 lanes of a ds_read_b128 group (4 row quads) then hit 16 distinct 4-bank slots.
constexpr int BT = 256, ROW64 = ROW_B;

#ifdef E3D_STAMPS   // lab builds only (tools/lab_gemm_stamps.py): in-kernel phase time stamps of one workgroup
__device__ long long e3d_stamps[2][32][8];
#define STAMP(slot)                                                                              \
    do {                                                                                         \
        if (stamp_on && kt < 32 && lane == 0) e3d_stamps[wr][kt][slot] = __builtin_readcyclecounter(); \
    } while (0)
#define STAMP_K(i)                                                                               \
    do {                                                                                         \
        if (blockIdx.x == 300 && (threadIdx.x & 255) == 0) e3d_stamps[threadIdx.x >> 8][i][7] = __builtin_readcyclecounter(); \
    } while (0)
#else
#define STAMP(slot) do {} while (0)
#define STAMP_K(i) do {} while (0)
#endif

// WR x WC waves, each 128x64: <2,4> = 256x256 tile, 512 threads, two LDS buffers, one workgroup per CU.
// (<2,2> = 256x128 with one buffer and two workgroups per CU was measured 10 % slower and is not dispatched.)
//
// PIPE (two buffers): the staging of tile t+1 is spread over the 8 MFMA groups of iteration t instead
// of being one phase at its end.  In-kernel time stamps showed the classic loop as a serial sum --
// ~2000 cycles for the 8 waves to push their 64 KB of global loads through the texture-address unit,
// ~2400 of MFMA + fragment reads, ~1400 of split + LDS writes, barrier -- with the matrix pipe idle
// in two of the three.  Here each MFMA group is preceded by 1/8 of the staging: split + ds_write of
// one register item of tile t+1 (loaded during iteration t-1), then the re-issue of that item's
// global load for tile t+2 (one register set, distance 2), so loads, VALU and LDS writes all run
// under the MFMAs of the other wave of the SIMD.
template <int NS, int ACT, int WR, int WC, int NBUF, bool PIPE, typename E>
__global__ __launch_bounds__(WR * WC * 64, 2) void gemm_split256_kernel(const float* __restrict__ A, int64_t lda,
                                                                        const float* __restrict__ W,
                                                                        const float* __restrict__ bias,
                                                                        float* __restrict__ out, int64_t ldc,
                                                                        int M, int N, int K, int tiles_m,
                                                                        int tiles_n, float* __restrict__ absmax, float out_scale) {
    static_assert(!PIPE || NBUF == 2, "PIPE needs two LDS buffers");
    STAMP_K(0);   // kernel entry
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int NT = WR * WC * 64, TM = WR * 128, TN = WC * 64;
    constexpr int A_BYTES = TM * ROW64, B_BYTES = TN * ROW64;   // one operand, one part
    constexpr int BUF_BYTES = NS * (A_BYTES + B_BYTES);         // A parts then B parts
    constexpr int NA = TM * 8 / NT, NB = TN * 8 / NT;           // float4 items per thread
    constexpr int NI = NA + NB;

    const int lid = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int tm = lid / tiles_n, tn = lid % tiles_n;
    const int row0 = tm * TM, col0 = tn * TN;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid / WC, wc = wid % WC;
    const int l31 = lane & 31, half = lane >> 5;

    // staging items: float4 f = tid + NT i -> row f>>3, k-group f&7 (4 floats); items [0, NA) are A's
    const float* src[NI];
    int dst[NI];   // byte offset of the item's first part inside a buffer
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const bool is_a = i < NA;
        const int f = tid + NT * (is_a ? i : i - NA), r = f >> 3, kg = f & 7;
        if (is_a) {
            int ar = row0 + r;
            ar = ar < M ? ar : M - 1;
            src[i] = A + (int64_t)ar * lda + kg * 4;
        } else {
            src[i] = W + (int64_t)(col0 + r) * K + kg * 4;
        }
        dst[i] = (is_a ? 0 : NS * A_BYTES) + swz_off(r, kg >> 1) + (kg & 1) * 8;
    }

    f32x16 acc[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    f32x4 rg[NI];
    auto item_load = [&](int i, int kt) { rg[i] = *reinterpret_cast<const f32x4*>(src[i] + kt * BK); };
    auto item_store = [&](int i, unsigned char* buf) {
        typename Vec<E>::x4 p[NS];
        split4<NS, E>(rg[i], p);
#pragma unroll
        for (int s = 0; s < NS; ++s)
            *reinterpret_cast<typename Vec<E>::x4*>(buf + s * (i < NA ? A_BYTES : B_BYTES) + dst[i]) = p[s];
    };

    STAMP_K(1);   // addressing done
    const int nk = K / BK;
#pragma unroll
    for (int i = 0; i < NI; ++i) item_load(i, 0);
#pragma unroll
    for (int i = 0; i < NI; ++i) item_store(i, smem_raw);
    if (PIPE) {
#pragma unroll
        for (int i = 0; i < NI; ++i) item_load(i, nk > 1 ? 1 : 0);
    }
    __syncthreads();

    int a_row[4], b_row[2];
#pragma unroll
    for (int m = 0; m < 4; ++m) a_row[m] = wr * 128 + m * 32 + l31;
#pragma unroll
    for (int n = 0; n < 2; ++n) b_row[n] = wc * 64 + n * 32 + l31;
    STAMP_K(2);   // prologue done (first tile staged)
    int cur = 0;
#ifdef E3D_STAMPS
    const bool stamp_on = blockIdx.x == 300 && (wid & 3) == 0;
#endif
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        STAMP(0);
        if (!PIPE && more) {
#pragma unroll
            for (int i = 0; i < NI; ++i) item_load(i, kt + 1);
        }
        STAMP(1);
        const unsigned char* base = smem_raw + cur * BUF_BYTES;
        unsigned char* next = smem_raw + (cur ^ 1) * BUF_BYTES;
        // PIPE: tile kt+1 sits in rg[]; tile kt+2 (clamped: the last two iterations re-load and
        // re-stage the last tile into the buffer nobody reads again) replaces it item by item
        const int kt2 = kt + 2 < nk ? kt + 2 : nk - 1;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            typename Vec<E>::x8 fa[NS][4], fb[NS][2];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    fb[s][n] = *reinterpret_cast<const typename Vec<E>::x8*>(base + NS * A_BYTES + s * B_BYTES +
                                                                swz_off(b_row[n], 2 * ks + half));
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    fa[s][m] = *reinterpret_cast<const typename Vec<E>::x8*>(base + s * A_BYTES + swz_off(a_row[m], 2 * ks + half));
            }
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                if (PIPE) {
                    const int g = ks * 4 + m;   // MFMA group 0..7 of this iteration
#pragma unroll
                    for (int i = 0; i < NI; ++i)
                        if (i * 8 / NI == g) {
                            item_store(i, next);
                            item_load(i, kt2);
                        }
                }
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    if constexpr (NS == 3) {
                        acc[m][n] = mma16(fa[1][m], fb[1][n], acc[m][n]);
                        acc[m][n] = mma16(fa[0][m], fb[NS - 1][n], acc[m][n]);
                        acc[m][n] = mma16(fa[NS - 1][m], fb[0][n], acc[m][n]);
                    }
                    if constexpr (NS >= 2) {
                        acc[m][n] = mma16(fa[0][m], fb[1][n], acc[m][n]);
                        acc[m][n] = mma16(fa[1][m], fb[0][n], acc[m][n]);
                    }
                    acc[m][n] = mma16(fa[0][m], fb[0][n], acc[m][n]);
                }
                if (PIPE) __builtin_amdgcn_sched_barrier(0);   // keep the staging slices where they were put
            }
            STAMP(2 + ks);   // MFMAs of this k-step issued
        }
        if (PIPE) {
            STAMP(5);
            __syncthreads();
            STAMP(6);
            cur ^= 1;
        } else if (NBUF == 2) {
#ifdef E3D_STAMPS
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            STAMP(4);        // next tile's global loads landed
#endif
            if (more) {
#pragma unroll
                for (int i = 0; i < NI; ++i) item_store(i, next);
            }
            STAMP(5);        // split + LDS writes issued
            __syncthreads();
            STAMP(6);
            cur ^= 1;
        } else {
            __syncthreads();
            if (more) {
#pragma unroll
                for (int i = 0; i < NI; ++i) item_store(i, smem_raw);
            }
            __syncthreads();
        }
    }

    STAMP_K(3);   // k loop done
    // (m outer, n inner: the other loop order -- one bias load per column block -- costs 36-84 B/lane of scratch)
    unsigned amax = 0;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int col = col0 + wc * 64 + n * 32 + l31;
            const float bv = bias ? bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + wr * 128 + m * 32 + mfma32_row(r, half);
                float v = fmaf(acc[m][n][r], out_scale, bv);   // (out_scale = 1: acc + bv, bit for bit)
                if (ACT == E3D_ACT_GELU) v = gelu_erf(v);
                if (ACT == E3D_ACT_SILU) v = silu(v);
                if (row < M) {
                    out[(int64_t)row * ldc + col] = v;
                    if (ACT == E3D_ACT_NONE && absmax) e3d_absmax_accum(amax, v);
                }
            }
        }
    }
    if (ACT == E3D_ACT_NONE && absmax) e3d_absmax_commit(amax, absmax, lane);
    STAMP_K(4);   // stores issued
#ifdef E3D_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP_K(5);   // stores retired
#endif
}

// ---------------------------------------------------------------------------------------------
// Persistent form of the interleaved-staging kernel (bf16x3, 256x256 tiles, M % 256 == 0): one
// workgroup per CU walks tiles b, b + G, b + 2G, ... and the k-tile stream does not stop at a tile
// boundary -- while the last two k-steps of a tile run, the first two k-tiles of the NEXT output
// tile are loaded and staged, so a tile costs its k loop plus its output stores; the first-load
// latency (~11k cycles per tile in the stamps) and the workgroup relaunch are paid once per CU.
// Staging addresses are a wave-uniform tile base (SGPRs) plus a per-item 32-bit offset that is the
// same for every tile.
// -DE3D_NT_STORES (lab): streaming (non-temporal) output stores.  +2-3.5 % per launch in isolation, nothing inside
// the model step (the consumer kernel then finds less of the output in the memory-side cache), so not the default.
#ifdef E3D_NT_STORES
#define E3D_STORE_OUT(ptr, val) __builtin_nontemporal_store((val), (ptr))
#else
#define E3D_STORE_OUT(ptr, val) (*(ptr) = (val))
#endif

template <int ACT, typename E>
__global__ __launch_bounds__(512, 2) void gemm_split256p_kernel(const float* __restrict__ A, int64_t lda,
                                                                 const float* __restrict__ W,
                                                                 const float* __restrict__ bias,
                                                                 float* __restrict__ out, int64_t ldc, int N, int K,
                                                                 int tiles_m, int tiles_n, float* __restrict__ absmax, float out_scale) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int NS = 2, T_BYTES = BT * ROW64, BUF_BYTES = 2 * NS * T_BYTES;
    constexpr int NI = 8;   // float4 items per thread and k-tile: 0..3 from A, 4..7 from W
    const int total = tiles_m * tiles_n, nk = K / BK;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 2, wc = wid & 3;
    const int l31 = lane & 31, half = lane >> 5;

    unsigned goff[NI];   // element offset from the tile base
    int dst[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int f = tid + 512 * (i & 3), r = f >> 3, kg = f & 7;
        goff[i] = (unsigned)(r * (i < 4 ? (int)lda : K) + kg * 4);
        dst[i] = (i < 4 ? 0 : NS * T_BYTES) + swz_off(r, kg >> 1) + (kg & 1) * 8;
    }

    // load cursor: the k-tile the next global loads fetch (wave-uniform)
    int ld_tile = blockIdx.x, ld_k = 0;
    const float* ld_a;
    const float* ld_w;
    auto cursor_bases = [&]() {
        const int lid = xcd_remap(ld_tile, total);
        ld_a = A + (int64_t)(lid / tiles_n) * BT * lda + ld_k * BK;
        ld_w = W + (int64_t)(lid % tiles_n) * BT * K + ld_k * BK;
    };
    auto cursor_advance = [&]() {
        if (ld_k + 1 < nk) {
            ++ld_k;
            ld_a += BK;
            ld_w += BK;
        } else if (ld_tile + (int)gridDim.x < total) {
            ld_tile += gridDim.x;
            ld_k = 0;
            cursor_bases();
        }   // else: stay on the stream's last k-tile (re-staged into a buffer nobody reads again)
    };
    f32x4 rg[NI];
    // (the timing-only ablations and operand-format prototypes of rounds 1-2 live in
    // tools/lab/archive/gemm_split_r02_lab_switches.hip.txt: nothing in this file computes a wrong result by a -D flag)
    auto item_load = [&](int i) { rg[i] = *reinterpret_cast<const f32x4*>((i < 4 ? ld_a : ld_w) + goff[i]); };
    auto item_store = [&](int i, unsigned char* buf) {
        typename Vec<E>::x4 p[NS];
        split4<NS, E>(rg[i], p);
#pragma unroll
        for (int s = 0; s < NS; ++s) *reinterpret_cast<typename Vec<E>::x4*>(buf + s * T_BYTES + dst[i]) = p[s];
    };
    cursor_bases();
#pragma unroll
    for (int i = 0; i < NI; ++i) item_load(i);
    cursor_advance();
#pragma unroll
    for (int i = 0; i < NI; ++i) item_store(i, smem_raw);
#pragma unroll
    for (int i = 0; i < NI; ++i) item_load(i);
    cursor_advance();
    __syncthreads();

    int a_row[4], b_row[2];
#pragma unroll
    for (int m = 0; m < 4; ++m) a_row[m] = wr * 128 + m * 32 + l31;
#pragma unroll
    for (int n = 0; n < 2; ++n) b_row[n] = wc * 64 + n * 32 + l31;
    int cur = 0;
    unsigned amax = 0;   // largest |out| of this lane over all its tiles (absmax != nullptr)

    for (int tile = blockIdx.x; tile < total; tile += gridDim.x) {
        f32x16 acc[4][2];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
        for (int kt = 0; kt < nk; ++kt) {
            const unsigned char* base = smem_raw + cur * BUF_BYTES;
            unsigned char* next = smem_raw + (cur ^ 1) * BUF_BYTES;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                typename Vec<E>::x8 fa[NS][4], fb[NS][2];
#pragma unroll
                for (int s = 0; s < NS; ++s) {
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        fb[s][n] = *reinterpret_cast<const typename Vec<E>::x8*>(base + (NS + s) * T_BYTES +
                                                                    swz_off(b_row[n], 2 * ks + half));
#pragma unroll
                    for (int m = 0; m < 4; ++m)
                        fa[s][m] = *reinterpret_cast<const typename Vec<E>::x8*>(base + s * T_BYTES + swz_off(a_row[m], 2 * ks + half));
                }
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const int g = ks * 4 + m;   // MFMA group g stages item g: stream k-tile +1 out, +2 in
                    item_store(g, next);
                    item_load(g);
#pragma unroll
                    for (int n = 0; n < 2; ++n) {
                        acc[m][n] = mma16(fa[0][m], fb[1][n], acc[m][n]);
                        acc[m][n] = mma16(fa[1][m], fb[0][n], acc[m][n]);
                        acc[m][n] = mma16(fa[0][m], fb[0][n], acc[m][n]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            cursor_advance();
            __syncthreads();
            cur ^= 1;
        }

        // epilogue: rows m * 32 .. of this wave's 128 x 64 part, one 32-row group at a time (this loop order keeps the
        // kernel free of scratch spills: the n-outer order with one bias load per column block cost 12-36 B/lane)
        const int lid = xcd_remap(tile, total);
        const int row0 = (lid / tiles_n) * BT, col0 = (lid % tiles_n) * BT;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const int col = col0 + wc * 64 + n * 32 + l31;
                const float bv = bias ? bias[col] : 0.f;
                float* o = out + (int64_t)(row0 + wr * 128 + m * 32 + 4 * half) * ldc + col;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = fmaf(acc[m][n][r], out_scale, bv);   // (out_scale = 1: acc + bv, bit for bit)
                    if (ACT == E3D_ACT_GELU) v = gelu_erf(v);
                    if (ACT == E3D_ACT_SILU) v = silu(v);
                    E3D_STORE_OUT(&o[(int64_t)((r & 3) + 8 * (r >> 2)) * ldc], v);
                    if (ACT == E3D_ACT_NONE && absmax) e3d_absmax_accum(amax, v);
                }
            }
        }
    }
    if (ACT == E3D_ACT_NONE && absmax) e3d_absmax_commit(amax, absmax, lane);
}

template <int ACT, typename E>
int launch256p(const float* A, int64_t lda, const float* W, const float* bias, float* out, int64_t ldc, int M, int N,
               int K, Epi epi, hipStream_t s) {
    const int tiles_m = M / BT, tiles_n = N / BT;
    constexpr size_t lds = 2 * 2 * 2 * BT * ROW64;
    static std::atomic<uint64_t> lds_ok{0};
    const int n_cu = e3d_cu_count();
    const int total = tiles_m * tiles_n;
    e3d_allow_lds(lds_ok, gemm_split256p_kernel<ACT, E>, lds);
    hipLaunchKernelGGL((gemm_split256p_kernel<ACT, E>), dim3(total < n_cu ? total : n_cu), dim3(512), lds, s, A, lda, W, bias, out,
                       ldc, N, K, tiles_m, tiles_n, epi.absmax, epi.scale);
    return e3d_launch_status("e3d_gemm_f32_split (persistent 256x256)");
}

int g_general_form = -1;   // E3D_GEMM_FORM, e3d_gemm_general_select: 0 = by shape
int g_tile_pref = -1;  // E3D_GEMM_TILE (A/B runs): 0 = 256x128 (8 waves of 64x64) for every shape, 1 = 256x256 classic
                       // loop, 3 = 256x256 with interleaved staging, 4 = + persistent (default), 5 = persistent for EVERY tile count (lab)

template <int NS, int ACT, int WR, int WC, int NBUF, bool PIPE, typename E>
int launch256(const float* A, int64_t lda, const float* W, const float* bias, float* out, int64_t ldc, int M, int N,
              int K, Epi epi, hipStream_t s) {
    constexpr int TM = WR * 128, TN = WC * 64;
    const int tiles_m = (M + TM - 1) / TM, tiles_n = N / TN;
    const size_t lds = (size_t)NBUF * NS * (TM + TN) * ROW64;
    static std::atomic<uint64_t> lds_ok{0};
    e3d_allow_lds(lds_ok, gemm_split256_kernel<NS, ACT, WR, WC, NBUF, PIPE, E>, lds);
    hipLaunchKernelGGL((gemm_split256_kernel<NS, ACT, WR, WC, NBUF, PIPE, E>), dim3(tiles_m * tiles_n), dim3(WR * WC * 64), lds, s,
                       A, lda, W, bias, out, ldc, M, N, K, tiles_m, tiles_n, epi.absmax, epi.scale);
    return e3d_launch_status("e3d_gemm_f32_split (128x64 wave tiles)");
}

template <int NS, int ACT, bool A_KMAJ, bool B_KMAJ, int WM, int WN, typename E, int TN = 2, bool TR = false>
int launch_general(const float* A, int64_t lda, const float* B, int64_t ldb, const float* bias, float* out, int64_t ldc,
                   int M, int N, int K, Epi epi, hipStream_t s);

// smallest number of 256x256 tiles for which the persistent kernel is chosen (E3D_GEMM_P_MIN, experiments)
int p_min() {
    static int v = 0;
    if (!v) {
        const char* e = getenv("E3D_GEMM_P_MIN");
        v = e ? atoi(e) : 160;   // measured: 192 tiles (M = 16384, N = 768) +9 % on the persistent kernel, 96 tiles -8 %
    }
    return v;
}

template <int NS, int ACT, bool A_KMAJ, bool B_KMAJ, typename E>
int launch(const float* A, int64_t lda, const float* B, int64_t ldb, const float* bias, float* out, int64_t ldc,
           int M, int N, int K, Epi epi, hipStream_t s) {
    if (g_tile_pref < 0) {
        const char* e = getenv("E3D_GEMM_TILE");
        g_tile_pref = e ? atoi(e) : 4;
    }
    if constexpr (!A_KMAJ && !B_KMAJ) {
        if constexpr (NS == 2) {
            // the persistent kernel from 160 tiles upwards -- and from 96 when the output is at least 6 tiles wide (N >= 1536):
            // there the 128x128 form needs 1.5+ rounds of 256 CUs (M = 4096, N = 1536: 384 tiles), one 256x256 tile per CU
            // is faster (structure training step 32.0 -> 31.1 ms); narrow outputs with 96-159 tiles (M = 8192, N = 768)
            // measured 6 % slower on it (sequence step 25.0 -> 26.6 ms) and stay on the 128x128 form
            const int64_t n_tiles = (int64_t)(M / BT) * (N / BT);
            if (N % BT == 0 && M % BT == 0 && ldb == K && g_tile_pref >= 4 && K >= 2 * BK && lda < (1 << 22) &&
                (g_tile_pref >= 5 || n_tiles >= p_min() || (n_tiles >= 96 && N >= 6 * BT && !getenv("E3D_GEMM_P_MIN"))))
                return launch256p<ACT, E>(A, lda, B, bias, out, ldc, M, N, K, epi, s);
            if (N % BT == 0 && ldb == K && g_tile_pref >= 3 && (int64_t)((M + BT - 1) / BT) * (N / BT) >= 256)
                return launch256<NS, ACT, 2, 4, 2, true, E>(A, lda, B, bias, out, ldc, M, N, K, epi, s);
        }
        if constexpr (NS == 1) {   // single-product form: the interleaved-staging 256x256 kernel from 160 tiles upwards
            if (N % BT == 0 && ldb == K && g_tile_pref >= 3 && (int64_t)((M + BT - 1) / BT) * (N / BT) >= 160)
                return launch256<NS, ACT, 2, 4, 2, true, E>(A, lda, B, bias, out, ldc, M, N, K, epi, s);
        }
        if (N % BT == 0 && ldb == K && g_tile_pref >= 1 &&
            (int64_t)((M + BT - 1) / BT) * (N / BT) >= 256)   // enough 256x256 tiles to fill the 256 CUs
            return launch256<NS, ACT, 2, 4, (NS <= 2 ? 2 : 1), false, E>(A, lda, B, bias, out, ldc, M, N, K, epi, s);
    }
    // general kernel, three tile forms: 1 = 256x128 (8 waves, one workgroup per CU: 96 KB of LDS), 2 = 128x128 (4 waves,
    // two per CU), 3 = 128x128 on 8 waves (each 64x32; forward and input-gradient layouts of the 2-term kernels).  Both are bound by the latency of a workgroup's own k-step chain at these sizes, so the choice is a
    // matter of rounds (tools/lab/gemm_forms_ab.py, K = 768: a 128x128 workgroup alone on its CU takes ~29 us, ~45 us
    // when it shares the CU; a 256x128 one ~37 us): E3D_GEMM_FORM / e3d_gemm_general_select force a form.
    if (g_general_form < 0) {
        const char* e = getenv("E3D_GEMM_FORM");
        g_general_form = e ? atoi(e) : 0;
    }
    int form = g_general_form;
    if (form != 1 && form != 2 && form != 3 && form != 4) {
        const int64_t g256 = (int64_t)((M + 255) / 256) * ((N + 127) / 128), g128 = (int64_t)((M + 127) / 128) * ((N + 127) / 128);
        const int64_t cus = e3d_cu_count();
        if (NS == 3) form = g256 < 128 && !A_KMAJ && !B_KMAJ ? 2 : 1;       // 3-term kernels: as measured in round 1
        else if (A_KMAJ && B_KMAJ) form = 1;                                // weight gradients (split-K over one resident round): +4 %
        else if (A_KMAJ) form = g128 <= cus ? 2 : (g256 <= cus ? 1 : (45 * ((g128 + 2 * cus - 1) / (2 * cus)) < 37 * ((g256 + cus - 1) / cus) ? 2 : 1));
        // forward / input-gradient layouts: the 8-wave 128x128 form (3) against the 256x128 form (1) by rounds of one
        // workgroup per CU (measured at K = 768: ~25 us and ~38 us per round; form 3 at M = 4096: N = 768 26.8 us
        // against 32.5 (4-wave 128x128) / 36.9, N = 2304 70 against 81; M = 8192: form 1 by 4 %)
        else form = 25 * ((g128 + cus - 1) / cus) <= 38 * ((g256 + cus - 1) / cus) ? 3 : 1;
        // small launches of the forward layout (M <= 2048 rows, N <= 1024: a decoder on trimmed frames, a few pockets): 128x64
        // tiles on four waves, up to three workgroups per CU with barriers of their own -- 18.8 against 22.4 us at M = 1024, N = K =
        // 768, 19.6 against 22.8 at N = 1024, 19.7 against 23.0 at M = 2048 (profiles/r04_gemm_mid_m_ab.log); in the step:
        // profiles/r04_gemm_form4_small_m_in_step.log.  (At M = 4096 and above the same form changed nothing in the step: below.)
        if (NS == 2 && !A_KMAJ && !B_KMAJ && M <= 2048 && N <= 1024) form = 4;
    }
    // form 4 (round 4, selectable only: E3D_GEMM_FORM=4 / e3d_gemm_general_select(4)): 128x64 tiles on FOUR waves -- 48 KB of LDS,
    // so up to three workgroups share a CU with barriers of their own.  Standalone (hot operands, tools/lab/gemm_forms_ab.py) a
    // CU carrying 1 / 2 / 3 such workgroups takes ~20 / 28 / 41 us at K = 768 against ~24.5 us per round of 128x128 tiles (M =
    // 2048 x N = 768: 20.1 vs 23.5 us, M = 8192 x N = 768: 42.6 vs 48.1) -- and inside the sequence training step (M = 8192,
    // cold operands, neighbours on the queue) a shape rule built on those figures changed nothing: 19.9 vs 19.9 ms, three
    // interleaved pairs (profiles/r04_gemm_form4_128x64_ab.log).  Dispatched by shape only for M <= 2048 (above).
    if constexpr (NS == 2 && !A_KMAJ && !B_KMAJ) {
        if (form == 4) return launch_general<NS, ACT, A_KMAJ, B_KMAJ, 2, 2, E, 1>(A, lda, B, ldb, bias, out, ldc, M, N, K, epi, s);
    }
    if (form == 4) form = 3;
    if constexpr (NS <= 2 && !A_KMAJ) {
        // form 3: the 128x128 tile on EIGHT waves (2 x 4, each 64x32): two waves per SIMD cover each other's staging.
        // (The same form with k-tiles of 64 -- half the k-steps, 128 KB of LDS -- was built and measured: 27.3 against
        // 26.6 us at M = 4096, N = K = 768, bit-identical results; a 12-workgroup launch takes 23.5 us either way, so
        // the k-step's cost is proportional to its bytes, not a fixed barrier / latency term: here the fragment reads
        // (1 ds_read_b128 per MFMA at 64x32 per wave) cost as much LDS time as the MFMAs cost matrix-pipe time.)
        if (form == 3) return launch_general<NS, ACT, A_KMAJ, B_KMAJ, 2, 4, E, 1>(A, lda, B, ldb, bias, out, ldc, M, N, K, epi, s);
    }
    if (form == 3) form = 2;
    if (form == 2) return launch_general<NS, ACT, A_KMAJ, B_KMAJ, 2, 2, E>(A, lda, B, ldb, bias, out, ldc, M, N, K, epi, s);
    if constexpr (A_KMAJ && B_KMAJ) {   // weight-gradient layout: the transposing staging when both operands come in whole quads
        static const bool tr_on = !getenv("E3D_WGRAD_TR") || atoi(getenv("E3D_WGRAD_TR")) != 0;
        // (the transposing stager addresses a k-tile's rows through a buffer descriptor: 32-bit byte counts, so the reduction
        //  length times the row stride must stay under 2 GiB -- e3d_tr_span_ok; wider problems keep the dword staging)
        if (tr_on && M % 4 == 0 && N % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 && ((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0 &&
            e3d_tr_span_ok(K, lda) && e3d_tr_span_ok(K, ldb))
            return launch_general<NS, ACT, A_KMAJ, B_KMAJ, 4, 2, E, 2, true>(A, lda, B, ldb, bias, out, ldc, M, N, K, epi, s);
    }
    return launch_general<NS, ACT, A_KMAJ, B_KMAJ, 4, 2, E>(A, lda, B, ldb, bias, out, ldc, M, N, K, epi, s);
}

template <int NS, int ACT, bool A_KMAJ, bool B_KMAJ, int WM, int WN, typename E, int TN, bool TR>
int launch_general(const float* A, int64_t lda, const float* B, int64_t ldb, const float* bias, float* out, int64_t ldc,
                   int M, int N, int K, Epi epi, hipStream_t s) {
    constexpr int BM = WM * 64, BNt = WN * 32 * TN;
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BNt - 1) / BNt;
    constexpr int NBUF = NS <= 2 ? 2 : 1;
    size_t lds = (size_t)NBUF * NS * (BM + BNt) * ROW_B;
    // A grid that fits the chip in one round at one workgroup per CU should run that way: two 128x128 workgroups fit a
    // CU's LDS, and the dispatcher does pair them up while other CUs stay empty (a workgroup alone on its CU ~29 us,
    // sharing it ~45 us at K = 768).  Asking for more than half of the LDS keeps them apart.  E3D_GEMM_SPREAD=0: off.
    static const bool spread = !getenv("E3D_GEMM_SPREAD") || atoi(getenv("E3D_GEMM_SPREAD")) != 0;
    if (spread && BM * BNt == 128 * 128 && (int64_t)tiles_m * tiles_n <= e3d_cu_count() && lds < (size_t)84 * 1024) lds = (size_t)84 * 1024;
    static std::atomic<uint64_t> lds_ok{0};
    e3d_allow_lds(lds_ok, gemm_split_kernel<NS, ACT, A_KMAJ, B_KMAJ, WM, WN, E, TN, TR>, (size_t)84 * 1024 > lds ? (size_t)84 * 1024 : lds);
    // split-K only for the K-major x K-major (weight-gradient) layout: few tiles, K = token count; one round of
    // resident workgroups (more slices only add atomics: measured in round 1)
    int splits = 1;
    const int resident = e3d_cu_count() * (WM * WN >= 8 ? 1 : 2);   // workgroups the chip holds at once (LDS-limited)
    if (A_KMAJ && B_KMAJ && ACT == E3D_ACT_NONE && tiles_m * tiles_n * 2 < resident && K >= 1024) {
        splits = resident / (tiles_m * tiles_n);
        splits = splits > 16 ? 16 : splits;
        while (splits > 1 && K / splits < 256) --splits;
    }
    int k_chunk = K;
    if (splits > 1) {
        k_chunk = ((K + splits - 1) / splits + BK - 1) / BK * BK;
        splits = (K + k_chunk - 1) / k_chunk;
        hipError_t e = hipSuccess;
        if (ldc == N) e = e3d_zero_async(out, (size_t)M * N, s);
        else {
            hipLaunchKernelGGL(e3d_zero2d_kernel, dim3(1024), dim3(256), 0, s, out, (size_t)ldc, (size_t)M, (size_t)N);
            e = hipGetLastError();
        }
        if (e != hipSuccess) {
            e3d_set_error("gemm_split: split-K memset failed: %s", hipGetErrorString(e));
            return (int)e;
        }
    }
    hipLaunchKernelGGL((gemm_split_kernel<NS, ACT, A_KMAJ, B_KMAJ, WM, WN, E, TN, TR>), dim3(tiles_m * tiles_n, splits),
                       dim3(WM * WN * 64), lds, s, A, lda, B, ldb, bias, out, ldc, M, N, K, tiles_m, tiles_n, k_chunk,
                       splits > 1 ? nullptr : epi.absmax, epi.scale);
    return e3d_launch_status("e3d_gemm_f32_split");
}

template <int NS, typename E>
int dispatch(int act, bool a_kmaj, bool b_kmaj, const float* A, int64_t lda, const float* B, int64_t ldb,
             const float* bias, float* out, int64_t ldc, int M, int N, int K, Epi epi, hipStream_t s) {
    if (!a_kmaj && !b_kmaj) {
        switch (act) {
            case E3D_ACT_NONE: return launch<NS, E3D_ACT_NONE, false, false, E>(A, lda, B, ldb, bias, out, ldc, M, N, K, epi, s);
            case E3D_ACT_GELU: return launch<NS, E3D_ACT_GELU, false, false, E>(A, lda, B, ldb, bias, out, ldc, M, N, K, epi, s);
            case E3D_ACT_SILU: return launch<NS, E3D_ACT_SILU, false, false, E>(A, lda, B, ldb, bias, out, ldc, M, N, K, epi, s);
        }
        e3d_set_error("gemm_split: unknown activation %d", act);
        return -1;
    }
    if (act != E3D_ACT_NONE) {
        e3d_set_error("gemm_split: activations are only fused for the forward (K-contiguous) layout");
        return -1;
    }
    // K-major operand layouts (the training GEMMs) exist for the bf16 terms only
    if (!a_kmaj && b_kmaj) return launch<NS, E3D_ACT_NONE, false, true, __bf16>(A, lda, B, ldb, bias, out, ldc, M, N, K, epi, s);
    if (a_kmaj && b_kmaj) return launch<NS, E3D_ACT_NONE, true, true, __bf16>(A, lda, B, ldb, bias, out, ldc, M, N, K, Epi{nullptr, epi.scale}, s);
    return launch<NS, E3D_ACT_NONE, true, false, __bf16>(A, lda, B, ldb, bias, out, ldc, M, N, K, epi, s);
}

}  // namespace

#ifdef GEMM_STAMPS
extern "C" int e3d_debug_gemm_stamps(long long* t) { return (int)hipMemcpyFromSymbol(t, HIP_SYMBOL(gemm_stamps), sizeof(long long) * 2 * 64 * 8); }
#endif

#ifdef E3D_STAMPS
extern "C" int e3d_debug_read_stamps(long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(e3d_stamps), sizeof(long long) * 2 * 32 * 8);
}
#endif

extern "C" int e3d_gemm_kernel_select(int pref) {
    if (g_tile_pref < 0) {
        const char* e = getenv("E3D_GEMM_TILE");
        g_tile_pref = e ? atoi(e) : 4;
    }
    const int prev = g_tile_pref;
    if (pref >= 0) g_tile_pref = pref;
    return prev;
}

extern "C" int e3d_gemm_general_select(int form) {
    if (g_general_form < 0) {
        const char* e = getenv("E3D_GEMM_FORM");
        g_general_form = e ? atoi(e) : 0;
    }
    const int prev = g_general_form;
    if (form >= 0) g_general_form = form;
    return prev;
}

static int gemm_split_general(const float* A, int64_t lda, int a_kmajor, const float* B, int64_t ldb, int b_kmajor,
                              const float* bias, float* out, int64_t ldc, int M, int N, int K, int act, int terms,
                              float* absmax, float out_scale, void* stream) {
    const Epi epi{absmax, out_scale};
    E3D_REQUIRE(out_scale == 1.0f || !(a_kmajor && b_kmajor), "gemm_split: out_scale is not available for the split-K layout");
    E3D_REQUIRE(A && B && out, "gemm_split: null pointer");
    E3D_REQUIRE(M > 0 && N > 0 && K > 0, "gemm_split: bad shape M=%d N=%d K=%d", M, N, K);
    E3D_REQUIRE(terms == E3D_TERMS_BF16 || terms == 3 || terms == 6 || terms == E3D_TERMS_F16X3,
                "gemm_split: terms must be 1, 3, 6 or 19 (got %d)", terms);
    E3D_REQUIRE(ldc >= N, "gemm_split: ldc=%lld < N=%d", (long long)ldc, N);
    if (!a_kmajor) E3D_REQUIRE(lda >= K && lda % 4 == 0 && K % BK == 0 && ((uintptr_t)A % 16) == 0,
                               "gemm_split: K-contiguous A needs K%%32==0, lda%%4==0, 16B alignment (K=%d lda=%lld)", K, (long long)lda);
    else E3D_REQUIRE(lda >= M, "gemm_split: K-major A needs lda >= M");
    if (!b_kmajor) E3D_REQUIRE(ldb >= K && ldb % 4 == 0 && K % BK == 0 && ((uintptr_t)B % 16) == 0,
                               "gemm_split: K-contiguous B needs K%%32==0, ldb%%4==0, 16B alignment (K=%d ldb=%lld)", K, (long long)ldb);
    else E3D_REQUIRE(ldb >= N, "gemm_split: K-major B needs ldb >= N");
    hipStream_t s = (hipStream_t)stream;
    E3D_REQUIRE(!absmax || (!(a_kmajor && b_kmajor) && act == E3D_ACT_NONE),
                "gemm_split: out_absmax exists for act = none and not for the split-K (weight-gradient) layout");
    if (terms == E3D_TERMS_BF16) return dispatch<1, __bf16>(act, a_kmajor != 0, b_kmajor != 0, A, lda, B, ldb, bias, out, ldc, M, N, K, epi, s);
    if (terms == 3) return dispatch<2, __bf16>(act, a_kmajor != 0, b_kmajor != 0, A, lda, B, ldb, bias, out, ldc, M, N, K, epi, s);
    if (terms == E3D_TERMS_F16X3) {
        // fp16 terms exist for the forward layout; a K-major operand (training GEMMs) runs the fp32-grade bf16x6 form
        if (!a_kmajor && !b_kmajor) return dispatch<2, _Float16>(act, false, false, A, lda, B, ldb, bias, out, ldc, M, N, K, epi, s);
        return dispatch<3, __bf16>(act, a_kmajor != 0, b_kmajor != 0, A, lda, B, ldb, bias, out, ldc, M, N, K, epi, s);
    }
    return dispatch<3, __bf16>(act, a_kmajor != 0, b_kmajor != 0, A, lda, B, ldb, bias, out, ldc, M, N, K, epi, s);
}

extern "C" int e3d_gemm_f32_split_general(const float* A, int64_t lda, int a_kmajor, const float* B,
                                          int64_t ldb, int b_kmajor, const float* bias, float* out,
                                          int64_t ldc, int M, int N, int K, int act, int terms,
                                          void* stream) {
    return gemm_split_general(A, lda, a_kmajor, B, ldb, b_kmajor, bias, out, ldc, M, N, K, act, terms, nullptr, 1.0f, stream);
}

extern "C" int e3d_gemm_wgrad_grouped_f32_split(const float* const* dz, const float* const* x, float* const* dw,
                                                float* const* db, uint64_t accumulate_bits, int count, int64_t ldz,
                                                int64_t ldx, int N, int K, int M, int terms, void* stream) {
    E3D_REQUIRE(dz && x && dw && count >= 1 && count <= 64, "gemm_wgrad_grouped: 1..64 problems (count=%d)", count);
    E3D_REQUIRE(M > 0 && N > 0 && K > 0 && ldz >= N && ldx >= K, "gemm_wgrad_grouped: bad shape N=%d K=%d M=%d ldz=%lld ldx=%lld",
                N, K, M, (long long)ldz, (long long)ldx);
    E3D_REQUIRE(terms == E3D_TERMS_BF16 || terms == 3 || terms == 6 || terms == E3D_TERMS_F16X3,
                "gemm_wgrad_grouped: terms must be 1, 3, 6 or 19 (got %d)", terms);
    WgradGroup g;
    for (int p = 0; p < 64; ++p) {
        const int q = p < count ? p : 0;
        E3D_REQUIRE(dz[q] && x[q] && dw[q], "gemm_wgrad_grouped: null pointer in problem %d", q);
        g.dz[p] = dz[q]; g.x[p] = x[q]; g.dw[p] = dw[q]; g.db[p] = db ? db[q] : nullptr;
    }
    g.accumulate = accumulate_bits;
    // transposing staging (float4 loads along the contiguous index, ds_read_b64_tr_b16 fragments): whole quads only
    static const bool tr_on = !getenv("E3D_WGRAD_TR") || atoi(getenv("E3D_WGRAD_TR")) != 0;
    bool tr = tr_on && N % 4 == 0 && K % 4 == 0 && ldz % 4 == 0 && ldx % 4 == 0 && e3d_tr_span_ok(M, ldz) && e3d_tr_span_ok(M, ldx);
    for (int p = 0; p < count && tr; ++p) tr = ((uintptr_t)dz[p] % 16) == 0 && ((uintptr_t)x[p] % 16) == 0;
    const int tiles_m = (N + 255) / 256, tiles_n = (K + 127) / 128;
    E3D_REQUIRE((int64_t)tiles_m * tiles_n * count < (1ll << 30), "gemm_wgrad_grouped: too many tiles");
    const dim3 grid(tiles_m * tiles_n * count), block(512);
    hipStream_t s = (hipStream_t)stream;
    auto go = [&](auto kern, size_t lds, std::atomic<uint64_t>& lds_ok) {
        e3d_allow_lds(lds_ok, kern, lds);
        hipLaunchKernelGGL(kern, grid, block, lds, s, g, ldz, ldx, N, K, M, tiles_m, tiles_n, count);
    };
    static std::atomic<uint64_t> ok1{0}, ok1t{0}, ok2{0}, ok2t{0}, ok3{0}, ok3t{0};
    if (terms == E3D_TERMS_BF16) {
        const size_t lds = (size_t)2 * 1 * (256 + 128) * ROW_B;
        if (tr) go(gemm_wgrad_grouped_kernel<1, true>, lds, ok1t);
        else go(gemm_wgrad_grouped_kernel<1, false>, lds, ok1);
    } else if (terms == 3) {
        const size_t lds = (size_t)2 * 2 * (256 + 128) * ROW_B;
        if (tr) go(gemm_wgrad_grouped_kernel<2, true>, lds, ok2t);
        else go(gemm_wgrad_grouped_kernel<2, false>, lds, ok2);
    } else {   // fp32-grade: three bf16 terms, six products (the K-major layouts have no fp16 form)
        const size_t lds = (size_t)1 * 3 * (256 + 128) * ROW_B;
        if (tr) go(gemm_wgrad_grouped_kernel<3, true>, lds, ok3t);
        else go(gemm_wgrad_grouped_kernel<3, false>, lds, ok3);
    }
    return e3d_launch_status("e3d_gemm_wgrad_grouped_f32_split");
}

extern "C" int e3d_gemm_wgrad_ragged_f32_split(const float* const* dz, const float* const* x, float* const* dw, float* const* db,
                                               const int* N, const int* K, const int64_t* ldz, const int64_t* ldx,
                                               uint64_t accumulate_bits, int count, int M, int terms, void* stream) {
    E3D_REQUIRE(dz && x && dw && N && K && ldz && ldx && count >= 1 && count <= 64, "gemm_wgrad_ragged: 1..64 problems (count=%d)", count);
    E3D_REQUIRE(M > 0, "gemm_wgrad_ragged: bad token count %d", M);
    E3D_REQUIRE(terms == E3D_TERMS_BF16 || terms == 3 || terms == 6 || terms == E3D_TERMS_F16X3,
                "gemm_wgrad_ragged: terms must be 1, 3, 6 or 19 (got %d)", terms);
    WgradRagged g;
    static const bool tr_on = !getenv("E3D_WGRAD_TR") || atoi(getenv("E3D_WGRAD_TR")) != 0;
    bool tr = tr_on;
    int64_t tiles = 0;
    for (int p = 0; p < 64; ++p) {
        const int q = p < count ? p : 0;
        E3D_REQUIRE(dz[q] && x[q] && dw[q], "gemm_wgrad_ragged: null pointer in problem %d", q);
        E3D_REQUIRE(N[q] > 0 && K[q] > 0 && ldz[q] >= N[q] && ldx[q] >= K[q] && ldz[q] < (1ll << 31) && ldx[q] < (1ll << 31),
                    "gemm_wgrad_ragged: bad shape in problem %d: N=%d K=%d ldz=%lld ldx=%lld", q, N[q], K[q], (long long)ldz[q],
                    (long long)ldx[q]);
        g.dz[p] = dz[q]; g.x[p] = x[q]; g.dw[p] = dw[q]; g.db[p] = db ? db[q] : nullptr;
        g.N[p] = N[q]; g.K[p] = K[q]; g.ldz[p] = (int)ldz[q]; g.ldx[p] = (int)ldx[q];
        g.tile_start[p] = (int)tiles;
        if (p < count) {
            tiles += (int64_t)((N[q] + 255) / 256) * ((K[q] + 127) / 128);
            tr = tr && N[q] % 4 == 0 && K[q] % 4 == 0 && ldz[q] % 4 == 0 && ldx[q] % 4 == 0 && ((uintptr_t)dz[q] % 16) == 0 &&
                 ((uintptr_t)x[q] % 16) == 0 && e3d_tr_span_ok(M, ldz[q]) && e3d_tr_span_ok(M, ldx[q]);
        }
    }
    E3D_REQUIRE(tiles < (1ll << 30), "gemm_wgrad_ragged: too many tiles");
    for (int p = count; p <= 64; ++p) g.tile_start[p] = (int)tiles;
    g.accumulate = accumulate_bits;
    const dim3 grid((unsigned)tiles), block(512);
    hipStream_t s = (hipStream_t)stream;
    auto go = [&](auto kern, size_t lds, std::atomic<uint64_t>& lds_ok) {
        e3d_allow_lds(lds_ok, kern, lds);
        hipLaunchKernelGGL(kern, grid, block, lds, s, g, M, count);
    };
    static std::atomic<uint64_t> ok1{0}, ok1t{0}, ok2{0}, ok2t{0}, ok3{0}, ok3t{0};
    if (terms == E3D_TERMS_BF16) {
        const size_t lds = (size_t)2 * 1 * (256 + 128) * ROW_B;
        if (tr) go(gemm_wgrad_ragged_kernel<1, true>, lds, ok1t);
        else go(gemm_wgrad_ragged_kernel<1, false>, lds, ok1);
    } else if (terms == 3) {
        const size_t lds = (size_t)2 * 2 * (256 + 128) * ROW_B;
        if (tr) go(gemm_wgrad_ragged_kernel<2, true>, lds, ok2t);
        else go(gemm_wgrad_ragged_kernel<2, false>, lds, ok2);
    } else {
        const size_t lds = (size_t)1 * 3 * (256 + 128) * ROW_B;
        if (tr) go(gemm_wgrad_ragged_kernel<3, true>, lds, ok3t);
        else go(gemm_wgrad_ragged_kernel<3, false>, lds, ok3);
    }
    return e3d_launch_status("e3d_gemm_wgrad_ragged_f32_split");
}

extern "C" int e3d_gemm_bias_act_f32_split(const float* A, int64_t lda, const float* W,
                                           const float* bias, float* out, int64_t ldc, int M,
                                           int N, int K, int act, int terms, void* stream) {
    E3D_REQUIRE(N % BN == 0, "gemm_split: need N%%128==0 (N=%d)", N);
    return gemm_split_general(A, lda, 0, W, (int64_t)K, 0, bias, out, ldc, M, N, K, act, terms, nullptr, 1.0f, stream);
}

extern "C" int e3d_gemm_bias_act_f32_split_ex(const float* A, int64_t lda, const float* W, const float* bias, float* out,
                                              int64_t ldc, int M, int N, int K, int act, int terms, float* out_absmax,
                                              float out_scale, void* stream) {
    E3D_REQUIRE(N % BN == 0, "gemm_split: need N%%128==0 (N=%d)", N);
    return gemm_split_general(A, lda, 0, W, (int64_t)K, 0, bias, out, ldc, M, N, K, act, terms, out_absmax, out_scale, stream);
}
