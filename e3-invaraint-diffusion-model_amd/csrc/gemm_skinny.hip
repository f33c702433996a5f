// Small-M ("skinny") forward GEMM for single-pocket and few-pocket sampling:
//     out[M,N] = act(A[M,K] . W[N,K]^T + bias),   M <= 128
// At M = 64 the tiled kernels of gemm_split.hip put 6 workgroups on 256 CUs and spend ~29 us walking 24 k-steps one
// after the other; a reverse step of one pocket is ~80 such launches (GPU 100 % busy with
// dependent small kernels).  What bounds a product this small is HBM LATENCY on the weight stream (2.4 MB = 18 000
// lines, each CU keeps a limited number of misses in flight), so the weight must be streamed by the WHOLE chip: the
// work is cut along K as well, one single-wave workgroup per (32-row, 32-column, K-slice) unit -- 400-900 independent
// waves per launch -- each streaming its operand fragments straight from L2 / HBM into registers (fragment-shaped
// loads: no LDS, no barrier), splitting them into two 16-bit terms and issuing the 3 cross products, and writing its
// partial tile to a slab.  A SECOND small launch sums the slabs in slice order (deterministic) and finishes the rows:
// bias + activation, or bias + residual + LayerNorm when the consumer is BertSelfOutput / BertOutput (one launch less).
// Forms measured and dropped (round 2, MI355X, one 64-residue pocket, graph replay, ms per reverse step):
//   * K cut across the 8 waves of ONE workgroup per tile, summed in LDS (one launch, no workspace): 14.3 us per product
//     against 6.7 + 5.3 here -- 48-144 CUs stream the weight instead of 256 (1.80 vs 1.57 ms per step);
//   * slabs summed inside the launch by the last arriver (ticket per tile; agent-scope release / acquire): 2.06 ms --
//     every wave pays a buffer_wbl2; with write-through slab stores it would save ~0.5 us per product over the second
//     launch, not worth a hand-off that is only measured-valid.
// Workspace (caller-provided, one per stream, no initialisation needed): e3d_gemm_skinny_workspace_bytes().
#include <type_traits>

#include "e3d_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <typename E> struct SV;
template <> struct SV<__bf16> { typedef bf16x8 x8; };
template <> struct SV<_Float16> { typedef f16x8 x8; };
__device__ __forceinline__ f32x16 mma16(const bf16x8 a, const bf16x8 b, const f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mma16(const f16x8 a, const f16x8 b, const f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ void split8(const f32x4 a, const f32x4 b, bf16x8& hi, bf16x8& lo) {
    const float x[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 p = (__bf16)x[j];
        hi[j] = p;
        lo[j] = (__bf16)(x[j] - (float)p);
    }
}
__device__ __forceinline__ void split8(const f32x4 a, const f32x4 b, f16x8& hi, f16x8& lo) {
    const float x[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        e3d_f16x2 h, l;
        e3d_split2_f16(x[j], x[j + 1], h, l);
        hi[j] = h[0]; hi[j + 1] = h[1];
        lo[j] = l[0]; lo[j + 1] = l[1];
    }
}

constexpr int KB_GROUP = 4;   // 16-wide k blocks whose fragment loads are in flight together (2 groups: 128 VGPRs)

// slab element of (row, col) of a 32x32 tile held in the MFMA accumulator layout [r][lane]:
// row = 8 (r / 4) + 4 half + r % 4, col = lane & 31, half = lane >> 5  ->  4 consecutive columns are 16 contiguous bytes
__device__ __forceinline__ int slab_offset(int row, int col) {
    return (((row >> 3) * 4 + (row & 3)) << 6) + (((row >> 2) & 1) << 5) + col;
}

constexpr int MAX_SLICES = 8;

// sum over the K slices, in slice order, of the 4 slab words at p: all MAX_SLICES loads are issued together (slices past
// n_slices re-load the last one and are not added -- no branch between the loads)
__device__ __forceinline__ f32x4 slab_sum(const float* __restrict__ p, int n_slices) {
    f32x4 part[MAX_SLICES];
#pragma unroll
    for (int s = 0; s < MAX_SLICES; ++s) part[s] = *reinterpret_cast<const f32x4*>(p + min(s, n_slices - 1) * 1024);
    f32x4 v = part[0];
#pragma unroll
    for (int s = 1; s < MAX_SLICES; ++s)
        if (s < n_slices) v += part[s];
    return v;
}

// first launch: one wave per (tile, K slice)
template <int ACT, typename E>
__global__ __launch_bounds__(64) void gemm_skinny_kernel(const float* __restrict__ A, int64_t lda,
                                                         const float* __restrict__ W, const float* __restrict__ bias,
                                                         float* __restrict__ out, int64_t ldc, int M, int N, int K,
                                                         int tiles_n, int k_slice, int to_slabs, float* __restrict__ slabs,
                                                         float* __restrict__ absmax, float out_scale) {
    typedef typename SV<E>::x8 X8;
    const int lane = threadIdx.x, l31 = lane & 31, half = lane >> 5;
    const int tile = blockIdx.x, tm = tile / tiles_n, tn = tile % tiles_n;
    const int slice = blockIdx.y, n_slices = gridDim.y;
    const int k0 = slice * k_slice, k1 = min(K, k0 + k_slice);
    const int n_kb = (k1 - k0) >> 4;
    const float* arow = A + (int64_t)min(tm * 32 + l31, M - 1) * lda + k0 + 8 * half;   // rows past M: clamped, discarded
    const float* wrow = W + (int64_t)(tn * 32 + l31) * K + k0 + 8 * half;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    f32x4 ra[2][KB_GROUP][2], rw[2][KB_GROUP][2];
    auto load_group = [&](auto buf_tag, int kb0) {
        constexpr int BUF = decltype(buf_tag)::value;
#pragma unroll
        for (int i = 0; i < KB_GROUP; ++i) {
            const int kb = min(kb0 + i, n_kb - 1);   // past the slice: harmless re-load of its last block (not used)
            ra[BUF][i][0] = *reinterpret_cast<const f32x4*>(arow + 16 * kb);
            ra[BUF][i][1] = *reinterpret_cast<const f32x4*>(arow + 16 * kb + 4);
            rw[BUF][i][0] = *reinterpret_cast<const f32x4*>(wrow + 16 * kb);
            rw[BUF][i][1] = *reinterpret_cast<const f32x4*>(wrow + 16 * kb + 4);
        }
    };
    auto mma_group = [&](auto buf_tag, int kb0) {
        constexpr int BUF = decltype(buf_tag)::value;
#pragma unroll
        for (int i = 0; i < KB_GROUP; ++i) {
            if (kb0 + i < n_kb) {   // wave-uniform
                X8 ah, al, wh, wl;
                split8(ra[BUF][i][0], ra[BUF][i][1], ah, al);
                split8(rw[BUF][i][0], rw[BUF][i][1], wh, wl);
                acc = mma16(ah, wl, acc);    // smallest terms first
                acc = mma16(al, wh, acc);
                acc = mma16(ah, wh, acc);
            }
        }
    };
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;
    load_group(B0{}, 0);
    for (int kb0 = 0; kb0 < n_kb; kb0 += 2 * KB_GROUP) {
        load_group(B1{}, kb0 + KB_GROUP);
        mma_group(B0{}, kb0);
        load_group(B0{}, kb0 + 2 * KB_GROUP);
        mma_group(B1{}, kb0 + KB_GROUP);
    }

    if (to_slabs) {
        // this slice's partial tile in the accumulator layout: 256 contiguous bytes per store instruction
        float* mine = slabs + ((int64_t)tile * n_slices + slice) * 1024;
#pragma unroll
        for (int r = 0; r < 16; ++r) mine[r * 64 + lane] = acc[r];
        return;
    }
    const int col = tn * 32 + l31;     // a single slice covers K: finish here
    const float bv = bias ? bias[col] : 0.f;
    unsigned amax = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = tm * 32 + mfma32_row(r, half);
        float v = fmaf(acc[r], out_scale, bv);
        if (ACT == E3D_ACT_GELU) v = gelu_erf(v);
        if (ACT == E3D_ACT_SILU) v = silu(v);
        if (row < M) {
            out[(int64_t)row * ldc + col] = v;
            if (ACT == E3D_ACT_NONE && absmax) e3d_absmax_accum(amax, v);
        }
    }
    if (ACT == E3D_ACT_NONE && absmax) e3d_absmax_commit(amax, absmax, lane);
}

// second launch, plain form: one wave per (output row, 256-column chunk), a lane finishes 4 consecutive columns
// (16-byte slab loads, all slices in flight together; 16-byte coalesced stores)
template <int ACT>
__global__ __launch_bounds__(256) void skinny_finish_kernel(const float* __restrict__ slabs, int n_slices, int tiles_n,
                                                            const float* __restrict__ bias, float* __restrict__ out,
                                                            int64_t ldc, int M, int N, float* __restrict__ absmax, float out_scale) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int c0 = blockIdx.y * 256 + 4 * lane;
    if (row >= M) return;        // (wave-uniform)
    unsigned amax = 0;           // lanes past the last column stay in the wave for the |out| maximum
    if (c0 < N) {
        const float* p = slabs + ((int64_t)(row >> 5) * tiles_n + (c0 >> 5)) * n_slices * 1024 + slab_offset(row & 31, c0 & 31);
        f32x4 v = slab_sum(p, n_slices);
        if (out_scale != 1.0f) v *= out_scale;      // (a power of two: exact)
        if (bias) v += *reinterpret_cast<const f32x4*>(bias + c0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (ACT == E3D_ACT_GELU) v[j] = gelu_erf(v[j]);
            if (ACT == E3D_ACT_SILU) v[j] = silu(v[j]);
            if (ACT == E3D_ACT_NONE) e3d_absmax_accum(amax, v[j]);
        }
        *reinterpret_cast<f32x4*>(out + (int64_t)row * ldc + c0) = v;
    }
    if (ACT == E3D_ACT_NONE && absmax) e3d_absmax_commit(amax, absmax, lane);
}

// second launch, BertSelfOutput / BertOutput form: out = LayerNorm(sum of slices + bias + residual) * gamma + beta.
// Same row layout and the same arithmetic, in the same order, as skinny_finish_kernel<0> followed by
// residual_layernorm_kernel<V> of rowops.hip (two-pass centred variance): bit-identical to the unfused pair.
template <int V>
__global__ __launch_bounds__(256) void skinny_finish_layernorm_kernel(
    const float* __restrict__ slabs, int n_slices, const float* __restrict__ bias, const float* __restrict__ res,
    const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float* __restrict__ out, int M,
    float out_scale) {
    constexpr int H = 256 * V, tiles_n = H / 32;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* base = slabs + (int64_t)(row >> 5) * tiles_n * n_slices * 1024 + slab_offset(row & 31, 0);
    f32x4 r[V], t[V], g[V], b[V];
#pragma unroll
    for (int i = 0; i < V; ++i) {
        const int c0 = 4 * (64 * i + lane);
        const float* p = base + (int64_t)(c0 >> 5) * n_slices * 1024 + (c0 & 31);
        r[i] = slab_sum(p, n_slices);
    }
#pragma unroll
    for (int i = 0; i < V; ++i) {
        const int c0 = 4 * (64 * i + lane);
        if (out_scale != 1.0f) r[i] *= out_scale;   // (a power of two: exact)
        if (bias) r[i] += *reinterpret_cast<const f32x4*>(bias + c0);
        if (res) r[i] += *reinterpret_cast<const f32x4*>(res + (int64_t)row * H + c0);
        g[i] = *reinterpret_cast<const f32x4*>(gamma + c0);
        b[i] = *reinterpret_cast<const f32x4*>(beta + c0);
    }
    constexpr float inv_h = 1.0f / H;
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i) sum += (r[i][0] + r[i][1]) + (r[i][2] + r[i][3]);
    const float mean = wave_sum(sum) * inv_h;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            r[i][j] -= mean;
            ss += r[i][j] * r[i][j];
        }
    const float rstd = 1.0f / sqrtf(wave_sum(ss) * inv_h + eps);
#pragma unroll
    for (int i = 0; i < V; ++i) {
        t[i] = r[i] * rstd;
        *reinterpret_cast<f32x4*>(out + (int64_t)row * H + 4 * (64 * i + lane)) = t[i] * g[i] + b[i];
    }
}

}  // namespace
extern "C" int64_t e3d_gemm_skinny_workspace_bytes(int M, int N, int K);
namespace {

int g_waves_per_cu_x2 = 3, g_min_k_slice = 96;   // plan knobs (e3d_gemm_skinny_plan_select)

struct Plan { int tiles_m, tiles_n, tiles, slices, k_slice; };
Plan make_plan(int M, int N, int K) {
    Plan p;
    p.tiles_m = (M + 31) / 32;
    p.tiles_n = N / 32;
    p.tiles = p.tiles_m * p.tiles_n;
    // ~1.5 waves per CU on the whole chip (sweep of tools/lab/skinny_ab.py: 1.5 <= 2 <= 3 < 4, the gap growing with N:
    // fewer slices = less slab traffic), at most MAX_SLICES slices, K slices of at least 96
    int s = (g_waves_per_cu_x2 * e3d_cu_count() / 2 + p.tiles - 1) / p.tiles;
    s = s < 1 ? 1 : (s > MAX_SLICES ? MAX_SLICES : s);
    int ks = ((K + s - 1) / s + 15) / 16 * 16;
    ks = ks < g_min_k_slice ? g_min_k_slice : ks;
    ks = ks > K ? (K + 15) / 16 * 16 : ks;
    p.k_slice = ks;
    p.slices = (K + ks - 1) / ks;
    return p;
}

template <typename E>
int launch_partials(const float* A, int64_t lda, const float* W, const float* bias, float* out, int64_t ldc, int M, int N,
                    int K, int act, const Plan& p, int to_slabs, float* slabs, float* absmax, float out_scale, hipStream_t s) {
    const dim3 grid(p.tiles, p.slices), block(64);
#define E3D_SKINNY_CASE(a)                                                                                              \
    case a:                                                                                                             \
        hipLaunchKernelGGL((gemm_skinny_kernel<a, E>), grid, block, 0, s, A, lda, W, bias, out, ldc, M, N, K, p.tiles_n, \
                           p.k_slice, to_slabs, slabs, absmax, out_scale);                                                                \
        break
    switch (act) {
        E3D_SKINNY_CASE(E3D_ACT_NONE);
        E3D_SKINNY_CASE(E3D_ACT_GELU);
        E3D_SKINNY_CASE(E3D_ACT_SILU);
        default:
            e3d_set_error("gemm_skinny: unknown activation %d", act);
            return -1;
    }
#undef E3D_SKINNY_CASE
    return 0;
}

int check_common(const float* A, int64_t lda, const float* W, const float* out, int M, int N, int K, int terms,
                 const void* workspace, int64_t workspace_bytes) {
    E3D_REQUIRE(A && W && out && workspace, "gemm_skinny: null pointer");
    E3D_REQUIRE(M > 0 && M <= 4096 && N > 0 && K > 0, "gemm_skinny: needs 0 < M <= 4096 (M=%d N=%d K=%d)", M, N, K);
    E3D_REQUIRE(N % 32 == 0 && K % 16 == 0 && lda % 4 == 0 && lda >= K,
                "gemm_skinny: needs N%%32==0, K%%16==0, lda%%4==0 (N=%d K=%d lda=%lld)", N, K, (long long)lda);
    E3D_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0 && ((uintptr_t)workspace % 16) == 0,
                "gemm_skinny: A, W and the workspace must be 16-byte aligned");
    E3D_REQUIRE(terms == E3D_TERMS_BF16X3 || terms == E3D_TERMS_F16X3, "gemm_skinny: terms must be 3 or 19 (got %d)", terms);
    E3D_REQUIRE(workspace_bytes >= e3d_gemm_skinny_workspace_bytes(M, N, K), "gemm_skinny: workspace too small (%lld < %lld)",
                (long long)workspace_bytes, (long long)e3d_gemm_skinny_workspace_bytes(M, N, K));
    return 0;
}

}  // namespace

extern "C" void e3d_gemm_skinny_plan_select(int waves_per_cu_x2, int min_k_slice) {
    g_waves_per_cu_x2 = waves_per_cu_x2 > 0 ? waves_per_cu_x2 : 3;
    g_min_k_slice = min_k_slice >= 16 ? (min_k_slice + 15) / 16 * 16 : 96;
}

extern "C" int64_t e3d_gemm_skinny_workspace_bytes(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const Plan p = make_plan(M, N, K);
    return (int64_t)p.tiles * p.slices * 1024 * sizeof(float);
}

extern "C" int e3d_gemm_skinny_f32_split_ex(const float* A, int64_t lda, const float* W, const float* bias, float* out,
                                            int64_t ldc, int M, int N, int K, int act, int terms, void* workspace,
                                            int64_t workspace_bytes, float* absmax, float out_scale, void* stream) {
    if (check_common(A, lda, W, out, M, N, K, terms, workspace, workspace_bytes)) return -1;
    E3D_REQUIRE(!absmax || act == E3D_ACT_NONE, "gemm_skinny: out_absmax exists for act = none");
    E3D_REQUIRE(ldc >= N && ldc % 4 == 0 && ((uintptr_t)out % 16) == 0 && (!bias || ((uintptr_t)bias % 16) == 0),
                "gemm_skinny: out / bias must be 16-byte aligned, ldc %% 4 == 0 (ldc=%lld)", (long long)ldc);
    hipStream_t s = (hipStream_t)stream;
    const Plan p = make_plan(M, N, K);
    float* slabs = reinterpret_cast<float*>(workspace);
    const int to_slabs = p.slices > 1;
    const int rc = terms == E3D_TERMS_F16X3
                       ? launch_partials<_Float16>(A, lda, W, bias, out, ldc, M, N, K, act, p, to_slabs, slabs, absmax, out_scale, s)
                       : launch_partials<__bf16>(A, lda, W, bias, out, ldc, M, N, K, act, p, to_slabs, slabs, absmax, out_scale, s);
    if (rc) return rc;
    if (to_slabs) {
        const dim3 grid((M + 3) / 4, (N + 255) / 256), block(256);
        switch (act) {
            case E3D_ACT_NONE: hipLaunchKernelGGL(skinny_finish_kernel<E3D_ACT_NONE>, grid, block, 0, s, slabs, p.slices, p.tiles_n, bias, out, ldc, M, N, absmax, out_scale); break;
            case E3D_ACT_GELU: hipLaunchKernelGGL(skinny_finish_kernel<E3D_ACT_GELU>, grid, block, 0, s, slabs, p.slices, p.tiles_n, bias, out, ldc, M, N, absmax, out_scale); break;
            default: hipLaunchKernelGGL(skinny_finish_kernel<E3D_ACT_SILU>, grid, block, 0, s, slabs, p.slices, p.tiles_n, bias, out, ldc, M, N, absmax, out_scale); break;
        }
    }
    return e3d_launch_status("e3d_gemm_skinny_f32_split");
}

extern "C" int e3d_gemm_skinny_f32_split(const float* A, int64_t lda, const float* W, const float* bias, float* out,
                                         int64_t ldc, int M, int N, int K, int act, int terms, void* workspace,
                                         int64_t workspace_bytes, void* stream) {
    return e3d_gemm_skinny_f32_split_ex(A, lda, W, bias, out, ldc, M, N, K, act, terms, workspace, workspace_bytes, nullptr,
                                        1.0f, stream);
}

extern "C" int e3d_gemm_skinny_residual_layernorm_f32_split_ex(const float* A, int64_t lda, const float* W, const float* bias,
                                                               const float* residual, const float* gamma, const float* beta,
                                                               float eps, float* out, int M, int H, int K, int terms,
                                                               void* workspace, int64_t workspace_bytes, float out_scale,
                                                               void* stream) {
    if (check_common(A, lda, W, out, M, H, K, terms, workspace, workspace_bytes)) return -1;
    E3D_REQUIRE(gamma && beta, "gemm_skinny_residual_layernorm: null gamma / beta");
    E3D_REQUIRE(H == 256 || H == 512 || H == 768 || H == 1024, "gemm_skinny_residual_layernorm: H must be 256/512/768/1024 (H=%d)", H);
    hipStream_t s = (hipStream_t)stream;
    const Plan p = make_plan(M, H, K);
    float* slabs = reinterpret_cast<float*>(workspace);
    const int rc = terms == E3D_TERMS_F16X3
                       ? launch_partials<_Float16>(A, lda, W, nullptr, out, H, M, H, K, E3D_ACT_NONE, p, 1, slabs, nullptr, 1.0f, s)
                       : launch_partials<__bf16>(A, lda, W, nullptr, out, H, M, H, K, E3D_ACT_NONE, p, 1, slabs, nullptr, 1.0f, s);
    if (rc) return rc;
    const dim3 grid((M + 3) / 4), block(256);
    switch (H) {
        case 256: hipLaunchKernelGGL(skinny_finish_layernorm_kernel<1>, grid, block, 0, s, slabs, p.slices, bias, residual, gamma, beta, eps, out, M, out_scale); break;
        case 512: hipLaunchKernelGGL(skinny_finish_layernorm_kernel<2>, grid, block, 0, s, slabs, p.slices, bias, residual, gamma, beta, eps, out, M, out_scale); break;
        case 768: hipLaunchKernelGGL(skinny_finish_layernorm_kernel<3>, grid, block, 0, s, slabs, p.slices, bias, residual, gamma, beta, eps, out, M, out_scale); break;
        default: hipLaunchKernelGGL(skinny_finish_layernorm_kernel<4>, grid, block, 0, s, slabs, p.slices, bias, residual, gamma, beta, eps, out, M, out_scale); break;
    }
    return e3d_launch_status("e3d_gemm_skinny_residual_layernorm_f32_split");
}

extern "C" int e3d_gemm_skinny_residual_layernorm_f32_split(const float* A, int64_t lda, const float* W, const float* bias,
                                                            const float* residual, const float* gamma, const float* beta,
                                                            float eps, float* out, int M, int H, int K, int terms,
                                                            void* workspace, int64_t workspace_bytes, void* stream) {
    return e3d_gemm_skinny_residual_layernorm_f32_split_ex(A, lda, W, bias, residual, gamma, beta, eps, out, M, H, K, terms,
                                                           workspace, workspace_bytes, 1.0f, stream);
}
