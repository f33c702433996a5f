// fp32 GEMM through the bf16 matrix cores with SPLIT operands and fp32 accumulation:
//     out[M,N] = act(A[M,K] @ W[N,K]^T + bias)
// Each fp32 operand x is split on the fly into bf16 terms x = x0 + x1 (+ x2) (xi = rne_bf16 of the
// running residual) and the product is rebuilt from the significant cross terms:
//     TERMS = 3:  a0*b0 + a0*b1 + a1*b0                         (error ~2^-17 per product)
//     TERMS = 6:  + a0*b2 + a2*b0 + a1*b1                       (error ~2^-24: fp32 grade)
// v_mfma_f32_32x32x16_bf16 runs at 16x the rate of the fp32 MFMA, so the 3-/6-term products cost
// 3/16 / 6/16 of the exact fp32 kernel (gemm_f32.hip).  Accumulation stays fp32 in the MFMA.
// Accuracy through the full 12+12-layer structure model (CPU emulation):
// 3 terms 2.6e-5, 6 terms 4.3e-7 relative (fp32 torch itself: 1.8e-6); tolerance 1e-4.
//
// Workgroup tile 256x128x32, 8 waves as 4(M) x 2(N), each wave 2x2 MFMA tiles of 32x32 (64
// accumulator registers).  Staging: global_load_dwordx4 (fp32) -> split in registers -> ds_write_b64
// into per-term bf16 images [rows][32 + 8 pad] (80-byte rows: the ds_read_b128 fragment reads
// of 16 lanes hit 16 distinct 4-bank slots), double buffered for TERMS = 3.
#include "e3d_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 256, BN = 128, BK = 32;
constexpr int ROW_B = 80;  // bytes per LDS row: 32 bf16 + 16 B pad

template <int NS>
__device__ __forceinline__ void split4(const f32x4 v, bf16x4 (&parts)[NS]) {
    f32x4 r = v;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const __bf16 p = (__bf16)r[j];
            parts[s][j] = p;
            r[j] -= (float)p;
        }
    }
}

template <int NS, int ACT>
__global__ __launch_bounds__(512) void gemm_nt_split(const float* __restrict__ A, int64_t lda,
                                                     const float* __restrict__ W,
                                                     const float* __restrict__ bias,
                                                     float* __restrict__ out, int64_t ldc, int M,
                                                     int N, int K, int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int NBUF = NS == 2 ? 2 : 1;
    constexpr int A_BYTES = BM * ROW_B, B_BYTES = BN * ROW_B;
    constexpr int BUF_BYTES = NS * (A_BYTES + B_BYTES);
    // layout per buffer: A part 0..NS-1, then B part 0..NS-1

    const int lid = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int tm = lid / tiles_n, tn = lid % tiles_n;
    const int row0 = tm * BM, col0 = tn * BN;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;
    const int l31 = lane & 31, half = lane >> 5;

    // staging map: float4 index f -> row f>>3, chunk f&7 ; A: 4 per thread, B: 2 per thread
    const float* a_src[4];
    const float* b_src[2];
    int a_off[4], b_off[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = tid + 512 * i, r = f >> 3, c = f & 7;
        int ar = row0 + r;
        ar = ar < M ? ar : M - 1;
        a_src[i] = A + (int64_t)ar * lda + c * 4;
        a_off[i] = r * ROW_B + c * 8;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int f = tid + 512 * i, r = f >> 3, c = f & 7;
        b_src[i] = W + (int64_t)(col0 + r) * K + c * 4;
        b_off[i] = r * ROW_B + c * 8;
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    f32x4 ra[4], rb[2];
    auto g_load = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) ra[i] = *reinterpret_cast<const f32x4*>(a_src[i] + kt * BK);
#pragma unroll
        for (int i = 0; i < 2; ++i) rb[i] = *reinterpret_cast<const f32x4*>(b_src[i] + kt * BK);
    };
    auto lds_store = [&](int buf) {
        unsigned char* base = smem_raw + buf * BUF_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            bf16x4 p[NS];
            split4<NS>(ra[i], p);
#pragma unroll
            for (int s = 0; s < NS; ++s)
                *reinterpret_cast<bf16x4*>(base + s * A_BYTES + a_off[i]) = p[s];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            bf16x4 p[NS];
            split4<NS>(rb[i], p);
#pragma unroll
            for (int s = 0; s < NS; ++s)
                *reinterpret_cast<bf16x4*>(base + NS * A_BYTES + s * B_BYTES + b_off[i]) = p[s];
        }
    };

    g_load(0);
    lds_store(0);
    __syncthreads();

    const int nk = K / BK;
    const int a_frag = (wr * 64 + l31) * ROW_B + half * 16;
    const int b_frag = (wc * 64 + l31) * ROW_B + half * 16;
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if (more) g_load(kt + 1);
        const unsigned char* ab = smem_raw + cur * BUF_BYTES + a_frag;
        const unsigned char* bb = smem_raw + cur * BUF_BYTES + NS * A_BYTES + b_frag;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 fa[NS][2], fb[NS][2];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
#pragma unroll
                for (int m = 0; m < 2; ++m)
                    fa[s][m] = *reinterpret_cast<const bf16x8*>(ab + s * A_BYTES + m * 32 * ROW_B + ks * 32);
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    fb[s][n] = *reinterpret_cast<const bf16x8*>(bb + s * B_BYTES + n * 32 * ROW_B + ks * 32);
            }
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    // smallest terms first
                    if (NS == 3) {
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][m], fb[1][n], acc[m][n], 0, 0, 0);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][m], fb[NS - 1][n], acc[m][n], 0, 0, 0);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[NS - 1][m], fb[0][n], acc[m][n], 0, 0, 0);
                    }
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][m], fb[1][n], acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][m], fb[0][n], acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][m], fb[0][n], acc[m][n], 0, 0, 0);
                }
        }
        if (NBUF == 2) {
            if (more) lds_store(cur ^ 1);
            __syncthreads();
            cur ^= 1;
        } else {
            __syncthreads();  // everyone done reading the single buffer
            if (more) lds_store(0);
            __syncthreads();
        }
    }

#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int col = col0 + wc * 64 + n * 32 + l31;
        const float bv = bias ? bias[col] : 0.f;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + wr * 64 + m * 32 + mfma32_row(r, half);
                float v = acc[m][n][r] + bv;
                if (ACT == E3D_ACT_GELU) v = gelu_erf(v);
                if (ACT == E3D_ACT_SILU) v = silu(v);
                if (row < M) out[(int64_t)row * ldc + col] = v;
            }
        }
    }
}

template <int NS, int ACT>
int launch(const float* A, int64_t lda, const float* W, const float* bias, float* out, int64_t ldc, int M, int N,
           int K, hipStream_t s) {
    const int tiles_m = (M + BM - 1) / BM, tiles_n = N / BN;
    constexpr int NBUF = NS == 2 ? 2 : 1;
    const size_t lds = (size_t)NBUF * NS * (BM + BN) * ROW_B;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_split<NS, ACT>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_nt_split<NS, ACT>), dim3(tiles_m * tiles_n), dim3(512), lds, s, A, lda, W, bias, out,
                       ldc, M, N, K, tiles_m, tiles_n);
    return e3d_launch_status("e3d_gemm_bias_act_f32_split");
}

template <int NS>
int dispatch_act(int act, const float* A, int64_t lda, const float* W, const float* bias, float* out, int64_t ldc,
                 int M, int N, int K, hipStream_t s) {
    switch (act) {
        case E3D_ACT_NONE: return launch<NS, E3D_ACT_NONE>(A, lda, W, bias, out, ldc, M, N, K, s);
        case E3D_ACT_GELU: return launch<NS, E3D_ACT_GELU>(A, lda, W, bias, out, ldc, M, N, K, s);
        case E3D_ACT_SILU: return launch<NS, E3D_ACT_SILU>(A, lda, W, bias, out, ldc, M, N, K, s);
    }
    e3d_set_error("gemm_split: unknown activation %d", act);
    return -1;
}

}  // namespace

extern "C" int e3d_gemm_bias_act_f32_split(const float* A, int64_t lda, const float* W,
                                           const float* bias, float* out, int64_t ldc, int M,
                                           int N, int K, int act, int terms, void* stream) {
    E3D_REQUIRE(A && W && out, "gemm_split: null pointer");
    E3D_REQUIRE(M > 0 && N > 0 && K > 0, "gemm_split: bad shape M=%d N=%d K=%d", M, N, K);
    E3D_REQUIRE(N % BN == 0 && K % BK == 0, "gemm_split: need N%%128==0 and K%%32==0 (N=%d K=%d)", N, K);
    E3D_REQUIRE(lda >= K && ldc >= N && lda % 4 == 0, "gemm_split: bad strides lda=%lld ldc=%lld", (long long)lda,
                (long long)ldc);
    E3D_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0, "gemm_split: operands must be 16B aligned");
    E3D_REQUIRE(terms == 3 || terms == 6, "gemm_split: terms must be 3 or 6 (got %d)", terms);
    hipStream_t s = (hipStream_t)stream;
    if (terms == 3) return dispatch_act<2>(act, A, lda, W, bias, out, ldc, M, N, K, s);
    return dispatch_act<3>(act, A, lda, W, bias, out, ldc, M, N, K, s);
}
