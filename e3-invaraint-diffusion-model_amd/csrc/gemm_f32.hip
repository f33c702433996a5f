// fp32 GEMM on the gfx950 matrix cores: out[M,N] = act(A[M,K] @ W[N,K]^T + bias).
//
// v_mfma_f32_32x32x2_f32 (exact fp32 fmaf chain, 64 FLOP/clk/SIMD).  Workgroup tile
// 128x128x32, 4 waves in a 2x2 grid, each wave 2x2 MFMA tiles of 32x32 (64 accumulator
// registers).  Both operands are K-contiguous ("NT"), so A and W tiles are staged the same way:
// global_load_dwordx4 -> registers -> ds_write_b128 into [128][36]-float LDS images (the 16-byte
// row pad makes the ds_read_b128 fragment reads conflict-free), double buffered with the next
// tile's global loads in flight under the current tile's 64 MFMAs.
//
// k-order trick: a lane (row r, half h) takes one float4 = k {8s+4h .. 8s+4h+3} and feeds element
// kk to MFMA kk; A and B use the same map, so each MFMA still pairs equal k.
//
// Blocked accumulation (round 4): every 32-deep k-tile is summed into a fresh partial accumulator (a chain of 16
// two-product MFMAs) which is then added to the running sum -- K / 32 + 16 roundings per output instead of one serial
// chain of K / 2 (384 at K = 768).  The serial chain was where this "exact fp32" kernel lost a factor of two against the
// CPU's blocked sgemm (teacher-forced per stage against the fp64 oracle: 1.5-2.0e-6 here, 0.7-0.9e-6 on the CPU, and
// 1.0e-6 for f16x3, whose MFMAs sum 16 products internally: profiles/r04_margin_bisect_x2_before.log).
#include "e3d_common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 32, LDS_LD = 36;
constexpr int TILE_F = BM * LDS_LD;  // floats per operand per buffer

template <int ACT>
__global__ __launch_bounds__(256) void gemm_nt_f32(const float* __restrict__ A, int64_t lda,
                                                   const float* __restrict__ W,
                                                   const float* __restrict__ bias,
                                                   float* __restrict__ out, int64_t ldc, int M,
                                                   int N, int K, int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;               // [2][128][36]
    float* Bs = smem + 2 * TILE_F;  // [2][128][36]

    // XCD-aware tile order: consecutive logical ids walk N fastest inside one M panel, so the
    // workgroups of one XCD share A rows (and all of W is L2/MALL resident anyway).
    const int lid = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int tm = lid / tiles_n, tn = lid % tiles_n;
    const int row0 = tm * BM, col0 = tn * BN;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;
    const int l31 = lane & 31, half = lane >> 5;

    // staging map: 4 float4 per operand per thread; f = tid + 256*i -> row f>>3, chunk f&7
    const float* a_src[4];
    const float* b_src[4];
    int lds_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = tid + 256 * i, r = f >> 3, c = (f & 7) * 4;
        int ar = row0 + r;
        ar = ar < M ? ar : M - 1;  // clamp: rows past M are computed and discarded
        a_src[i] = A + (int64_t)ar * lda + c;
        b_src[i] = W + (int64_t)(col0 + r) * K + c;
        lds_off[i] = r * LDS_LD + c;
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    f32x4 ra[4], rb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        ra[i] = *reinterpret_cast<const f32x4*>(a_src[i]);
        rb[i] = *reinterpret_cast<const f32x4*>(b_src[i]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        *reinterpret_cast<f32x4*>(As + lds_off[i]) = ra[i];
        *reinterpret_cast<f32x4*>(Bs + lds_off[i]) = rb[i];
    }
    __syncthreads();

    const int nk = K / BK;
    const int a_frag = (wr * 64 + l31) * LDS_LD + 4 * half;
    const int b_frag = (wc * 64 + l31) * LDS_LD + 4 * half;
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if (more) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ra[i] = *reinterpret_cast<const f32x4*>(a_src[i] + (kt + 1) * BK);
                rb[i] = *reinterpret_cast<const f32x4*>(b_src[i] + (kt + 1) * BK);
            }
        }
        const float* as = As + cur * TILE_F + a_frag;
        const float* bs = Bs + cur * TILE_F + b_frag;
        f32x16 part[2][2];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) part[m][n][r] = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            f32x4 fa[2], fb[2];
#pragma unroll
            for (int m = 0; m < 2; ++m)
                fa[m] = *reinterpret_cast<const f32x4*>(as + m * 32 * LDS_LD + s * 8);
#pragma unroll
            for (int n = 0; n < 2; ++n)
                fb[n] = *reinterpret_cast<const f32x4*>(bs + n * 32 * LDS_LD + s * 8);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        part[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[m][kk], fb[n][kk],
                                                                          part[m][n], 0, 0, 0);
        }
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n) acc[m][n] += part[m][n];
        if (more) {
            float* ad = As + (cur ^ 1) * TILE_F;
            float* bd = Bs + (cur ^ 1) * TILE_F;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                *reinterpret_cast<f32x4*>(ad + lds_off[i]) = ra[i];
                *reinterpret_cast<f32x4*>(bd + lds_off[i]) = rb[i];
            }
        }
        __syncthreads();
        cur ^= 1;
    }

    // epilogue: bias + activation, 128-byte row segments per half-wave
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

}  // namespace

extern "C" int e3d_gemm_bias_act_f32(const float* A, int64_t lda, const float* W,
                                     const float* bias, float* out, int64_t ldc, int M, int N,
                                     int K, int act, void* stream) {
    E3D_REQUIRE(A && W && out, "gemm: null pointer");
    E3D_REQUIRE(M > 0 && N > 0 && K > 0, "gemm: bad shape M=%d N=%d K=%d", M, N, K);
    E3D_REQUIRE(N % BN == 0 && K % BK == 0, "gemm: need N%%128==0 and K%%32==0 (N=%d K=%d)", N, K);
    E3D_REQUIRE(lda >= K && ldc >= N && lda % 4 == 0, "gemm: bad strides lda=%lld ldc=%lld",
                (long long)lda, (long long)ldc);
    E3D_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0, "gemm: operands must be 16B aligned");
    const int tiles_m = (M + BM - 1) / BM, tiles_n = N / BN;
    const dim3 grid(tiles_m * tiles_n), block(256);
    const size_t lds = 4 * TILE_F * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
    switch (act) {
        case E3D_ACT_NONE:
            hipLaunchKernelGGL(gemm_nt_f32<E3D_ACT_NONE>, grid, block, lds, s, A, lda, W, bias, out, ldc, M, N, K, tiles_m, tiles_n);
            break;
        case E3D_ACT_GELU:
            hipLaunchKernelGGL(gemm_nt_f32<E3D_ACT_GELU>, grid, block, lds, s, A, lda, W, bias, out, ldc, M, N, K, tiles_m, tiles_n);
            break;
        case E3D_ACT_SILU:
            hipLaunchKernelGGL(gemm_nt_f32<E3D_ACT_SILU>, grid, block, lds, s, A, lda, W, bias, out, ldc, M, N, K, tiles_m, tiles_n);
            break;
        default:
            E3D_REQUIRE(false, "gemm: unknown activation %d", act);
    }
    return e3d_launch_status("e3d_gemm_bias_act_f32");
}
