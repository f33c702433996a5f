// HBM-bound row kernels: LayerNorm epilogues, adaLN gate, feature embeddings, output head.
// One wavefront per row of H = 256*V floats; a lane keeps its V float4 (columns 4*(64*i+lane))
// in registers, so every row is read once and written once with 16-byte coalesced accesses and
// the mean/variance are two wave reductions (shuffles, no LDS).
#include "e3d_common.h"

namespace {

template <int V>
__device__ __forceinline__ void row_load(f32x4 (&r)[V], const float* p, int lane) {
#pragma unroll
    for (int i = 0; i < V; ++i) r[i] = *reinterpret_cast<const f32x4*>(p + 4 * (64 * i + lane));
}

template <int V>
__device__ __forceinline__ void row_store(const f32x4 (&r)[V], float* p, int lane) {
#pragma unroll
    for (int i = 0; i < V; ++i) *reinterpret_cast<f32x4*>(p + 4 * (64 * i + lane)) = r[i];
}

// in-place normalise: r <- (r - mean) * rstd ; two-pass (centered) variance, biased (1/H)
template <int V>
__device__ __forceinline__ void row_normalize(f32x4 (&r)[V], float eps) {
    constexpr float inv_h = 1.0f / (256 * V);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i) s += (r[i][0] + r[i][1]) + (r[i][2] + r[i][3]);
    const float mean = wave_sum(s) * inv_h;
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
    for (int i = 0; i < V; ++i) r[i] *= rstd;
}

// DROP: x is the output of a dense layer that nn.Dropout follows (BertSelfOutput / BertOutput in training mode): the
// multipliers of e3d_dropout_f32 for the same (p, seed) -- element index = row * H + column -- are applied as x is read,
// instead of by a pass of their own over the [M, H] tensor
template <int V, bool DROP = false>
__global__ __launch_bounds__(256) void residual_layernorm_kernel(
    const float* __restrict__ x, const float* __restrict__ res, const float* __restrict__ gamma,
    const float* __restrict__ beta, float eps, float* __restrict__ s_out, float* __restrict__ out, int M, E3dDrop drop_in = E3dDrop{}) {
    constexpr int H = 256 * V;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    f32x4 r[V], t[V];
    row_load<V>(r, x + (int64_t)row * H, lane);
    if (DROP) {
        const E3dDrop drop = e3d_drop_resolve(drop_in);
#pragma unroll
        for (int i = 0; i < V; ++i) {
            float m[4];
            e3d_drop_mult4(drop, (uint64_t)row * (H / 4) + 64 * i + lane, m);
#pragma unroll
            for (int j = 0; j < 4; ++j) r[i][j] *= m[j];
        }
    }
    if (res) {
        row_load<V>(t, res + (int64_t)row * H, lane);
#pragma unroll
        for (int i = 0; i < V; ++i) r[i] += t[i];
    }
    if (s_out) row_store<V>(r, s_out + (int64_t)row * H, lane);  // pre-norm sum, kept for the backward
    row_normalize<V>(r, eps);
    f32x4 g[V], b[V];
    row_load<V>(g, gamma, lane);
    row_load<V>(b, beta, lane);
#pragma unroll
    for (int i = 0; i < V; ++i) r[i] = r[i] * g[i] + b[i];
    row_store<V>(r, out + (int64_t)row * H, lane);
}

template <int V>
__global__ __launch_bounds__(256) void adaln_gate_kernel(const float* __restrict__ x,
                                                         const float* __restrict__ y,
                                                         const float* __restrict__ mod, int branch,
                                                         int rows_per_cond, float* __restrict__ out,
                                                         int M) {
    constexpr int H = 256 * V;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    f32x4 r[V], xr[V], sh[V], sc[V], ga[V];
    row_load<V>(r, y + (int64_t)row * H, lane);
    row_normalize<V>(r, 1e-5f);
    const float* mrow = mod + (int64_t)(row / rows_per_cond) * 6 * H + (int64_t)branch * 3 * H;
    row_load<V>(sh, mrow, lane);
    row_load<V>(sc, mrow + H, lane);
    row_load<V>(ga, mrow + 2 * H, lane);
    row_load<V>(xr, x + (int64_t)row * H, lane);
#pragma unroll
    for (int i = 0; i < V; ++i) r[i] = xr[i] + ga[i] * (r[i] * (1.0f + sc[i]) + sh[i]);
    row_store<V>(r, out + (int64_t)row * H, lane);
}

template <int V>
__global__ __launch_bounds__(256) void embed_layernorm_kernel(
    const float* __restrict__ x, int F, const float* __restrict__ W, const float* __restrict__ bias,
    const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
    const float* __restrict__ post_add, int rows_per_add, float* __restrict__ z_out,
    float* __restrict__ out, int M) {
    constexpr int H = 256 * V;
    // W^T [F][H] lives in LDS for the whole workgroup (W is [H][F]: reading it per row straight from
    // global memory is a stride-F gather); each wave then walks rows wave, wave + n_waves, ...
    extern __shared__ __attribute__((aligned(16))) float wt[];
    for (int i = threadIdx.x; i < H * F; i += blockDim.x) wt[(i % F) * H + i / F] = W[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = gridDim.x * 4;
    f32x4 g[V], b[V], bs[V];
    row_load<V>(g, gamma, lane);
    row_load<V>(b, beta, lane);
    row_load<V>(bs, bias, lane);
    // FOUR rows per trip (round 4): a W^T fragment read from LDS serves four rows -- the one-row form read F x V x 1 KB of
    // LDS per row (60 KB at F = 20), which, not the 3-KB output row, set its time (131 -> 82 us for 65 536 rows at F = 20, 80 -> 53 at F = 8; eight rows per trip:
    // 77 us at 228 registers, not kept); each row's sum
    // runs over f in the same order with the same fma, so results are unchanged
    constexpr int R = 4;
    for (int row0 = wave * R; row0 < M; row0 += n_waves * R) {
        const float* xr[R];
#pragma unroll
        for (int k = 0; k < R; ++k) xr[k] = x + (int64_t)min(row0 + k, M - 1) * F;
        f32x4 r[R][V];
#pragma unroll
        for (int k = 0; k < R; ++k)
#pragma unroll
            for (int i = 0; i < V; ++i) r[k][i] = bs[i];
        for (int f = 0; f < F; ++f) {
            float xv[R];
#pragma unroll
            for (int k = 0; k < R; ++k) xv[k] = xr[k][f];
#pragma unroll
            for (int i = 0; i < V; ++i) {   // explicit fma: the per-row kernel below must round the same way
                const f32x4 w = *reinterpret_cast<const f32x4*>(wt + f * H + 4 * (64 * i + lane));
#pragma unroll
                for (int k = 0; k < R; ++k)
#pragma unroll
                    for (int j = 0; j < 4; ++j) r[k][i][j] = __builtin_fmaf(xv[k], w[j], r[k][i][j]);
            }
        }
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int row = row0 + k;
            if (row >= M) break;
            if (z_out) row_store<V>(r[k], z_out + (int64_t)row * H, lane);  // pre-LayerNorm rows for the backward
            row_normalize<V>(r[k], eps);
#pragma unroll
            for (int i = 0; i < V; ++i) r[k][i] = r[k][i] * g[i] + b[i];
            if (post_add) {
                f32x4 a[V];
                row_load<V>(a, post_add + (int64_t)(row / rows_per_add) * H, lane);
#pragma unroll
                for (int i = 0; i < V; ++i) r[k][i] += a[i];
            }
            row_store<V>(r[k], out + (int64_t)row * H, lane);
        }
    }
}

// Few rows (single-pocket sampling): one wave per row and W [H][F] read where it lies -- a lane's 4 columns are 4F
// contiguous floats, the few rows of the launch share them through L2; the LDS-staged form above walks 128 rows per
// workgroup and would put 64 rows on ONE workgroup (40 us measured at M = 64 against 4 us here).  Same arithmetic in
// the same order as the staged form.
template <int V>
__global__ __launch_bounds__(256) void embed_layernorm_rows_kernel(
    const float* __restrict__ x, int F, const float* __restrict__ W, const float* __restrict__ bias,
    const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
    const float* __restrict__ post_add, int rows_per_add, float* __restrict__ z_out,
    float* __restrict__ out, int M) {
    constexpr int H = 256 * V;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = x + (int64_t)row * F;
    f32x4 r[V], g[V], b[V];
    row_load<V>(r, bias, lane);
    row_load<V>(g, gamma, lane);
    row_load<V>(b, beta, lane);
    if ((F & 3) == 0 && ((((uintptr_t)x | (uintptr_t)W) & 15) == 0)) {   // 16-byte loads of W (F = 8 angle features, 20 classes): same per-element order of the adds
        for (int f0 = 0; f0 < F; f0 += 4) {
            const f32x4 xv = *reinterpret_cast<const f32x4*>(xr + f0);
#pragma unroll
            for (int i = 0; i < V; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 w = *reinterpret_cast<const f32x4*>(W + (int64_t)(4 * (64 * i + lane) + j) * F + f0);
#pragma unroll
                    for (int q = 0; q < 4; ++q) r[i][j] = __builtin_fmaf(xv[q], w[q], r[i][j]);
                }
        }
    } else {
        for (int f = 0; f < F; ++f) {
            const float xv = xr[f];
#pragma unroll
            for (int i = 0; i < V; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) r[i][j] = __builtin_fmaf(xv, W[(int64_t)(4 * (64 * i + lane) + j) * F + f], r[i][j]);
        }
    }
    if (z_out) row_store<V>(r, z_out + (int64_t)row * H, lane);
    row_normalize<V>(r, eps);
#pragma unroll
    for (int i = 0; i < V; ++i) r[i] = r[i] * g[i] + b[i];
    if (post_add) {
        f32x4 a[V];
        row_load<V>(a, post_add + (int64_t)(row / rows_per_add) * H, lane);
#pragma unroll
        for (int i = 0; i < V; ++i) r[i] += a[i];
    }
    row_store<V>(r, out + (int64_t)row * H, lane);
}

template <int V>
__global__ __launch_bounds__(256) void head_linear_kernel(const float* __restrict__ x,
                                                          const float* __restrict__ W,
                                                          const float* __restrict__ b,
                                                          float* __restrict__ out, int M, int Nout) {
    constexpr int H = 256 * V;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    f32x4 r[V];
    row_load<V>(r, x + (int64_t)row * H, lane);
    float mine = 0.f;
    for (int n = 0; n < Nout; ++n) {
        f32x4 w[V];
        row_load<V>(w, W + (int64_t)n * H, lane);
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < V; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) s = fmaf(r[i][j], w[i][j], s);
        s = wave_sum(s);
        if (lane == n) mine = s + b[n];
    }
    if (lane < Nout) out[(int64_t)row * Nout + lane] = mine;
}

#define DISPATCH_V(H, CALL)                                             \
    switch (H) {                                                        \
        case 256: { constexpr int V = 1; CALL; } break;                 \
        case 512: { constexpr int V = 2; CALL; } break;                 \
        case 768: { constexpr int V = 3; CALL; } break;                 \
        case 1024: { constexpr int V = 4; CALL; } break;                \
        default: E3D_REQUIRE(false, "row op: H must be 256/512/768/1024 (H=%d)", H); \
    }

}  // namespace

extern "C" int e3d_residual_layernorm_fwd(const float* x, const float* residual,
                                          const float* gamma, const float* beta, float eps,
                                          float* s_out, float* out, int M, int H, void* stream) {
    E3D_REQUIRE(x && gamma && beta && out && M > 0, "residual_layernorm: bad arguments");
    const dim3 grid((M + 3) / 4), block(256);
    DISPATCH_V(H, hipLaunchKernelGGL(residual_layernorm_kernel<V>, grid, block, 0, (hipStream_t)stream, x,
                                     residual, gamma, beta, eps, s_out, out, M));
    return e3d_launch_status("e3d_residual_layernorm_fwd");
}

extern "C" int e3d_residual_layernorm_drop_fwd(const float* x, const float* residual, const float* gamma, const float* beta,
                                               float eps, float* s_out, float* out, int M, int H, float drop_p, uint64_t drop_seed,
                                               void* stream) {
    E3D_REQUIRE(x && gamma && beta && out && M > 0, "residual_layernorm_drop: bad arguments");
    E3D_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "residual_layernorm_drop: p = %f", (double)drop_p);
    const dim3 grid((M + 3) / 4), block(256);
    const E3dDrop d = e3d_drop_make(drop_p, drop_seed);
    DISPATCH_V(H, hipLaunchKernelGGL((residual_layernorm_kernel<V, true>), grid, block, 0, (hipStream_t)stream, x, residual, gamma,
                                     beta, eps, s_out, out, M, d));
    return e3d_launch_status("e3d_residual_layernorm_drop_fwd");
}

extern "C" int e3d_adaln_gate_fwd(const float* x, const float* y, const float* mod, int branch,
                                  int rows_per_cond, float* out, int M, int H, void* stream) {
    E3D_REQUIRE(x && y && mod && out && M > 0, "adaln_gate: bad arguments");
    E3D_REQUIRE((branch == 0 || branch == 1) && rows_per_cond >= 1, "adaln_gate: branch=%d rows_per_cond=%d",
                branch, rows_per_cond);
    const dim3 grid((M + 3) / 4), block(256);
    DISPATCH_V(H, hipLaunchKernelGGL(adaln_gate_kernel<V>, grid, block, 0, (hipStream_t)stream, x, y, mod, branch,
                                     rows_per_cond, out, M));
    return e3d_launch_status("e3d_adaln_gate_fwd");
}

extern "C" int e3d_embed_layernorm_fwd(const float* x, int F, const float* W, const float* b,
                                       const float* gamma, const float* beta, float eps,
                                       const float* post_add, int rows_per_add, float* z_out,
                                       float* out, int M, int H, void* stream) {
    E3D_REQUIRE(x && W && b && gamma && beta && out && M > 0, "embed_layernorm: bad arguments");
    E3D_REQUIRE(F >= 1 && F <= 32, "embed_layernorm: F must be in [1,32] (F=%d)", F);
    E3D_REQUIRE(!post_add || rows_per_add >= 1, "embed_layernorm: rows_per_add=%d", rows_per_add);
    if (M <= 512) {   // few rows: one wave per row, no staging
        const dim3 grid((M + 3) / 4), block(256);
        DISPATCH_V(H, hipLaunchKernelGGL(embed_layernorm_rows_kernel<V>, grid, block, 0, (hipStream_t)stream, x, F, W, b,
                                         gamma, beta, eps, post_add, rows_per_add, z_out, out, M));
        return e3d_launch_status("e3d_embed_layernorm_fwd");
    }
    // every workgroup re-stages W^T (F*H*4 bytes of LDS): each one amortises it over >= 32 rows, and a launch that does not
    // fill the chip at 128 rows per workgroup spreads out first (round 4: M = 8192 ran on 64 CUs -- 59 us in the training
    // steps; 32 rows per workgroup = 256 workgroups; M = 65 536 keeps its 512 x 128 rows)
    int rpb = (M + 255) / 256;
    rpb = rpb < 32 ? 32 : (rpb > 128 ? 128 : rpb);
    int blocks = (M + rpb - 1) / rpb;
    blocks = blocks < 1 ? 1 : (blocks > 512 ? 512 : blocks);
    const dim3 grid(blocks), block(256);
    const size_t lds = (size_t)F * H * sizeof(float);
    DISPATCH_V(H, {
        static std::atomic<uint64_t> lds_ok{0};
        e3d_allow_lds(lds_ok, embed_layernorm_kernel<V>, 32 * (size_t)H * sizeof(float));
        hipLaunchKernelGGL(embed_layernorm_kernel<V>, grid, block, lds, (hipStream_t)stream, x, F, W, b, gamma, beta,
                           eps, post_add, rows_per_add, z_out, out, M);
    });
    return e3d_launch_status("e3d_embed_layernorm_fwd");
}

extern "C" int e3d_head_linear_fwd(const float* x, const float* W, const float* b, float* out,
                                   int M, int H, int Nout, void* stream) {
    E3D_REQUIRE(x && W && b && out && M > 0, "head_linear: bad arguments");
    E3D_REQUIRE(Nout >= 1 && Nout <= 32, "head_linear: Nout must be in [1,32] (Nout=%d)", Nout);
    const dim3 grid((M + 3) / 4), block(256);
    DISPATCH_V(H, hipLaunchKernelGGL(head_linear_kernel<V>, grid, block, 0, (hipStream_t)stream, x, W, b, out, M,
                                     Nout));
    return e3d_launch_status("e3d_head_linear_fwd");
}

// ------------------------------------------------------------------------------------------------ |x| maximum
namespace {
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ target) {
    unsigned m = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) e3d_absmax_accum(m, x[i]);
    e3d_absmax_commit(m, target, threadIdx.x & 63);
}
}  // namespace

extern "C" int e3d_absmax_f32(const float* x, int64_t n, float* target, void* stream) {
    E3D_REQUIRE(x && target && n > 0, "absmax: bad arguments");
    int64_t blocks = (n + 256 * 8 - 1) / (256 * 8);
    blocks = blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks);
    hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, n, target);
    return e3d_launch_status("e3d_absmax_f32");
}
