// Backward / training-side HBM-bound kernels: activation fwd+bwd on saved pre-activations,
// LayerNorm and adaLN-gate backward, bias (column) and broadcast (group) reductions, and the
// weight gradients of the small-K embedding / small-N head linears.
// Row kernels follow rowops.hip: one wavefront per row of H = 256*V floats kept in registers.
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
template <int V>
__device__ __forceinline__ void row_atomic_add(const f32x4 (&r)[V], float* p, int lane) {
#pragma unroll
    for (int i = 0; i < V; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) atomicAdd(p + 4 * (64 * i + lane) + j, r[i][j]);
}

// r <- xhat = (r - mean) * rstd ; returns rstd
template <int V>
__device__ __forceinline__ float row_normalize(f32x4 (&r)[V], float eps) {
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
    return rstd;
}

// g <- rstd * (g - mean(g) - xhat * mean(g * xhat))   (LayerNorm input gradient)
template <int V>
__device__ __forceinline__ void ln_input_grad(f32x4 (&g)[V], const f32x4 (&xhat)[V], float rstd) {
    constexpr float inv_h = 1.0f / (256 * V);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s1 += g[i][j];
            s2 += g[i][j] * xhat[i][j];
        }
    const float m1 = wave_sum(s1) * inv_h, m2 = wave_sum(s2) * inv_h;
#pragma unroll
    for (int i = 0; i < V; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) g[i][j] = rstd * (g[i][j] - m1 - xhat[i][j] * m2);
}

// ---------------------------------------------------------------- LayerNorm backward
// Each wave walks rows wave_id, wave_id + n_waves, ...; dgamma/dbeta partials stay in registers, are
// summed over the block in LDS and added once per block (float atomics: order-dependent last bits).
// DROP: the LayerNorm's first input was dropout(x) (e3d_residual_layernorm_drop_fwd): besides ds (the gradient of the
// pre-norm sum = of the residual) the kernel writes ds * multipliers (the gradient of x) to ``dsd``
template <int V, bool DROP = false>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ dy,
                                                            const float* __restrict__ s,
                                                            const float* __restrict__ gamma, float eps,
                                                            float* __restrict__ ds, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, int M, float* __restrict__ dsd = nullptr,
                                                            E3dDrop drop_in = E3dDrop{}, float* __restrict__ part = nullptr) {
    constexpr int H = 256 * V;
    const int lane = threadIdx.x & 63;
    const E3dDrop drop = DROP ? e3d_drop_resolve(drop_in) : drop_in;
    auto store_dropped = [&](const f32x4 (&g)[V], int row) {
        f32x4 d[V];
#pragma unroll
        for (int i = 0; i < V; ++i) {
            float m[4];
            e3d_drop_mult4(drop, (uint64_t)row * (H / 4) + 64 * i + lane, m);
#pragma unroll
            for (int j = 0; j < 4; ++j) d[i][j] = g[i][j] * m[j];
        }
        row_store<V>(d, dsd + (int64_t)row * H, lane);
    };
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = gridDim.x * 4;
    f32x4 ga[V], acc_g[V], acc_b[V];
    if (gamma) row_load<V>(ga, gamma, lane);
#pragma unroll
    for (int i = 0; i < V; ++i) { acc_g[i] = 0.f; acc_b[i] = 0.f; }
    // two rows per trip, all four row loads issued before either row's reductions (a wave's rows were one dependent
    // load -> reduce -> store chain each: 16 us at M = 4096); partial sums in the same row order as before
    for (int row = wave; row < M; row += 2 * n_waves) {
        const int row2 = row + n_waves;
        const bool two = row2 < M;
        f32x4 x[V], g[V], x2[V], g2[V];
        row_load<V>(x, s + (int64_t)row * H, lane);
        row_load<V>(g, dy + (int64_t)row * H, lane);
        row_load<V>(x2, s + (int64_t)(two ? row2 : row) * H, lane);
        row_load<V>(g2, dy + (int64_t)(two ? row2 : row) * H, lane);
        const float rstd = row_normalize<V>(x, eps);
#pragma unroll
        for (int i = 0; i < V; ++i) {
            acc_g[i] += g[i] * x[i];
            acc_b[i] += g[i];
            if (gamma) g[i] *= ga[i];
        }
        ln_input_grad<V>(g, x, rstd);
        row_store<V>(g, ds + (int64_t)row * H, lane);
        if (DROP) store_dropped(g, row);
        if (two) {
            const float rstd2 = row_normalize<V>(x2, eps);
#pragma unroll
            for (int i = 0; i < V; ++i) {
                acc_g[i] += g2[i] * x2[i];
                acc_b[i] += g2[i];
                if (gamma) g2[i] *= ga[i];
            }
            ln_input_grad<V>(g2, x2, rstd2);
            row_store<V>(g2, ds + (int64_t)row2 * H, lane);
            if (DROP) store_dropped(g2, row2);
        }
    }
    // the block's 4 waves reduce through LDS first: one atomic per column per block
    __shared__ float red[4][2][H];
    const int w = threadIdx.x >> 6;
    row_store<V>(acc_g, red[w][0], lane);
    row_store<V>(acc_b, red[w][1], lane);
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * H; c += 256) {
        const int which = c / H, col = c - which * H;
        float* dst = which ? dbeta : dgamma;
        const float v = red[0][which][col] + red[1][which][col] + red[2][which][col] + red[3][which][col];
        // ``part`` (the _ws entry points): the block's sums go to row blockIdx.x of a [blocks][2 H] workspace and
        // ln_param_grad_sum_kernel adds the rows up in a fixed order -- no zero-fill launch, no atomics (256-512 adders per
        // address were ~9 us of a 15-us launch at M = 4096: twice the blocks cost +9 us per launch in the training step)
        if (part) part[(int64_t)blockIdx.x * 2 * H + c] = v;
        else if (dst) atomicAdd(dst + col, v);
    }
}

// dgamma | dbeta [2 H] = column sums of the per-block partial rows [n_part][2 H], n_part <= 512 (fixed order: deterministic).
// 64 columns x 16 row slices per block; a thread's <= 32 rows are ALL in flight before the first add (a loop of 8 loads per
// trip took 12 us for 3 MB at n_part = 512: four exposed latencies on 24 workgroups).
__global__ __launch_bounds__(1024) void ln_param_grad_sum_kernel(const float* __restrict__ part, int n_part, int H2,
                                                                 float* __restrict__ dgamma, float* __restrict__ dbeta, int H) {
    const int cl = threadIdx.x & 63, c = blockIdx.x * 64 + cl, sl = threadIdx.x >> 6;
    __shared__ float red[16][64];
    float a[32];
    if (c < H2) {
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            const int r = sl + 16 * u;
            a[u] = r < n_part ? part[(int64_t)r * H2 + c] : 0.f;
        }
    } else {
#pragma unroll
        for (int u = 0; u < 32; ++u) a[u] = 0.f;
    }
#pragma unroll
    for (int st = 16; st > 0; st >>= 1)
#pragma unroll
        for (int u = 0; u < st; ++u) a[u] += a[u + st];
    red[sl][cl] = a[0];
    __syncthreads();
    if (sl == 0 && c < H2) {
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) v += red[k][cl];
        float* dst = c < H ? dgamma : dbeta;
        if (dst) dst[c < H ? c : c - H] = v;
    }
}

// ---------------------------------------------------------------- adaLN gate backward
template <int V>
__global__ __launch_bounds__(256) void adaln_gate_bwd_kernel(const float* __restrict__ dout,
                                                             const float* __restrict__ y,
                                                             const float* __restrict__ mod, int branch,
                                                             int rows_per_cond, float* __restrict__ dy,
                                                             float* __restrict__ dmod, int M) {
    constexpr int H = 256 * V;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    f32x4 yh[V], go[V], sh[V], sc[V], ga[V];
    row_load<V>(yh, y + (int64_t)row * H, lane);
    const float rstd = row_normalize<V>(yh, 1e-5f);
    row_load<V>(go, dout + (int64_t)row * H, lane);
    const int64_t moff = (int64_t)(row / rows_per_cond) * 6 * H + (int64_t)branch * 3 * H;
    row_load<V>(sh, mod + moff, lane);
    row_load<V>(sc, mod + moff + H, lane);
    row_load<V>(ga, mod + moff + 2 * H, lane);
    f32x4 dsh[V], dsc[V], dga[V];
#pragma unroll
    for (int i = 0; i < V; ++i) {
        dga[i] = go[i] * (yh[i] * (1.0f + sc[i]) + sh[i]);
        dsh[i] = go[i] * ga[i];
        dsc[i] = dsh[i] * yh[i];
        go[i] = dsh[i] * (1.0f + sc[i]);  // d(yhat)
    }
    ln_input_grad<V>(go, yh, rstd);
    row_store<V>(go, dy + (int64_t)row * H, lane);
    if (rows_per_cond == 1) {
        row_store<V>(dsh, dmod + moff, lane);
        row_store<V>(dsc, dmod + moff + H, lane);
        row_store<V>(dga, dmod + moff + 2 * H, lane);
    } else {
        row_atomic_add<V>(dsh, dmod + moff, lane);
        row_atomic_add<V>(dsc, dmod + moff + H, lane);
        row_atomic_add<V>(dga, dmod + moff + 2 * H, lane);
    }
}

// The same for conditioning rows shared by rows_per_cond % 16 == 0 consecutive rows (one timestep row per item): a
// workgroup takes 16 rows of ONE conditioning row (4 per wave), sums the three modulation gradients in registers, then
// over its waves in LDS, and adds them once -- 1/16 of the float atomics of the per-row form (88 -> ~25 us at
// 4096 x 768: 9.4 M atomics onto 32 x 2304 addresses were the kernel).
template <int V>
__global__ __launch_bounds__(256) void adaln_gate_bwd_shared_kernel(const float* __restrict__ dout,
                                                                    const float* __restrict__ y,
                                                                    const float* __restrict__ mod, int branch,
                                                                    int rows_per_cond, float* __restrict__ dy,
                                                                    float* __restrict__ dmod, int M) {
    constexpr int H = 256 * V;
    __shared__ float red[4][3][H];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int row0 = blockIdx.x * 16;
    const int64_t moff = (int64_t)(row0 / rows_per_cond) * 6 * H + (int64_t)branch * 3 * H;
    f32x4 sh[V], sc[V], ga[V], a_sh[V], a_sc[V], a_ga[V];
    row_load<V>(sh, mod + moff, lane);
    row_load<V>(sc, mod + moff + H, lane);
    row_load<V>(ga, mod + moff + 2 * H, lane);
#pragma unroll
    for (int i = 0; i < V; ++i) { a_sh[i] = 0.f; a_sc[i] = 0.f; a_ga[i] = 0.f; }
#pragma unroll 2
    for (int r = 0; r < 4; ++r) {
        const int row = row0 + 4 * r + w;
        if (row >= M) break;
        f32x4 yh[V], go[V];
        row_load<V>(yh, y + (int64_t)row * H, lane);
        row_load<V>(go, dout + (int64_t)row * H, lane);
        const float rstd = row_normalize<V>(yh, 1e-5f);
#pragma unroll
        for (int i = 0; i < V; ++i) {
            a_ga[i] += go[i] * (yh[i] * (1.0f + sc[i]) + sh[i]);
            const f32x4 dsh = go[i] * ga[i];
            a_sh[i] += dsh;
            a_sc[i] += dsh * yh[i];
            go[i] = dsh * (1.0f + sc[i]);  // d(yhat)
        }
        ln_input_grad<V>(go, yh, rstd);
        row_store<V>(go, dy + (int64_t)row * H, lane);
    }
    row_store<V>(a_sh, red[w][0], lane);
    row_store<V>(a_sc, red[w][1], lane);
    row_store<V>(a_ga, red[w][2], lane);
    __syncthreads();
    for (int c = threadIdx.x; c < 3 * H; c += 256) {
        const int which = c / H, col = c - which * H;
        atomicAdd(dmod + moff + (int64_t)which * H + col, red[0][which][col] + red[1][which][col] + red[2][which][col] + red[3][which][col]);
    }
}

// ---------------------------------------------------------------- activations on saved z
__device__ __forceinline__ float act_apply(float z, int act) {
    return act == E3D_ACT_GELU ? gelu_erf(z) : (act == E3D_ACT_SILU ? silu(z) : z);
}
__device__ __forceinline__ float act_grad(float z, int act) {
    if (act == E3D_ACT_GELU)  // 0.5(1+erf(z/sqrt2)) + z exp(-z^2/2)/sqrt(2 pi)
        return 0.5f * (1.0f + erff(z * 0.70710678118654752440f)) + z * expf(-0.5f * z * z) * 0.39894228040143267794f;
    if (act == E3D_ACT_SILU) {
        const float sg = 1.0f / (1.0f + expf(-z));
        return sg * (1.0f + z * (1.0f - sg));
    }
    return 1.0f;
}

__global__ __launch_bounds__(256) void act_fwd_kernel(const float* __restrict__ z, int act, float* __restrict__ out,
                                                      int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = act_apply(z[i], act);
}
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ dh, const float* __restrict__ z, int act,
                                                      float* __restrict__ dz, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        dz[i] = dh[i] * act_grad(z[i], act);
}

// ---------------------------------------------------------------- reductions
// out[n] += sum over this block's row span of x[m, n]   (out pre-zeroed by the launcher)
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, int64_t ld, float* __restrict__ out,
                                                     int M, int N, int rows_per_block) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const int m0 = blockIdx.y * rows_per_block, m1 = min(M, m0 + rows_per_block);
    float acc = 0.f;
    for (int m = m0; m < m1; ++m) acc += x[(int64_t)m * ld + n];
    atomicAdd(out + n, acc);
}

// out[g, h] = sum_{i < rows_per_group} x[g * rows_per_group + i, h]   (deterministic)
__global__ __launch_bounds__(256) void group_sum_kernel(const float* __restrict__ x, int rows_per_group,
                                                        float* __restrict__ out, int H) {
    const int h = blockIdx.x * 256 + threadIdx.x, grp = blockIdx.y;
    if (h >= H) return;
    const float* p = x + (int64_t)grp * rows_per_group * H + h;
    float acc = 0.f;
    for (int i = 0; i < rows_per_group; ++i) acc += p[(int64_t)i * H];
    out[(int64_t)grp * H + h] = acc;
}

// dW[h, f] (or [f, h]) += sum_m g[m, h] * x[m, f],  db[h] += sum_m g[m, h];  F <= 32.
__global__ __launch_bounds__(256) void small_k_wgrad_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                            float* __restrict__ dW, float* __restrict__ db, int M,
                                                            int H, int F, int transpose_out, int rows_per_block) {
    // A block = 64 output columns x rows_per_block (<= 256) rows: wave w takes every 4th row, a lane one column h.  The
    // block's x rows (contiguous) are staged in LDS once and read back as broadcasts, the g values are fetched 8 rows at a
    // time (independent 256-byte loads), the four waves' partials meet in LDS and leave as ONE atomic per output word
    // and block.  (History: one thread per column over 64 rows, 21 atomics per thread -- 2 M float atomics onto 16 K
    // addresses were the kernel: 114 us at M = 8192, H = 768, F = 20.)
    __shared__ __attribute__((aligned(16))) float xs[256 * 32];
    __shared__ float red[4][33][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int h = blockIdx.x * 64 + lane;
    const int m0 = blockIdx.y * rows_per_block, rows = min(M, m0 + rows_per_block) - m0;
    // x rows padded to 32 floats in LDS (zeros past F): the inner loop is then 8 broadcast 16-byte reads and 32 FMAs per row
    // without a branch per feature (round 4: the F-dependent `if` inside the unrolled loop was a scalar branch per feature and
    // row -- 63 us at M = 8192, H = 768, F = 20 for 25 MB of g)
    {   // (all 32 loads of a thread in flight: a rolled loop paid one load latency per trip -- most of the kernel's time)
        const int f = threadIdx.x & 31, r0 = threadIdx.x >> 5;
        float xv[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            const int r = r0 + 8 * u;
            xv[u] = (f < F && r < rows) ? x[(int64_t)(m0 + r) * F + f] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 32; ++u) xs[(r0 + 8 * u) * 32 + f] = xv[u];
    }
    __syncthreads();
    float acc[32];
#pragma unroll
    for (int f = 0; f < 32; ++f) acc[f] = 0.f;
    float accb = 0.f;
    const float* gp = g + (int64_t)m0 * H + min(h, H - 1);
    for (int m = w; m < rows; m += 32) {
        float gv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) gv[j] = m + 4 * j < rows ? gp[(int64_t)(m + 4 * j) * H] : 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            accb += gv[j];
            const f32x4* xr = reinterpret_cast<const f32x4*>(xs + min(m + 4 * j, rows - 1) * 32);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const f32x4 xv = xr[q];
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[4 * q + t] = fmaf(gv[j], xv[t], acc[4 * q + t]);
            }
        }
    }
#pragma unroll
    for (int f = 0; f < 32; ++f)
        if (f < F) red[w][f][lane] = acc[f];
    red[w][32][lane] = accb;
    __syncthreads();
    if (h >= H) return;
    for (int f = w; f <= 32; f += 4) {      // f == 32: the bias column
        if (f < F || f == 32) {
            const float v = (red[0][f][lane] + red[1][f][lane]) + (red[2][f][lane] + red[3][f][lane]);
            if (f < 32) atomicAdd(dW + (transpose_out ? (int64_t)f * H + h : (int64_t)h * F + f), v);
            else if (db) atomicAdd(db + h, v);
        }
    }
}

// The same sums for H % 64 == 0 (every model width) with 1-KB loads: a lane takes FOUR consecutive columns (float4) of one of
// the four rows a wave reads per instruction (lane = 16 row-group + column quad), 4 x 32 accumulators per lane; the four row
// groups meet through two shuffles per accumulator at the end, the four waves through LDS, the blocks through one atomic per
// output word as above.  (Round 4: the form above read 256 bytes per load instruction and took 53 us for 25 MB of g at
// M = 8192, H = 768 -- latency, not bytes.)
__global__ __launch_bounds__(256) void small_k_wgrad4_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                             float* __restrict__ dW, float* __restrict__ db, int M, int H, int F,
                                                             int transpose_out, int rows_per_block) {
    __shared__ __attribute__((aligned(16))) float xs[256 * 32];
    __shared__ float red[4][33][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, rg = lane >> 4, c4 = lane & 15;
    const int h0 = blockIdx.x * 64;
    const int m0 = blockIdx.y * rows_per_block, rows = min(M, m0 + rows_per_block) - m0;
    {   // (all 32 loads of a thread in flight: a rolled loop paid one load latency per trip -- most of the kernel's time)
        const int f = threadIdx.x & 31, r0 = threadIdx.x >> 5;
        float xv[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            const int r = r0 + 8 * u;
            xv[u] = (f < F && r < rows) ? x[(int64_t)(m0 + r) * F + f] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 32; ++u) xs[(r0 + 8 * u) * 32 + f] = xv[u];
    }
    __syncthreads();
    float acc[4][32], accb[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        accb[c] = 0.f;
#pragma unroll
        for (int f = 0; f < 32; ++f) acc[c][f] = 0.f;
    }
    const float* gp = g + (int64_t)m0 * H + h0 + 4 * c4;
    for (int it0 = 0; it0 * 16 < rows; it0 += 4) {
        f32x4 gv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int row = 16 * (it0 + u) + 4 * w + rg;
            gv[u] = row < rows ? *reinterpret_cast<const f32x4*>(gp + (int64_t)row * H) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int row = min(16 * (it0 + u) + 4 * w + rg, rows - 1);
            const f32x4* xr = reinterpret_cast<const f32x4*>(xs + row * 32);
#pragma unroll
            for (int c = 0; c < 4; ++c) accb[c] += gv[u][c];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const f32x4 xv = xr[q];
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[c][4 * q + t] = fmaf(gv[u][c], xv[t], acc[c][4 * q + t]);
            }
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int f = 0; f < 32; ++f) {
            float v = acc[c][f];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (rg == 0) red[w][f][4 * c4 + c] = v;
        }
        float b = accb[c];
        b += __shfl_xor(b, 16, 64);
        b += __shfl_xor(b, 32, 64);
        if (rg == 0) red[w][32][4 * c4 + c] = b;
    }
    __syncthreads();
    // one atomic per output word and block, consecutive lanes on consecutive words: dW[h][f] of the block's 64 columns is one
    // contiguous run of 64 F floats (a lane per column h, as the form above does it, puts the 64 lanes of every atomic
    // instruction on 64 different cache lines F floats apart -- that, not the loads, was the kernel's time)
    for (int o = threadIdx.x; o < 64 * F; o += 256) {
        const int hh = transpose_out ? (o & 63) : o / F, f = transpose_out ? (o >> 6) : o - hh * F;
        const float v = (red[0][f][hh] + red[1][f][hh]) + (red[2][f][hh] + red[3][f][hh]);
        atomicAdd(dW + (transpose_out ? (int64_t)f * H + h0 + hh : (int64_t)h0 * F + o), v);
    }
    if (db && threadIdx.x < 64) {
        const int hh = threadIdx.x;
        atomicAdd(db + h0 + hh, (red[0][32][hh] + red[1][32][hh]) + (red[2][32][hh] + red[3][32][hh]));
    }
}

// dx[m, :] = sum_n dout[m, n] * W[n, :]   (output head, n <= 32)
template <int V>
__global__ __launch_bounds__(256) void head_linear_bwd_dx_kernel(const float* __restrict__ dout,
                                                                 const float* __restrict__ W, float* __restrict__ dx,
                                                                 int M, int Nout) {
    constexpr int H = 256 * V;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    f32x4 r[V];
#pragma unroll
    for (int i = 0; i < V; ++i) r[i] = 0.f;
    for (int n = 0; n < Nout; ++n) {
        const float d = dout[(int64_t)row * Nout + n];
        f32x4 w[V];
        row_load<V>(w, W + (int64_t)n * H, lane);
#pragma unroll
        for (int i = 0; i < V; ++i) r[i] += d * w[i];
    }
    row_store<V>(r, dx + (int64_t)row * H, lane);
}

#define DISPATCH_V(H, CALL)                                             \
    switch (H) {                                                        \
        case 256: { constexpr int V = 1; CALL; } break;                 \
        case 512: { constexpr int V = 2; CALL; } break;                 \
        case 768: { constexpr int V = 3; CALL; } break;                 \
        case 1024: { constexpr int V = 4; CALL; } break;                \
        default: E3D_REQUIRE(false, "row op: H must be 256/512/768/1024 (H=%d)", H); \
    }

int elementwise_blocks(int64_t n) {
    int64_t b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

// ---------------------------------------------------------------- grouped transposes (W^T of many weights at once)
struct TransposeGroup {
    const float* src[64];
    float* dst[64];
};

// dst_p[c * ld_dst + r] = src_p[r * ld_src + c] for up to 64 matrices of one shape: 32x32 tiles through LDS
__global__ __launch_bounds__(256) void transpose_grouped_kernel(const TransposeGroup g, int rows, int cols, int64_t ld_src,
                                                                int64_t ld_dst, int tiles_c) {
    __shared__ float tile[32][33];
    const float* __restrict__ src = g.src[blockIdx.y];
    float* __restrict__ dst = g.dst[blockIdx.y];
    const int tr = blockIdx.x / tiles_c, tc = blockIdx.x % tiles_c;
    const int x = threadIdx.x & 31, y0 = threadIdx.x >> 5;
#pragma unroll
    for (int y = y0; y < 32; y += 8) {
        const int r = tr * 32 + y, c = tc * 32 + x;
        if (r < rows && c < cols) tile[y][x] = src[(int64_t)r * ld_src + c];
    }
    __syncthreads();
#pragma unroll
    for (int y = y0; y < 32; y += 8) {
        const int c = tc * 32 + y, r = tr * 32 + x;
        if (r < rows && c < cols) dst[(int64_t)c * ld_dst + r] = tile[x][y];
    }
}

}  // namespace

extern "C" int e3d_layernorm_bwd(const float* dy, const float* s, const float* gamma, float eps, float* ds,
                                 float* dgamma, float* dbeta, int M, int H, void* stream) {
    E3D_REQUIRE(dy && s && ds && M > 0, "layernorm_bwd: bad arguments");
    hipError_t e = hipSuccess;
    if (dgamma && dbeta == dgamma + H) {   // one buffer (autograd.layernorm_bwd allocates them so): one memset
        e = e3d_zero_async(dgamma, (size_t)2 * H, (hipStream_t)stream);
    } else {
        if (dgamma) e = e3d_zero_async(dgamma, (size_t)H, (hipStream_t)stream);
        if (e == hipSuccess && dbeta) e = e3d_zero_async(dbeta, (size_t)H, (hipStream_t)stream);
    }
    E3D_REQUIRE(e == hipSuccess, "layernorm_bwd: memset failed: %s", hipGetErrorString(e));
    const int blocks = (M + 15) / 16 < 512 ? (M + 15) / 16 : 512;
    DISPATCH_V(H, hipLaunchKernelGGL(layernorm_bwd_kernel<V>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dy, s,
                                     gamma, eps, ds, dgamma, dbeta, M));
    return e3d_launch_status("e3d_layernorm_bwd");
}

static int ln_bwd_blocks(int M) { return (M + 15) / 16 < 512 ? (M + 15) / 16 : 512; }

extern "C" int64_t e3d_layernorm_bwd_workspace_floats(int M, int H) { return (M <= 0 || H <= 0) ? -1 : (int64_t)ln_bwd_blocks(M) * 2 * H; }

// As e3d_layernorm_bwd / e3d_layernorm_bwd_drop (ds_dropped == NULL: no dropout), with the parameter gradients summed through
// ``workspace`` (e3d_layernorm_bwd_workspace_floats(M, H) floats) instead of atomics on a zeroed output: deterministic.
extern "C" int e3d_layernorm_bwd_ws(const float* dy, const float* s, const float* gamma, float eps, float* ds, float* ds_dropped,
                                    float* dgamma, float* dbeta, int M, int H, float drop_p, uint64_t drop_seed, float* workspace,
                                    int64_t workspace_floats, void* stream) {
    E3D_REQUIRE(dy && s && ds && M > 0, "layernorm_bwd_ws: bad arguments");
    E3D_REQUIRE(drop_p >= 0.f && drop_p < 1.f && (ds_dropped || drop_p == 0.f), "layernorm_bwd_ws: p = %f", (double)drop_p);
    const bool affine = dgamma || dbeta;
    const int blocks = ln_bwd_blocks(M);
    E3D_REQUIRE(!affine || (workspace && workspace_floats >= (int64_t)blocks * 2 * H),
                "layernorm_bwd_ws: workspace of %lld floats, %lld needed", (long long)workspace_floats, (long long)blocks * 2 * H);
    float* part = affine ? workspace : nullptr;
    hipStream_t st = (hipStream_t)stream;
    if (ds_dropped) {
        const E3dDrop d = e3d_drop_make(drop_p, drop_seed);
        DISPATCH_V(H, hipLaunchKernelGGL((layernorm_bwd_kernel<V, true>), dim3(blocks), dim3(256), 0, st, dy, s, gamma, eps, ds,
                                         dgamma, dbeta, M, ds_dropped, d, part));
    } else {
        DISPATCH_V(H, hipLaunchKernelGGL(layernorm_bwd_kernel<V>, dim3(blocks), dim3(256), 0, st, dy, s, gamma, eps, ds, dgamma,
                                         dbeta, M, (float*)nullptr, E3dDrop{}, part));
    }
    if (affine)
        hipLaunchKernelGGL(ln_param_grad_sum_kernel, dim3((2 * H + 63) / 64), dim3(1024), 0, st, part, blocks, 2 * H, dgamma, dbeta, H);
    return e3d_launch_status("e3d_layernorm_bwd_ws");
}

extern "C" int e3d_layernorm_bwd_drop(const float* dy, const float* s, const float* gamma, float eps, float* ds, float* ds_dropped,
                                      float* dgamma, float* dbeta, int M, int H, float drop_p, uint64_t drop_seed, void* stream) {
    E3D_REQUIRE(dy && s && ds && ds_dropped && M > 0, "layernorm_bwd_drop: bad arguments");
    E3D_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "layernorm_bwd_drop: p = %f", (double)drop_p);
    hipError_t e = hipSuccess;
    if (dgamma && dbeta == dgamma + H) {
        e = e3d_zero_async(dgamma, (size_t)2 * H, (hipStream_t)stream);
    } else {
        if (dgamma) e = e3d_zero_async(dgamma, (size_t)H, (hipStream_t)stream);
        if (e == hipSuccess && dbeta) e = e3d_zero_async(dbeta, (size_t)H, (hipStream_t)stream);
    }
    E3D_REQUIRE(e == hipSuccess, "layernorm_bwd_drop: zero-fill failed: %s", hipGetErrorString(e));
    const int blocks = (M + 15) / 16 < 512 ? (M + 15) / 16 : 512;
    const E3dDrop d = e3d_drop_make(drop_p, drop_seed);
    DISPATCH_V(H, hipLaunchKernelGGL((layernorm_bwd_kernel<V, true>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, dy, s, gamma,
                                     eps, ds, dgamma, dbeta, M, ds_dropped, d));
    return e3d_launch_status("e3d_layernorm_bwd_drop");
}

extern "C" int e3d_adaln_gate_bwd(const float* dout, const float* y, const float* mod, int branch,
                                  int rows_per_cond, float* dy, float* dmod, int M, int H, void* stream) {
    E3D_REQUIRE(dout && y && mod && dy && dmod && M > 0, "adaln_gate_bwd: bad arguments");
    E3D_REQUIRE((branch == 0 || branch == 1) && rows_per_cond >= 1, "adaln_gate_bwd: branch=%d rows_per_cond=%d", branch,
                rows_per_cond);
    if (rows_per_cond % 16 == 0) {
        DISPATCH_V(H, hipLaunchKernelGGL(adaln_gate_bwd_shared_kernel<V>, dim3((M + 15) / 16), dim3(256), 0, (hipStream_t)stream,
                                         dout, y, mod, branch, rows_per_cond, dy, dmod, M));
        return e3d_launch_status("e3d_adaln_gate_bwd");
    }
    DISPATCH_V(H, hipLaunchKernelGGL(adaln_gate_bwd_kernel<V>, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, dout,
                                     y, mod, branch, rows_per_cond, dy, dmod, M));
    return e3d_launch_status("e3d_adaln_gate_bwd");
}

extern "C" int e3d_act_fwd(const float* z, int act, float* out, int64_t n, void* stream) {
    E3D_REQUIRE(z && out && n > 0 && act >= 0 && act <= 2, "act_fwd: bad arguments");
    hipLaunchKernelGGL(act_fwd_kernel, dim3(elementwise_blocks(n)), dim3(256), 0, (hipStream_t)stream, z, act, out, n);
    return e3d_launch_status("e3d_act_fwd");
}

extern "C" int e3d_act_bwd(const float* dh, const float* z, int act, float* dz, int64_t n, void* stream) {
    E3D_REQUIRE(dh && z && dz && n > 0 && act >= 0 && act <= 2, "act_bwd: bad arguments");
    hipLaunchKernelGGL(act_bwd_kernel, dim3(elementwise_blocks(n)), dim3(256), 0, (hipStream_t)stream, dh, z, act, dz, n);
    return e3d_launch_status("e3d_act_bwd");
}

extern "C" int e3d_colsum(const float* x, int64_t ld, float* out, int M, int N, void* stream) {
    E3D_REQUIRE(x && out && M > 0 && N > 0 && ld >= N, "colsum: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = e3d_zero_async(out, (size_t)N, s);
    E3D_REQUIRE(e == hipSuccess, "colsum: memset failed: %s", hipGetErrorString(e));
    // ~1024 blocks of 256 columns x rpb rows
    const int col_blocks = (N + 255) / 256;
    int rpb = (int)(((int64_t)M * col_blocks + 1023) / 1024);
    rpb = rpb < 16 ? 16 : rpb;
    hipLaunchKernelGGL(colsum_kernel, dim3((N + 255) / 256, (M + rpb - 1) / rpb), dim3(256), 0, s, x, ld, out, M, N, rpb);
    return e3d_launch_status("e3d_colsum");
}

extern "C" int e3d_transpose_grouped_f32(const float* const* src, float* const* dst, int count, int rows, int cols,
                                        int64_t ld_src, int64_t ld_dst, void* stream) {
    E3D_REQUIRE(src && dst && count >= 1 && count <= 64, "transpose_grouped: 1..64 matrices (count=%d)", count);
    E3D_REQUIRE(rows > 0 && cols > 0 && ld_src >= cols && ld_dst >= rows, "transpose_grouped: bad shape %d x %d (ld %lld / %lld)",
                rows, cols, (long long)ld_src, (long long)ld_dst);
    TransposeGroup g;
    for (int p = 0; p < 64; ++p) {
        const int q = p < count ? p : 0;
        E3D_REQUIRE(src[q] && dst[q], "transpose_grouped: null pointer in matrix %d", q);
        g.src[p] = src[q];
        g.dst[p] = dst[q];
    }
    const int tiles_r = (rows + 31) / 32, tiles_c = (cols + 31) / 32;
    hipLaunchKernelGGL(transpose_grouped_kernel, dim3(tiles_r * tiles_c, count), dim3(256), 0, (hipStream_t)stream, g, rows, cols,
                       ld_src, ld_dst, tiles_c);
    return e3d_launch_status("e3d_transpose_grouped_f32");
}

extern "C" int e3d_group_sum(const float* x, int rows_per_group, float* out, int M, int H, void* stream) {
    E3D_REQUIRE(x && out && M > 0 && H > 0 && rows_per_group >= 1 && M % rows_per_group == 0,
                "group_sum: bad arguments (M=%d rows_per_group=%d)", M, rows_per_group);
    hipLaunchKernelGGL(group_sum_kernel, dim3((H + 255) / 256, M / rows_per_group), dim3(256), 0, (hipStream_t)stream, x,
                       rows_per_group, out, H);
    return e3d_launch_status("e3d_group_sum");
}

extern "C" int e3d_small_k_wgrad(const float* g, const float* x, float* dW, float* db, int M, int H, int F,
                                 int transpose_out, void* stream) {
    E3D_REQUIRE(g && x && dW && M > 0 && H > 0 && F >= 1 && F <= 32, "small_k_wgrad: bad arguments (F=%d)", F);
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = e3d_zero_async(dW, (size_t)H * F, s);
    if (e == hipSuccess && db) e = e3d_zero_async(db, (size_t)H, s);
    E3D_REQUIRE(e == hipSuccess, "small_k_wgrad: memset failed: %s", hipGetErrorString(e));
    const int rpb = 256;   // = the kernels' LDS tile of x rows
    if (H % 64 == 0 && ((uintptr_t)g % 16) == 0)
        hipLaunchKernelGGL(small_k_wgrad4_kernel, dim3(H / 64, (M + rpb - 1) / rpb), dim3(256), 0, s, g, x, dW, db, M, H, F,
                           transpose_out, rpb);
    else
        hipLaunchKernelGGL(small_k_wgrad_kernel, dim3((H + 63) / 64, (M + rpb - 1) / rpb), dim3(256), 0, s, g, x, dW, db, M,
                           H, F, transpose_out, rpb);
    return e3d_launch_status("e3d_small_k_wgrad");
}

extern "C" int e3d_head_linear_bwd_dx(const float* dout, const float* W, float* dx, int M, int H, int Nout,
                                      void* stream) {
    E3D_REQUIRE(dout && W && dx && M > 0 && Nout >= 1 && Nout <= 32, "head_linear_bwd_dx: bad arguments");
    DISPATCH_V(H, hipLaunchKernelGGL(head_linear_bwd_dx_kernel<V>, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                                     dout, W, dx, M, Nout));
    return e3d_launch_status("e3d_head_linear_bwd_dx");
}
