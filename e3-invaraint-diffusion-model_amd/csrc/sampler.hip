// Diffusion-process kernels: DDPM ancestral update + angular wrap, forward noising, and the
// discrete (BLOSUM / uniform transition) posterior + categorical draw.  All HBM-bound.
#include "e3d_common.h"

namespace {

// modulo_with_wrapped_range(v, -pi, pi): ((v - (-pi)) % 2pi) + (-pi) with torch's floored
// remainder (fmod, then +b when the sign differs); constants rounded to fp32 like torch does
// for python scalars.  (structure_model/utils.py:20-40)
__device__ __forceinline__ float wrap_pi(float v) {
    const float pi_f = 3.14159265358979323846f;
    const float top = 6.28318530717958647692f;
    const float sft = v + pi_f;
    float r = fmodf(sft, top);
    if (r != 0.f && r < 0.f) r += top;
    return r - pi_f;
}

__global__ __launch_bounds__(256) void ddpm_step_wrap_kernel(
    const float* __restrict__ x, const float* __restrict__ eps_hat, const float* __restrict__ noise,
    float sra, float beta, float s1m, float sigma, const float* __restrict__ coef_table,
    const int64_t* __restrict__ t_dev, int wrap, float* __restrict__ out, int64_t n4, int64_t n) {
    if (coef_table) {   // graph-replayable form: the step index lives on the device, the coefficients in a [T,4] table
        const int64_t t = t_dev[0];
        sra = coef_table[4 * t];
        beta = coef_table[4 * t + 1];
        s1m = coef_table[4 * t + 2];
        sigma = coef_table[4 * t + 3];
        if (sigma == 0.f) noise = nullptr;   // t == 0: the mean, exactly as the scalar form
    }
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const f32x4 xv = reinterpret_cast<const f32x4*>(x)[i];
        const f32x4 ev = reinterpret_cast<const f32x4*>(eps_hat)[i];
        f32x4 nv = {0.f, 0.f, 0.f, 0.f};
        if (noise) nv = reinterpret_cast<const f32x4*>(noise)[i];
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float mean = sra * (xv[j] - beta * ev[j] / s1m);
            if (noise) mean = mean + sigma * nv[j];
            o[j] = wrap ? wrap_pi(mean) : mean;
        }
        reinterpret_cast<f32x4*>(out)[i] = o;
    }
    // tail (n % 4)
    const int64_t t = n4 * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) {
        float mean = sra * (x[t] - beta * eps_hat[t] / s1m);
        if (noise) mean = mean + sigma * noise[t];
        out[t] = wrap ? wrap_pi(mean) : mean;
    }
}

__global__ __launch_bounds__(256) void q_sample_wrap_kernel(
    const float* __restrict__ x0, const float* __restrict__ noise, const int64_t* __restrict__ t,
    const float* __restrict__ sqrt_ab, const float* __restrict__ sqrt_1mab, float* __restrict__ out,
    int64_t per, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int64_t ti = t[i / per];
        out[i] = wrap_pi(sqrt_ab[ti] * x0[i] + sqrt_1mab[ti] * noise[i]);
    }
}

constexpr int CMAX = 32;

// inverse-CDF categorical draw / argmax on a row of C probabilities held in registers
__device__ __forceinline__ int pick_class(const float (&p)[CMAX], int C, float total, int mode, float u) {
    int best = 0;
    if (mode == 0) {
        float bv = p[0];
#pragma unroll
        for (int c = 1; c < CMAX; ++c)
            if (c < C && p[c] > bv) { bv = p[c]; best = c; }
    } else {
        // index = #{c : cumsum[c] <= u * total}, clamped to C-1
        const float thr = u * total;
        float cum = 0.f;
        int cnt = 0;
#pragma unroll
        for (int c = 0; c < CMAX; ++c)
            if (c < C) { cum += p[c]; cnt += (cum <= thr) ? 1 : 0; }
        best = cnt < C - 1 ? cnt : C - 1;
    }
    return best;
}

// One workgroup per batch item: the three CxC matrices of the item live in LDS, each thread
// walks rows l = tid, tid+256, ...
__global__ __launch_bounds__(256) void discrete_posterior_kernel(
    const int32_t* __restrict__ xt_idx, const float* __restrict__ logits, const float* __restrict__ Qsb,
    const float* __restrict__ Qtb, const float* __restrict__ u, int mode, int32_t* __restrict__ out_idx,
    float* __restrict__ prob_out, int L, int C) {
    __shared__ float s_qsb[CMAX * CMAX], s_qtb[CMAX * CMAX], s_qt[CMAX * CMAX], s_rs[CMAX];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int CC = C * C;
    for (int i = tid; i < CC; i += blockDim.x) {
        s_qsb[i] = Qsb[(int64_t)b * CC + i];
        s_qtb[i] = Qtb[(int64_t)b * CC + i];
    }
    __syncthreads();
    // Qt = (Qsb/Qtb) / rowsum(Qsb/Qtb)   (sequence_model/sample.py:160)
    if (tid < C) {
        float rs = 0.f;
        for (int j = 0; j < C; ++j) rs += s_qsb[tid * C + j] / s_qtb[tid * C + j];
        s_rs[tid] = rs;
    }
    __syncthreads();
    for (int i = tid; i < CC; i += blockDim.x) s_qt[i] = (s_qsb[i] / s_qtb[i]) / s_rs[i / C];
    __syncthreads();

    for (int l = tid; l < L; l += blockDim.x) {
        const int64_t n = (int64_t)b * L + l;
        const int xt = xt_idx[n];
        const float* lg = logits + n * C;
        float pred[CMAX], prob[CMAX];
        float mx = -INFINITY;
#pragma unroll
        for (int c = 0; c < CMAX; ++c) {
            pred[c] = c < C ? lg[c] : -INFINITY;
            mx = fmaxf(mx, pred[c]);
        }
        float se = 0.f;
#pragma unroll
        for (int c = 0; c < CMAX; ++c) {
            pred[c] = c < C ? expf(pred[c] - mx) : 0.f;
            se += pred[c];
        }
#pragma unroll
        for (int c = 0; c < CMAX; ++c) { pred[c] = pred[c] / se; prob[c] = 0.f; }
        // prob[c] = sum_x0 pred[x0] * (Qt[c][xt] * Qsb[x0][c]) / Qtb[x0][xt]   (sample.py:129-139,162-165)
#pragma unroll 1
        for (int x0 = 0; x0 < C; ++x0) {
            float den = s_qtb[x0 * C + xt];
            if (den == 0.f) den = 1e-6f;
            const float w = pred[x0];
#pragma unroll
            for (int c = 0; c < CMAX; ++c)
                if (c < C) prob[c] += w * ((s_qt[c * C + xt] * s_qsb[x0 * C + c]) / den);
        }
        float tot = 0.f;
#pragma unroll
        for (int c = 0; c < CMAX; ++c)
            if (c < C) tot += prob[c];
        if (tot == 0.f) {  // sample.py:166
#pragma unroll
            for (int c = 0; c < CMAX; ++c) prob[c] = c < C ? 1e-5f : 0.f;
            tot = 0.f;
#pragma unroll
            for (int c = 0; c < CMAX; ++c)
                if (c < C) tot += prob[c];
        }
        float ntot = 0.f;
#pragma unroll
        for (int c = 0; c < CMAX; ++c) {
            prob[c] = c < C ? prob[c] / tot : 0.f;
            ntot += prob[c];
        }
        if (prob_out) {
#pragma unroll
            for (int c = 0; c < CMAX; ++c)
                if (c < C) prob_out[n * C + c] = prob[c];
        }
        out_idx[n] = pick_class(prob, C, ntot, mode, u ? u[n] : 0.f);
    }
}

__global__ __launch_bounds__(256) void discrete_q_sample_kernel(
    const int32_t* __restrict__ x0_idx, const float* __restrict__ Qtb, const float* __restrict__ u,
    int mode, int32_t* __restrict__ out_idx, int L, int C, int64_t n_rows) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_rows) return;
    const int x0 = x0_idx[n];
    if (x0 < 0) { out_idx[n] = 0; return; }  // all-zero (padding) row -> class 0, model.py:305-308
    const float* q = Qtb + (n / L) * C * C;
    float p[CMAX];
    float tot = 0.f;
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {
        p[c] = c < C ? q[c * C + x0] : 0.f;  // (Qtb @ onehot)[c] = Qtb[c][x0]
        tot += p[c];
    }
    out_idx[n] = pick_class(p, C, tot, mode, u ? u[n] : 0.f);
}

}  // namespace

extern "C" int e3d_ddpm_step_wrap(const float* x, const float* eps_hat, const float* noise,
                                  float sqrt_recip_alpha, float beta, float sqrt_one_minus_ab,
                                  float sigma, int wrap, float* out, int64_t n, void* stream) {
    E3D_REQUIRE(x && eps_hat && out && n > 0, "ddpm_step_wrap: bad arguments");
    E3D_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)eps_hat % 16) == 0 && ((uintptr_t)out % 16) == 0 &&
                    ((uintptr_t)noise % 16) == 0, "ddpm_step_wrap: pointers must be 16B aligned");
    const int64_t n4 = n / 4;
    int64_t blocks = (n4 + 255) / 256;
    blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
    hipLaunchKernelGGL(ddpm_step_wrap_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, eps_hat,
                       sigma != 0.f ? noise : nullptr, sqrt_recip_alpha, beta, sqrt_one_minus_ab, sigma, nullptr, nullptr, wrap,
                       out, n4, n);
    return e3d_launch_status("e3d_ddpm_step_wrap");
}

extern "C" int e3d_ddpm_step_wrap_table(const float* x, const float* eps_hat, const float* noise,
                                        const float* coef_table, const int64_t* t_dev, int wrap, float* out,
                                        int64_t n, void* stream) {
    E3D_REQUIRE(x && eps_hat && noise && coef_table && t_dev && out && n > 0, "ddpm_step_wrap_table: bad arguments");
    E3D_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)eps_hat % 16) == 0 && ((uintptr_t)out % 16) == 0 &&
                    ((uintptr_t)noise % 16) == 0, "ddpm_step_wrap_table: pointers must be 16B aligned");
    const int64_t n4 = n / 4;
    int64_t blocks = (n4 + 255) / 256;
    blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
    hipLaunchKernelGGL(ddpm_step_wrap_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, eps_hat, noise,
                       0.f, 0.f, 1.f, 0.f, coef_table, t_dev, wrap, out, n4, n);
    return e3d_launch_status("e3d_ddpm_step_wrap_table");
}

extern "C" int e3d_q_sample_wrap(const float* x0, const float* noise, const int64_t* t,
                                 const float* sqrt_ab, const float* sqrt_1mab, float* out, int B,
                                 int64_t per, void* stream) {
    E3D_REQUIRE(x0 && noise && t && sqrt_ab && sqrt_1mab && out && B > 0 && per > 0, "q_sample_wrap: bad arguments");
    const int64_t n = (int64_t)B * per;
    int64_t blocks = (n + 255) / 256;
    blocks = blocks > 2048 ? 2048 : blocks;
    hipLaunchKernelGGL(q_sample_wrap_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x0, noise, t,
                       sqrt_ab, sqrt_1mab, out, per, n);
    return e3d_launch_status("e3d_q_sample_wrap");
}

extern "C" int e3d_discrete_posterior_sample(const int32_t* xt_idx, const float* logits,
                                             const float* Qsb, const float* Qtb, const float* u,
                                             int mode, int32_t* out_idx, float* prob_out, int B,
                                             int L, int C, void* stream) {
    E3D_REQUIRE(xt_idx && logits && Qsb && Qtb && out_idx && B > 0 && L > 0, "discrete_posterior: bad arguments");
    E3D_REQUIRE(C >= 2 && C <= CMAX, "discrete_posterior: C must be in [2,%d] (C=%d)", CMAX, C);
    E3D_REQUIRE(mode == 0 || (mode == 1 && u), "discrete_posterior: mode 1 needs uniforms");
    hipLaunchKernelGGL(discrete_posterior_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, xt_idx, logits, Qsb, Qtb,
                       u, mode, out_idx, prob_out, L, C);
    return e3d_launch_status("e3d_discrete_posterior_sample");
}

extern "C" int e3d_discrete_q_sample(const int32_t* x0_idx, const float* Qtb, const float* u, int mode,
                                     int32_t* out_idx, int B, int L, int C, void* stream) {
    E3D_REQUIRE(x0_idx && Qtb && out_idx && B > 0 && L > 0, "discrete_q_sample: bad arguments");
    E3D_REQUIRE(C >= 2 && C <= CMAX, "discrete_q_sample: C must be in [2,%d] (C=%d)", CMAX, C);
    E3D_REQUIRE(mode == 0 || (mode == 1 && u), "discrete_q_sample: mode 1 needs uniforms");
    const int64_t n = (int64_t)B * L;
    hipLaunchKernelGGL(discrete_q_sample_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, x0_idx, Qtb, u, mode, out_idx, L, C, n);
    return e3d_launch_status("e3d_discrete_q_sample");
}
