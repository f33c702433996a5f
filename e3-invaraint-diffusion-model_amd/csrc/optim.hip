// Global-norm gradient clip + AdamW over ALL parameters of a model in three launches (squared-norm partials, their
// ordered sum, the update) -- what Lightning's ``gradient_clip_val`` + torch.optim.AdamW do around the reference's
// training_step (structure_model/train_model.py:99-110, structure_model/model.py:361-366) in ~100 multi-tensor launches
// (a norm pass, a multiply pass that rewrites every gradient, the update: 1.45 ms for 146 M parameters).  Here the
// gradients are read twice and never written (the clip coefficient is a device scalar the update multiplies in), every
// tensor is reached through pointer tables in device memory (no 4-KB kernel-argument limit: one launch whatever the
// number of parameters), and nothing synchronises with the host.  HBM-bound: 4 B (norm) + 28 B (update) per parameter.
//
// Work unit = a CHUNK of 8192 consecutive elements of one tensor (chunk -> (tensor, first element) map built once by
// the host); 256 threads, float4 accesses when all four pointers of the tensor are 16-byte aligned.
#include "e3d_common.h"

namespace {

constexpr int CHUNK = 8192, NT = 256;

__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) red[wid] = v;
    __syncthreads();
    float t = 0.f;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 0; w < NT / 64; ++w) t += red[w];
    }
    return t;   // valid in thread 0
}

__global__ __launch_bounds__(NT) void sqnorm_partial_kernel(const float* const* __restrict__ grads, const int64_t* __restrict__ numel,
                                                            const int* __restrict__ chunk_tensor,
                                                            const int64_t* __restrict__ chunk_first, float* __restrict__ partial) {
    __shared__ float red[NT / 64];
    const int t = chunk_tensor[blockIdx.x];
    const int64_t first = chunk_first[blockIdx.x];
    const float* g = grads[t] + first;
    const int n = (int)min((int64_t)CHUNK, numel[t] - first);
    float s0 = 0.f, s1 = 0.f;
    if (((uintptr_t)g & 15) == 0) {
        const int n4 = n >> 2;
        for (int i = threadIdx.x; i < n4; i += NT) {
            const f32x4 v = reinterpret_cast<const f32x4*>(g)[i];
            s0 = fmaf(v[0], v[0], s0); s1 = fmaf(v[1], v[1], s1);
            s0 = fmaf(v[2], v[2], s0); s1 = fmaf(v[3], v[3], s1);
        }
        for (int i = 4 * n4 + threadIdx.x; i < n; i += NT) s0 = fmaf(g[i], g[i], s0);
    } else {
        for (int i = threadIdx.x; i < n; i += NT) s0 = fmaf(g[i], g[i], s0);
    }
    const float tot = block_sum(s0 + s1, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

// out[0] = total norm, out[1] = clip coefficient min(max_norm / (norm + 1e-6), 1) -- NaN / inf propagate as in
// torch.nn.utils.clip_grad_norm_(error_if_nonfinite=False).  One block, double accumulation, fixed order.
__global__ __launch_bounds__(NT) void sqnorm_final_kernel(const float* __restrict__ partial, int n, float max_norm, float* __restrict__ out) {
    __shared__ double red[NT];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += NT) s += (double)partial[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = NT / 2; w > 0; w >>= 1) {
        if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float norm = (float)sqrt(red[0]);
        const float r = max_norm / (norm + 1e-6f);
        out[0] = norm;
        out[1] = (r < 1.0f || r != r) ? r : 1.0f;
    }
}

struct AdamScalars {
    float lr, beta1, beta2, eps, weight_decay, bc1, bc2_sqrt;
};

__device__ __forceinline__ void adamw_one(float& p, float g, float& m, float& v, const AdamScalars& a) {
    // torch's fused AdamW (torch/csrc .. fused_adam_utils.cuh, ADAMW mode), in the same order of operations
    p -= a.lr * a.weight_decay * p;
    m = m + (g - m) * (1.0f - a.beta1);                       // lerp(exp_avg, grad, 1 - beta1)
    v = a.beta2 * v + (1.0f - a.beta2) * g * g;
    const float step_size = a.lr / a.bc1;
    const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
    p -= step_size * m / denom;
}

__global__ __launch_bounds__(NT) void adamw_kernel(float* const* __restrict__ params, const float* const* __restrict__ grads,
                                                   float* const* __restrict__ exp_avg, float* const* __restrict__ exp_avg_sq,
                                                   const int64_t* __restrict__ numel, const int* __restrict__ chunk_tensor,
                                                   const int64_t* __restrict__ chunk_first, const float* __restrict__ clip,
                                                   AdamScalars a, const float* __restrict__ dyn) {
    if (dyn) {   // learning rate and step count from device memory (a captured graph cannot carry them as arguments)
        a.lr = dyn[0];
        if (a.beta1 < 0.f) {   // e3d_adamw_step_dev: the other hyper-parameters too (OneCycleLR cycles beta1 every step)
            a.beta1 = dyn[2];
            a.beta2 = dyn[3];
            a.eps = dyn[4];
            a.weight_decay = dyn[5];
        }
        const double step = (double)dyn[1] + 1.0;
        a.bc1 = (float)(1.0 - pow((double)a.beta1, step));
        a.bc2_sqrt = (float)sqrt(1.0 - pow((double)a.beta2, step));
    }
    const int t = chunk_tensor[blockIdx.x];
    const int64_t first = chunk_first[blockIdx.x];
    float* p = params[t] + first;
    const float* g = grads[t] + first;
    float* m = exp_avg[t] + first;
    float* v = exp_avg_sq[t] + first;
    const int n = (int)min((int64_t)CHUNK, numel[t] - first);
    const float coef = clip ? clip[1] : 1.0f;
    const bool vec = (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0;
    const int n4 = vec ? n >> 2 : 0;
    for (int i = threadIdx.x; i < n4; i += NT) {
        f32x4 pp = reinterpret_cast<f32x4*>(p)[i];
        const f32x4 gg = reinterpret_cast<const f32x4*>(g)[i];
        f32x4 mm = reinterpret_cast<f32x4*>(m)[i], vv = reinterpret_cast<f32x4*>(v)[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float pj = pp[j], mj = mm[j], vj = vv[j];
            adamw_one(pj, gg[j] * coef, mj, vj, a);
            pp[j] = pj; mm[j] = mj; vv[j] = vj;
        }
        reinterpret_cast<f32x4*>(p)[i] = pp;
        reinterpret_cast<f32x4*>(m)[i] = mm;
        reinterpret_cast<f32x4*>(v)[i] = vv;
    }
    for (int i = 4 * n4 + threadIdx.x; i < n; i += NT) adamw_one(p[i], g[i] * coef, m[i], v[i], a);
}

}  // namespace

extern "C" int e3d_optim_chunk_elems(void) { return CHUNK; }

extern "C" int e3d_grad_global_norm(const float* const* grads, const int64_t* numel, const int* chunk_tensor,
                                    const int64_t* chunk_first, int n_chunks, float max_norm, float* partial, float* norm_and_clip,
                                    void* stream) {
    E3D_REQUIRE(grads && numel && chunk_tensor && chunk_first && partial && norm_and_clip, "grad_global_norm: null pointer");
    E3D_REQUIRE(n_chunks > 0, "grad_global_norm: no chunks");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(sqnorm_partial_kernel, dim3(n_chunks), dim3(NT), 0, s, grads, numel, chunk_tensor, chunk_first, partial);
    hipLaunchKernelGGL(sqnorm_final_kernel, dim3(1), dim3(NT), 0, s, partial, n_chunks, max_norm, norm_and_clip);
    return e3d_launch_status("e3d_grad_global_norm");
}

extern "C" int e3d_adamw_step(float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                              const int64_t* numel, const int* chunk_tensor, const int64_t* chunk_first, int n_chunks,
                              const float* norm_and_clip, float lr, float beta1, float beta2, float eps, float weight_decay,
                              int step, void* stream) {
    E3D_REQUIRE(params && grads && exp_avg && exp_avg_sq && numel && chunk_tensor && chunk_first, "adamw_step: null pointer");
    E3D_REQUIRE(n_chunks > 0 && step >= 1, "adamw_step: bad n_chunks=%d / step=%d", n_chunks, step);
    AdamScalars a;
    a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.weight_decay = weight_decay;
    // bias corrections in double on the host, as torch computes them from the step count
    a.bc1 = (float)(1.0 - pow((double)beta1, (double)step));
    a.bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
    hipLaunchKernelGGL(adamw_kernel, dim3(n_chunks), dim3(NT), 0, (hipStream_t)stream, params, grads, exp_avg, exp_avg_sq, numel,
                       chunk_tensor, chunk_first, norm_and_clip, a, (const float*)nullptr);
    return e3d_launch_status("e3d_adamw_step");
}

extern "C" int e3d_adamw_step_dyn(float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                                  const int64_t* numel, const int* chunk_tensor, const int64_t* chunk_first, int n_chunks,
                                  const float* norm_and_clip, const float* lr_and_step, float beta1, float beta2, float eps,
                                  float weight_decay, void* stream) {
    E3D_REQUIRE(params && grads && exp_avg && exp_avg_sq && numel && chunk_tensor && chunk_first && lr_and_step,
                "adamw_step_dyn: null pointer");
    E3D_REQUIRE(n_chunks > 0, "adamw_step_dyn: no chunks");
    AdamScalars a;
    a.lr = 0.f; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.weight_decay = weight_decay; a.bc1 = 1.f; a.bc2_sqrt = 1.f;
    hipLaunchKernelGGL(adamw_kernel, dim3(n_chunks), dim3(NT), 0, (hipStream_t)stream, params, grads, exp_avg, exp_avg_sq, numel,
                       chunk_tensor, chunk_first, norm_and_clip, a, lr_and_step);
    return e3d_launch_status("e3d_adamw_step_dyn");
}

extern "C" int e3d_adamw_step_dev(float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                                  const int64_t* numel, const int* chunk_tensor, const int64_t* chunk_first, int n_chunks,
                                  const float* norm_and_clip, const float* hyper, void* stream) {
    E3D_REQUIRE(params && grads && exp_avg && exp_avg_sq && numel && chunk_tensor && chunk_first && hyper,
                "adamw_step_dev: null pointer");
    E3D_REQUIRE(n_chunks > 0, "adamw_step_dev: no chunks");
    AdamScalars a;
    a.lr = 0.f; a.beta1 = -1.f; a.beta2 = 0.f; a.eps = 0.f; a.weight_decay = 0.f; a.bc1 = 1.f; a.bc2_sqrt = 1.f;
    hipLaunchKernelGGL(adamw_kernel, dim3(n_chunks), dim3(NT), 0, (hipStream_t)stream, params, grads, exp_avg, exp_avg_sq, numel,
                       chunk_tensor, chunk_first, norm_and_clip, a, hyper);
    return e3d_launch_status("e3d_adamw_step_dev");
}
