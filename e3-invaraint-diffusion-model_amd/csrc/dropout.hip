// Dropout for the training path (reference: nn.Dropout(p = 0.1) in BertEmbeddings, BertSelfOutput,
// BertOutput, the SELayer MLP -- structure_model/model.py:45-47,109-117 -- and on the attention
// probabilities in transformers 4.38.2 BertSelfAttention).  Decisions come from the counter-based
// generator of e3d_common.h, so the backward pass re-applies the same op to the gradient.
#include "e3d_common.h"

namespace {

__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, E3dDrop d_in, float* __restrict__ out,
                                                      int64_t n) {
    const E3dDrop d = e3d_drop_resolve(d_in);
    const int64_t n4 = n >> 2, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4 + 1; i += stride) {
        float m[4];
        e3d_drop_mult4(d, (uint64_t)i, m);
        if (i < n4) {
            f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] *= m[j];
            reinterpret_cast<f32x4*>(out)[i] = v;
        } else {
            for (int j = 0; 4 * n4 + j < n; ++j) out[4 * n4 + j] = x[4 * n4 + j] * m[j];
        }
    }
}

// multipliers of the attention-probability dropout, [B*nh, Lq, Lk] (test aid: the kernels never store them)
__global__ __launch_bounds__(256) void attn_drop_mask_kernel(E3dDrop d_in, float* __restrict__ out, int Lq, int Lk,
                                                             int64_t n_rows) {
    const E3dDrop d = e3d_drop_resolve(d_in);
    const int64_t row = blockIdx.x;   // (bh, q)
    if (row >= n_rows) return;
    const int bh = (int)(row / Lq), q = (int)(row % Lq);
    for (int key0 = 4 * threadIdx.x; key0 < Lk; key0 += 4 * blockDim.x) {
        float m[4];
        e3d_drop_mult4(d, e3d_attn_drop_idx4(bh, Lq, Lk, q, key0), m);
        for (int j = 0; j < 4 && key0 + j < Lk; ++j) out[row * Lk + key0 + j] = m[j];
    }
}

}  // namespace

extern "C" int e3d_dropout_f32(const float* x, float p, uint64_t seed, float* out, int64_t n, void* stream) {
    E3D_REQUIRE(x && out && n > 0, "dropout: bad arguments");
    E3D_REQUIRE(p >= 0.f && p < 1.f, "dropout: p=%g outside [0, 1)", (double)p);
    E3D_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)out % 16) == 0, "dropout: pointers must be 16B aligned");
    int64_t blocks = ((n >> 2) + 1 + 255) / 256;
    blocks = blocks > 8192 ? 8192 : blocks;
    hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, e3d_drop_make(p, seed),
                       out, n);
    return e3d_launch_status("e3d_dropout_f32");
}

extern "C" int e3d_attn_dropout_mask(int B, int nh, int Lq, int Lk, float p, uint64_t seed, float* out, void* stream) {
    E3D_REQUIRE(out && B > 0 && nh > 0 && Lq > 0 && Lk > 0, "attn_dropout_mask: bad arguments");
    E3D_REQUIRE(p >= 0.f && p < 1.f, "attn_dropout_mask: p=%g outside [0, 1)", (double)p);
    const int64_t rows = (int64_t)B * nh * Lq;
    E3D_REQUIRE(rows < (1ll << 31), "attn_dropout_mask: too many rows");
    hipLaunchKernelGGL(attn_drop_mask_kernel, dim3((unsigned)rows), dim3(64), 0, (hipStream_t)stream, e3d_drop_make(p, seed),
                       out, Lq, Lk, rows);
    return e3d_launch_status("e3d_attn_dropout_mask");
}
