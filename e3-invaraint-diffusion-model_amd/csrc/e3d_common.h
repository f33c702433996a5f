// Shared host/device helpers for the gfx950 kernels.  Wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/e3d_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

void e3d_set_error(const char* fmt, ...);

// attn_relkey_coop.hip: workgroup-cooperative bf16x3 attention forward (internal; arguments validated by the caller)
int e3d_attn_coop_launch(const float* q, int64_t q_bs, int64_t q_rs, const float* k, int64_t k_bs, int64_t k_rs,
                         const float* v, int64_t v_bs, int64_t v_rs, const float* dist_emb, int P,
                         const float* key_mask, float* out, float* lse, int B, int nh, int Lq, int Lk, int skip,
                         hipStream_t s);

#define E3D_REQUIRE(cond, ...)       \
    do {                             \
        if (!(cond)) {               \
            e3d_set_error(__VA_ARGS__); \
            return -1;               \
        }                            \
    } while (0)

static inline int e3d_launch_status(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        e3d_set_error("%s: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

// Row index inside a 32x32 MFMA accumulator: reg in [0,16), half = lane >> 5.
__device__ __forceinline__ int mfma32_row(int reg, int half) {
    return (reg & 3) + 8 * (reg >> 2) + 4 * half;
}

__device__ __forceinline__ float gelu_erf(float x) {
    // x * 0.5 * (1 + erf(x / sqrt(2)))  -- ATen's exact GELU
    return x * 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
}

__device__ __forceinline__ float silu(float x) { return x / (1.0f + expf(-x)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// XCD-aware remap (cdna_hip_programming.md T1, bijective form): consecutive logical ids
// land on one XCD so that workgroups sharing operands share an L2.
__device__ __forceinline__ int xcd_remap(int id, int n) {
    const int q = n >> 3, r = n & 7, xcd = id & 7, slot = id >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}
