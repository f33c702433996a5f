// Shared host/device helpers for the gfx950 kernels.  Wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <math.h>
#include <atomic>

#include "../../include/e3d_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

void e3d_set_error(const char* fmt, ...);

// device scalars bounding |Q|, |K| and |distance table| elements of an attention call (e3d_mask_skip_is_exact below)
struct E3dBounds { const float* q; const float* k; const float* e; };


#define E3D_REQUIRE(cond, ...)       \
    do {                             \
        if (!(cond)) {               \
            e3d_set_error(__VA_ARGS__); \
            return -1;               \
        }                            \
    } while (0)

// One-time launch setup PER DEVICE (dynamic-LDS opt-in of a kernel, CU count): function attributes and device
// properties belong to the device that is current when they are set / read, so a process that drives several GPUs
// (or several host threads) must not share one process-wide flag.  ``mask`` holds one bit per device ordinal;
// ``f(device)`` is idempotent, so two threads racing through the first launch is harmless.
static inline int e3d_current_device() {
    int d = 0;
    (void)hipGetDevice(&d);
    return d;
}
template <typename F>
static inline void e3d_once_per_device(std::atomic<uint64_t>& mask, F&& f) {
    const int d = e3d_current_device();
    const uint64_t bit = 1ull << (d & 63);
    if (!(mask.load(std::memory_order_acquire) & bit)) {
        f(d);
        mask.fetch_or(bit, std::memory_order_release);
    }
}
// CUs of the current device, rounded down to whole XCD rounds (tile t then runs on XCD t % 8, as xcd_remap assumes)
static inline int e3d_cu_count() {
    static std::atomic<int> n_cu[64];
    const int d = e3d_current_device() & 63;
    int n = n_cu[d].load(std::memory_order_relaxed);
    if (!n) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || n <= 0) n = 256;
        n -= n % 8;
        if (n <= 0) n = 8;
        n_cu[d].store(n, std::memory_order_relaxed);
    }
    return n;
}
template <typename K>
static inline void e3d_allow_lds(std::atomic<uint64_t>& mask, K kernel, size_t lds_bytes) {
    e3d_once_per_device(mask, [&](int) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds_bytes);
    });
}

// Zero-fill by a kernel launch.  hipMemsetAsync nodes inside a captured graph did not clear their destinations reliably on
// this stack (tools/lab/graph_stale_pointer_hunt.py: gradients accumulated by atomics on top of whatever the buffer held);
// a kernel node always runs where it was captured.
static __global__ void e3d_zero_kernel(float* __restrict__ p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0.f;
}
static inline hipError_t e3d_zero_async(float* p, size_t n_floats, hipStream_t s) {
    if (n_floats == 0) return hipSuccess;
    const size_t blocks = (n_floats + 1023) / 1024;
    hipLaunchKernelGGL(e3d_zero_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, s, p, n_floats);
    return hipGetLastError();
}
// rows x cols floats of a matrix with row stride ld (floats)
static __global__ void e3d_zero2d_kernel(float* __restrict__ p, size_t ld, size_t rows, size_t cols) {
    const size_t n = rows * cols;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[(i / cols) * ld + i % cols] = 0.f;
}

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
    // x * 0.5 * (1 + erf(x / sqrt(2)))  -- ATen's exact (erf) GELU, evaluated branch-free as
    //     relu(x) - 0.5 |x| erfc(|x| / sqrt(2)),   erfc(z) = t (a1 + t (a2 + t (a3 + t (a4 + t a5)))) exp(-z^2),  t = 1 / (1 + p z)
    // (Abramowitz & Stegun 7.1.26, |error| <= 1.5e-7 on erfc, i.e. <= 0.75e-7 |x| on the result: below fp32
    // rounding of the GEMM sum it is applied to).  libm's erff costs ~30 VALU instructions per element once both of its
    // range branches run in one wave -- 128 elements per lane in the 256x256 epilogue; this form is 14 with two
    // transcendentals, and has no cancellation on the negative side (0.5 x erfc is formed directly).
    const float ax = fabsf(x), z = ax * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float erfc_z = p * t * __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);
    return fmaxf(x, 0.f) - 0.5f * ax * erfc_z;
}

__device__ __forceinline__ float silu(float x) { return x / (1.0f + expf(-x)); }

// fp16 split of two fp32 values in 4 instructions: hi = v_cvt_pk_f16_f32 (RNE), the residuals by v_fma_mix_f32 reading
// the packed halves directly (x - float(hi), exact), lo = v_cvt_pk_f16_f32.  The compiler's own lowering of the same
// C expression takes 7 (separate conversions back to fp32); with the bf16 form at 6 this makes the f16x3 staging the
// cheaper one.  hi is formed once and the residual reads that very register, so the two terms always complement
// each other (no second, differently rounded evaluation of a fused producer -- see attn_relkey_coop.hip).
typedef _Float16 e3d_f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void e3d_split2_f16(float x0, float x1, e3d_f16x2& hi, e3d_f16x2& lo) {
    e3d_f16x2 p;
    p[0] = (_Float16)x0;
    p[1] = (_Float16)x1;
    float r0, r1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(p), "v"(x0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(p), "v"(x1));
    hi = p;
    lo[0] = (_Float16)r0;
    lo[1] = (_Float16)r1;
}


__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---------------------------------------------------------------- running |x| maximum of a kernel's outputs
// GEMM epilogues can report the largest |out| they wrote (``absmax`` arguments of the *_ex entry points) so that the
// attention kernels can PROVE that a padded key cannot reach the softmax of a valid one before they skip all-padding key
// tiles: the reference masks additively with -10000 (structure_model/model.py:226-231), which only silences a key
// while the score spread stays below ~9900.  Kept as the bit pattern of |x| in an unsigned: the integer order equals
// the float order for finite values and puts inf / NaN above everything (a NaN anywhere then reads as "no bound").
// The target is only ever raised (atomic max): a value left over from an earlier, larger launch is a valid bound too.
__device__ __forceinline__ void e3d_absmax_accum(unsigned& m, float v) {
    const unsigned b = __float_as_uint(v) & 0x7fffffffu;
    m = m > b ? m : b;
}
__device__ __forceinline__ void e3d_absmax_commit(unsigned m, float* target, int lane) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned t = (unsigned)__shfl_xor((int)m, o, 64);
        m = m > t ? m : t;
    }
    // thousands of waves raise ONE address: same-address atomics serialise in L2 (~3 us on a 9-us small-M launch), so a
    // wave first reads the current value (device-scope load, served by L2) and only joins the queue when it would raise it
    // -- the target only ever grows, so a stale read can cost an unnecessary atomic, never a missed maximum
    if (lane == 0) {
        unsigned* t = reinterpret_cast<unsigned*>(target);
        if (m > __hip_atomic_load(t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(t, m);
    }
}
// May all-padding key tiles be skipped?  Scores are s = (q.k + q.e) / 8 with |q.k| <= 64 qa ka and |q.e| <= 64 qa ea
// (qa, ka, ea: largest |element| of Q, K and the distance table), so any two scores of a row differ by at most
// 16 qa (ka + ea); a padded key contributes exp(s_pad - 10000 - m) with m >= the row's valid scores, which is 0.0f --
// also without flush-to-zero, exp(-104) < 2^-149 -- while 16 qa (ka + ea) < 10000 - 104.  Missing bounds (null) or a
// NaN / inf bound: no skipping, the full sweep is always exact.
__device__ __forceinline__ bool e3d_mask_skip_is_exact(const E3dBounds b) {
    if (!b.q || !b.k) return false;
    const float qa = *b.q, ka = *b.k, ea = b.e ? *b.e : 0.f;
    return 16.0f * qa * (ka + ea) < 9890.0f;   // false for NaN
}

// ---------------------------------------------------------------- dropout decisions
// Counter-based: one splitmix64 hash per group of 4 consecutive elements, one 16-bit field per
// element; an element is KEPT iff its field >= thr, kept values are scaled by 65536 / (65536 - thr)
// (the exact inverse keep probability of this generator, so the op is unbiased).  Forward and
// backward kernels regenerate the same decisions from (seed, element index): no mask is stored.
struct E3dDrop {
    uint64_t seed;
    uint32_t thr;
    float scale;
    const uint64_t* epoch;   // device word added to the seed inside the kernel (may be null): see e3d_dropout_set_epoch_ptr
};
// The word registered for the calling thread's current device (capi.hip), or null.  A captured HIP graph bakes the
// ``seed`` argument of every dropout launch; a training step replayed from a graph advances this word instead, so every
// replay draws fresh decisions while forward and backward of one step still agree.
const uint64_t* e3d_dropout_epoch_ptr();
static inline E3dDrop e3d_drop_make(float p, uint64_t seed) {
    E3dDrop d;
    long t = lrintf(p * 65536.0f);
    t = t < 0 ? 0 : (t > 65535 ? 65535 : t);
    d.seed = seed;
    d.epoch = e3d_dropout_epoch_ptr();
    d.thr = (uint32_t)t;
    d.scale = 65536.0f / (float)(65536 - t);
    return d;
}
__device__ __forceinline__ E3dDrop e3d_drop_resolve(E3dDrop d) {   // once, at kernel entry
    if (d.epoch) d.seed += *d.epoch * 0xD1342543DE82EF95ull;
    d.epoch = nullptr;
    return d;
}
__device__ __forceinline__ void e3d_drop_mult4(const E3dDrop d, uint64_t idx4, float (&m)[4]) {
    uint64_t z = idx4 + d.seed * 0x9E3779B97F4A7C15ull + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
#pragma unroll
    for (int j = 0; j < 4; ++j) m[j] = ((uint32_t)(z >> (16 * j)) & 0xFFFFu) >= d.thr ? d.scale : 0.f;
}
// attention probabilities P[b, h, q, key]: group index of keys key0 .. key0+3 (key0 % 4 == 0)
__device__ __forceinline__ uint64_t e3d_attn_drop_idx4(int bh, int Lq, int Lk, int q, int key0) {
    return ((uint64_t)bh * Lq + q) * (uint64_t)((Lk + 3) >> 2) + (uint64_t)(key0 >> 2);
}

int e3d_attn_fill_planes(const float* dist_emb, int P, int Lk, void* scratch, int f16, float* e_absmax, hipStream_t s);

// attn_relkey_coop.hip: workgroup-cooperative bf16x3 / f16x3 attention forward (internal; arguments validated by the
// caller); ``dropping``: dropout on the probabilities (bf16x3 only)
int e3d_attn_coop_launch(const float* q, int64_t q_bs, int64_t q_rs, const float* k, int64_t k_bs, int64_t k_rs,
                         const float* v, int64_t v_bs, int64_t v_rs, const float* dist_emb, int P,
                         const float* key_mask, float* out, float* lse, int B, int nh, int Lq, int Lk, int skip,
                         E3dBounds bnd, void* e_scratch, int e_ready, int f16, E3dDrop drop, bool dropping, hipStream_t s);

// attn_bwd_split.hip: launches A and B of the attention backward in bf16x3 arithmetic (internal)
int e3d_attn_bwd_split_launch(const float* q, int64_t q_bs, int64_t q_rs, const float* k, int64_t k_bs, int64_t k_rs,
                              const float* v, int64_t v_bs, int64_t v_rs, const float* dist_emb, int P,
                              const float* key_mask, const float* out, const float* lse, const float* dout, float* dq,
                              int64_t dq_bs, int64_t dq_rs, float* dk, int64_t dk_bs, int64_t dk_rs, float* dv,
                              int64_t dv_bs, int64_t dv_rs, float* Pm, float* dSm, float* part, int B, int nh, int Lq,
                              int Lk, E3dDrop drop, bool dropping, hipStream_t s);

// attn_bwd_coop.hip: the fused, recomputing backward (bf16x3, Lq, Lk <= 128); internal
int64_t e3d_attn_bwd_coop_scratch_bytes(int Lk);
int64_t e3d_attn_bwd_coop_de_floats(int Lq, int Lk);
bool e3d_attn_bwd_coop_supported(int Lq, int Lk, bool dropping);
int e3d_attn_bwd_coop_launch(const float* q, int64_t q_bs, int64_t q_rs, const float* k, int64_t k_bs, int64_t k_rs,
                             const float* v, int64_t v_bs, int64_t v_rs, const float* dist_emb, int P,
                             const float* key_mask, const float* out, const float* lse, const float* dout, float* dq,
                             int64_t dq_bs, int64_t dq_rs, float* dk, int64_t dk_bs, int64_t dk_rs, float* dv,
                             int64_t dv_bs, int64_t dv_rs, float* d_dist_emb, void* e_scratch, float* part, int B, int nh,
                             int Lq, int Lk, E3dDrop drop, bool dropping, hipStream_t s);

// XCD-aware remap (cdna_hip_programming.md T1, bijective form): consecutive logical ids
// land on one XCD so that workgroups sharing operands share an L2.
__device__ __forceinline__ int xcd_remap(int id, int n) {
    const int q = n >> 3, r = n & 7, xcd = id & 7, slot = id >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}
