// Lab prototype (not product), TWO INDEPENDENT 4-WAVE WORKGROUPS PER CU: 128x256 output tiles (each wave 128x64 as in the
// product kernel), one wave per SIMD and workgroup, so the two waves of a SIMD belong to different workgroups whose barriers and
// DMA waits are independent -- one workgroup's stalls are the other's MFMA time.  Ring of THREE k16 half-tile buffers (24 KB each:
// two workgroups = 144 KB), DMA issued two half-tiles ahead, fragments read at the start of their step.
// Derived from gemm_planes_glds4.hip: the 256x256 persistent split GEMM with BOTH operands handed over as pre-split 16-bit planes
// and staged by LDS-DMA (global_load_lds) through a ring of FOUR half-tile buffers (k16 each):
//   * per k16 step a wave issues its 4 one-KB DMA pieces of half-tile h+3, reads the fragments of half-tile h+1 (landed
//     and barriered one step earlier) and runs the 24 MFMAs of half-tile h from registers;
//   * one counted wait (vmcnt(4): only the newest half-tile may still be in flight) and ONE raw s_barrier per k16 step --
//     never vmcnt(0), never __syncthreads() inside the loop (it would drain the DMA queue);
//   * no staging registers, no split arithmetic, no ds_write.
// Operand layout (the bytes of an fp32 matrix): per row and k16 block, 16 hi terms then 16 lo terms (64 contiguous bytes).  out[M,N] = A . W^T (no bias / activation), three cross products hi*lo + lo*hi + hi*hi, fp32 accumulate.
//   build:  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Ie3-invaraint-diffusion-model_amd/csrc -Iinclude \
//               tools/lab/gemm_planes_glds4.hip -o lab_build/libgemm_glds4.so
//   run:    python tools/lab/gemm_glds4_ab.py
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <typename E> struct V8;
template <> struct V8<_Float16> { typedef f16x8 t; };
template <> struct V8<__bf16> { typedef bf16x8 t; };
__device__ __forceinline__ f32x16 mma16(const bf16x8 a, const bf16x8 b, const f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mma16(const f16x8 a, const f16x8 b, const f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int xcd_remap(int id, int n) {
    const int q = n >> 3, r = n & 7, xcd = id & 7, slot = id >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}

constexpr int BT = 256, BMT = 128;
constexpr int A_B = BMT * 64, B_B = BT * 64;   // operands of a half-tile: 128 / 256 rows x (16 hi + 16 lo terms) x 2 B
constexpr int HB = A_B + B_B;                  // half-tile buffer: A rows, B rows = 24 KB
#undef LAB_NBUF
#define LAB_NBUF 3
constexpr int NBUF = LAB_NBUF;            // 4: barrier every k16 step; 5 (= all 160 KB): issue distance 4, barrier every other step

// LDS rows are 64 bytes: [hi terms 0-7 | hi 8-15 | lo 0-7 | lo 8-15] with the four 16-byte slots XOR-swizzled by
// (row >> 2) & 3 (the product kernel's rule: the 16 rows of a ds_read_b128 lane group cover the 64 banks once)
__device__ __forceinline__ int frag_off(int row, int plane, int half) {
    return row * 64 + (((plane * 2 + half) ^ ((row >> 2) & 3)) << 4);
}

template <typename E>
__global__ __launch_bounds__(256, 2) void gemm_planes_glds4_kernel(const float* __restrict__ A, const float* __restrict__ W,
                                                                   float* __restrict__ out, int N, int K, int tiles_m,
                                                                   int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef typename V8<E>::t X8;
    const int total = tiles_m * tiles_n, nk = K / 32, n_half = 2 * nk;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wc = wid;
    const int l31 = lane & 31, half = lane >> 5;

    // DMA pieces of this wave: 4 per half-tile, 1 KB = 16 rows x 64 bytes (the row's hi and lo terms of one k16 block are
    // 64 contiguous bytes in memory: operand layout [row][k16 block][plane][16 terms])
    constexpr int NP = 6;      // pieces per wave and half-tile: 24 KB / 4 waves / 1 KB
    unsigned goff[NP];
    int dst[NP];
    bool is_a[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int piece = wid * NP + i;                                   // 0..7: A rows, 8..23: B rows
        const bool a = piece < 8;
        const int prow = a ? piece : piece - 8;
        const int row = prow * 16 + (lane >> 2), slot = lane & 3;
        const int src_slot = slot ^ ((row >> 2) & 3);
        is_a[i] = a;
        goff[i] = (unsigned)(row * K + src_slot * 4);
        dst[i] = __builtin_amdgcn_readfirstlane((a ? 0 : A_B) + prow * 1024);
    }

    // DMA cursor: (tile, k16 block) of the next half-tile to fetch (wave-uniform)
    int ld_tile = blockIdx.x, ld_k = 0;
    const float* ld_a;
    const float* ld_w;
    auto cursor_bases = [&]() {
        const int lid = xcd_remap(ld_tile, total);
        ld_a = A + (int64_t)(lid / tiles_n) * BMT * K + ld_k * 16;
        ld_w = W + (int64_t)(lid % tiles_n) * BT * K + ld_k * 16;
    };
    auto cursor_advance = [&]() {
        if (ld_k + 1 < n_half) {
            ++ld_k;
            ld_a += 16;
            ld_w += 16;
        } else if (ld_tile + (int)gridDim.x < total) {
            ld_tile += gridDim.x;
            ld_k = 0;
            cursor_bases();
        }   // else: end of the stream: keep re-fetching its last half-tile (into buffers nobody reads again)
    };
    auto issue = [&](int buf) {
        unsigned char* base = smem + buf * HB;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const float* g = (is_a[i] ? ld_a : ld_w) + goff[i];
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)(base + dst[i]), 16, 0, 0);
        }
        cursor_advance();
    };

    int a_off[2][4], b_off[2][2];     // [plane][m or n]
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int m = 0; m < 4; ++m) a_off[s][m] = frag_off(m * 32 + l31, s, half);
#pragma unroll
        for (int n = 0; n < 2; ++n) b_off[s][n] = A_B + frag_off(wc * 64 + n * 32 + l31, s, half);
    }
    X8 fa[2][4], fb[2][2];     // [plane][m or n]
    auto read_frags = [&](X8 (&xa)[2][4], X8 (&xb)[2][2], int buf) {
        const unsigned char* base = smem + buf * HB;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int n = 0; n < 2; ++n) xb[s][n] = *reinterpret_cast<const X8*>(base + b_off[s][n]);
#pragma unroll
            for (int m = 0; m < 4; ++m) xa[s][m] = *reinterpret_cast<const X8*>(base + a_off[s][m]);
        }
    };

    // prologue: half-tiles 0 and 1 in flight, 0 landed and visible
    cursor_bases();
    issue(0);
    issue(1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    int b_cur = 0, b_issue = 2;
    for (int tile = blockIdx.x; tile < total; tile += gridDim.x) {
        f32x16 acc[4][2];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

        for (int ht = 0; ht < n_half; ++ht) {
            // buffer b_issue held half-tile h - 1: every wave of this workgroup finished reading it before the barrier
            // that ended step h - 1
            issue(b_issue);
            read_frags(fa, fb, b_cur);
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        acc[m][n] = mma16(fa[t == 1 ? 1 : 0][m], fb[t == 0 ? 1 : 0][n], acc[m][n]);
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // all but this step's 6 pieces: half-tile h + 1 has landed
            __builtin_amdgcn_s_barrier();
            b_cur = b_cur + 1 == NBUF ? 0 : b_cur + 1;
            b_issue = b_issue + 1 == NBUF ? 0 : b_issue + 1;
        }

        const int lid = xcd_remap(tile, total);
        const int row0 = (lid / tiles_n) * BMT, col0 = (lid % tiles_n) * BT;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const int col = col0 + wc * 64 + n * 32 + l31;
                float* o = out + (int64_t)(row0 + m * 32 + 4 * half) * N + col;
#pragma unroll
                for (int r = 0; r < 16; ++r) o[(int64_t)((r & 3) + 8 * (r >> 2)) * N] = acc[m][n][r];
            }
    }
#ifdef LAB_STAGGER
    if (wr == 0) __builtin_amdgcn_s_barrier();         // pairs with the other group's extra barrier at the start
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the trailing re-fetches before the LDS is released
}

}  // namespace

extern "C" int lab_gemm_planes_glds_2wg(const float* A, const float* W, float* out, int M, int N, int K, int f16, void* stream) {
    if (M % BMT || N % BT || K % 32 || K < 64) return -1;
    int dev = 0, cus = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    cus = cus / 8 * 8;
    const int tiles_m = M / BMT, tiles_n = N / BT, total = tiles_m * tiles_n;
    const size_t lds = (size_t)NBUF * HB;
    const dim3 grid(total < 2 * cus ? total : 2 * cus), block(256);
    if (f16) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_planes_glds4_kernel<_Float16>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(gemm_planes_glds4_kernel<_Float16>, grid, block, lds, (hipStream_t)stream, A, W, out, N, K, tiles_m, tiles_n);
    } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_planes_glds4_kernel<__bf16>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(gemm_planes_glds4_kernel<__bf16>, grid, block, lds, (hipStream_t)stream, A, W, out, N, K, tiles_m, tiles_n);
    }
    return (int)hipGetLastError();
}
