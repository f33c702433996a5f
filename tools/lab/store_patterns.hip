// Lab: how fast does a 256-CU burst of GEMM-epilogue stores drain?  Three lane->address patterns for the
// same 65536 x 768 fp32 output, one 256x256 tile per workgroup of 8 waves (wave = 128 x 64 sub-tile):
//   A  MFMA accumulator order: dword stores, one instruction = 2 rows x 128 B          (128 instr / wave)
//   B  row order dwordx4: one instruction = 4 rows x 256 B                            ( 32 instr / wave)
//   C  row order dword: one instruction = 1 row x 256 B                               (128 instr / wave)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int PAT>
__global__ __launch_bounds__(512) void k(float* out, int ldc, int tiles_n) {
    extern __shared__ float lds_pad[];   // dynamic LDS only to pin the occupancy (128 KB: one workgroup per CU, as the GEMM)
    if (threadIdx.x == 9999) lds_pad[0] = 1.f;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, wr = wid >> 2, wc = wid & 3;
    float* base = out + (size_t)(tm * 256 + wr * 128) * ldc + tn * 256 + wc * 64;
    const float v = (float)threadIdx.x;
    if (PAT == 0) {
        const int l31 = lane & 31, half = lane >> 5;
        for (int n = 0; n < 2; ++n)
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    base[(size_t)(m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * ldc + n * 32 + l31] = v + r;
    } else if (PAT == 1) {
#pragma unroll 8
        for (int i = 0; i < 32; ++i) {
            f32x4 x = {v, v + 1, v + 2, v + i};
            *reinterpret_cast<f32x4*>(base + (size_t)(4 * i + (lane >> 4)) * ldc + 4 * (lane & 15)) = x;
        }
    } else {
#pragma unroll 8
        for (int i = 0; i < 128; ++i) base[(size_t)i * ldc + lane] = v + i;
    }
}

int main(int argc, char** argv) {
    const int M = 65536, N = 768;
    float* out;
    hipMalloc(&out, (size_t)M * N * 4);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const int lds = argc > 1 ? atoi(argv[1]) : 0;
    if (lds) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(k<0>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipFuncSetAttribute(reinterpret_cast<const void*>(k<1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipFuncSetAttribute(reinterpret_cast<const void*>(k<2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    }
    printf("dynamic LDS per workgroup: %d bytes\n", lds);
    for (int pat = 0; pat < 3; ++pat) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(a);
            for (int it = 0; it < 10; ++it) {
                if (pat == 0) hipLaunchKernelGGL(k<0>, dim3(256 * 3), dim3(512), lds, 0, out, N, 3);
                if (pat == 1) hipLaunchKernelGGL(k<1>, dim3(256 * 3), dim3(512), lds, 0, out, N, 3);
                if (pat == 2) hipLaunchKernelGGL(k<2>, dim3(256 * 3), dim3(512), lds, 0, out, N, 3);
            }
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            if (rep == 2) printf("pattern %c: %.1f us per 201 MB output = %.2f TB/s\n", 'A' + pat, ms * 100, (double)M * N * 4 / (ms / 10 * 1e-3) / 1e12);
        }
    }
    return 0;
}
