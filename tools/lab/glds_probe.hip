// Lab probe: global_load_lds (LDS-DMA) semantics on gfx950 -- the LDS image of one wave instruction is lane-linear
// (lane l lands at base + 16 l), the source address is per lane: a swizzled LDS layout is made by permuting the SOURCE.
//   /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/lab/glds_probe.hip -o lab_build/glds_probe && lab_build/glds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* __restrict__ src, float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) float lds[2048];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    // each wave copies 1 KB: lane l fetches 16 B from a PERMUTED source position; LDS image is lane-linear
    const int srcpos = (lane ^ 5);   // 16-byte units
    const float* g = src + wid * 256 + srcpos * 4;
    float* l = lds + __builtin_amdgcn_readfirstlane(wid * 256);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)");
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += blockDim.x) out[i] = lds[i];
}
int main() {
    float h[512], o[512]; for (int i = 0; i < 512; ++i) h[i] = i;
    float *d, *e; hipMalloc(&d, 2048); hipMalloc(&e, 2048);
    hipMemcpy(d, h, 2048, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(128), 0, 0, d, e);
    hipMemcpy(o, e, 2048, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int w = 0; w < 2; ++w) for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) {
        float want = w * 256 + (l ^ 5) * 4 + j;
        if (o[w * 256 + l * 4 + j] != want) ++bad;
    }
    printf("glds lane-linear image with permuted source: %s (%d mismatches) first words %g %g %g %g | %g\n", bad ? "WRONG" : "ok", bad, o[0], o[1], o[2], o[3], o[4]);
    return bad != 0;
}
