// Lab: do hipMemsetAsync NODES of a captured graph clear their destinations on every replay?  ("The bug the
// soak found": gradients accumulated by atomics on top of whatever the buffer held.)  Pure HIP, no torch, no library of ours.
//   /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/lab/graph_memset_repro.hip -o lab_build/graph_memset_repro && lab_build/graph_memset_repro
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 2; } } while (0)
__global__ void add1(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) atomicAdd(p + i, 1.0f); }
__global__ void fill(float* p, size_t n, float v) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v; }
int main(int argc, char** argv) {
    const int sizes[] = {768, 1024, 2304, 3072, 20, 8, 768 * 768, 1};   // bias / LayerNorm / embedding / split-K weight gradients (floats)
    const int NB = sizeof(sizes) / sizeof(int);
    hipStream_t s; CK(hipStreamCreate(&s));
    // argv[1] = "carve": the buffers are 512-byte-aligned pieces of ONE allocation, back to back, as a caching allocator hands them out
    const bool carve = argc > 1 && argv[1][0] == 'c';
    const bool null_launch = argc > 2;      // argv[2] present: poison fills and graph launches on the NULL stream (torch's default stream)
    float* buf[NB]; char* block = nullptr; size_t off = 0;
    if (carve) CK(hipMalloc(&block, 8 << 20));
    for (int i = 0; i < NB; ++i) {
        if (carve) { buf[i] = (float*)(block + off); off += ((size_t)sizes[i] * 4 + 511) / 512 * 512; }
        else CK(hipMalloc(&buf[i], sizes[i] * 4));
    }
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < NB; ++i) { CK(hipMemsetAsync(buf[i], 0, sizes[i] * 4, s)); add1<<<(sizes[i] + 255) / 256, 256, 0, s>>>(buf[i], sizes[i]); }
    CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    long bad = 0; std::vector<float> h(768 * 768);
    hipStream_t ls = null_launch ? nullptr : s;
    for (int r = 0; r < 1000; ++r) {
        float* junk; const size_t n = (size_t)64 << (r % 14);          // the caller allocates, poisons and frees between replays
        CK(hipMalloc(&junk, n * 4)); fill<<<64, 256, 0, ls>>>(junk, n, 1.2345e30f); CK(hipStreamSynchronize(ls)); CK(hipFree(junk));
        if (r % 3 == 0) for (int i = 0; i < NB; ++i) fill<<<64, 256, 0, ls>>>(buf[i], sizes[i], 1.2345e30f);   // stale values in the targets
        CK(hipGraphLaunch(ge, ls)); CK(hipStreamSynchronize(ls));
        for (int i = 0; i < NB; ++i) { CK(hipMemcpy(h.data(), buf[i], sizes[i] * 4, hipMemcpyDeviceToHost)); for (int j = 0; j < sizes[i]; ++j) bad += h[j] != 1.0f; }
    }
    printf("graph memset repro (pure HIP, %s): 1000 replays, %d buffers, wrong words: %ld -> %s\n", carve ? (null_launch ? "carved buffers, NULL-stream launches" : "buffers carved from one allocation") : "one hipMalloc per buffer", NB, bad, bad ? "MEMSET NODES DO NOT CLEAR" : "memset nodes clear every time");
    return bad != 0;
}
