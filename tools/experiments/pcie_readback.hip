// pcie_readback.hip -- what the box's PCIe link carries device -> page-locked host for a 1080p RGBA32F frame (33 MB): the floor of
// any step that hands a whole frame to the host.  hipMemcpyAsync (SDMA) and a copy kernel with 1 ... 256
// workgroups, back to back (link saturated) -- GB/s and ms per frame.
// Build + run ON THE GPU BOX:  hipcc --offload-arch=gfx950 -O3 -o /tmp/pcie tools/experiments/pcie_readback.hip && /tmp/pcie
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ void k_copy(const float4 *__restrict__ src, float4 *__restrict__ dst, unsigned n)
{
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        dst[i] = src[i];
}

int main()
{
    const size_t bytes = 1920ull * 1080 * 16;
    const unsigned n = (unsigned)(bytes / 16);
    float4 *dev, *host, *hostDev;
    hipMalloc(&dev, bytes);
    hipMemset(dev, 1, bytes);
    hipHostMalloc(&host, bytes, hipHostMallocDefault);
    hipHostGetDevicePointer((void **)&hostDev, host, 0);
    hipStream_t s; hipStreamCreate(&s);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int reps = 20;
    float ms;
    for (int w = 0; w < 2; w++)
    {
        hipEventRecord(a, s);
        for (int r = 0; r < reps; r++) hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, s);
        hipEventRecord(b, s); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
    }
    printf("hipMemcpyAsync D2H          : %6.3f ms per frame, %5.1f GB/s\n", ms / reps, bytes * reps / ms / 1e6);
    const int groups[] = {1, 2, 4, 8, 16, 32, 64, 256};
    for (int g : groups)
    {
        for (int w = 0; w < 2; w++)
        {
            hipEventRecord(a, s);
            for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k_copy, dim3(g), dim3(256), 0, s, dev, hostDev, n);
            hipEventRecord(b, s); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
        }
        printf("copy kernel, %3d workgroups : %6.3f ms per frame, %5.1f GB/s\n", g, ms / reps, bytes * reps / ms / 1e6);
    }
    return 0;
}
