// fetch_size_calibration.hip -- what rocprofv3's FETCH_SIZE counts for the access patterns of this repository (VERDICT round 4,
// task 6).  The guide (/opt/skills/guides/MI355X_MICROARCH.md, "HBM") establishes that on gfx950 FETCH_SIZE reports HALF the bytes
// of a wide coalesced streaming read (16 B per lane); tools/pmc_traffic.py doubled it for every kernel, also for the traversal
// kernels, whose lanes each fetch their OWN 64-byte node (four dwordx4) or 48-byte triangle (three).  Known byte counts:
//   k_stream      every lane reads consecutive float4s of a 1 GiB buffer                      -> 1 GiB, coalesced
//   k_gather64    every lane reads ONE 64-byte record, four dwordx4, at a permuted index      -> 1 GiB, every record exactly once
//   k_gather48    every lane reads ONE 48-byte record (three dwordx4) of a 48-byte-stride array-> 0.75 GiB, every record exactly once
//   k_gather64x2  the 64-byte gather with every record read by TWO lanes of different waves   -> 2 GiB requested, 1 GiB distinct
// Build and run ON THE GPU BOX:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/calib tools/experiments/fetch_size_calibration.hip
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/r05_calib -o calib -- /tmp/calib
// then tools/experiments/fetch_size_calibration.py prints counter KiB / known KiB per kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

constexpr uint64_t kBytes = 1ull << 30;
constexpr uint32_t kRecords64 = (uint32_t)(kBytes / 64), kRecords48 = (uint32_t)(kBytes / 64); // same count: 0.75 GiB of 48-byte records

__global__ void k_stream(const float4 *__restrict__ in, float *__restrict__ out, uint32_t n4)
{
    float acc = 0.0f;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x)
    {
        const float4 v = in[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 123.456f) out[0] = acc;
}
__device__ __forceinline__ uint32_t permute(uint32_t i, uint32_t n) { return (uint32_t)(((uint64_t)i * 2654435761ull + 12345ull) & (n - 1)); } // n = 2^k: a bijection
__global__ void k_gather64(const float4 *__restrict__ in, float *__restrict__ out, uint32_t records, uint32_t copies)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= records * copies)
        return;
    const uint32_t r = permute(t % records, records); // copy c of a record sits `records` threads later: another wave, long after
    const float4 *p = in + (size_t)r * 4;
    const float4 a = p[0], b = p[1], c = p[2], d = p[3];
    const float s = a.x + b.y + c.z + d.w;
    if (s == 123.456f) out[0] = s;
}
__global__ void k_gather48(const float4 *__restrict__ in, float *__restrict__ out, uint32_t records)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= records)
        return;
    const float4 *p = in + (size_t)permute(t, records) * 3;
    const float4 a = p[0], b = p[1], c = p[2];
    const float s = a.x + b.y + c.z;
    if (s == 123.456f) out[0] = s;
}

int main()
{
    float4 *buf; float *out;
    hipMalloc(&buf, kBytes); hipMalloc(&out, 4);
    hipMemset(buf, 0, kBytes);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; rep++)
    {
        hipLaunchKernelGGL(k_stream, dim3(256 * 8), dim3(256), 0, 0, buf, out, (uint32_t)(kBytes / 16));
        hipLaunchKernelGGL(k_gather64, dim3(kRecords64 / 256), dim3(256), 0, 0, buf, out, kRecords64, 1u);
        hipLaunchKernelGGL(k_gather48, dim3(kRecords48 / 256), dim3(256), 0, 0, buf, out, kRecords48);
        hipLaunchKernelGGL(k_gather64, dim3(kRecords64 / 256 * 2), dim3(256), 0, 0, buf, out, kRecords64, 2u);
    }
    hipDeviceSynchronize();
    printf("done: stream %llu B, gather64 %llu B, gather48 %llu B, gather64 x2 %llu B requested\n", (unsigned long long)kBytes,
           (unsigned long long)kRecords64 * 64, (unsigned long long)kRecords48 * 48, (unsigned long long)kRecords64 * 128);
    return 0;
}
