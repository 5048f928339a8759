// rcp_sqrt_exhaustive.hip -- experiment behind the round-5 arithmetic conventions (docs/EXPERIMENTS.md).
//
// For ALL 2^32 float bit patterns: the cheap device sequences
//   rcp_(x)  = v_rcp_f32 + one FMA Newton step, guarded            (5 VALU;  IEEE 1/x by hipcc: 11)
//   sqrt_(x) = v_sqrt_f32 + the two-neighbour residual test behind a wave-uniform branch for |x| < 2^-96 (10 VALU; hipcc: 16)
// are compared with the SPECIFICATION the oracle implements with plain IEEE arithmetic:
//   rcp_(x):  |x| < 2^-126 -> copysign(inf, x);  |x| > 2^126 -> copysign(0, x);  NaN -> NaN;  else RN(1/x)
//   sqrt_(x): RN(sqrt(x)), the IEEE result, everywhere (also shown: the fast path alone, class 4 = 2^-126 <= |x| < 2^-96)
// Mismatches are counted per input class and the first few are printed.
// Build + run:  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o rcp_sqrt_exhaustive rcp_sqrt_exhaustive.hip && ./rcp_sqrt_exhaustive
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

__device__ __forceinline__ float asf(uint32_t u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ uint32_t asu(float f) { return __builtin_bit_cast(uint32_t, f); }

__device__ __forceinline__ float rcp_dev(float x)
{
    const float r0 = __builtin_amdgcn_rcpf(x);
    const float e = __builtin_fmaf(-x, r0, 1.0f);
    const float r1 = __builtin_fmaf(r0, e, r0);
    return __builtin_fabsf(e) < 1.0f ? r1 : r0;
}
__device__ __forceinline__ float rcp_spec(float x)
{
    const float ax = __builtin_fabsf(x);
    if (x != x) return x;
    if (ax < 1.17549435e-38f) return __builtin_copysignf(__builtin_inff(), x);
    if (ax > 8.50705917e37f) return __builtin_copysignf(0.0f, x);
    return 1.0f / x;
}
__device__ __forceinline__ float sqrt_dev(float x) // pt_device.hpp sqrt_
{
    if (__builtin_amdgcn_ballot_w64(__builtin_fabsf(x) < 1.262177448e-29f) != 0) // 2^-96: the residuals below would underflow
        return __builtin_sqrtf(x);
    const float s = __builtin_amdgcn_sqrtf(x);
    const float sm = asf(asu(s) - 1u), sp = asf(asu(s) + 1u);
    const float rm = __builtin_fmaf(-sm, s, x), rp = __builtin_fmaf(-sp, s, x);
    float r = rm <= 0.0f ? sm : s;
    r = rp > 0.0f ? sp : r;
    return r;
}
// the same without the branch: what the fast path alone does on EVERY input (shows where it needs the branch)
__device__ __forceinline__ float sqrt_fast_only(float x)
{
    const float s = __builtin_amdgcn_sqrtf(x);
    const float sm = asf(asu(s) - 1u), sp = asf(asu(s) + 1u);
    const float rm = __builtin_fmaf(-sm, s, x), rp = __builtin_fmaf(-sp, s, x);
    float r = rm <= 0.0f ? sm : s;
    r = rp > 0.0f ? sp : r;
    return r;
}
__device__ __forceinline__ float sqrt_spec(float x) { return __builtin_sqrtf(x); } // IEEE, hipcc's 16-instruction expansion

// round 6: rsq_(x) of pt_device.hpp -- v_rsq_f32 + one compensated Newton step with the second-order term (8 VALU + class test + select)
// against the SPECIFICATION: RN(1 / sqrt(x)) for positive normal x = (float)(1.0 / sqrt((double)x)) (shown exact on the CPU for all 2^24
// mantissa / exponent-parity classes), +-inf for +-0 and denormals, +0 for +inf, NaN for negative numbers and NaN
__device__ __forceinline__ float rsq_dev(float x)
{
    const float y0 = __builtin_amdgcn_rsqf(x);
    const float h = x * y0;
    const float l = __builtin_fmaf(x, y0, -h);
    float e = __builtin_fmaf(-h, y0, 1.0f);
    e = __builtin_fmaf(-l, y0, e);
    const float p = __builtin_fmaf(0.375f, e, 0.5f);
    const float y1 = __builtin_fmaf(y0 * e, p, y0);
    return __builtin_amdgcn_classf(x, 0x100) ? y1 : y0;
}
__device__ __forceinline__ float rsq_spec(float x)
{
    const uint32_t b = asu(x);
    if (b - 0x00800000u < 0x7f800000u - 0x00800000u) return (float)(1.0 / __builtin_sqrt((double)x));
    if (x != x) return x;
    if ((b & 0x7fffffffu) < 0x00800000u) return __builtin_copysignf(__builtin_inff(), x);
    return b == 0x7f800000u ? 0.0f : __builtin_nanf("");
}

struct Report { unsigned long long bad[4][5]; uint32_t first[4][5][8]; uint32_t firstGot[4][5][8]; uint32_t firstWant[4][5][8]; };

__device__ int classOf(float x)
{
    const float ax = __builtin_fabsf(x);
    if (x != x) return 3;                                    // NaN
    if (ax < 1.17549435e-38f) return 1;                      // zero / denormal
    if (ax > 8.50705917e37f) return 2;                       // > 2^126 (incl. inf)
    if (ax < 1.262177448e-29f) return 4;                     // normal, below 2^-96
    return 0;                                                // the range the shading arithmetic lives in
}

__global__ void k_check(Report *rep, uint32_t base)
{
    const uint32_t bits = base + blockIdx.x * blockDim.x + threadIdx.x;
    const float x = asf(bits);
    const int c = classOf(x);
    for (int f = 0; f < 4; f++)
    {
        const float got = f == 0 ? rcp_dev(x) : f == 1 ? sqrt_dev(x) : f == 2 ? sqrt_fast_only(x) : rsq_dev(x);
        const float want = f == 0 ? rcp_spec(x) : f == 3 ? rsq_spec(x) : sqrt_spec(x);
        const bool same = asu(got) == asu(want) || (got != got && want != want);  // any NaN equals any NaN
        if (!same)
        {
            const unsigned long long k = atomicAdd(&rep->bad[f][c], 1ull);
            if (k < 8) { rep->first[f][c][k] = bits; rep->firstGot[f][c][k] = asu(got); rep->firstWant[f][c][k] = asu(want); }
        }
    }
}

// raw hardware behaviour on a few telling inputs
__global__ void k_probe(const uint32_t *in, uint32_t *out, int n)
{
    const int i = threadIdx.x;
    if (i < n) { out[3 * i] = asu(__builtin_amdgcn_rcpf(asf(in[i]))); out[3 * i + 1] = asu(__builtin_amdgcn_sqrtf(asf(in[i]))); out[3 * i + 2] = asu(__builtin_amdgcn_rsqf(asf(in[i]))); }
}

int main()
{
    Report *rep; hipMalloc(&rep, sizeof(Report)); hipMemset(rep, 0, sizeof(Report));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); hipEventRecord(a);
    for (uint32_t hi = 0; hi < 256; hi++)
        hipLaunchKernelGGL(k_check, dim3(1u << 16), dim3(256), 0, 0, rep, hi << 24);
    hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b);
    Report h; hipMemcpy(&h, rep, sizeof(h), hipMemcpyDeviceToHost);
    const char *fn[4] = {"rcp_", "sqrt_", "sqrt fast path alone", "rsq_"}, *cl[5] = {"normal range", "zero/denormal", "> 2^126 / inf", "NaN", "[2^-126, 2^-96)"};
    printf("all 2^32 inputs in %.1f ms\n", ms);
    for (int f = 0; f < 4; f++)
        for (int c = 0; c < 5; c++)
        {
            printf("%-20s %-16s mismatches %llu\n", fn[f], cl[c], h.bad[f][c]);
            for (unsigned k = 0; k < 8 && k < h.bad[f][c]; k++)
                printf("        x %08x got %08x want %08x\n", h.first[f][c][k], h.firstGot[f][c][k], h.firstWant[f][c][k]);
        }
    const uint32_t probes[] = {0x00000000u, 0x80000000u, 0x00000001u, 0x00400000u, 0x007fffffu, 0x80400000u, 0x00800000u, 0x7e800000u, 0x7e800001u,
                               0x7f000000u, 0x7f7fffffu, 0x7f800000u, 0xff800000u, 0x7fc00000u, 0x3f800000u, 0x3fffffffu};
    const int n = sizeof(probes) / 4;
    uint32_t *din, *dout; hipMalloc(&din, sizeof(probes)); hipMalloc(&dout, sizeof(probes) * 3);
    hipMemcpy(din, probes, sizeof(probes), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, din, dout, n);
    uint32_t out[3 * 16]; hipMemcpy(out, dout, sizeof(out), hipMemcpyDeviceToHost);
    for (int i = 0; i < n; i++) printf("probe x %08x  v_rcp %08x  v_sqrt %08x  v_rsq %08x\n", probes[i], out[3 * i], out[3 * i + 1], out[3 * i + 2]);
    return 0;
}
