// pt_wavefront.hpp -- the render kernels of the path-tracing pass for gfx950 (MI355X).  Device code only: the renderer object,
// its buffers and the bounce schedule are in pt_runtime.hpp, the C-ABI of include/ptx.h in ptx_capi.hip.
//
// The reference runs ONE ray-tracing pipeline dispatch per frame,
//   vkCmdTraceRaysKHR(W, H, 1)            (Renderer/Renderer.cpp:911-917)
// whose raygen shader (Shaders/raygen.rgen:36-118) loops over samples and bounces and
// calls traceRayEXT twice per bounce (closest hit :68, occlusion :31).  Recursion depth
// is 1, i.e. the path is an iterative loop in raygen: that loop is cut here at the two
// traceRayEXT calls into a queue-per-stage WAVEFRONT:
//
//   k_generate        raygen.rgen:38-60   RNG seed, primary ray
//   k_trace_closest   raygen.rgen:68      closest-hit query over the active queue
//   k_shade           closestHit.rchit / miss.rmiss + raygen.rgen:71-96 bookkeeping
//   k_trace_shadow    raygen.rgen:22-34   occlusion query: one answer per shadow-queue entry
//   k_apply_shadow    raygen.rgen:79-81   NEE add of the visible lights, finish of the paths that ended on this bounce
//   k_accumulate      raygen.rgen:115-117  image += radiance, in frame order
//
// Every path slot is (frame, pixel); its state lives in SoA arrays in HBM; queues hold
// slot indices and are compacted by wave-aggregated atomics.  A bring-up MEGAKERNEL
// (one thread per slot running the loop 1:1) shares all device functions and is kept as
// the in-tree A/B reference of the wavefront.
//
// No CPU fallback exists: without a HIP device ptx_create fails.
#pragma once

#include <hip/hip_runtime.h>

#include "pt_bvh.hpp"

using namespace ptd;

// =====================================================================================
// Launch parameters
// =====================================================================================

// Per-slot path state is written by one kernel and read once by the next: a stream.  Its loads and stores carry the
// non-temporal hint (global_load / global_store ... nt), so that the 4 MB of L2 an XCD has keep tree nodes and texels instead
// of records nobody reads twice.  Measured (1 MI355X, 1080p, 8 spp, two runs each in one call, plain -> nt): atrium_like
// 788 / 789 -> 811 / 823 Msamples/s, chess_like 2,248 / 2,229 -> 2,267 / 2,262, temple_like 898 / 883 -> 905 / 895, street_like
// flat; the hint on the loads alone or on the stores alone gives half of it; on the ShadeTri reads it costs 4 % (the samples
// of one pixel sit in neighbouring lanes and share them).
template <typename T> struct StreamWord { typedef T type; };
template <> struct StreamWord<float4> { typedef float type __attribute__((ext_vector_type(4))); };
template <> struct StreamWord<uint4> { typedef uint32_t type __attribute__((ext_vector_type(4))); };
template <typename T> struct StreamRef
{
    T *p;
    typedef typename StreamWord<T>::type W;
    PT_DEV operator T() const
    {
        const W w = __builtin_nontemporal_load(reinterpret_cast<const W *>(p));
        T v;
        __builtin_memcpy(&v, &w, sizeof(T));
        return v;
    }
    PT_DEV void operator=(const T &v) const
    {
        W w;
        __builtin_memcpy(&w, &v, sizeof(T));
        __builtin_nontemporal_store(w, reinterpret_cast<W *>(p));
    }
};
template <typename T> struct Stream // wf.rayO[slot] reads and writes as before; wf.rayO.p[slot] is the plain access
{
    T *p;
    PT_DEV StreamRef<T> operator[](size_t i) const { return StreamRef<T>{p + i}; }
    __host__ __device__ Stream &operator=(T *q) { p = q; return *this; }
    __host__ __device__ explicit operator bool() const { return p != nullptr; }
};

struct Wavefront // device pointers of the per-slot state (SoA)
{
    Stream<float4> rayO;   // origin.xyz, w = MaxRoughness (payload.MaxRoughness)
    Stream<float4> rayD;   // direction.xyz
    Stream<float4> thr;    // throughput.rgb
    Stream<float4> rad;    // radiance.rgb accumulated over the samples of this launch
    Stream<uint4> meta;    // x = rngState, y = pixel (y*W+x) or 0xffffffff, z = bounce | smpl<<16, w = frame
    Stream<float4> hit;    // t, u, v, triangle slot in leaf order (bits)
    Stream<uint32_t> hitPair;
    Stream<float4> shO;    // shadow origin.xyz, w = tmax (LightDistance)
    Stream<float4> shD;    // shadow direction.xyz, w = 1 if the path ends after this bounce
    Stream<float4> shC;    // NEE contribution throughput * DirectLight / DirectLightPdf
    Stream<float4> slotRad; // final radiance of the slot (consumed by k_accumulate)
    Stream<float4> decal;   // nearest ignored any-hit candidate: (triangle slot, u, v, pair) -- k_shade fetches its colour and alpha
    Stream<float> decalT;   // (payload.LightDirection / LightDistance) if the hit lies behind it; null unless the scene has non-opaque
                     // geometry.  decalT = its distance or -1 (payload.DirectLightPdf)
    Stream<float4> diff[3]; // payload.RayDifferentials0..2 (rx origin, rx dir, ry origin, ry dir); null unless the scene has textures
    uint32_t *queue[2];
    uint32_t *shadowQueue;
    uint8_t *shadowResult; // per shadow queue entry: bit 0 = the light is visible, bit 1 = the path ends here (k_apply_shadow)
    uint32_t *restartQueue;
    uint32_t *counters; // see enum Counter
    uint32_t *spill;    // traversal stack overflow region [kGlobalSpill][kMaxPersistentThreads]
};

// Every counter sits on its own 128-byte line: atomics to one L2 line serialise (~11 ns each on MI355X) whatever
// word they touch, and the queue, chunk and statistics counters are all hot in the same kernels.
constexpr int kCounterStride = 32; // uint32 words
constexpr int kMaxTimedBounces = 64; // per-bounce bookkeeping (live counts for the statistics, kernel timing events) up to this depth
enum Counter
{
    C_ACTIVE0 = 0 * kCounterStride,
    C_ACTIVE1 = 1 * kCounterStride,
    C_SHADOW = 2 * kCounterStride,    // shadow queue of even bounces (the shadow kernel of bounce b runs beside bounce b + 1: two sets)
    C_HITS = 3 * kCounterStride,      // closest-hit shader invocations (= occlusion queries of the reference)
    C_SAMPLES = 4 * kCounterStride,   // completed pixel-samples incl. retries
    C_RETRIES = 5 * kCounterStride,
    C_SEGMENTS = 6 * kCounterStride,  // closest-hit queries traced inside k_tail / the megakernel
    C_OVERFLOW = 7 * kCounterStride,
    C_CHUNK = 8 * kCounterStride,        // next unclaimed queue entry of k_trace_closest
    C_CHUNK_SHADOW = 9 * kCounterStride, // ... of k_trace_shadow, even bounces
    C_RESTART = 10 * kCounterStride, // slots re-queued for their next sample (multi-sample launch, NaN restart), drained after the bounce loop
    C_SHADOW1 = 11 * kCounterStride,       // odd bounces
    C_CHUNK_SHADOW1 = 12 * kCounterStride,
    C_WAVE_SEGMENTS = 13 * kCounterStride, // 64-bit: closest-hit queries traced by k_trace_closest (k_prologue adds each bounce's queue length)
    C_TAIL_PATHS = 14 * kCounterStride,    // paths k_tail took over
    C_BOUNCE_ACTIVE = 15 * kCounterStride, // [kMaxTimedBounces + 1]: queue length at the start of each bounce of the last round
    C_COUNT = C_BOUNCE_ACTIVE + ((kMaxTimedBounces + 1 + kCounterStride - 1) / kCounterStride) * kCounterStride
};
PT_DEV int queueCounter(int q) { return q ? (int)C_ACTIVE1 : (int)C_ACTIVE0; }
PT_DEV int shadowCounter(int parity) { return parity ? (int)C_SHADOW1 : (int)C_SHADOW; }
PT_DEV int shadowChunkCounter(int parity) { return parity ? (int)C_CHUNK_SHADOW1 : (int)C_CHUNK_SHADOW; }

// The bounce loop is driven from the device: every kernel of a bounce takes its queue length from the counter block, so
// the host enqueues the whole schedule (BounceCount bounces) without a single read-back in between.
//   * A queue at or below `tailBelow` paths (after the first bounce) belongs to k_tail, which runs them to the end of
//     their sample in one launch: the wavefront kernels of the following bounces see that and return at once.
//   * k_prologue, one thread ahead of each bounce, clears the counters that bounce appends to (the shadow queue has two
//     sets: the shadow kernel of bounce b runs beside bounce b + 1) and keeps the statistics.
struct BounceCtl
{
    uint32_t bounce;    // 1-based index inside the round
    uint32_t tailBelow; // queues of at most this many paths go to k_tail (never the first bounce of a round)
    uint32_t sortShade; // k_shade puts its block's queue entries in material-type order first (scenes that mix types)
};
PT_DEV bool bounceRuns(const BounceCtl &c, uint32_t count) { return count != 0u && (c.bounce <= 1u || count > c.tailBelow); }

__global__ void k_prologue(Wavefront wf, int qin, BounceCtl ctl)
{
    const uint32_t count = wf.counters[queueCounter(qin)];
    const int parity = (int)(ctl.bounce & 1u);
    wf.counters[queueCounter(qin ^ 1)] = 0u;
    wf.counters[shadowCounter(parity)] = 0u;
    wf.counters[shadowChunkCounter(parity)] = 0u;
    wf.counters[C_CHUNK] = 0u;
    if (ctl.bounce <= (uint32_t)kMaxTimedBounces)
        wf.counters[C_BOUNCE_ACTIVE + ctl.bounce] = count;
    if (bounceRuns(ctl, count))
    {
        unsigned long long *seg = reinterpret_cast<unsigned long long *>(&wf.counters[C_WAVE_SEGMENTS]);
        *seg += count;
    }
}


struct LaunchParams
{
    PtxRaygenUniformData u;
    uint32_t width, height;
    uint32_t rank, worldSize, tileSize, tilesX, numTiles, ownedTiles;
    uint32_t slotsPerFrame; // ownedTiles * tileSize^2
    uint32_t frames, firstFrame;
    uint32_t numSlots;
    uint32_t ownedPixels; // slots of one frame that map to a pixel inside the image
    uint32_t framesPerWave; // 1, 2, 4 or 8 (divides frames): a wave of 64 slots = 64 / framesPerWave pixels x framesPerWave frames
};

// slot <-> (frame of the batch, slot inside the frame).  The samples a batch adds to ONE pixel sit in neighbouring lanes: their
// primary rays differ by the sub-pixel jitter only, they reach the same triangles and the same texels (a wave = 8 pixels of a
// row x 8 frames instead of an 8x8 pixel block of one frame).  Measured, 1 / 2 / 4 / 8 frames per wave (PTX_FRAMES_PER_WAVE,
// two runs each): atrium_like 770, 766 / 769, 777 / 779, 776 / 788, 782 Msamples/s (k_shade<true> 44.8 -> 39.5 ms of kernel time
// per step), street_like +1 %, chess_like and temple_like flat; a 4x2 pixel footprint instead of the row: flat.
PT_DEV void slotFrame(const LaunchParams &p, uint32_t slot, uint32_t &f, uint32_t &s)
{
    const uint32_t g = p.framesPerWave, pixelsPerWave = 64u / g, wave = slot >> 6, lane = slot & 63u;
    const uint32_t chunks = p.slotsPerFrame / pixelsPerWave; // slotsPerFrame is a multiple of 64
    f = (wave / chunks) * g + lane % g;
    s = (wave % chunks) * pixelsPerWave + lane / g;
}

// slot -> pixel.  Owned tiles are rank, rank+world, ...; inside a tile pixels are laid
// out in 8x8 blocks so that one wave64 = one 8x8 pixel block (coherent primary rays).
PT_DEV uint32_t slotPixel(const LaunchParams &p, uint32_t slotInFrame)
{
    const uint32_t ts = p.tileSize, perTile = ts * ts;
    const uint32_t k = slotInFrame / perTile, o = slotInFrame % perTile;
    const uint32_t tile = p.rank + k * p.worldSize;
    const uint32_t bpr = ts / 8, blk = o / 64, ib = o % 64;
    const uint32_t x = (tile % p.tilesX) * ts + (blk % bpr) * 8 + (ib % 8);
    const uint32_t y = (tile / p.tilesX) * ts + (blk / bpr) * 8 + (ib / 8);
    if (tile >= p.numTiles || x >= p.width || y >= p.height)
        return 0xffffffffu;
    return y * p.width + x;
}

// raygen.rgen:44-60: start one sample of a slot (jitter draws, primary ray; with DIFF also the
// offset rays of raygen.rgen:56-58)
template <bool DIFF>
PT_DEV void startSample(const LaunchParams &p, uint32_t pixel, uint32_t &rng, f3 &origin, f3 &direction, DiffRays &diff)
{
    f2 u;
    u.x = rnd(rng);
    u.y = rnd(rng);
    const uint32_t px = pixel % p.width, py = pixel / p.width;
    f3 rx = F3s(0.0f), ry = F3s(0.0f);
    if (p.u.LensRadius > 0)
    {
        f2 u2;
        u2.x = rnd(rng);
        u2.y = rnd(rng);
        constructPrimaryRayLens<DIFF>(px, py, p.width, p.height, p.u.ViewInverse, p.u.ProjInverse, u, u2, p.u.LensRadius, p.u.FocalDistance,
                                      origin, direction, rx, ry);
    }
    else
        constructPrimaryRay<DIFF>(px, py, p.width, p.height, p.u.ViewInverse, p.u.ProjInverse, u, origin, direction, rx, ry);
    if (DIFF)
    {
        diff.rxOrigin = origin;
        diff.rxDirection = rx;
        diff.ryOrigin = origin;
        diff.ryDirection = ry;
    }
}

// the payload packing of raygen.rgen:56-58 / closestHit.rchit:157-159
PT_DEV void storeDiff(const Wavefront &wf, uint32_t slot, const DiffRays &d)
{
    wf.diff[0][slot] = make_float4(d.rxOrigin.x, d.rxOrigin.y, d.rxOrigin.z, d.rxDirection.x);
    wf.diff[1][slot] = make_float4(d.rxDirection.y, d.rxDirection.z, d.ryOrigin.x, d.ryOrigin.y);
    wf.diff[2][slot] = make_float4(d.ryOrigin.z, d.ryDirection.x, d.ryDirection.y, d.ryDirection.z);
}
PT_DEV DiffRays loadDiff(const Wavefront &wf, uint32_t slot)
{
    const float4 a = wf.diff[0][slot], b = wf.diff[1][slot], c = wf.diff[2][slot];
    DiffRays d;
    d.rxOrigin = F3(a.x, a.y, a.z);
    d.rxDirection = F3(a.w, b.x, b.y);
    d.ryOrigin = F3(b.z, b.w, c.x);
    d.ryDirection = F3(c.y, c.z, c.w);
    return d;
}

// new primary ray of a slot; the differentials go straight to the slot state when the scene carries them
PT_DEV void startSlotSample(const LaunchParams &p, const Wavefront &wf, uint32_t slot, uint32_t pixel, uint32_t &rng, f3 &o, f3 &d)
{
    DiffRays diff;
    if (wf.diff[0])
    {
        startSample<true>(p, pixel, rng, o, d, diff);
        storeDiff(wf, slot, diff);
    }
    else
        startSample<false>(p, pixel, rng, o, d, diff);
}

PT_DEV bool badRadiance(f3 r) // raygen.rgen:101,107
{
    return __builtin_isnan(r.x) || __builtin_isnan(r.y) || __builtin_isnan(r.z) || __builtin_isinf(r.x) ||
           __builtin_isinf(r.y) || __builtin_isinf(r.z);
}

// =====================================================================================
// Wavefront kernels
// =====================================================================================

constexpr int kBlock = 256;

// Queue append with ONE atomic per wave: ballot the pushing lanes, the first of them
// reserves popcount slots, every lane takes base + its rank.  Must be reached by all lanes
// of the wave that are still in the (wave-uniform) loop.
PT_DEV void wavePush(uint32_t *__restrict__ queue, uint32_t *__restrict__ counter, bool push, uint32_t value)
{
    const uint64_t mask = __ballot(push);
    if (mask == 0)
        return;
    const uint32_t lane = threadIdx.x & 63u;
    const int leader = __ffsll((unsigned long long)mask) - 1;
    uint32_t base = 0;
    if ((int)lane == leader)
        base = atomicAdd(counter, (uint32_t)__popcll(mask));
    base = __shfl(base, leader);
    if (push)
        queue[base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = value;
}

// statistics: lanes count in registers, one atomic per wave at kernel exit
PT_DEV void waveAddCounter(uint32_t *__restrict__ counter, uint32_t v)
{
    for (int off = 32; off > 0; off >>= 1)
        v += __shfl_down(v, off);
    if ((threadIdx.x & 63u) == 0 && v)
        atomicAdd(counter, v);
}

// the same per block (all threads of the block must call it): one global atomic per block instead of per wave
PT_DEV void blockAddCounter(uint32_t *__restrict__ counter, uint32_t v)
{
    __shared__ uint32_t s_sum;
    if (threadIdx.x == 0)
        s_sum = 0;
    __syncthreads();
    for (int off = 32; off > 0; off >>= 1)
        v += __shfl_down(v, off);
    if ((threadIdx.x & 63u) == 0 && v)
        atomicAdd(&s_sum, v);
    __syncthreads();
    if (threadIdx.x == 0 && s_sum)
        atomicAdd(counter, s_sum);
    __syncthreads();
}

#ifndef PT_SHADE_ITEMS
#define PT_SHADE_ITEMS 4
#endif
constexpr uint32_t kShadeItems = PT_SHADE_ITEMS; // queue entries per thread per block-wide append in k_shade
constexpr uint32_t kDeadPair = 0xfffffffeu; // hitPair of a slot outside the image (ragged edge tiles)

// No queue atomics here: queue 0 is the identity over all slots (the host sets its count);
// slots of edge tiles that fall outside the image are flagged dead through rayD.w < 0.
__global__ void __launch_bounds__(kBlock) k_generate(LaunchParams p, Wavefront wf)
{
    for (uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x; slot < p.numSlots; slot += gridDim.x * blockDim.x)
    {
        uint32_t f, s;
        slotFrame(p, slot, f, s);
        const uint32_t pixel = slotPixel(p, s);
        const uint32_t frame = p.firstFrame + f;
        uint4 meta = make_uint4(0u, pixel, 0u, frame);
        wf.queue[0][slot] = slot;
        if (pixel != 0xffffffffu)
        {
            uint32_t rng = initRng(pixel % p.width, pixel / p.width, p.width, frame); // raygen.rgen:38
            f3 o, d;
            startSlotSample(p, wf, slot, pixel, rng, o, d);
            meta.x = rng;
            wf.rayO[slot] = make_float4(o.x, o.y, o.z, 0.0f); // MaxRoughness = 0, raygen.rgen:60
            wf.rayD[slot] = make_float4(d.x, d.y, d.z, 0.0f);
            // thr[slot] = 1 and rad[slot] = 0 are implied by meta.z == 0 (first bounce of the first sample): 32 bytes per
            // slot neither written here nor read by the first k_shade
        }
        else
        {
            wf.rayD[slot] = make_float4(0.0f, 0.0f, 0.0f, -1.0f);
            wf.slotRad[slot] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
        wf.meta[slot] = meta;
    }
}

struct ClosestIO
{
    static constexpr float kFixedTmin = 0.00001f; // ray.glsl:79
    static constexpr float kFixedTmax = 10000.0f; // ray.glsl:80
    static constexpr bool kNeedsPrim = false;    // the hit record carries (t, u, v, slot) and the pair
    static constexpr bool kHasQueue = true;
    const Wavefront &wf;
    const uint32_t *queue;
    uint32_t slot;
    PT_DEV uint32_t queueEntry(uint32_t item) const { return queue[item]; }
    PT_DEV void setEntry(uint32_t s) { slot = s; }
    PT_DEV bool load(uint32_t item, f3 &o, f3 &d, float &tmin, float &tmax)
    {
        const float4 d4 = wf.rayD[slot];
        if (d4.w < 0.0f)
        {
            wf.hitPair[slot] = kDeadPair;
            return false;
        }
        const float4 o4 = wf.rayO[slot];
        o = F3(o4.x, o4.y, o4.z);
        d = F3(d4.x, d4.y, d4.z);
        tmin = 0.00001f; // ray.glsl:79-80: tmin = 1e-5, tmax = 1e4 on every segment
        tmax = 10000.0f;
        if (wf.decalT)
            wf.decalT[slot] = -1.0f; // anyhit.rahit state of a fresh ray: nothing ignored yet
        return true;
    }
    PT_DEV void improve(uint32_t, float t, float u, float v, uint32_t triSlot)
    {
        wf.hit[slot] = make_float4(t, u, v, __uint_as_float(triSlot)); // w = triangle slot in leaf order
    }
    PT_DEV uint32_t bestSlot(uint32_t) const { return __float_as_uint(wf.hit.p[slot].w); }
    PT_DEV void store(uint32_t, const Hit &h, bool, bool) { wf.hitPair[slot] = h.pair; }
    // anyhit.rahit:54-61: the nearest ignored candidate (ties: smaller (pair, prim)) is the decal.  It lives in the slot's
    // record -- (triangle slot, u, v, pair) + its distance -- and k_shade fetches its colour if the hit lies behind it.
    // The ids come from the triangle record when they are needed (the tie, the store), not as arguments held in registers.
    PT_DEV void ignored(float t, float u, float v, uint32_t triSlot, const TraceScene &sc)
    {
        const float cur = wf.decalT[slot];
        bool nearer = cur == -1.0f || t < cur;
        if (!nearer && t == cur)
        {
            const float4 mine = sc.tris[triSlot].c, other = sc.tris[__float_as_uint(wf.decal.p[slot].x)].c;
            const uint32_t pair = __float_as_uint(mine.y), curPair = __float_as_uint(other.y);
            nearer = pair < curPair || (pair == curPair && __float_as_uint(mine.z) < __float_as_uint(other.z));
        }
        if (nearer)
        {
            wf.decalT[slot] = t;
            wf.decal[slot] = make_float4(__uint_as_float(triSlot), u, v, sc.tris[triSlot].c.y);
        }
    }
};

// Occupancy of the traversal kernels (waves per SIMD; overridable for A/B builds through tools/kernel_resources.py -- -D...).
// The opaque variants run at the hardware's 8: 58 / 56 VGPRs without a spill.  The ALPHA variants hold the any-hit record of a
// non-opaque triangle beside the triangle (70 / 68 VGPRs) and run at 7; at 8 the shadow variant fits (63, no vector spill)
// and the closest variant spills 11 registers -- measured: atrium_like 732 / 726 -> 727 / 720 (shadow at 8) and 711 / 713 (both).
#ifndef PT_TRACE_WAVES
#define PT_TRACE_WAVES 8
#endif
#define PT_FULL_OCCUPANCY __attribute__((amdgpu_waves_per_eu(PT_TRACE_WAVES, PT_TRACE_WAVES)))
#ifndef PT_ALPHA_CLOSEST_WAVES
#define PT_ALPHA_CLOSEST_WAVES 7
#endif
#define PT_ALPHA_CLOSEST_ATTR __attribute__((amdgpu_waves_per_eu(PT_ALPHA_CLOSEST_WAVES, PT_ALPHA_CLOSEST_WAVES)))
#ifndef PT_ALPHA_SHADOW_WAVES
#define PT_ALPHA_SHADOW_WAVES 7
#endif
#define PT_ALPHA_SHADOW_ATTR __attribute__((amdgpu_waves_per_eu(PT_ALPHA_SHADOW_WAVES, PT_ALPHA_SHADOW_WAVES)))
template <bool ALPHA>
PT_DEV void traceClosestBody(const TraceScene &sc, const Wavefront &wf, int qin, const BounceCtl &ctl)
{
    const uint32_t count = wf.counters[queueCounter(qin)];
    if (!bounceRuns(ctl, count))
        return;
    PT_DECLARE_STACK(st, kLdsStack, wf.spill)
    ClosestIO io = { wf, wf.queue[qin], 0u };
    persistentTrace<false, ALPHA>(sc, io, count, &wf.counters[C_CHUNK], st);
    if (st.overflow)
        atomicAdd(&wf.counters[C_OVERFLOW], 1u);
}
template <bool ALPHA>
__global__ void __launch_bounds__(kBlock) k_trace_closest(TraceScene sc, Wavefront wf, int qin, BounceCtl ctl);
template <>
__global__ void __launch_bounds__(kBlock) PT_FULL_OCCUPANCY k_trace_closest<false>(TraceScene sc, Wavefront wf, int qin, BounceCtl ctl)
{
    traceClosestBody<false>(sc, wf, qin, ctl);
}
template <>
__global__ void __launch_bounds__(kBlock) PT_ALPHA_CLOSEST_ATTR k_trace_closest<true>(TraceScene sc, Wavefront wf, int qin, BounceCtl ctl)
{
    traceClosestBody<true>(sc, wf, qin, ctl);
}

// raygen.rgen:99-112 + sample loop control for a slot whose path has ended.
// Returns true if the slot has samples left in this launch (next sample of a multi-sample launch, NaN restart): the
// caller appends it to the restart queue and k_restart draws its next primary ray before that queue is consumed.
// The ray is NOT constructed here: the camera matrices, the lens and the differential code would sit in the register
// and instruction-cache budget of the shading and traversal kernels for a path the canonical schedule
// (SampleCount = 1) takes only after a NaN.
PT_DEV bool finishSample(const LaunchParams &p, const Wavefront &wf, uint32_t slot, uint4 &meta, f3 &radiance,
                         uint32_t &nSamples, uint32_t &nRetries)
{
    uint32_t smpl = meta.z >> 16;
    nSamples++;
    if (badRadiance(radiance))
    {
        radiance = F3s(0.0f);
        smpl = 0; // "smpl = -1; continue" restarts ALL samples of the launch, RNG carried on
        nRetries++;
    }
    else
        smpl = smpl + 1;
    if (smpl < p.u.SampleCount)
    {
        meta.z = smpl << 16; // bounce = 0
        wf.rad[slot] = make_float4(radiance.x, radiance.y, radiance.z, 0.0f);
        return true;
    }
    wf.slotRad[slot] = make_float4(radiance.x, radiance.y, radiance.z, 0.0f);
    return false;
}

// Appends the slots of the calling lanes (restart == true) to the restart queue with one atomic per wave.  May be
// called under divergent control flow: the ballot sees the active lanes only.
PT_DEV void pushRestarts(const Wavefront &wf, bool restart, uint32_t slot)
{
    const uint64_t mask = __ballot(restart);
    if (!mask)
        return;
    const uint32_t lane = threadIdx.x & 63u;
    const int leader = __ffsll((unsigned long long)mask) - 1;
    uint32_t base = 0;
    if ((int)lane == leader)
        base = atomicAdd(&wf.counters[C_RESTART], (uint32_t)__popcll(mask));
    base = __shfl(base, leader);
    if (restart)
        wf.restartQueue[base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = slot;
}

template <bool TEX>
PT_DEV void shadeBody(const LaunchParams &p, const SceneView &sv, const Wavefront &wf, int qin, const BounceCtl &ctl);
template <bool TEX>
__global__ void __launch_bounds__(kBlock) k_shade(LaunchParams p, SceneView sv, Wavefront wf, int qin, BounceCtl ctl);
// 178 VGPRs by itself (195 with the SLP vectoriser); held at 168 = three waves per SIMD, which costs nothing now (15 spilled
// dwords before round 3).  Four waves (128 VGPRs, 49 spilled) lose: chess_like 2,290 / 2,324 -> 2,164 / 2,185 Msamples/s.
#ifndef PT_SHADE_ATTR
#define PT_SHADE_ATTR __attribute__((amdgpu_waves_per_eu(3, 3)))
#endif
template <>
__global__ void __launch_bounds__(kBlock) PT_SHADE_ATTR k_shade<false>(LaunchParams p, SceneView sv, Wavefront wf, int qin, BounceCtl ctl)
{
    shadeBody<false>(p, sv, wf, qin, ctl);
}
// the textured variant: 221 VGPRs = two waves per SIMD (round 1: a few registers past 256, i.e. ONE wave, held at two for four
// spilled registers).  Three waves (168 VGPRs, 54 spilled, 164 B scratch) measure flat: atrium_like 725 / 730 -> 723 / 724.
#ifndef PT_SHADE_TEX_ATTR
#define PT_SHADE_TEX_ATTR __attribute__((amdgpu_waves_per_eu(2, 2)))
#endif
template <>
__global__ void __launch_bounds__(kBlock) PT_SHADE_TEX_ATTR k_shade<true>(LaunchParams p, SceneView sv, Wavefront wf, int qin, BounceCtl ctl)
{
    shadeBody<true>(p, sv, wf, qin, ctl);
}
template <bool TEX>
PT_DEV void shadeBody(const LaunchParams &p, const SceneView &sv, const Wavefront &wf, int qin, const BounceCtl &ctl)
{
    __shared__ uint32_t s_cnt[2], s_base[2];
    const int qout = qin ^ 1;
    const uint32_t count = wf.counters[queueCounter(qin)];
    if (!bounceRuns(ctl, count)) // empty, or k_tail's
        return;
    const int shadowSet = shadowCounter((int)(ctl.bounce & 1u));
    uint32_t nHits = 0, nSamples = 0, nRetries = 0;
    // kShadeItems queue entries per thread between two block-wide appends: the appends cost one global atomic per
    // block and queue, and same-address atomics serialise at ~11 ns -- at one entry per thread the 65 K blocks x 2
    // queues of a 16.6 M-slot launch would keep the counter line busy for 1.4 ms of a 2 ms kernel.
    for (uint32_t base = blockIdx.x * blockDim.x * kShadeItems; base < count; base += gridDim.x * blockDim.x * kShadeItems)
    {
      // Material-sorted shade queue: the block's kBlock x kShadeItems entries are put in the order
      //   sky (miss.rmiss) | MetallicRoughness | SpecularGlossiness | Phong | unknown type | the three types again for materials
      //   that sample a scene texture | dead slot
      // (ShaderTypes.incl:143-145, the dispatch of material.glsl:144-166) before they are shaded, so that a wave runs one
      // branch of sampleMaterial / the miss stage instead of all that its 64 entries happen to need, and the software sampler
      // -- up to sixteen anisotropic taps, a seventh of an atrium_like step -- runs in waves of textured hits only instead of in
      // every wave that holds one.  A stable counting sort: per wave one ballot per (entry, key) gives the counts, a prefix
      // over (key, wave) the bases, the same ballots the ranks; the sorted slots go through LDS.  Deterministic: the order
      // inside a key is the queue order.
      // Measured (1 MI355X, 1080p, 8 spp, one frame in flight): materials_test (three material types + an unknown one side
      // by side) 1,313 -> 1,536 Msamples/s, k_shade 8.06 -> 6.29 ms; scenes of ONE material type pay for the sort and get
      // nothing back (temple_like 614 -> 600, chess_like +-0.5 %), hence the switch.
      __shared__ uint32_t s_sorted[kBlock * kShadeItems];
      constexpr uint32_t kPadSlot = 0xffffffffu; // never a slot: 184 B of state per slot bound the count far below
      if (ctl.sortShade)
      {
        constexpr uint32_t kKeys = 9, kWaves = kBlock / 64;
        __shared__ uint32_t s_keyCount[kWaves][kKeys], s_keyBase[kWaves][kKeys];
        const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
        const uint64_t lower = (1ull << lane) - 1ull;
        uint32_t mySlot[kShadeItems], myKey[kShadeItems];
        uint32_t cnt[kKeys];
        for (uint32_t k = 0; k < kKeys; k++)
            cnt[k] = 0;
        for (uint32_t item = 0; item < kShadeItems; item++)
        {
            const uint32_t i = base + item * blockDim.x + threadIdx.x;
            // dead slots and the padding past the end of the queue are sorted last.  Inside that key padding and real dead
            // entries interleave (the order is wave, item, lane), so a position below the block's share of the queue can hold
            // padding: it carries kPadSlot, which the reader below treats as a dead slot -- slot 0 must not be shaded for it
            uint32_t sl = kPadSlot, key = kKeys - 1u;
            if (i < count)
            {
                sl = wf.queue[qin][i];
                const uint32_t pr = wf.hitPair[sl];
                if (pr == 0xffffffffu)
                    key = 0u;
                else if (pr != kDeadPair)
                {
                    const uint32_t type = sv.pairs[pr].materialId & 0xffu;
                    key = type <= PTX_MATERIAL_TYPE_PHONG ? ((sv.pairs[pr].flags & kPairTextured) ? 5u : 1u) + type : 4u;
                }
            }
            mySlot[item] = sl;
            myKey[item] = key;
            for (uint32_t k = 0; k < kKeys; k++)
                cnt[k] += (uint32_t)__popcll(__ballot(key == k));
        }
        if (lane < kKeys)
        {
            uint32_t c = 0;
            for (uint32_t k = 0; k < kKeys; k++) // no dynamic register indexing
                c = lane == k ? cnt[k] : c;
            s_keyCount[wave][lane] = c;
        }
        __syncthreads();
        if (threadIdx.x == 0)
        {
            uint32_t run = 0;
            for (uint32_t k = 0; k < kKeys; k++)
                for (uint32_t w = 0; w < kWaves; w++)
                {
                    s_keyBase[w][k] = run;
                    run += s_keyCount[w][k];
                }
        }
        __syncthreads();
        uint32_t done[kKeys];
        for (uint32_t k = 0; k < kKeys; k++)
            done[k] = s_keyBase[wave][k];
        for (uint32_t item = 0; item < kShadeItems; item++)
            for (uint32_t k = 0; k < kKeys; k++)
            {
                const uint64_t m = __ballot(myKey[item] == k);
                if (myKey[item] == k)
                    s_sorted[done[k] + (uint32_t)__popcll(m & lower)] = mySlot[item];
                done[k] += (uint32_t)__popcll(m);
            }
        __syncthreads();
      }
      uint32_t slots[kShadeItems];
      uint32_t pushBits = 0; // bit 2k: entry k joins the shadow queue, bit 2k+1: the next queue
#pragma nounroll
      for (uint32_t item = 0; item < kShadeItems; item++)
      {
        const uint32_t i = base + item * blockDim.x + threadIdx.x;
        bool pushNext = false, pushShadow = false, restart = false;
        uint32_t slot = 0, pair = kDeadPair;
        if (i < count)
        {
            // entries past the block's share of the queue were sorted last, with the dead slots
            slot = ctl.sortShade ? s_sorted[item * blockDim.x + threadIdx.x] : wf.queue[qin][i];
            if (slot != kPadSlot)
                pair = wf.hitPair[slot];
            else
                slot = 0;
        }
        if (pair != kDeadPair)
        {
            uint4 meta = wf.meta[slot];
            const float4 hit = wf.hit[slot];
            // first bounce of a sample: throughput = 1 (raygen.rgen:52); and of the first sample: radiance = 0 (:42)
            float4 r4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), t4 = make_float4(1.0f, 1.0f, 1.0f, 0.0f);
            if (meta.z != 0u)
                r4 = wf.rad[slot];
            if ((meta.z & 0xffffu) != 0u)
                t4 = wf.thr[slot];
            f3 radiance = F3(r4.x, r4.y, r4.z), throughput = F3(t4.x, t4.y, t4.z);

            if (pair == 0xffffffffu)
            {
                // miss.rmiss:16-39: sky colour / skybox lookup, Pdf = -1 -> raygen.rgen:71-75
                const float4 d4 = wf.rayD[slot];
                radiance = radiance + throughput * missEmissive(sv, F3(d4.x, d4.y, d4.z));
                restart = finishSample(p, wf, slot, meta, radiance, nSamples, nRetries);
            }
            else
            {
                const float4 o4 = wf.rayO[slot], d4 = wf.rayD[slot];
                HitOut out;
                DiffRays diff;
                if (TEX)
                    diff = loadDiff(wf, slot);
                Decal decal = noDecal();
                if (TEX && wf.decalT)
                {
                    decal.dist = wf.decalT[slot];
                    if (decal.dist != -1.0f)
                    {
                        const float4 dq = wf.decal[slot];
                        decal.slot = __float_as_uint(dq.x);
                        decal.u = dq.y;
                        decal.v = dq.z;
                        decal.pair = __float_as_uint(dq.w);
                    }
                }
                closestHit<TEX>(sv, F3(d4.x, d4.y, d4.z), hit.x, hit.y, hit.z, pair, __float_as_uint(hit.w), o4.w, meta.x, out, diff, decal);
                nHits++;

                radiance = radiance + throughput * out.Emissive; // raygen.rgen:77

                // raygen.rgen:79-81, evaluated with the pre-update throughput
                f3 contribution = F3s(0.0f);
                if (out.DirectLightPdf > 0.0f)
                {
                    contribution = (throughput * out.DirectLight) / out.DirectLightPdf;
                    // adding an exact zero cannot change radiance (it is never -0): skip the query
                    pushShadow = !(contribution.x == 0.0f && contribution.y == 0.0f && contribution.z == 0.0f);
                }

                if (out.Pdf > 0.001f) // :83-84
                    throughput = throughput * (out.Bsdf / out.Pdf);

                bool finished = false;
                const float prob = fmin_(maxComponent(throughput), 1.0f); // :86
                uint32_t bounce = meta.z & 0xffffu;
                if (prob < 0.001f)
                    finished = true;
                else if (prob < rnd(meta.x)) // :90
                    finished = true;
                else
                {
                    throughput = throughput / prob; // :93
                    bounce = bounce + 1;
                    if (bounce >= p.u.BounceCount)
                        finished = true;
                }
                meta.z = (meta.z & 0xffff0000u) | bounce;

                if (pushShadow)
                {
                    const f3 sd = -normalize(out.LightDirection); // raygen.rgen:24
                    wf.shO[slot] = make_float4(out.Position.x, out.Position.y, out.Position.z, out.LightDistance);
                    wf.shD[slot] = make_float4(sd.x, sd.y, sd.z, finished ? 1.0f : 0.0f);
                    wf.shC[slot] = make_float4(contribution.x, contribution.y, contribution.z, 0.0f);
                }
                if (finished && !pushShadow)
                    restart = finishSample(p, wf, slot, meta, radiance, nSamples, nRetries);
                else
                {
                    wf.rad[slot] = make_float4(radiance.x, radiance.y, radiance.z, 0.0f);
                    if (!finished)
                    {
                        wf.rayO[slot] = make_float4(out.Position.x, out.Position.y, out.Position.z, out.MaxRoughness);
                        wf.rayD[slot] = make_float4(out.Direction.x, out.Direction.y, out.Direction.z, 0.0f);
                        wf.thr[slot] = make_float4(throughput.x, throughput.y, throughput.z, 0.0f);
                        if (TEX)
                            storeDiff(wf, slot, diff);
                        pushNext = true; // a pending shadow query only adds to rad[slot] before the next bounce
                    }
                }
            }
            wf.meta[slot] = meta;
        }
        pushRestarts(wf, restart, slot);
        // queue appends with ONE global atomic per block and queue: same-address atomics
        // serialise at ~11 ns each on MI355X, so per-wave appends would cost more than the shading
        for (uint32_t k = 0; k < kShadeItems; k++) // no dynamic register indexing
            if (k == item)
                slots[k] = slot;
        pushBits |= (pushShadow ? 1u : 0u) << (2 * item) | (pushNext ? 2u : 0u) << (2 * item);
      }
        // queue appends with ONE global atomic per block and queue for all kShadeItems x 256 entries
        if (threadIdx.x < 2)
            s_cnt[threadIdx.x] = 0;
        __syncthreads();
        const uint32_t lane = threadIdx.x & 63u;
        uint32_t waveS[kShadeItems], waveN[kShadeItems];
        for (uint32_t k = 0; k < kShadeItems; k++)
        {
            const uint64_t maskS = __ballot((pushBits >> (2 * k)) & 1u), maskN = __ballot((pushBits >> (2 * k)) & 2u);
            uint32_t ws = 0, wn = 0;
            if (lane == 0)
            {
                if (maskS)
                    ws = atomicAdd(&s_cnt[0], (uint32_t)__popcll(maskS));
                if (maskN)
                    wn = atomicAdd(&s_cnt[1], (uint32_t)__popcll(maskN));
            }
            waveS[k] = __shfl(ws, 0);
            waveN[k] = __shfl(wn, 0);
        }
        __syncthreads();
        if (threadIdx.x < 2 && s_cnt[threadIdx.x])
            s_base[threadIdx.x] = atomicAdd(&wf.counters[threadIdx.x == 0 ? shadowSet : queueCounter(qout)], s_cnt[threadIdx.x]);
        __syncthreads();
        const uint64_t below = (1ull << lane) - 1ull;
        for (uint32_t k = 0; k < kShadeItems; k++)
        {
            const uint64_t maskS = __ballot((pushBits >> (2 * k)) & 1u), maskN = __ballot((pushBits >> (2 * k)) & 2u);
            if ((pushBits >> (2 * k)) & 1u)
                wf.shadowQueue[s_base[0] + waveS[k] + (uint32_t)__popcll(maskS & below)] = slots[k];
            if ((pushBits >> (2 * k)) & 2u)
                wf.queue[qout][s_base[1] + waveN[k] + (uint32_t)__popcll(maskN & below)] = slots[k];
        }
    }
    blockAddCounter(&wf.counters[C_HITS], nHits);
    blockAddCounter(&wf.counters[C_SAMPLES], nSamples);
    blockAddCounter(&wf.counters[C_RETRIES], nRetries);
}

struct ShadowIO
{
    static constexpr float kFixedTmin = 0.00001f; // raygen.rgen:26
    static constexpr float kFixedTmax = -1.0f;    // per ray: the distance to the light
    static constexpr bool kNeedsPrim = false;
    static constexpr bool kHasQueue = true;
    const Wavefront &wf;
    float finished;
    uint32_t staged;
    PT_DEV uint32_t queueEntry(uint32_t item) const { return wf.shadowQueue[item]; }
    PT_DEV void setEntry(uint32_t s) { staged = s; }
    PT_DEV bool load(uint32_t item, f3 &o, f3 &d, float &tmin, float &tmax)
    {
        const uint32_t slot = staged;
        const float4 o4 = wf.shO[slot], d4 = wf.shD[slot];
        o = F3(o4.x, o4.y, o4.z);
        d = F3(d4.x, d4.y, d4.z);
        tmin = 0.00001f; // raygen.rgen:26-31: tmin = 1e-5, tmax = LightDistance, terminate on first hit
        tmax = o4.w;
        finished = d4.w;
        return true;
    }
    PT_DEV void ignored(float, float, float, uint32_t, const TraceScene &) {} // shadow rays keep no decal
    PT_DEV void improve(uint32_t, float, float, float, uint32_t) {}
    PT_DEV uint32_t bestSlot(uint32_t) const { return 0u; }
    // The traversal only records the answer.  What follows from it -- the NEE add into rad[slot], finishing the sample of a
    // path that ended on this bounce -- is k_apply_shadow's: inside the traversal loop those dependent loads and stores
    // sat in the retire phase of nearly every round for a handful of lanes (shadow rounds took 1.8x a closest round).
    PT_DEV void store(uint32_t item, const Hit &, bool occluded, bool) { wf.shadowResult[item] = (uint8_t)((occluded ? 0u : 1u) | (finished != 0.0f ? 2u : 0u)); }
};

template <bool ALPHA>
PT_DEV void traceShadowBody(const LaunchParams &p, const TraceScene &sc, const Wavefront &wf, int qout, int parity)
{
    const uint32_t count = wf.counters[shadowCounter(parity)];
    if (count == 0u)
        return;
    PT_DECLARE_STACK(st, kLdsStack, wf.spill)
    ShadowIO io = { wf, 0.0f, 0u };
    persistentTrace<true, ALPHA>(sc, io, count, &wf.counters[shadowChunkCounter(parity)], st);
    if (st.overflow)
        atomicAdd(&wf.counters[C_OVERFLOW], 1u);
    (void)p;
    (void)qout;
}

// raygen.rgen:79-81 after the occlusion query, one thread per shadow queue entry: a visible light adds the contribution
// k_shade prepared; a path that ended on this bounce is finished (its slot may be due a new sample: restart queue).
__global__ void __launch_bounds__(kBlock) k_apply_shadow(LaunchParams p, Wavefront wf, int parity)
{
    const uint32_t count = wf.counters[shadowCounter(parity)];
    uint32_t nSamples = 0, nRetries = 0;
    for (uint32_t base = blockIdx.x * blockDim.x; base < count; base += gridDim.x * blockDim.x)
    {
        const uint32_t item = base + threadIdx.x;
        bool restart = false;
        uint32_t slot = 0;
        if (item < count)
        {
            const uint32_t result = wf.shadowResult[item];
            slot = wf.shadowQueue[item];
            if (result)
            {
                float4 r4 = wf.rad[slot];
                if (result & 1u)
                {
                    const float4 c = wf.shC[slot];
                    r4.x = r4.x + c.x;
                    r4.y = r4.y + c.y;
                    r4.z = r4.z + c.z;
                }
                if (result & 2u)
                {
                    uint4 meta = wf.meta[slot];
                    f3 radiance = F3(r4.x, r4.y, r4.z);
                    restart = finishSample(p, wf, slot, meta, radiance, nSamples, nRetries);
                    if (restart)
                        wf.meta.p[slot].z = meta.z;
                }
                else
                    wf.rad[slot] = r4; // the slot is already in the next queue (k_shade)
            }
        }
        // the slot cannot join the next queue directly: k_trace_closest of the next bounce may already be consuming it
        pushRestarts(wf, restart, slot);
    }
    blockAddCounter(&wf.counters[C_SAMPLES], nSamples);
    blockAddCounter(&wf.counters[C_RETRIES], nRetries);
}
template <bool ALPHA>
__global__ void __launch_bounds__(kBlock) k_trace_shadow(LaunchParams p, TraceScene sc, Wavefront wf, int qout, int parity);
template <>
__global__ void __launch_bounds__(kBlock) PT_FULL_OCCUPANCY k_trace_shadow<false>(LaunchParams p, TraceScene sc, Wavefront wf, int qout, int parity)
{
    traceShadowBody<false>(p, sc, wf, qout, parity);
}
template <>
__global__ void __launch_bounds__(kBlock) PT_ALPHA_SHADOW_ATTR k_trace_shadow<true>(LaunchParams p, TraceScene sc, Wavefront wf, int qout, int parity)
{
    traceShadowBody<true>(p, sc, wf, qout, parity);
}

// The second half of finishSample for the slots the shadow kernel re-queued: next primary ray, RNG carried on.
__global__ void __launch_bounds__(kBlock) k_restart(LaunchParams p, Wavefront wf, uint32_t count)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x)
    {
        const uint32_t slot = wf.restartQueue[i];
        uint4 meta = wf.meta[slot];
        f3 o, d;
        startSlotSample(p, wf, slot, meta.y, meta.x, o, d);
        wf.rayO[slot] = make_float4(o.x, o.y, o.z, 0.0f);
        wf.rayD[slot] = make_float4(d.x, d.y, d.z, 0.0f);
        wf.meta[slot] = meta; // bounce 0: thr[slot] = 1 is implied
    }
}

// raygen.rgen:115-117 for `frames` launches in frame order: bit-identical to issuing the
// launches one after another.
// pendingRestarts: counter of slots still waiting for another sample (multi-sample launch, NaN restart): their slotRad is
// not final, the host runs further rounds and accumulates afterwards
// shardMajor: `image` is the dense tile-major shard buffer of this rank (ptx_bind_shard_accumulation: the message of the gather is
// accumulated in place, entry = slot inside the frame, k_pack_shard's layout) instead of the row-major frame.
__global__ void __launch_bounds__(kBlock) k_accumulate(LaunchParams p, const float4 *__restrict__ slotRad, float4 *__restrict__ image,
                                                        const uint32_t *__restrict__ pendingRestarts, uint32_t shardMajor)
{
    if (pendingRestarts && *pendingRestarts != 0u)
        return;
    // One thread per slot of a frame group: a wave reads the 64 slots of its wave-chunk with one coalesced load (8 pixels x 8
    // frames when framesPerWave = 8), then the first lane of each pixel adds its frames in frame order through lane shuffles.
    const uint32_t g = p.framesPerWave, pixelsPerWave = 64u / g, groups = p.frames / g;
    const uint32_t chunks = p.slotsPerFrame / pixelsPerWave;
    const uint32_t lane = threadIdx.x & 63u, sub = lane % g;
    for (uint32_t base = (blockIdx.x * blockDim.x + threadIdx.x) & ~63u; base < chunks * 64u; base += gridDim.x * blockDim.x)
    {
        const uint32_t chunk = base >> 6, s = chunk * pixelsPerWave + lane / g;
        const uint32_t pixel = slotPixel(p, s);
        const bool owner = sub == 0u && pixel != 0xffffffffu;
        const uint32_t at = shardMajor ? s : pixel;
        float4 acc = owner ? image[at] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        for (uint32_t group = 0; group < groups; group++)
        {
            const float4 r = slotRad[((size_t)(group * chunks + chunk) << 6) + lane];
            for (uint32_t k = 0; k < g; k++) // every lane runs the shuffles; only the owners' sums are kept
            {
                const float rx = __shfl(r.x, (int)(lane + k)), ry = __shfl(r.y, (int)(lane + k)), rz = __shfl(r.z, (int)(lane + k));
                acc.x = rx + acc.x;
                acc.y = ry + acc.y;
                acc.z = rz + acc.z;
            }
        }
        if (owner)
        {
            acc.w = 1.0f;
            image[at] = acc;
        }
    }
}

// =====================================================================================
// Fused path loop: raygen.rgen:36-118 as one device function.  Used by
//   * k_megakernel  -- one thread per slot from the first sample (bring-up / A-B reference)
//   * k_tail        -- finishes the paths still alive once the wavefront has thinned out:
//                      late bounces have few rays and every per-bounce kernel then costs the
//                      latency of its LONGEST ray (~0.4 ms measured) whatever the ray count
// =====================================================================================

struct PathCounters
{
    uint32_t nSeg = 0, nHit = 0, nSmp = 0, nRetry = 0;
    bool stuck = false; // some path never produced a finite sample and was given up (kMaxSampleRetries)
};
// A slot whose samples keep coming out NaN / Inf would spin for ever (it hangs the GPU in the reference): after this many
// restarts in a row it is given up with radiance 0 and the launch reports an error.
constexpr uint32_t kMaxSampleRetries = 256;

// Runs a slot to the end of its launch -- or, with ONE_SAMPLE, to the end of the sample it is in (smpl then tells
// the caller whether samples remain).  `fresh` = start with a new sample (primary ray); otherwise continue the
// current sample at `bounce` with the given ray / throughput.
// MODE 0: opaque geometry, fixed 1x1 textures; 1: + ray differentials and the sampler (TEX); 2: + any-hit stages (ALPHA)
template <int MODE, bool ONE_SAMPLE = false>
PT_DEV f3 runPath(const LaunchParams &p, const SceneView &sv, const TraceScene &sc, Stack &st, uint32_t pixel, uint32_t &rng,
                  f3 radiance, f3 throughput, f3 ro, f3 rd, DiffRays diff, float maxRoughness, uint32_t bounce, int &smpl, bool fresh,
                  PathCounters &pc)
{
    constexpr bool TEX = MODE >= 1, ALPHA = MODE == 2;
    uint32_t restartsInARow = 0;
    for (;;)
    {
        if (fresh)
        {
            if (ONE_SAMPLE || smpl >= (int)p.u.SampleCount) // ONE_SAMPLE (k_tail): no sample is ever started here
                break;
            throughput = F3s(1.0f);
            startSample<TEX>(p, pixel, rng, ro, rd, diff);
            maxRoughness = 0.0f;
            bounce = 0;
            fresh = false;
        }
        for (; bounce < p.u.BounceCount; bounce++)
        {
            Hit h;
            pc.nSeg++;
            Decal decal = noDecal();
            if (!traceRay<false, false, ALPHA>(sc, ro, rd, 0.00001f, 10000.0f, st, h, nullptr, nullptr, &decal))
            {
                radiance = radiance + throughput * missEmissive(sv, rd);
                break;
            }
            HitOut out;
            closestHit<TEX>(sv, rd, h.t, h.u, h.v, h.pair, h.slot, maxRoughness, rng, out, diff, decal);
            pc.nHit++;
            maxRoughness = out.MaxRoughness;
            radiance = radiance + throughput * out.Emissive;
            if (out.DirectLightPdf > 0.0f)
            {
                const f3 c = (throughput * out.DirectLight) / out.DirectLightPdf;
                if (!(c.x == 0.0f && c.y == 0.0f && c.z == 0.0f))
                {
                    Hit sh;
                    if (!traceRay<true, false, ALPHA>(sc, out.Position, -normalize(out.LightDirection), 0.00001f, out.LightDistance, st, sh))
                        radiance = radiance + c;
                }
            }
            if (out.Pdf > 0.001f)
                throughput = throughput * (out.Bsdf / out.Pdf);
            const float prob = fmin_(maxComponent(throughput), 1.0f);
            if (prob < 0.001f)
                break;
            if (prob < rnd(rng))
                break;
            throughput = throughput / prob;
            ro = out.Position;
            rd = out.Direction;
        }
        pc.nSmp++;
        if (badRadiance(radiance)) // raygen.rgen:99-112: restart ALL samples, RNG carried on
        {
            radiance = F3s(0.0f);
            smpl = 0;
            pc.nRetry++;
            if (++restartsInARow >= kMaxSampleRetries)
            {
                pc.stuck = true;
                smpl = (int)p.u.SampleCount;
            }
        }
        else
        {
            smpl++;
            restartsInARow = 0;
        }
        fresh = true;
    }
    return radiance;
}

template <int MODE>
__global__ void __launch_bounds__(kBlock) k_megakernel(LaunchParams p, SceneView sv, TraceScene sc, float4 *__restrict__ slotRad,
                                                        uint32_t *__restrict__ counters)
{
    PT_DECLARE_STACK(st, kLdsStackMega, (uint32_t *)nullptr)
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t f = 0, s = 0;
    if (slot < p.numSlots)
        slotFrame(p, slot, f, s);
    const uint32_t pixel = slot < p.numSlots ? slotPixel(p, s) : 0xffffffffu;
    PathCounters pc;
    f3 radiance = F3s(0.0f);
    if (pixel != 0xffffffffu)
    {
        uint32_t rng = initRng(pixel % p.width, pixel / p.width, p.width, p.firstFrame + f);
        DiffRays diff;
        diff.rxOrigin = diff.rxDirection = diff.ryOrigin = diff.ryDirection = F3s(0.0f);
        int smpl = 0;
        radiance = runPath<MODE>(p, sv, sc, st, pixel, rng, F3s(0.0f), F3s(1.0f), F3s(0.0f), F3s(0.0f), diff, 0.0f, 0u, smpl, true, pc);
    }
    if (slot < p.numSlots)
        slotRad[slot] = make_float4(radiance.x, radiance.y, radiance.z, 0.0f);
    if (st.overflow)
        atomicAdd(&counters[C_OVERFLOW], 1u);
    if (pc.stuck)
        atomicAdd(&counters[C_OVERFLOW + 1], 1u);
    waveAddCounter(&counters[C_SEGMENTS], pc.nSeg);
    waveAddCounter(&counters[C_HITS], pc.nHit);
    waveAddCounter(&counters[C_SAMPLES], pc.nSmp);
    waveAddCounter(&counters[C_RETRIES], pc.nRetry);
}

// The slots of queue `qin` sit at a bounce boundary (ray, throughput, radiance, RNG and
// bounce/sample counters in the SoA state, no shadow query pending): run each to the end of its sample.  A slot
// with samples left (multi-sample launch, NaN restart) goes back through the restart queue and the wavefront
// kernels: finishing ALL its samples here, one thread per path at 2 waves / SIMD, made a SampleCount = 8 launch six
// times slower than eight one-sample launches.
// k_tail is latency-bound at whatever occupancy it gets: a 32-entry LDS stack (overflow into the global region of the
// traversal kernels) instead of 64 entries lifts the LDS limit of two blocks per CU, and 168 VGPRs (7 spilled dwords in
// mode 0) make it three waves per SIMD: 1.50 -> 1.08 ms per chess_like step
#ifndef PT_TAIL_ATTR
#define PT_TAIL_ATTR __attribute__((amdgpu_waves_per_eu(3, 3)))
#endif
#ifndef PT_TAIL_LDS
#define PT_TAIL_LDS 32
#endif
// the textured tails (k_tail<1>, <2>, k_finish_restarts) at TWO waves per SIMD: at three they spilled 63-105 VGPRs into 148-188
// bytes of scratch per lane, and both settings measure the same (round 4: atrium_like's 1 / 8 shard 2.88-2.91 ms per step, configs[3]
// 827-835 Msamples/s either way) -- no scratch, then
#ifndef PT_TAIL_TEX_ATTR
#define PT_TAIL_TEX_ATTR __attribute__((amdgpu_waves_per_eu(2, 2)))
#endif
template <int MODE>
PT_DEV void tailBody(const LaunchParams &p, const SceneView &sv, const TraceScene &sc, const Wavefront &wf, int qin, const BounceCtl &ctl);
template <int MODE>
__global__ void __launch_bounds__(kBlock) k_tail(LaunchParams p, SceneView sv, TraceScene sc, Wavefront wf, int qin, BounceCtl ctl);
template <>
__global__ void __launch_bounds__(kBlock) PT_TAIL_ATTR k_tail<0>(LaunchParams p, SceneView sv, TraceScene sc, Wavefront wf, int qin, BounceCtl ctl)
{
    tailBody<0>(p, sv, sc, wf, qin, ctl);
}
template <>
__global__ void __launch_bounds__(kBlock) PT_TAIL_TEX_ATTR k_tail<1>(LaunchParams p, SceneView sv, TraceScene sc, Wavefront wf, int qin, BounceCtl ctl)
{
    tailBody<1>(p, sv, sc, wf, qin, ctl);
}
template <>
__global__ void __launch_bounds__(kBlock) PT_TAIL_TEX_ATTR k_tail<2>(LaunchParams p, SceneView sv, TraceScene sc, Wavefront wf, int qin, BounceCtl ctl)
{
    tailBody<2>(p, sv, sc, wf, qin, ctl);
}
// Launched after the shadow kernel of every bounce, on that kernel's stream (so the NEE adds of the bounce have landed
// in rad[slot]): it takes the queue over once it is short enough -- the complement of bounceRuns() for the bounces that
// follow, which then find the queue is not theirs and return.
template <int MODE>
PT_DEV void tailBody(const LaunchParams &p, const SceneView &sv, const TraceScene &sc, const Wavefront &wf, int qin, const BounceCtl &ctl)
{
    const uint32_t count = wf.counters[queueCounter(qin)];
    if (count == 0u || count > ctl.tailBelow) // ctl.bounce = the bounce whose shade kernel filled the queue
        return;
    if (blockIdx.x == 0 && threadIdx.x == 0)
    {
        wf.counters[C_TAIL_PATHS] = count;
        wf.counters[C_TAIL_PATHS + 1] = ctl.bounce;
    }
    PT_DECLARE_STACK(st, PT_TAIL_LDS, wf.spill)
    PathCounters pc;
    // (Dealing the paths to every 2nd / 4th / 8th lane -- a wave runs each bounce for as long as its slowest path takes, so fewer
    // paths per wave shorten every wave's chain and put more waves on a SIMD -- was measured: with frames in flight the lanes
    // it wastes are not free.  chess_like whole frame 7.36 / 7.12 -> 7.39 / 7.26 ms per step, a rank's shard of 8 1.29 / 1.27 ->
    // 1.35 / 1.34, of 4 2.07 -> 2.23, street_like's shard of 8 1.96 -> 2.04.)
    for (uint32_t base = blockIdx.x * blockDim.x; base < count; base += gridDim.x * blockDim.x)
    {
        const uint32_t i = base + threadIdx.x;
        bool restart = false;
        uint32_t restartSlot = 0;
        if (i < count)
        {
            const uint32_t slot = wf.queue[qin][i];
            const uint4 meta = wf.meta[slot];
            const float4 o4 = wf.rayO[slot], d4 = wf.rayD[slot];
            float4 r4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), t4 = make_float4(1.0f, 1.0f, 1.0f, 0.0f); // see k_shade
            if (meta.z != 0u)
                r4 = wf.rad[slot];
            if ((meta.z & 0xffffu) != 0u)
                t4 = wf.thr[slot];
            uint32_t rng = meta.x;
            DiffRays diff;
            if (MODE >= 1)
                diff = loadDiff(wf, slot);
            else
                diff.rxOrigin = diff.rxDirection = diff.ryOrigin = diff.ryDirection = F3s(0.0f);
            int smpl = (int)(meta.z >> 16);
            const f3 radiance = runPath<MODE, true>(p, sv, sc, st, meta.y, rng, F3(r4.x, r4.y, r4.z), F3(t4.x, t4.y, t4.z), F3(o4.x, o4.y, o4.z),
                                                   F3(d4.x, d4.y, d4.z), diff, o4.w, meta.z & 0xffffu, smpl, false, pc);
            if (smpl < (int)p.u.SampleCount)
            {
                wf.rad[slot] = make_float4(radiance.x, radiance.y, radiance.z, 0.0f);
                wf.meta[slot] = make_uint4(rng, meta.y, (uint32_t)smpl << 16, meta.w);
                restart = true;
                restartSlot = slot;
            }
            else
                wf.slotRad[slot] = make_float4(radiance.x, radiance.y, radiance.z, 0.0f);
        }
        pushRestarts(wf, restart, restartSlot); // k_restart draws the next primary ray before the queue is consumed
    }
    if (st.overflow)
        atomicAdd(&wf.counters[C_OVERFLOW], 1u);
    waveAddCounter(&wf.counters[C_SEGMENTS], pc.nSeg);
    waveAddCounter(&wf.counters[C_HITS], pc.nHit);
    waveAddCounter(&wf.counters[C_SAMPLES], pc.nSmp);
    waveAddCounter(&wf.counters[C_RETRIES], pc.nRetry);
}

// The rare slots whose sample came out NaN / Inf in a canonical (SampleCount = 1) launch: raygen.rgen:99-112 restarts
// the sample with the RNG carried on.  They sit in the restart queue with radiance 0 and smpl = 0; this kernel runs each
// of them to the end of the launch the way the megakernel would (new primary ray, whole path, again if the radiance is
// bad again), so that a launch completes on the device without the host looking at the queue.  Multi-sample launches do
// NOT come here: their restart queue holds every slot once per extra sample, and goes through the wavefront kernels
// round by round (renderImpl).
template <int MODE>
__global__ void __launch_bounds__(kBlock) PT_TAIL_TEX_ATTR k_finish_restarts(LaunchParams p, SceneView sv, TraceScene sc, Wavefront wf)
{
    const uint32_t count = wf.counters[C_RESTART];
    if (count == 0u)
        return;
    PT_DECLARE_STACK(st, PT_TAIL_LDS, wf.spill)
    PathCounters pc;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x)
    {
        const uint32_t slot = wf.restartQueue[i];
        const uint4 meta = wf.meta[slot];
        const float4 r4 = wf.rad[slot];
        uint32_t rng = meta.x;
        DiffRays diff;
        diff.rxOrigin = diff.rxDirection = diff.ryOrigin = diff.ryDirection = F3s(0.0f);
        int smpl = (int)(meta.z >> 16);
        const f3 radiance = runPath<MODE>(p, sv, sc, st, meta.y, rng, F3(r4.x, r4.y, r4.z), F3s(1.0f), F3s(0.0f), F3s(0.0f), diff, 0.0f, 0u, smpl,
                                          true, pc);
        wf.slotRad[slot] = make_float4(radiance.x, radiance.y, radiance.z, 0.0f);
    }
    if (st.overflow)
        atomicAdd(&wf.counters[C_OVERFLOW], 1u);
    if (pc.stuck)
        atomicAdd(&wf.counters[C_OVERFLOW + 1], 1u);
    waveAddCounter(&wf.counters[C_SEGMENTS], pc.nSeg);
    waveAddCounter(&wf.counters[C_HITS], pc.nHit);
    waveAddCounter(&wf.counters[C_SAMPLES], pc.nSmp);
    waveAddCounter(&wf.counters[C_RETRIES], pc.nRetry);
}

// =====================================================================================
// Utility kernels
// =====================================================================================

// skinning.comp:21-50: 4-bone linear-blend skinning of one output vertex.  bones = mat3x4[]: "vec4 * mat3x4" is the
// dot product with each stored row, i.e. a bone is the affine matrix in 3 rows x 4; the normal goes through the
// inverse transpose of its linear part.
__global__ void k_skin(const PtxAnimatedVertex *__restrict__ in, const uint32_t *__restrict__ source, uint32_t count,
                       const PtxTransform *__restrict__ bones, uint32_t boneCount, PtxVertex *__restrict__ out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count)
        return;
    const PtxAnimatedVertex a = in[source[i]];
    f3 P = F3s(0.0f), N = F3s(0.0f), T = F3s(0.0f), B = F3s(0.0f);
    float totalWeight = 0;
    for (int k = 0; k < 4 && totalWeight < 1.0f; k++)
    {
        const uint32_t boneIndex = a.BoneIndices[k];
        const float w = a.BoneWeights[k];
        float M[12] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0 };
        if (boneIndex < boneCount)
            for (int q = 0; q < 12; q++)
                M[q] = bones[boneIndex].m[q];
        P = P + xformPoint(M, ld3(a.Position)) * w;
        T = T + normalize(xformVector(M, ld3(a.Tangent))) * w;
        B = B + normalize(xformVector(M, ld3(a.Bitangent))) * w;
        mat3 R;
        R.c0 = F3(M[0], M[4], M[8]);
        R.c1 = F3(M[1], M[5], M[9]);
        R.c2 = F3(M[2], M[6], M[10]);
        const mat3 Ri = inverse(R);
        const f3 n = ld3(a.Normal);
        N = N + normalize(F3(dot(n, Ri.c0), dot(n, Ri.c1), dot(n, Ri.c2))) * w;
        totalWeight += w;
    }
    PtxVertex o;
    o.Position[0] = P.x; o.Position[1] = P.y; o.Position[2] = P.z;
    o.TexCoords[0] = a.TexCoords[0]; o.TexCoords[1] = a.TexCoords[1];
    o.Normal[0] = N.x; o.Normal[1] = N.y; o.Normal[2] = N.z;
    o.Tangent[0] = T.x; o.Tangent[1] = T.y; o.Tangent[2] = T.z;
    o.Bitangent[0] = B.x; o.Bitangent[1] = B.y; o.Bitangent[2] = B.z;
    out[i] = o;
}

// traceRayEXT stand-in over explicit rays (o.xyz, tmin, d.xyz, tmax): traversal parity tests
struct RaysIO
{
    static constexpr float kFixedTmin = -1.0f, kFixedTmax = -1.0f; // per ray
    static constexpr bool kNeedsPrim = true;     // ptx_trace_rays reports (pair, prim)
    static constexpr bool kHasQueue = false;
    PT_DEV uint32_t queueEntry(uint32_t item) const { return item; }
    PT_DEV void setEntry(uint32_t) {}
    const float4 *rays;
    float4 *outHit;
    uint2 *outIds;
    PT_DEV bool load(uint32_t item, f3 &o, f3 &d, float &tmin, float &tmax)
    {
        const float4 o4 = rays[2 * item], d4 = rays[2 * item + 1];
        o = F3(o4.x, o4.y, o4.z);
        d = F3(d4.x, d4.y, d4.z);
        tmin = o4.w;
        tmax = d4.w;
        return true;
    }
    PT_DEV void ignored(float, float, float, uint32_t, const TraceScene &) {}
    // closest-hit queries: (u, v) and the triangle of the best hit so far wait in the output record
    PT_DEV void improve(uint32_t item, float t, float u, float v, uint32_t triSlot) { outHit[item] = make_float4(t, u, v, __uint_as_float(triSlot)); }
    PT_DEV uint32_t bestSlot(uint32_t item) const { return __float_as_uint(outHit[item].w); }
    PT_DEV void store(uint32_t item, const Hit &h, bool hitAny, bool anyHitQuery)
    {
        const float4 cur = outHit[item];
        const bool kept = hitAny && !anyHitQuery; // improve() has written (u, v)
        outHit[item] = make_float4(h.t, kept ? cur.y : 0.0f, kept ? cur.z : 0.0f, hitAny ? 1.0f : 0.0f);
        outIds[item] = make_uint2(h.pair, h.prim);
    }
};

template <bool ALPHA>
__global__ void __launch_bounds__(kBlock) k_trace_rays(TraceScene sc, const float4 *__restrict__ rays, uint32_t n, int anyHit,
                                                        float4 *__restrict__ outHit, uint2 *__restrict__ outIds, uint32_t *chunkCounter, uint32_t *spill)
{
    PT_DECLARE_STACK(st, kLdsStack, spill)
    if (anyHit == 2) // diagnostics: closest hit, returning (node visits, triangle tests) instead of ids
    {
        for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        {
            const float4 o = rays[2 * i], d = rays[2 * i + 1];
            Hit h;
            uint32_t nv = 0, nt = 0;
            const bool hitAny = traceRay<false, true, ALPHA>(sc, F3(o.x, o.y, o.z), F3(d.x, d.y, d.z), o.w, d.w, st, h, &nv, &nt);
            outHit[i] = make_float4(h.t, h.u, h.v, hitAny ? 1.0f : 0.0f);
            outIds[i] = make_uint2(nv, nt);
        }
        return;
    }
    RaysIO io = { rays, outHit, outIds };
    if (anyHit)
        persistentTrace<true, ALPHA>(sc, io, n, chunkCounter, st);
    else
        persistentTrace<false, ALPHA>(sc, io, n, chunkCounter, st);
}

// shard pack / unpack: tile-major dense buffer [ownedTile][tileSize^2] of RGBA32F
// the image to page-locked host memory with a few workgroups: posted writes over PCIe
__global__ void __launch_bounds__(kBlock) k_copy_out(const float4 *__restrict__ src, float4 *__restrict__ dst, uint32_t n)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        dst[i] = src[i];
}

__global__ void k_pack_shard(LaunchParams p, const float4 *__restrict__ image, float4 *__restrict__ dst)
{
    for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < p.slotsPerFrame; s += gridDim.x * blockDim.x)
    {
        const uint32_t pixel = slotPixel(p, s);
        dst[s] = pixel == 0xffffffffu ? make_float4(0, 0, 0, 0) : image[pixel];
    }
}
// host != nullptr: the same pixels also go straight to the page-locked frame of the host (ptx_unpack_shard_host: rank 0 of an
// N-GPU step hands the gathered frame to the host while it unpacks it -- no snapshot, no second pass over the image)
__global__ void k_unpack_shard(LaunchParams p, const float4 *__restrict__ src, float4 *__restrict__ image, float4 *__restrict__ host)
{
    for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < p.slotsPerFrame; s += gridDim.x * blockDim.x)
    {
        const uint32_t pixel = slotPixel(p, s);
        if (pixel != 0xffffffffu)
        {
            const float4 v = src[s];
            image[pixel] = v;
            if (host)
                host[pixel] = v;
        }
    }
}

// The owner of a gathered frame composes it from ALL the ranks' shards in ONE launch (ptx_unpack_shards; round 6: eight thin
// k_unpack_shard launches per step sat on the rank that was already the slowest).  One thread per PIXEL in row-major order, the
// source entry by the inverse of slotPixel: tile -> (rank = tile % world, position among that rank's tiles = tile / world), 8x8
// block and lane inside the tile.  The stores -- to the device image, to the host's page-locked frame, or both -- are one
// contiguous stream over the whole frame (a wave writes 1 KB); the loads are the 128-byte rows of the 8x8 blocks.
struct GatherParams
{
    uint32_t width, height, tileSize, tilesX, worldSize;
    uint32_t strideSlots; // float4 entries between the pieces of consecutive ranks in `src`
};
__global__ void __launch_bounds__(kBlock) k_gather_frame(GatherParams g, const float4 *__restrict__ src, float4 *__restrict__ image, float4 *__restrict__ host)
{
    const uint32_t n = g.width * g.height, ts = g.tileSize, bpr = ts / 8u, perTile = ts * ts;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    {
        const uint32_t y = i / g.width, x = i - y * g.width;
        const uint32_t tyq = y / ts, txq = x / ts, ty = y - tyq * ts, tx = x - txq * ts;
        const uint32_t tile = tyq * g.tilesX + txq;
        const uint32_t k = tile % g.worldSize, kt = tile / g.worldSize;
        const uint32_t o = ((ty >> 3) * bpr + (tx >> 3)) * 64u + (ty & 7u) * 8u + (tx & 7u);
        const float4 v = src[(size_t)k * g.strideSlots + kt * perTile + o];
        if (image)
            image[i] = v;
        if (host)
            host[i] = v;
    }
}

// What a tree costs the rays of a path tracer: node visits + triangle tests of closest-hit queries along `n` sampled rays.
// ptx_build_accel builds a few candidate trees and keeps the cheapest: "lower surface-area cost" does not always mean "fewer
// visits" (street_like: a wider PLOC search gives 13 % MORE visits per ray), and results never depend on the tree.
//   k_sample_segments   the rays, drawn ONCE per build from the first candidate's triangle array and kept: every candidate is
//                       priced on the same rays (leaf order differs from tree to tree, so indices into it would name other
//                       triangles -- sampling noise that could decide between candidates a per cent apart).  A ray leaves the
//                       centroid of a pseudo-random triangle in a cosine-distributed direction about its normal (either side) and
//                       runs until it lands -- a diffuse bounce, what most rays of a path are.  (Rounds 2-3 sampled the segment
//                       between the centroids of two random triangles: two triangles of one floor or wall give a ray IN that
//                       surface, which crosses every leaf box along it -- up to 265,000 visits for one street_like segment, 7 % of
//                       the sum over all 65,536 -- and no path ray does that; the candidates were being told apart by a dozen
//                       such outliers.)
//   k_sample_tree_cost  per ray visits + tests, one number per ray: the host takes the mean without the top 0.1 % and the tail
__global__ void k_sample_segments(TraceScene sc, uint32_t n, float4 *__restrict__ rays)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    float4 o = make_float4(0.0f, 0.0f, 0.0f, 0.0f), d = o; // d.w = 0: no ray
    if (sc.triCount > 1u)
    {
        uint32_t h = jenkinsHash(2u * i + 1u);
        const Tri t = sc.tris[h % sc.triCount];
        const f3 e1 = F3(t.a.w, t.b.x, t.b.y), e2 = F3(t.b.z, t.b.w, t.c.x);
        const float third = 1.0f / 3.0f;
        const f3 c = F3(t.a.x + (e1.x + e2.x) * third, t.a.y + (e1.y + e2.y) * third, t.a.z + (e1.z + e2.z) * third);
        f3 nrm = cross(e1, e2);
        const float len = __builtin_sqrtf(dot(nrm, nrm));
        if (len > 0.0f)
        {
            h = jenkinsHash(h);
            nrm = nrm * (((h & 1u) ? 1.0f : -1.0f) / len);
            const mat3 frame = computeTangentSpace(nrm);
            const f3 tx = frame.c0, ty = frame.c1;
            h = jenkinsHash(h);
            const float u1 = (float)(h >> 8) * (1.0f / 16777216.0f);
            h = jenkinsHash(h);
            const float u2 = (float)(h >> 8) * (1.0f / 16777216.0f);
            const float rr = __builtin_sqrtf(u1), phi = 6.2831853f * u2;
            const float lx = rr * __builtin_cosf(phi), ly = rr * __builtin_sinf(phi), lz = __builtin_sqrtf(fmaxf(0.0f, 1.0f - u1));
            f3 dir = tx * lx + ty * ly + nrm * lz;
            dir = dir * (1.0f / __builtin_sqrtf(dot(dir, dir)));
            const float size = __builtin_sqrtf(len); // ~ the triangle's edge length
            o = make_float4(c.x + nrm.x * 1e-3f * size, c.y + nrm.y * 1e-3f * size, c.z + nrm.z * 1e-3f * size, 1e-5f);
            d = make_float4(dir.x, dir.y, dir.z, 1e4f);
        }
    }
    rays[2 * i] = o;
    rays[2 * i + 1] = d;
}

template <bool ALPHA>
__global__ void __launch_bounds__(kBlock) k_sample_tree_cost(TraceScene sc, const float4 *__restrict__ rays, uint32_t n, uint32_t *spill,
                                                              uint32_t *__restrict__ perRay)
{
    PT_DECLARE_STACK(st, kLdsStack, spill)
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    uint32_t visits = 0, tests = 0;
    const float4 o = rays[2 * i], d = rays[2 * i + 1];
    if (d.w > 0.0f)
    {
        Hit best;
        traceRay<false, true, ALPHA>(sc, F3(o.x, o.y, o.z), F3(d.x, d.y, d.z), o.w, d.w, st, best, &visits, &tests);
    }
    perRay[i] = visits + tests; // one dependent fetch each
}

