// pt_bvh.hpp -- software LBVH for gfx950: builder kernels (Morton codes + hand-written
// LSD radix sort + Karras hierarchy + bottom-up refit) and the stack traversal used by
// the closest-hit and any-hit queries.
//
// Stands in for the driver's VK_KHR_acceleration_structure build
// (Path-Tracing/Renderer/AccelerationStructure.cpp:64-301) and for traceRayEXT
// (Shaders/raygen.rgen:31,68).  The reference gives only the INPUT layout (one BLAS per
// Model, one geometry per Mesh with an optional baked mesh transform, one TLAS instance
// per ModelInstance with instanceShaderBindingTableRecordOffset = MeshOffset); the
// algorithm is new.  MI355X-first choice: with 288 GB of HBM the instances are
// flattened into ONE world-space triangle soup and ONE tree -- no per-instance ray
// transform, no two-level walk.
//
// Layout in HBM
//   BvhNode   64 B  FOUR children: node origin + per-axis power-of-two scale, 8-bit
//                   quantised child boxes (conservative: they only grow), 4 child refs.
//                   One visit = 4 x dwordx4 for four boxes -- half the bytes per child of a
//                   binary node and half the depth.  Measured reason: with binary 64-B nodes
//                   the traversal kernels were bound by the per-CU address/L1 pipeline (every
//                   lane fetches its own node: 4 divergent dwordx4 per visit), not by HBM.
//   Tri       48 B  v0, e1, e2 (Moeller-Trumbore form) + (pair, prim) ids, in leaf order
// child ref >= 0: internal node index; < 0: leaf, ~ref = triangle slot (bits 0..29) | kLeafNonOpaque for a triangle of a
// non-opaque geometry (so that its any-hit record can be fetched WITH the triangle, not after it); kEmptyRef: no child.
#pragma once

#include "pt_device.hpp"

namespace ptd
{

constexpr int kNodeWidth = 4;
struct BvhNode
{
    float4 a; // origin.xyz, w = bits: biased exponents ex | ey << 8 | ez << 16 (scale = 2^(e-127))
    int4 refs; // child refs
    uint4 q0; // x = lo.x bytes of children 0..3, y = hi.x bytes, z = lo.y bytes, w = hi.y bytes
    uint4 q1; // x = lo.z bytes, y = hi.z bytes, z, w unused
};
static_assert(sizeof(BvhNode) == 64, "BvhNode is 64 B");
constexpr int kEmptyRef = 0x7ffffffe;

struct Tri
{
    float4 a; // v0.xyz, e1.x
    float4 b; // e1.yz, e2.xy
    float4 c; // e2.z, pair (bits), prim (bits), w (bits): bit 31 = non-opaque geometry (any-hit stages run); for those, bits
              // 0..14 / 15..29 = width - 1 / height - 1 of the base level of its alpha texture (k_alpha_tris)
};
static_assert(sizeof(Tri) == 48, "Tri is 48 B");
constexpr uint32_t kTriNonOpaque = 0x80000000u; // Tri::c.w
constexpr uint32_t kLeafNonOpaque = 0x40000000u, kLeafSlotMask = 0x3fffffffu; // ~ref of a leaf
constexpr uint32_t kMaxTriangles = 0x3fffffffu;

struct Hit
{
    float t, u, v;
    uint32_t pair, prim; // pair == 0xffffffff: miss
    uint32_t slot;       // triangle slot in leaf order (index into tris / shadeTris)
};

// ---------------------------------------------------------------------------------
// Builder
// ---------------------------------------------------------------------------------

PT_DEV uint32_t orderedFloat(float f) // monotone float -> uint map for atomicMin/Max
{
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
PT_DEV float unorderedFloat(uint32_t u)
{
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

// One thread per flattened triangle (global id g, in instance-then-mesh-then-primitive
// order): world-space vertices, Moeller-Trumbore edges, padded bounds, scene bounds.
__global__ void k_tri_setup(uint32_t n, uint32_t pairCount, const uint32_t *__restrict__ pairFirst,
                            const DevPair *__restrict__ pairs, const PtxVertex *__restrict__ vertices,
                            const uint32_t *__restrict__ indices, Tri *__restrict__ triTmp, float4 *__restrict__ boxLo,
                            float4 *__restrict__ boxHi, uint32_t *__restrict__ sceneBounds, uint8_t *__restrict__ inert, int refit)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n)
        return;
    uint32_t lo = 0, hi = pairCount; // last pair with pairFirst[p] <= g
    while (hi - lo > 1)
    {
        const uint32_t mid = (lo + hi) >> 1;
        if (pairFirst[mid] <= g)
            lo = mid;
        else
            hi = mid;
    }
    const uint32_t p = lo, prim = g - pairFirst[p];
    const DevPair *pr = &pairs[p];
    f3 w[3];
    for (int k = 0; k < 3; k++)
    {
        const uint32_t idx = indices[pr->indexOffset + prim * 3 + k];
        w[k] = xformPoint(pr->M, ld3(vertices[pr->vertexOffset + idx].Position));
    }
    f3 e1 = w[1] - w[0], e2 = w[2] - w[0];
    // A zero-area triangle (exactly vanishing edge cross product: repeated or collinear vertices) is never hit, as in
    // Vulkan.  Otherwise det = e1 . (d x e2) is a rounding residue instead of 0 and the test reports a meaningless t
    // (found by the full-size sweep on atrium_like: e1 == e2, "hit" at t = 16 for a ray passing the vertex at 29.65).
    bool isInert;
    {
        const f3 n = cross(e1, e2);
        isInert = n.x == 0.0f && n.y == 0.0f && n.z == 0.0f;
        if (isInert)
            e1 = e2 = F3s(0.0f);
    }
    // Inert triangles take no part in the tree: a full build sorts them behind the others (k_morton) and builds over the
    // rest.  (Kept in the tree as point boxes they cannot be hit either, but the rings of them at the poles of lathed
    // meshes tie in PLOC's area order and merge one pair per iteration: 346 instead of 107 iterations for chess_like.)
    // A refit keeps the order of the last full build: a triangle that has gone inert since stays where it is, unhittable;
    // one that has come to life is not in the tree, and the caller has to rebuild (sceneBounds[7]).
    if (!refit)
        inert[g] = isInert ? 1 : 0;
    else if (inert[g] && !isInert)
        sceneBounds[7] = 1u;
    Tri t;
    t.a = make_float4(w[0].x, w[0].y, w[0].z, e1.x);
    t.b = make_float4(e1.y, e1.z, e2.x, e2.y);
    t.c = make_float4(e2.z, __uint_as_float(p), __uint_as_float(prim), __uint_as_float((pr->flags & kPairNonOpaque) ? kTriNonOpaque : 0u));
    triTmp[g] = t;

    // bounds from the same p0, p0+e1, p0+e2 the intersection test sees, padded so the slab test does not reject a
    // ray the triangle test accepts: 1e-5 of the coordinates (position of the hit point along the ray) + 5e-4 of the
    // triangle's extent per axis (the two-pass triangle test is good to ~1e-4 of the size; per axis because that error
    // moves the accepted point WITHIN the triangle's plane -- padding every axis by the largest extent makes
    // floor-grazing shadow rays start inside their neighbours' boxes: -4 %).  Same rule as in the oracle.
    float l[3], h[3];
    const float p0[3] = { w[0].x, w[0].y, w[0].z }, a1[3] = { e1.x, e1.y, e1.z }, a2[3] = { e2.x, e2.y, e2.z };
    for (int a = 0; a < 3; a++)
    {
        const float q1 = p0[a] + a1[a], q2 = p0[a] + a2[a];
        const float mn = fminf(p0[a], fminf(q1, q2)), mx = fmaxf(p0[a], fmaxf(q1, q2));
        const float pad = 1e-5f * fmaxf(fabsf(mn), fabsf(mx)) + 5e-4f * (mx - mn) + 1e-7f;
        l[a] = mn - pad;
        h[a] = mx + pad;
    }
    boxLo[g] = make_float4(l[0], l[1], l[2], 0.0f);
    boxHi[g] = make_float4(h[0], h[1], h[2], 0.0f);
    for (int a = 0; a < 3 && !isInert; a++)
    {
        const float c = 0.5f * (l[a] + h[a]);
        if (c == c && fabsf(c) < 3.0e38f)
        {
            atomicMin(&sceneBounds[a], orderedFloat(c));
            atomicMax(&sceneBounds[3 + a], orderedFloat(c));
        }
    }
}

PT_DEV uint64_t expandBits21(uint32_t v) // 21 bits -> every third bit of 63
{
    uint64_t x = v & 0x1fffffu;
    x = (x | x << 32) & 0x1f00000000ffffULL;
    x = (x | x << 16) & 0x1f0000ff0000ffULL;
    x = (x | x << 8) & 0x100f00f00f00f00fULL;
    x = (x | x << 4) & 0x10c30c30c30c30c3ULL;
    x = (x | x << 2) & 0x1249249249249249ULL;
    return x;
}

constexpr uint64_t kInertKey = ~0ull; // above every 63-bit Morton code: inert triangles end up behind the sorted rest

// cubic: one scale for the three axes (cells of the curve are cubes) instead of each axis normalised to its own extent
// (cells have the proportions of the scene: in a street 80 x 16 x 16 units they are five times longer than wide)
__global__ void k_morton(uint32_t n, const float4 *__restrict__ boxLo, const float4 *__restrict__ boxHi,
                         const uint32_t *__restrict__ sceneBounds, const uint8_t *__restrict__ inert, uint64_t *__restrict__ keys,
                         uint32_t *__restrict__ vals, int cubic)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n)
        return;
    if (inert[g])
    {
        keys[g] = kInertKey;
        vals[g] = g;
        return;
    }
    const float4 lo = boxLo[g], hi = boxHi[g];
    const float c[3] = { 0.5f * (lo.x + hi.x), 0.5f * (lo.y + hi.y), 0.5f * (lo.z + hi.z) };
    uint32_t q[3];
    float widest = 0.0f;
    for (int a = 0; a < 3; a++)
        widest = fmaxf(widest, unorderedFloat(sceneBounds[3 + a]) - unorderedFloat(sceneBounds[a]));
    for (int a = 0; a < 3; a++)
    {
        const float mn = unorderedFloat(sceneBounds[a]), mx = unorderedFloat(sceneBounds[3 + a]);
        const float ext = cubic ? widest : mx - mn;
        float f = ext > 0.0f ? (c[a] - mn) / ext : 0.0f;
        f = f == f ? fminf(fmaxf(f, 0.0f), 1.0f) : 0.0f;
        const uint32_t v = (uint32_t)(f * 2097151.0f);
        q[a] = v > 2097151u ? 2097151u : v;
    }
    keys[g] = (expandBits21(q[0]) << 2) | (expandBits21(q[1]) << 1) | expandBits21(q[2]);
    vals[g] = g;
}

// number of sorted keys below the inert sentinel (one thread: ~log2 n dependent loads)
__global__ void k_count_valid(uint32_t n, const uint64_t *__restrict__ sortedKeys, uint32_t *__restrict__ out)
{
    uint32_t lo = 0, hi = n; // first index whose key is the sentinel
    while (lo < hi)
    {
        const uint32_t mid = (lo + hi) >> 1;
        if (sortedKeys[mid] == kInertKey)
            hi = mid;
        else
            lo = mid + 1;
    }
    *out = lo;
}

// ---- LSD radix sort, 8-bit digits, 64-bit keys + 32-bit values --------------------
// Pass = histogram (per tile) -> exclusive scan over (digit, tile) -> stable scatter.
// One wave per tile: the in-tile rank of an element is (earlier chunks' digit count) +
// (lanes below me in this 64-element chunk with my digit), the latter by 8 ballots.
constexpr uint32_t kSortTile = 2048; // elements per tile (one wave, 32 chunks of 64)

__global__ void __launch_bounds__(64) k_sort_hist(uint32_t n, const uint64_t *__restrict__ keys, uint32_t shift,
                                                  uint32_t numTiles, uint32_t *__restrict__ hist)
{
    __shared__ uint32_t h[256];
    const uint32_t lane = threadIdx.x, tile = blockIdx.x;
    for (uint32_t i = lane; i < 256; i += 64)
        h[i] = 0;
    __syncthreads();
    const uint32_t base = tile * kSortTile;
    for (uint32_t i = lane; i < kSortTile; i += 64)
        if (base + i < n)
            atomicAdd(&h[(uint32_t)(keys[base + i] >> shift) & 0xffu], 1u);
    __syncthreads();
    for (uint32_t i = lane; i < 256; i += 64)
        hist[i * numTiles + tile] = h[i];
}

// exclusive scan of `count` uints by ONE block of 1024 threads (count <= a few 100k)
__global__ void __launch_bounds__(1024) k_scan_exclusive(uint32_t count, uint32_t *__restrict__ data)
{
    __shared__ uint32_t partial[1024];
    const uint32_t tid = threadIdx.x;
    const uint32_t per = (count + 1023u) / 1024u;
    const uint32_t begin = tid * per, end = begin + per < count ? begin + per : count;
    uint32_t sum = 0;
    for (uint32_t i = begin; i < end; i++)
        sum += data[i];
    partial[tid] = sum;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1)
    {
        const uint32_t v = tid >= off ? partial[tid - off] : 0;
        __syncthreads();
        partial[tid] += v;
        __syncthreads();
    }
    uint32_t run = tid ? partial[tid - 1] : 0;
    for (uint32_t i = begin; i < end; i++)
    {
        const uint32_t v = data[i];
        data[i] = run;
        run += v;
    }
}

// the same scan in three passes over any number of blocks (the single block above takes 0.4 ms for the 250 K counters of
// a 2 M-key pass: eight of them were a quarter of the sort): block sums, scan of the sums, apply
constexpr uint32_t kScan32Block = 2048;
__global__ void __launch_bounds__(256) k_scan32_sums(uint32_t count, const uint32_t *__restrict__ data, uint32_t *__restrict__ sums)
{
    __shared__ uint32_t part[256];
    const uint32_t base = blockIdx.x * kScan32Block;
    uint32_t s = 0;
    for (uint32_t k = threadIdx.x; k < kScan32Block; k += 256)
        if (base + k < count)
            s += data[base + k];
    part[threadIdx.x] = s;
    __syncthreads();
    for (uint32_t off = 128; off > 0; off >>= 1)
    {
        if (threadIdx.x < off)
            part[threadIdx.x] += part[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0)
        sums[blockIdx.x] = part[0];
}
__global__ void __launch_bounds__(256) k_scan32_apply(uint32_t count, uint32_t *__restrict__ data, const uint32_t *__restrict__ sums)
{
    __shared__ uint32_t part[256];
    const uint32_t base = blockIdx.x * kScan32Block + threadIdx.x * 8;
    uint32_t v[8], s = 0;
    for (int k = 0; k < 8; k++)
    {
        v[k] = base + k < count ? data[base + k] : 0u;
        s += v[k];
    }
    part[threadIdx.x] = s;
    __syncthreads();
    for (uint32_t off = 1; off < 256; off <<= 1)
    {
        const uint32_t t = threadIdx.x >= off ? part[threadIdx.x - off] : 0u;
        __syncthreads();
        part[threadIdx.x] += t;
        __syncthreads();
    }
    uint32_t run = sums[blockIdx.x] + (threadIdx.x ? part[threadIdx.x - 1] : 0u);
    for (int k = 0; k < 8; k++)
        if (base + k < count)
        {
            data[base + k] = run;
            run += v[k];
        }
}

__global__ void __launch_bounds__(64) k_sort_scatter(uint32_t n, const uint64_t *__restrict__ keysIn,
                                                     const uint32_t *__restrict__ valsIn, uint64_t *__restrict__ keysOut,
                                                     uint32_t *__restrict__ valsOut, uint32_t shift, uint32_t numTiles,
                                                     const uint32_t *__restrict__ hist)
{
    __shared__ uint32_t offs[256];
    const uint32_t lane = threadIdx.x, tile = blockIdx.x;
    for (uint32_t i = lane; i < 256; i += 64)
        offs[i] = hist[i * numTiles + tile];
    __syncthreads();
    const uint32_t base = tile * kSortTile;
    const uint64_t laneMaskLt = (1ull << lane) - 1ull;
    for (uint32_t c = 0; c < kSortTile; c += 64)
    {
        const uint32_t i = base + c + lane;
        const bool valid = i < n;
        const uint64_t key = valid ? keysIn[i] : 0;
        const uint32_t val = valid ? valsIn[i] : 0;
        const uint32_t digit = (uint32_t)(key >> shift) & 0xffu;
        uint64_t peers = __ballot(valid);
        for (int b = 0; b < 8; b++)
        {
            const uint64_t m = __ballot((digit >> b) & 1u);
            peers &= ((digit >> b) & 1u) ? m : ~m;
        }
        const uint32_t rank = (uint32_t)__popcll(peers & laneMaskLt);
        uint32_t dst = 0;
        if (valid)
            dst = offs[digit] + rank;
        __syncthreads();
        if (valid && rank == (uint32_t)__popcll(peers) - 1u) // last peer bumps the running offset
            offs[digit] += (uint32_t)__popcll(peers);
        __syncthreads();
        if (valid)
        {
            keysOut[dst] = key;
            valsOut[dst] = val;
        }
    }
}

// ---- Karras 2012: one internal node per thread ---------------------------------------
PT_DEV int karrasDelta(const uint64_t *keys, int n, int i, int j)
{
    if (j < 0 || j >= n)
        return -1;
    const uint64_t a = keys[i], b = keys[j];
    if (a == b)
        return 64 + __clz((uint32_t)i ^ (uint32_t)j);
    return __clzll((long long)(a ^ b));
}

__global__ void k_karras(int n, const uint64_t *__restrict__ keys, int2 *__restrict__ children, int *__restrict__ parentOfNode,
                         int *__restrict__ parentOfLeaf)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1)
        return;
    const int d = (karrasDelta(keys, n, i, i + 1) - karrasDelta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    const int dmin = karrasDelta(keys, n, i, i - d);
    int lmax = 2;
    while (karrasDelta(keys, n, i, i + lmax * d) > dmin)
        lmax <<= 1;
    int l = 0;
    for (int t = lmax >> 1; t >= 1; t >>= 1)
        if (karrasDelta(keys, n, i, i + (l + t) * d) > dmin)
            l += t;
    const int j = i + l * d;
    const int dnode = karrasDelta(keys, n, i, j);
    int s = 0;
    for (int t = (l + 1) >> 1;; t = (t + 1) >> 1)
    {
        if (karrasDelta(keys, n, i, i + (s + t) * d) > dnode)
            s += t;
        if (t == 1)
            break;
    }
    const int gamma = i + s * d + (d < 0 ? -1 : 0);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    const int left = (lo == gamma) ? ~gamma : gamma;            // leaf refs are ~index
    const int right = (hi == gamma + 1) ? ~(gamma + 1) : gamma + 1;
    children[i] = make_int2(left, right);
    if (left < 0)
        parentOfLeaf[~left] = i;
    else
        parentOfNode[left] = i;
    if (right < 0)
        parentOfLeaf[~right] = i;
    else
        parentOfNode[right] = i;
    if (i == 0)
        parentOfNode[0] = -1;
}

// ---- PLOC: parallel locally-ordered clustering (Meister & Bittner 2017) ------------------------
// Alternative to the Karras topology over the same Morton order: clusters (initially the sorted leaves)
// repeatedly look for the neighbour within +-kPlocRadius positions whose union box has the smallest
// surface area; mutual nearest neighbours merge into a new internal node, the sequence is compacted, and
// the loop runs until one cluster is left.  The result is a bottom-up agglomerative tree guided by the
// surface-area metric instead of by Morton prefixes -- lower SAH cost, i.e. fewer node visits per ray --
// in the SAME arrays (children / parentOfNode / parentOfLeaf, root = node 0, leaf ref = ~sorted position),
// so k_refit / k_emit and the refit path are unchanged.  Node ids are handed out downwards from n - 2 by
// the prefix scan of the merge flags: deterministic, and the last merge (the root) gets id 0.
// Search radius and shape weight.  Round 1, chess_like alone: radius 4 -> 1265, 8 -> 1277, 16 -> 1270, 32 -> 1285, 64 -> 1297, 128 -> 1298
// Msamples/s (build 11.7 .. 21 ms) -> 32.  Round 3, all four stand-ins, Msamples/s at 8 / 16 / 32 / 64 / 128 (pairs = two runs):
// street_like 1274, 1283 / 1270, 1261 / 1220, 1214 / 1191 / 1186 -- a wider search makes ITS tree worse, monotonically: 15.7 node
// visits per primary ray at 16, 18.7 at 32 --, chess_like 2270, 2274 / 2301, 2292 / 2302, 2295, temple_like 896, 892 / 896, 896 / 902,
// 907, atrium_like - / 777, 757 / 783, 764 / 780 / 772; and the cost of 65,536 sampled rays (k_sample_tree_cost) over radius {8, 16,
// 32, 64} x shape {0, 0.25, 1} moves by 5-10 % per scene with no setting best everywhere (temple_like: 2.86 M at (64, 1), 3.18 M at
// (16, 0); street_like: 3.96 M at (8, 0), 4.75 M at (64, 0), 3.78 M at (8, 1) over a Morton curve with cubic cells, which costs
// chess_like and atrium_like 3-5 %).  So ptx_build_accel builds a few candidates and keeps the tree that costs the sampled rays
// least (kTreeCandidates).  These constants are the parameters of builds that skip the comparison.
constexpr int kPlocRadius = 16;
constexpr float kPlocShape = 0.0f;

__global__ void k_ploc_init(uint32_t n, const uint32_t *__restrict__ vals, const float4 *__restrict__ boxLo, const float4 *__restrict__ boxHi,
                            int *__restrict__ cluster, float4 *__restrict__ cLo, float4 *__restrict__ cHi)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const uint32_t g = vals[i];
    cluster[i] = ~(int)i;
    cLo[i] = boxLo[g];
    cHi[i] = boxHi[g];
}

// pairs are ordered by (union area, lower position, higher position): a strict total order, so the globally
// smallest pair is always mutual and every iteration merges at least once
// `shape`: weight of a compactness term in the merge metric, area + shape * (longest extent)^2 -- the surface area of the union of
// two flat boxes does not tell a square from a strip.  Symmetric in (i, j) like the area, so the order stays total.
__global__ void k_ploc_nearest(uint32_t count, uint32_t radius, float shape, const float4 *__restrict__ cLo, const float4 *__restrict__ cHi, uint32_t *__restrict__ nn)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count)
        return;
    const float4 lo = cLo[i], hi = cHi[i];
    const uint32_t first = i > radius ? i - radius : 0u;
    const uint32_t last = i + radius < count ? i + radius : count - 1u;
    float best = 3.0e38f;
    uint32_t bestJ = i == first ? last : first, bestA = 0xffffffffu, bestB = 0xffffffffu;
    for (uint32_t j = first; j <= last; j++)
    {
        if (j == i)
            continue;
        const float4 l = cLo[j], h = cHi[j];
        const float dx = fmaxf(hi.x, h.x) - fminf(lo.x, l.x), dy = fmaxf(hi.y, h.y) - fminf(lo.y, l.y), dz = fmaxf(hi.z, h.z) - fminf(lo.z, l.z);
        const float longest = fmaxf(dx, fmaxf(dy, dz));
        const float area = (dx * dy + dy * dz + dz * dx) + shape * longest * longest;
        const uint32_t a = i < j ? i : j, b = i < j ? j : i;
        if (area < best || (area == best && (a < bestA || (a == bestA && b < bestB))) || bestA == 0xffffffffu)
        {
            best = area;
            bestJ = j;
            bestA = a;
            bestB = b;
        }
    }
    nn[i] = bestJ;
}

// flags for the scan: low word = the position survives, high word = it leads a merge
__global__ void k_ploc_flags(uint32_t count, const uint32_t *__restrict__ nn, unsigned long long *__restrict__ flags)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count)
        return;
    const uint32_t j = nn[i];
    const bool mutual = nn[j] == i;
    const unsigned long long keep = (mutual && i > j) ? 0ull : 1ull, lead = (mutual && i < j) ? 1ull : 0ull;
    flags[i] = keep | (lead << 32);
}

// three-pass exclusive scan of packed (32 + 32 bit) counters: block sums, scan of the sums, apply
constexpr uint32_t kScanBlock = 1024;
__global__ void __launch_bounds__(256) k_scan64_sums(uint32_t count, const unsigned long long *__restrict__ data, unsigned long long *__restrict__ sums)
{
    __shared__ unsigned long long part[256];
    const uint32_t base = blockIdx.x * kScanBlock;
    unsigned long long s = 0;
    for (uint32_t k = threadIdx.x; k < kScanBlock; k += 256)
        if (base + k < count)
            s += data[base + k];
    part[threadIdx.x] = s;
    __syncthreads();
    for (uint32_t off = 128; off > 0; off >>= 1)
    {
        if (threadIdx.x < off)
            part[threadIdx.x] += part[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0)
        sums[blockIdx.x] = part[0];
}
__global__ void __launch_bounds__(1024) k_scan64_top(uint32_t blocks, unsigned long long *__restrict__ sums, unsigned long long *__restrict__ total)
{
    __shared__ unsigned long long part[1024];
    const uint32_t tid = threadIdx.x, per = (blocks + 1023u) / 1024u;
    const uint32_t begin = tid * per, end = begin + per < blocks ? begin + per : blocks;
    unsigned long long s = 0;
    for (uint32_t i = begin; i < end; i++)
        s += sums[i];
    part[tid] = s;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1)
    {
        const unsigned long long v = tid >= off ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    unsigned long long run = tid ? part[tid - 1] : 0;
    for (uint32_t i = begin; i < end; i++)
    {
        const unsigned long long v = sums[i];
        sums[i] = run;
        run += v;
    }
    if (tid == 1023)
        *total = part[1023];
}
__global__ void __launch_bounds__(256) k_scan64_apply(uint32_t count, unsigned long long *__restrict__ data, const unsigned long long *__restrict__ sums)
{
    // one block scans its kScanBlock elements serially per thread-chunk of 4, then adds the block offset
    __shared__ unsigned long long part[256];
    const uint32_t base = blockIdx.x * kScanBlock + threadIdx.x * 4;
    unsigned long long v[4], s = 0;
    for (int k = 0; k < 4; k++)
    {
        v[k] = base + k < count ? data[base + k] : 0ull;
        s += v[k];
    }
    part[threadIdx.x] = s;
    __syncthreads();
    for (uint32_t off = 1; off < 256; off <<= 1)
    {
        const unsigned long long t = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
        __syncthreads();
        part[threadIdx.x] += t;
        __syncthreads();
    }
    unsigned long long run = sums[blockIdx.x] + (threadIdx.x ? part[threadIdx.x - 1] : 0ull);
    for (int k = 0; k < 4; k++)
        if (base + k < count)
        {
            data[base + k] = run;
            run += v[k];
        }
}

__global__ void k_ploc_merge(uint32_t count, const int *__restrict__ cluster, const float4 *__restrict__ cLo, const float4 *__restrict__ cHi,
                             const uint32_t *__restrict__ nn, const unsigned long long *__restrict__ prefix, int firstId,
                             int *__restrict__ outCluster, float4 *__restrict__ outLo, float4 *__restrict__ outHi, int2 *__restrict__ children,
                             int *__restrict__ parentOfNode, int *__restrict__ parentOfLeaf, float4 *__restrict__ nodeLo, float4 *__restrict__ nodeHi)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count)
        return;
    const uint32_t j = nn[i];
    const bool mutual = nn[j] == i;
    if (mutual && i > j)
        return; // absorbed by its partner
    const unsigned long long p = prefix[i];
    const uint32_t dst = (uint32_t)p;
    if (!mutual)
    {
        outCluster[dst] = cluster[i];
        outLo[dst] = cLo[i];
        outHi[dst] = cHi[i];
        return;
    }
    const int id = firstId - (int)(p >> 32);
    const int a = cluster[i], b = cluster[j];
    children[id] = make_int2(a, b);
    if (a < 0) parentOfLeaf[~a] = id; else parentOfNode[a] = id;
    if (b < 0) parentOfLeaf[~b] = id; else parentOfNode[b] = id;
    const float4 l0 = cLo[i], h0 = cHi[i], l1 = cLo[j], h1 = cHi[j];
    const float4 lo = make_float4(fminf(l0.x, l1.x), fminf(l0.y, l1.y), fminf(l0.z, l1.z), 0.0f);
    const float4 hi = make_float4(fmaxf(h0.x, h1.x), fmaxf(h0.y, h1.y), fmaxf(h0.z, h1.z), 0.0f);
    nodeLo[id] = lo;
    nodeHi[id] = hi;
    if (id == 0)
        parentOfNode[0] = -1;
    outCluster[dst] = id;
    outLo[dst] = lo;
    outHi[dst] = hi;
}

typedef float v4f_native __attribute__((ext_vector_type(4)));
PT_DEV float4 loadUncached(const float4 *p) // bypasses the (incoherent) L1 for cross-CU data
{
    const v4f_native v = __builtin_nontemporal_load(reinterpret_cast<const v4f_native *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}

// Bottom-up AABB refit: the second thread to arrive at a node owns it.
__global__ void k_refit(int n, const uint32_t *__restrict__ vals, const float4 *__restrict__ boxLo,
                        const float4 *__restrict__ boxHi, const int2 *__restrict__ children,
                        const int *__restrict__ parentOfNode, const int *__restrict__ parentOfLeaf,
                        float4 *__restrict__ nodeLo, float4 *__restrict__ nodeHi, uint32_t *__restrict__ flags)
{
    const int leaf = blockIdx.x * blockDim.x + threadIdx.x;
    if (leaf >= n)
        return;
    int node = parentOfLeaf[leaf];
    while (node >= 0)
    {
        __threadfence(); // release my child's box / acquire the sibling's
        if (atomicAdd(&flags[node], 1u) == 0u)
            return;
        __threadfence();
        const int2 ch = children[node];
        float4 l0, h0, l1, h1;
        if (ch.x < 0) { const uint32_t g = vals[~ch.x]; l0 = boxLo[g]; h0 = boxHi[g]; }
        else { l0 = loadUncached(&nodeLo[ch.x]); h0 = loadUncached(&nodeHi[ch.x]); }
        if (ch.y < 0) { const uint32_t g = vals[~ch.y]; l1 = boxLo[g]; h1 = boxHi[g]; }
        else { l1 = loadUncached(&nodeLo[ch.y]); h1 = loadUncached(&nodeHi[ch.y]); }
        nodeLo[node] = make_float4(fminf(l0.x, l1.x), fminf(l0.y, l1.y), fminf(l0.z, l1.z), 0.0f);
        nodeHi[node] = make_float4(fmaxf(h0.x, h1.x), fmaxf(h0.y, h1.y), fmaxf(h0.z, h1.z), 0.0f);
        node = parentOfNode[node];
    }
}

// Final layout.  Every binary LBVH node i becomes one 4-wide node: start from its two
// children and, twice, replace the internal child with the largest surface area by that
// child's two children (greedy SAH-style collapse).  Nodes that end up inside another
// node's expansion are simply never referenced (the tree is walked from node 0), so the
// collapse needs no top-down pass.  Child boxes are quantised to 8 bits inside the node's
// own box with a per-axis power-of-two scale; quantisation is conservative and is checked
// against the exact decode arithmetic of the traversal (o + q * scale).
constexpr int kDefaultLeafTris = 1;

struct ChildBox
{
    float lo[3], hi[3];
    int ref;
};

PT_DEV void fetchChild(int ref, const uint32_t *vals, const float4 *boxLo, const float4 *boxHi, const float4 *nodeLo,
                       const float4 *nodeHi, ChildBox &c)
{
    float4 l, h;
    if (ref < 0) { const uint32_t g = vals[~ref]; l = boxLo[g]; h = boxHi[g]; }
    else { l = nodeLo[ref]; h = nodeHi[ref]; }
    c.lo[0] = l.x; c.lo[1] = l.y; c.lo[2] = l.z;
    c.hi[0] = h.x; c.hi[1] = h.y; c.hi[2] = h.z;
    c.ref = ref;
}

PT_DEV float childArea(const ChildBox &c)
{
    const float dx = c.hi[0] - c.lo[0], dy = c.hi[1] - c.lo[1], dz = c.hi[2] - c.lo[2];
    return dx * dy + dy * dz + dz * dx;
}

// exact value of o + q * scale (24 + 8 significant bits fit a double): the quantised box must
// contain the child box in REAL arithmetic, whatever rounding the traversal's slab test applies
PT_DEV double decodeQ(float o, uint32_t q, float scale) { return (double)o + (double)q * (double)scale; }

// the deindexed vertices of one triangle, next to its Tri record (see ShadeTri)
PT_DEV void writeShadeTri(const Tri &t, const DevPair *pairs, const PtxVertex *vertices, const uint32_t *indices, ShadeTri *out)
{
    const DevPair pr = pairs[__float_as_uint(t.c.y)];
    const uint32_t prim = __float_as_uint(t.c.z);
    float f[68];
    Vtx o[3];
    for (int k = 0; k < 3; k++)
    {
        const PtxVertex *v = &vertices[pr.vertexOffset + indices[pr.indexOffset + prim * 3 + k]];
        float *d = &f[14 * k];
        d[0] = v->Position[0]; d[1] = v->Position[1]; d[2] = v->Position[2];
        d[3] = v->TexCoords[0]; d[4] = v->TexCoords[1];
        d[5] = v->Normal[0]; d[6] = v->Normal[1]; d[7] = v->Normal[2];
        d[8] = v->Tangent[0]; d[9] = v->Tangent[1]; d[10] = v->Tangent[2];
        d[11] = v->Bitangent[0]; d[12] = v->Bitangent[1]; d[13] = v->Bitangent[2];
        o[k] = loadVertex(v);
    }
    f[42] = f[43] = 0.0f;
    f3 wp[3], wn[3], gn;
    worldCorners(pr, o[0], o[1], o[2], wp, wn, gn);
    for (int k = 0; k < 3; k++)
    {
        f[44 + 3 * k] = wp[k].x; f[45 + 3 * k] = wp[k].y; f[46 + 3 * k] = wp[k].z;
        f[53 + 3 * k] = wn[k].x; f[54 + 3 * k] = wn[k].y; f[55 + 3 * k] = wn[k].z;
    }
    f[62] = gn.x; f[63] = gn.y; f[64] = gn.z;
    f[65] = f[66] = f[67] = 0.0f;
    for (int k = 0; k < 17; k++)
        out->v[k] = make_float4(f[4 * k], f[4 * k + 1], f[4 * k + 2], f[4 * k + 3]);
}

__global__ void k_emit(int n, const uint32_t *__restrict__ vals, const float4 *__restrict__ boxLo,
                       const float4 *__restrict__ boxHi, const int2 *__restrict__ children, const float4 *__restrict__ nodeLo,
                       const float4 *__restrict__ nodeHi, const Tri *__restrict__ triTmp, BvhNode *__restrict__ nodes,
                       Tri *__restrict__ tris, const DevPair *__restrict__ pairs, const PtxVertex *__restrict__ vertices,
                       const uint32_t *__restrict__ indices, ShadeTri *__restrict__ shadeTris)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
    {
        const Tri t = triTmp[vals[i]];
        tris[i] = t;
        writeShadeTri(t, pairs, vertices, indices, &shadeTris[i]);
    }
    if (i >= n - 1)
        return;
    ChildBox c[kNodeWidth];
    int count = 2;
    {
        const int2 ch = children[i];
        fetchChild(ch.x, vals, boxLo, boxHi, nodeLo, nodeHi, c[0]);
        fetchChild(ch.y, vals, boxLo, boxHi, nodeLo, nodeHi, c[1]);
    }
    for (int round = 0; round < kNodeWidth - 2; round++)
    {
        int pick = -1;
        float best = -1.0f;
        for (int k = 0; k < kNodeWidth; k++)
            if (k < count && c[k].ref >= 0)
            {
                const float a = childArea(c[k]);
                if (a > best) { best = a; pick = k; }
            }
        if (pick < 0)
            break;
        const int2 ch = children[c[pick].ref];
        ChildBox a, b;
        fetchChild(ch.x, vals, boxLo, boxHi, nodeLo, nodeHi, a);
        fetchChild(ch.y, vals, boxLo, boxHi, nodeLo, nodeHi, b);
        for (int k = 0; k < kNodeWidth; k++) // no dynamic register indexing
            if (k == pick)
                c[k] = a;
        for (int k = 0; k < kNodeWidth; k++)
            if (k == count)
                c[k] = b;
        count++;
    }

    const float4 nl = nodeLo[i], nh = nodeHi[i];
    const float o[3] = { nl.x, nl.y, nl.z }, top[3] = { nh.x, nh.y, nh.z };
    uint32_t ebits[3], qlo[3] = { 0, 0, 0 }, qhi[3] = { 0, 0, 0 };
    for (int a = 0; a < 3; a++)
    {
        // smallest power of two with 255 * scale >= extent, then grow until every child box
        // survives the round trip through the traversal's decode
        const float ext = top[a] - o[a];
        int e = 1;
        if (ext > 0.0f)
        {
            const int be = (int)((__float_as_uint(ext / 255.0f) >> 23) & 0xffu); // floor(log2) + 127
            e = be + 1;
            if (e < 1) e = 1;
            if (e > 254) e = 254;
        }
        for (;;)
        {
            const float scale = __uint_as_float((uint32_t)e << 23);
            bool ok = true;
            uint32_t wl = 0, wh = 0;
            for (int k = 0; k < kNodeWidth; k++)
            {
                uint32_t ql = 255, qh = 0; // empty slot: inverted box, never hit
                if (k < count)
                {
                    float f = floorf((c[k].lo[a] - o[a]) / scale);
                    ql = f < 0.0f ? 0u : (f > 255.0f ? 255u : (uint32_t)f);
                    while (ql > 0 && decodeQ(o[a], ql, scale) > (double)c[k].lo[a])
                        ql--;
                    if (decodeQ(o[a], ql, scale) > (double)c[k].lo[a])
                        ok = false;
                    f = ceilf((c[k].hi[a] - o[a]) / scale);
                    qh = f < 0.0f ? 0u : (f > 255.0f ? 255u : (uint32_t)f);
                    while (qh < 255 && decodeQ(o[a], qh, scale) < (double)c[k].hi[a])
                        qh++;
                    if (decodeQ(o[a], qh, scale) < (double)c[k].hi[a])
                        ok = false;
                }
                {
                    wl |= ql << (8 * k);
                    wh |= qh << (8 * k);
                }
            }
            if (ok || e >= 254)
            {
                ebits[a] = (uint32_t)e;
                qlo[a] = wl;
                qhi[a] = wh;
                break;
            }
            e++;
        }
    }
    BvhNode nd;
    nd.a = make_float4(o[0], o[1], o[2], __uint_as_float(ebits[0] | (ebits[1] << 8) | (ebits[2] << 16)));
    // a leaf of a non-opaque geometry says so in its ref: the traversal fetches its any-hit record beside the triangle
    for (int k = 0; k < count; k++)
        if (c[k].ref < 0 && (__float_as_uint(triTmp[vals[~c[k].ref]].c.w) & kTriNonOpaque))
            c[k].ref = ~(int)((uint32_t)~c[k].ref | kLeafNonOpaque);
    nd.refs = make_int4(c[0].ref, c[1].ref, count > 2 ? c[2].ref : kEmptyRef, count > 3 ? c[3].ref : kEmptyRef);
    nd.q0 = make_uint4(qlo[0], qhi[0], qlo[1], qhi[1]);
    nd.q1 = make_uint4(qlo[2], qhi[2], 0u, 0u);
    nodes[i] = nd;
}

// Breadth-first relayout of the emitted nodes.  k_emit writes a 4-wide node at the index of every BINARY node, but only
// about half of them is ever referenced from the root (the others sit inside another node's expansion): in that array
// the live nodes are scattered between dead ones, about one live node per 128-byte line, and a node's children are
// anywhere.  One launch per level copies the live nodes into a compact array in breadth-first order -- the top of the tree
// is a dense prefix, the (up to four) children of a node are adjacent, a line holds two live nodes -- and rewrites the child
// refs.  Nodes [lo, hi) of the new array are placed already (oldOf[i] = index in the emitted array); their internal
// children take the next free indices, one atomic per wave.  The order of the waves' atomics is not fixed, so the layout
// below the root can differ between two builds; what a ray hits does not depend on it.
__global__ void __launch_bounds__(256) k_relayout_level(uint32_t lo, uint32_t hi, const BvhNode *__restrict__ raw, uint32_t *__restrict__ oldOf,
                                                        uint32_t *__restrict__ nextFree, BvhNode *__restrict__ out)
{
    const uint32_t i = lo + blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63u;
    BvhNode nd;
    nd.refs = make_int4(kEmptyRef, kEmptyRef, kEmptyRef, kEmptyRef);
    if (i < hi)
        nd = raw[oldOf[i]];
    int refs[kNodeWidth] = { nd.refs.x, nd.refs.y, nd.refs.z, nd.refs.w };
    uint32_t c = 0;
    for (int k = 0; k < kNodeWidth; k++)
        c += (refs[k] >= 0 && refs[k] != kEmptyRef) ? 1u : 0u;
    uint32_t incl = c; // inclusive prefix over the wave
    for (uint32_t d = 1; d < 64; d <<= 1)
    {
        const uint32_t up = __shfl_up(incl, d);
        if (lane >= d)
            incl += up;
    }
    const uint32_t total = __shfl(incl, 63);
    uint32_t base = 0;
    if (lane == 0 && total)
        base = atomicAdd(nextFree, total);
    base = __shfl(base, 0) + incl - c;
    if (i >= hi)
        return;
    for (int k = 0; k < kNodeWidth; k++)
        if (refs[k] >= 0 && refs[k] != kEmptyRef)
        {
            oldOf[base] = (uint32_t)refs[k];
            refs[k] = (int)base++;
        }
    nd.refs = make_int4(refs[0], refs[1], refs[2], refs[3]);
    out[i] = nd;
}

// a one-triangle scene has no internal node: give it a root with one leaf child
__global__ void k_single_leaf_root(const uint32_t *vals, const float4 *boxLo, const float4 *boxHi, const Tri *triTmp, BvhNode *nodes, Tri *tris,
                                   const DevPair *pairs, const PtxVertex *vertices, const uint32_t *indices, ShadeTri *shadeTris)
{
    const uint32_t g = vals[0]; // the one triangle of the tree (the others, if any, are inert)
    writeShadeTri(triTmp[g], pairs, vertices, indices, &shadeTris[0]);
    tris[0] = triTmp[g];
    BvhNode nd;
    // origin below the box, scale covering it: child 0 spans the whole quantised range
    const float lo[3] = { boxLo[g].x, boxLo[g].y, boxLo[g].z }, hi[3] = { boxHi[g].x, boxHi[g].y, boxHi[g].z };
    uint32_t eb[3];
    for (int a = 0; a < 3; a++)
    {
        int e = 1;
        while (e < 254 && decodeQ(lo[a], 255u, __uint_as_float((uint32_t)e << 23)) < (double)hi[a])
            e++;
        eb[a] = (uint32_t)e;
    }
    nd.a = make_float4(lo[0], lo[1], lo[2], __uint_as_float(eb[0] | (eb[1] << 8) | (eb[2] << 16)));
    nd.refs = make_int4((__float_as_uint(triTmp[g].c.w) & kTriNonOpaque) ? ~(int)kLeafNonOpaque : ~0, kEmptyRef, kEmptyRef, kEmptyRef);
    nd.q0 = make_uint4(0xffffff00u, 0x000000ffu, 0xffffff00u, 0x000000ffu);
    nd.q1 = make_uint4(0xffffff00u, 0x000000ffu, 0u, 0u);
    nodes[0] = nd;
}

// ---------------------------------------------------------------------------------
// Traversal
// ---------------------------------------------------------------------------------

constexpr uint32_t kRefillMin = 16; // idle lanes before the refill runs (while any lane has work); measured: 1 -> 1590, 8 -> 1619, 16 -> 1630, 24 -> 1632, 32 -> 1613, 48 -> 1575 Msamples/s (chess_like, one frame in flight)
constexpr int kLdsStack = 16;       // entries per lane kept in LDS (lane-interleaved: no bank conflicts); measured flat from 8 to 24
constexpr int kLdsStackMega = 64;       // the megakernel runs 2 blocks/CU anyway (195 VGPRs): deep LDS stack, no spill
constexpr int kGlobalSpill = 80; // overflow entries per persistent thread, in a global buffer (LDS 16 + 80 >= 3 x 32 levels)
constexpr uint32_t kMaxPersistentThreads = 2048u * 256u;
constexpr uint32_t kMaxNodeVisits = 1u << 20;

// What the any-hit stages read, laid out for them (the ALPHA kernel variants only).  anyhit.rahit:38-52 and
// occlusionAnyhit.rahit:37-50 need ONE number per candidate, texture(textures[colorIdx], uv).a * colorFactor.a.  Through the
// general path (hitBaseColor: pair -> material -> texture table -> four texels, uv from five float4 of the 272-byte shading
// record) that was four dependent fetches and enough live state to cost the traversal kernels their eighth wave per SIMD.
//   AlphaTri   32 B per triangle slot: the three texture coordinates, the material's alpha factor and where the quads of its
//              alpha texture start (kNoAlphaTex: no texture, the factor is the alpha); the texture's base-level extent rides in
//              the free bits of Tri::c.w.  A leaf of a non-opaque geometry is marked in its ref (kLeafNonOpaque), so the record is
//              fetched WITH the triangle: behind the intersection test one dependent load is left (the quad) instead of three
//              (record -> texture table -> quad).
//   quads      per base-level texel (x, y) the alphas of the 2 x 2 bilinear footprint whose top-left texel it is --
//              (x, y), (x+1, y), (x, y+1), (x+1, y+1) with repeat addressing -- so the footprint is ONE dwordx4 load
struct AlphaTri
{
    float4 a; // u0, v0, u1, v1
    float4 b; // u2, v2, colour factor alpha (or the constant alpha itself), first quad of the alpha texture (bits; kNoAlphaTex: none)
};
struct AlphaTex // build time only (k_alpha_tris): per colour texture, base-level extent and where its quads start
{
    uint32_t width, height, offset, pad;
};
constexpr uint32_t kNoAlphaTex = 0xffffffffu;

struct TraceScene
{
    const BvhNode *nodes;
    const Tri *tris;
    uint32_t triCount;
    const AlphaTri *alphaTris; // ALPHA variants only
    const float4 *alphaQuads;
};

// An index the compiler cannot prove equal to the one it came from, available only once `after` has been computed: what is
// loaded through it is loaded THEN, not kept in registers from an earlier load of the same address.
PT_DEV uint32_t fetchAgainAfter(uint32_t index, float after)
{
    asm volatile("" : "+v"(index) : "v"(after));
    return index;
}

// texture(textures[colorIdx], uv).a * colorFactor.a at a candidate hit: the .w of hitBaseColor(), bit for bit -- the same
// interpolation of the texture coordinates, the same bilinear weights and operation order as sampleLevel / lerp4 on the
// alpha channel alone.  (ta, tb) = the triangle's AlphaTri, triW = its Tri::c.w.
PT_DEV float hitAlpha(const TraceScene &sc, float4 ta, float4 tb, uint32_t triW, float u, float v)
{
    const uint32_t first = __float_as_uint(tb.w);
    if (first == kNoAlphaTex)
        return tb.z;
    const f3 bary = F3(1.0f - u - v, u, v);
    float tu = (ta.x * bary.x + ta.z * bary.y) + tb.x * bary.z;
    float tv = (ta.y * bary.x + ta.w * bary.y) + tb.y * bary.z;
    const float4 *quads = sc.alphaQuads + first;
    const uint32_t w = (triW & 0x7fffu) + 1u, h = ((triW >> 15) & 0x7fffu) + 1u;
    float alpha;
    if (w == 1 && h == 1)
        alpha = quads[0].x;
    else
    {
        if (!(abs_(tu) < 1e9f)) tu = 0.0f;
        if (!(abs_(tv) < 1e9f)) tv = 0.0f;
        const float x = tu * (float)w - 0.5f, y = tv * (float)h - 0.5f;
        const float x0 = __builtin_floorf(x), y0 = __builtin_floorf(y);
        const float ax = x - x0, ay = y - y0;
        const uint32_t ix0 = wrapRepeat(x0, w), ix1 = wrapRepeat(x0 + 1.0f, w), iy0 = wrapRepeat(y0, h), iy1 = wrapRepeat(y0 + 1.0f, h);
        float a00, a10, a01, a11;
        if (ix1 == (ix0 + 1 == w ? 0u : ix0 + 1) && iy1 == (iy0 + 1 == h ? 0u : iy0 + 1))
        {
            const float4 q = quads[(size_t)iy0 * w + ix0];
            a00 = q.x; a10 = q.y; a01 = q.z; a11 = q.w;
        }
        else
        {
            // coordinates so large that x0 + 1 is not the next texel in float arithmetic: the four texels one by one (a quad's
            // first entry is its own texel)
            a00 = a10 = a01 = a11 = 0.0f;
#pragma nounroll
            for (int k = 0; k < 4; k++)
            {
                const float a = quads[(size_t)((k & 2) ? iy1 : iy0) * w + ((k & 1) ? ix1 : ix0)].x;
                if (k == 0) a00 = a;
                else if (k == 1) a10 = a;
                else if (k == 2) a01 = a;
                else a11 = a;
            }
        }
        const float top = a00 * (1.0f - ax) + a10 * ax, bot = a01 * (1.0f - ax) + a11 * ax;
        alpha = top * (1.0f - ay) + bot * ay;
    }
    return alpha * tb.z;
}
PT_DEV float hitAlpha(const TraceScene &sc, uint32_t slot, uint32_t triW, float u, float v)
{
    return hitAlpha(sc, sc.alphaTris[slot].a, sc.alphaTris[slot].b, triW, u, v);
}

// One thread per triangle slot of the tree: the any-hit record of a triangle of a non-opaque geometry.  The texture
// coordinates come from the shading record k_emit wrote (floats 3..4, 17..18, 31..32), the colour texture and factor by
// the rules of material.glsl:25-54 (getColorTextureIdx / getColorFactor; an unknown material type: texture 0, factor 1).
// The extent of the alpha texture goes into the free bits of the triangle's own record (Tri::c.w).
__global__ void k_alpha_tris(uint32_t n, Tri *__restrict__ tris, const ShadeTri *__restrict__ shadeTris, SceneView sv,
                             const uint32_t *__restrict__ alphaTexOf, const AlphaTex *__restrict__ alphaTex, AlphaTri *__restrict__ out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    AlphaTri r;
    r.a = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    r.b = make_float4(0.0f, 0.0f, 1.0f, __uint_as_float(kNoAlphaTex));
    const float4 tc = tris[i].c;
    if (__float_as_uint(tc.w) & kTriNonOpaque)
    {
        const ShadeTri *st = &shadeTris[i];
        const float4 q0 = st->v[0], q1 = st->v[1], q4 = st->v[4], q7 = st->v[7], q8 = st->v[8];
        r.a = make_float4(q0.w, q1.x, q4.y, q4.z);
        const DevPair *pr = &sv.pairs[__float_as_uint(tc.y)];
        const uint32_t materialType = pr->materialId & 0xffu, materialIndex = pr->materialId >> 8;
        uint32_t idx = 0;
        float factor = 1.0f;
        if (materialType == PTX_MATERIAL_TYPE_METALLIC_ROUGHNESS)
        {
            idx = sv.mr[materialIndex].ColorIdx;
            factor = sv.mr[materialIndex].Color[3];
        }
        else if (materialType == PTX_MATERIAL_TYPE_SPECULAR_GLOSSINESS)
        {
            idx = sv.sg[materialIndex].ColorIdx;
            factor = sv.sg[materialIndex].Color[3];
        }
        else if (materialType == PTX_MATERIAL_TYPE_PHONG)
        {
            idx = sv.phong[materialIndex].ColorIdx;
            factor = sv.phong[materialIndex].Color[3];
        }
        uint32_t first = kNoAlphaTex, extent = 0u;
        if (idx >= PTX_SCENE_TEXTURE_OFFSET && idx - PTX_SCENE_TEXTURE_OFFSET < sv.tex.textureCount)
        {
            const AlphaTex at = alphaTex[alphaTexOf[idx - PTX_SCENE_TEXTURE_OFFSET]];
            first = at.offset;
            extent = (at.width - 1u) | (at.height - 1u) << 15; // both at most 32768 (checked at upload)
        }
        else
            factor = sampleTexture(idx).w * factor; // a fixed 1x1 default (or the white placeholder past the table): constant alpha
        r.b = make_float4(q7.w, q8.x, factor, __uint_as_float(first));
        tris[i].c.w = __uint_as_float(kTriNonOpaque | extent);
    }
    out[i] = r;
}

// The quads of one colour texture from its decoded base level.
__global__ void k_alpha_quads(uint32_t w, uint32_t h, const float4 *__restrict__ level0, float4 *__restrict__ quads)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= w * h)
        return;
    const uint32_t x = k % w, y = k / w, x1 = x + 1 == w ? 0u : x + 1, y1 = y + 1 == h ? 0u : y + 1;
    quads[k] = make_float4(level0[(size_t)y * w + x].w, level0[(size_t)y * w + x1].w, level0[(size_t)y1 * w + x].w, level0[(size_t)y1 * w + x1].w);
}

// The any-hit stage for one candidate of a non-opaque geometry; true = the candidate stays.
//   closest rays  anyhit.rahit:36-64: alpha < 0.5 -> remembered as the decal if it is the nearest so far, ignored
//   shadow rays   occlusionAnyhit.rahit:35-53: alpha < 1 -> ignored
template <bool ANY_HIT>
PT_DEV bool anyHitKeeps(const TraceScene &sc, uint32_t pair, uint32_t prim, uint32_t slot, uint32_t triW, float t, float u, float v, Decal &decal)
{
    const float alpha = hitAlpha(sc, slot, triW, u, v);
    if (ANY_HIT)
        return !(alpha < 1.0f);
    if (alpha < 0.5f)
    {
        if (decal.dist == -1.0f || t < decal.dist || (t == decal.dist && (pair < decal.pair || (pair == decal.pair && prim < decal.prim))))
        {
            decal.dist = t;
            decal.slot = slot;
            decal.u = u;
            decal.v = v;
            decal.pair = pair;
            decal.prim = prim;
        }
        return false;
    }
    return true;
}

// explicit LDS address space: through a generic pointer hipcc emits flat_load/flat_store
// (checked in the ISA), which go down the vector-memory path instead of ds_read/ds_write
typedef __attribute__((address_space(3))) uint32_t lds_u32;

// Traversal stack: LDS first, then an explicit GLOBAL overflow region [entry][thread].
// Deliberately no private (scratch) array: a 724-B/lane scratch segment cut the traversal
// kernels' throughput by ~30 % on MI355X (fewer resident waves), measured.
struct Stack
{
    lds_u32 *lds; // &s_stack[0][lane]
    uint32_t stride;
    int ldsDepth;
    uint32_t *spill; // &spillBuffer[thread] or nullptr
    uint32_t spillStride;
    int spillDepth;
    int sp = 0;
    bool overflow = false;
    PT_DEV void push(uint32_t v)
    {
        if (sp < ldsDepth)
            lds[sp * stride] = v;
        else if (sp - ldsDepth < spillDepth)
            spill[(size_t)(sp - ldsDepth) * spillStride] = v;
        else
            overflow = true;
        sp++;
    }
    PT_DEV uint32_t pop()
    {
        sp--;
        if (sp < ldsDepth)
            return lds[sp * stride];
        if (sp - ldsDepth < spillDepth)
            return spill[(size_t)(sp - ldsDepth) * spillStride];
        return 0xffffffffu; // overflow: a harmless leaf (the launch reports the overflow)
    }
};

#define PT_DECLARE_STACK(st, DEPTH, spillPtr)                                                                              \
    __shared__ uint32_t s_stack[DEPTH][kBlock];                                                                            \
    Stack st;                                                                                                              \
    st.lds = (lds_u32 *)&s_stack[0][threadIdx.x];                                                                          \
    st.stride = kBlock;                                                                                                    \
    st.ldsDepth = DEPTH;                                                                                                   \
    st.spill = (spillPtr) ? (spillPtr) + (blockIdx.x * blockDim.x + threadIdx.x) : nullptr;                                \
    st.spillStride = gridDim.x * blockDim.x;                                                                               \
    st.spillDepth = (spillPtr) ? kGlobalSpill : 0;

// slab test against one child box; returns entry distance in tn
PT_DEV bool slab(float lx, float ly, float lz, float hx, float hy, float hz, f3 o, f3 id, float tmin, float tmax, float &tn)
{
    float t0 = (lx - o.x) * id.x, t1 = (hx - o.x) * id.x;
    float lo = fminf(t0, t1), hi = fmaxf(t0, t1);
    t0 = (ly - o.y) * id.y;
    t1 = (hy - o.y) * id.y;
    lo = fmaxf(lo, fminf(t0, t1));
    hi = fminf(hi, fmaxf(t0, t1));
    t0 = (lz - o.z) * id.z;
    t1 = (hz - o.z) * id.z;
    lo = fmaxf(lo, tmin);
    lo = fmaxf(lo, fminf(t0, t1));
    hi = fminf(hi, fmaxf(t0, t1));
    hi = fminf(hi, tmax);
    tn = lo;
    return lo <= hi * 1.0000004f;
}

#define PT_BYTE(w, k) ((float)(((w) >> (8 * (k))) & 0xffu))

// One visit of a 4-wide node: slab-test the four children, then order the hit ones by
// entry distance (r0 nearest).  Returns the number of children hit.
// The quantised planes are never decoded: with A = scale * id and B = (origin - o) * id
// (per node and axis) every plane distance is ONE fma, t = q * A + B.  Box tests only have
// to be conservative (section "Arithmetic" of DESIGN.md): the builder guarantees the real
// box o + q * scale contains the child, leaf boxes are padded by 1e-5 relative, and the
// interval test keeps the (1 + 2^-21) slack.  NaNs (0 * inf for axis-parallel rays) drop out
// of fminf / fmaxf, which only makes the test more permissive.
PT_DEV int visitNode(const BvhNode *__restrict__ np, f3 o, f3 id, float tmin, float lim, int &r0, int &r1, int &r2, int &r3)
{
    const float4 na = np->a;
    const int4 refs = np->refs;
    const uint4 q0 = np->q0, q1 = np->q1;
    const uint32_t eb = __float_as_uint(na.w);
    const float ax = __uint_as_float((eb & 0xffu) << 23) * id.x, ay = __uint_as_float(((eb >> 8) & 0xffu) << 23) * id.y,
                az = __uint_as_float(((eb >> 16) & 0xffu) << 23) * id.z;
    const float bx = (na.x - o.x) * id.x, by = (na.y - o.y) * id.y, bz = (na.z - o.z) * id.z;
    // entry / exit plane per axis from the sign of the direction: no per-child min/max pairs
    const bool ngx = id.x < 0.0f, ngy = id.y < 0.0f, ngz = id.z < 0.0f;
    const uint32_t nx = ngx ? q0.y : q0.x, fx = ngx ? q0.x : q0.y;
    const uint32_t ny = ngy ? q0.w : q0.z, fy = ngy ? q0.z : q0.w;
    const uint32_t nz = ngz ? q1.y : q1.x, fz = ngz ? q1.x : q1.y;
    float t0, t1, t2, t3;
    bool h0, h1, h2, h3;
#define PT_CHILD(k, tn, hk)                                                                                                \
    {                                                                                                                      \
        const float lo = fmaxf(fmaxf(__builtin_fmaf(PT_BYTE(nx, k), ax, bx), __builtin_fmaf(PT_BYTE(ny, k), ay, by)),      \
                               fmaxf(__builtin_fmaf(PT_BYTE(nz, k), az, bz), tmin));                                       \
        const float hi = fminf(fminf(__builtin_fmaf(PT_BYTE(fx, k), ax, bx), __builtin_fmaf(PT_BYTE(fy, k), ay, by)),      \
                               fminf(__builtin_fmaf(PT_BYTE(fz, k), az, bz), lim));                                        \
        tn = lo;                                                                                                           \
        hk = lo <= hi * 1.0000004f;                                                                                        \
    }
    PT_CHILD(0, t0, h0)
    PT_CHILD(1, t1, h1)
    PT_CHILD(2, t2, h2)
    PT_CHILD(3, t3, h3)
#undef PT_CHILD
    h0 = h0 && refs.x != kEmptyRef; // an inverted box is not a miss for the min/max slab form
    h1 = h1 && refs.y != kEmptyRef;
    h2 = h2 && refs.z != kEmptyRef;
    h3 = h3 && refs.w != kEmptyRef;
    const float inf = __uint_as_float(0x7f800000u);
    float k0 = h0 ? t0 : inf, k1 = h1 ? t1 : inf, k2 = h2 ? t2 : inf, k3 = h3 ? t3 : inf;
    r0 = refs.x; r1 = refs.y; r2 = refs.z; r3 = refs.w;
    // 5-comparator sorting network on (key, ref); misses carry +inf and sink to the end
#define PT_CSWAP(ka, ra, kb, rb)                                                                                           \
    {                                                                                                                      \
        const bool sw = kb < ka;                                                                                           \
        const float tk = sw ? kb : ka;                                                                                     \
        kb = sw ? ka : kb;                                                                                                 \
        ka = tk;                                                                                                           \
        const int tr = sw ? rb : ra;                                                                                       \
        rb = sw ? ra : rb;                                                                                                 \
        ra = tr;                                                                                                           \
    }
    PT_CSWAP(k0, r0, k1, r1)
    PT_CSWAP(k2, r2, k3, r3)
    PT_CSWAP(k0, r0, k2, r2)
    PT_CSWAP(k1, r1, k3, r3)
    PT_CSWAP(k1, r1, k2, r2)
#undef PT_CSWAP
    return (int)h0 + (int)h1 + (int)h2 + (int)h3;
}

// Culling against the current best leaves room for the triangle test's own error in t (Moeller-Trumbore from a far
// origin: ~1e-5 relative): two triangles in one plane can report the SAME t while the point o + t d lies a few 1e-5
// outside the second one's padded box; it must still be visited for the id tie-break (seen by the oracle's tree on
// street_like at 1920x1080; this tree's quantisation margin happened to cover it).
constexpr float kCullSlack = 1.0001f;

// A ray whose origin/direction is not finite, whose direction is zero or whose interval is
// empty hits nothing (the triangle test rejects it); without this early-out its NaN plane
// distances would drop out of every fminf/fmaxf and it would walk the whole tree.
PT_DEV bool rayIsTraceable(f3 o, f3 d, float tmin, float tmax)
{
    const float s = (o.x + o.y + o.z) + (d.x + d.y + d.z); // NaN or inf if any component is
    const bool finite = abs_(s) < __uint_as_float(0x7f800000u);
    const bool nonzero = d.x != 0.0f || d.y != 0.0f || d.z != 0.0f;
    return finite && nonzero && tmax > tmin;
}

// 1/d for the slab tests only (never for the triangle test): v_rcp_f32, 1 ulp
PT_DEV f3 fastInverse(f3 d) { return F3(__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y), __builtin_amdgcn_rcpf(d.z)); }

// Closest hit = min t over all triangles the ray hits in (tmin, tmax); ties go to the
// smaller (pair, prim), i.e. the smaller global triangle id -- independent of tree shape.
template <bool ANY_HIT, bool STATS = false, bool ALPHA = false>
PT_DEV bool traceRay(const TraceScene &sc, f3 o, f3 d, float tmin, float tmax, Stack &st, Hit &best, uint32_t *nodeVisits = nullptr,
                     uint32_t *triTests = nullptr, Decal *decalOut = nullptr)
{
    Decal decal = noDecal();
    if (ALPHA && decalOut)
        *decalOut = decal;
    best.t = tmax;
    best.u = best.v = 0.0f;
    best.pair = 0xffffffffu;
    best.prim = 0xffffffffu;
    best.slot = 0u;
    if (sc.triCount == 0 || !rayIsTraceable(o, d, tmin, tmax))
        return false;
    const f3 id = fastInverse(d);
    st.sp = 0;
    int ref = 0;
    // a legitimate ray visits a few hundred nodes; the bound only turns a corrupted tree
    // into a wrong pixel instead of a hung GPU
    for (uint32_t visits = 0; visits < kMaxNodeVisits; visits++)
    {
        if (ref >= 0)
        {
            if (STATS)
                (*nodeVisits)++;
            int r0, r1, r2, r3;
            const int h = visitNode(&sc.nodes[ref], o, id, tmin, ANY_HIT ? best.t : best.t * kCullSlack, r0, r1, r2, r3);
            if (h > 3) st.push((uint32_t)r3);
            if (h > 2) st.push((uint32_t)r2);
            if (h > 1) st.push((uint32_t)r1);
            if (h > 0)
                ref = r0;
            else
            {
                if (st.sp == 0)
                    break;
                ref = (int)st.pop();
            }
        }
        else
        {
            if (STATS)
                (*triTests)++;
            const uint32_t leafSlot = ALPHA ? (uint32_t)~ref & kLeafSlotMask : (uint32_t)~ref;
            const Tri *tp = &sc.tris[leafSlot];
            const float4 ta = tp->a, tb = tp->b, tc = tp->c;
            float t, u, v;
            if (intersectTri(F3(ta.x, ta.y, ta.z), F3(ta.w, tb.x, tb.y), F3(tb.z, tb.w, tc.x), o, d, tmin, tmax, t, u, v) &&
                (!ALPHA || !(__float_as_uint(tc.w) & kTriNonOpaque) ||
                 anyHitKeeps<ANY_HIT>(sc, __float_as_uint(tc.y), __float_as_uint(tc.z), leafSlot, __float_as_uint(tc.w), t, u, v, decal)))
            {
                const uint32_t pair = __float_as_uint(tc.y), prim = __float_as_uint(tc.z);
                if (ANY_HIT)
                {
                    best.pair = pair;
                    best.prim = prim;
                    best.slot = leafSlot;
                    return true;
                }
                if (t < best.t || (t == best.t && (pair < best.pair || (pair == best.pair && prim < best.prim))))
                {
                    best.t = t;
                    best.u = u;
                    best.v = v;
                    best.pair = pair;
                    best.prim = prim;
                    best.slot = leafSlot;
                }
            }
            if (st.sp == 0)
                break;
            ref = (int)st.pop();
        }
    }
    if (ALPHA && decalOut)
        *decalOut = decal;
    return best.pair != 0xffffffffu;
}

// ---------------------------------------------------------------------------------
// Persistent-threads traversal with per-lane ray refill
// ---------------------------------------------------------------------------------
// A wave keeps all 64 lanes busy: a lane whose ray is finished takes the next ray of the
// wave's current chunk (a contiguous run of kTraceChunk queue entries, so neighbouring
// lanes still hold neighbouring pixels); only when the chunk is used up does lane 0 grab a
// new one with ONE global atomic.  Per-ray atomics would serialise at ~11 ns each on
// MI355X (same address), hence chunks.  Each round runs up to kNodeStepsPerRound node
// visits per lane, then one leaf phase for every lane that reached a leaf, then the
// refill -- so the triangle code is issued once per round, not once per node visit.
//
// IO supplies the queue: kHasQueue + queueEntry(item) / setEntry(entry) (what item `item` of the queue names -- a path slot --
// read by the wave a chunk at a time and handed back to the lane that takes the item), bool load(item, o, d, tmin, tmax) (false =
// nothing to trace, e.g. a dead slot), for closest-hit
// queries void improve(item, t, u, v, triSlot) (a nearer hit was found: the IO keeps where) and uint32_t bestSlot(item)
// (its triangle), and void store(item, hit, hitAny, anyHitQuery) when the ray is done (hit.t and hit.pair; u, v, slot are the IO's).
constexpr uint32_t kTraceChunk = 128;   // measured: 64 -> 1327, 128 -> 1347, 256 -> 1310 Msamples/s
constexpr int kNodeStepsPerRound = 2;   // measured: 1 -> 1040, 2 -> 1064, 3 -> 1055, 4 -> 1034, 6 -> 975, 10 -> 872 Msamples/s
constexpr int kRefDone = 0x7fffffff;


// IO::kFixedTmin >= 0: every ray of the queue has this tmin (the wavefront queues: 1e-5) -> a literal, not a register
// IO::kFixedTmax likewise (closest-hit queues: 1e4)
#define PT_TMIN (IO::kFixedTmin >= 0.0f ? IO::kFixedTmin : tmin)
#define PT_TMAX (IO::kFixedTmax >= 0.0f ? IO::kFixedTmax : tmax)
template <bool ANY_HIT, bool ALPHA, typename IO>
PT_DEV void persistentTrace(const TraceScene &sc, IO &io, uint32_t count, uint32_t *__restrict__ chunkCounter, Stack &st)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t below = (1ull << lane) - 1ull;
    uint32_t cursor = 0, end = 0; // wave-uniform
    bool exhausted = false;       // wave-uniform
    // Chunk schedule.  Same-address atomics serialise at ~11 ns on MI355X, so a 16.6 M-ray launch (130 K chunks)
    // would spend 1.4 ms of L2 time on the chunk counter alone.  Most chunks are therefore dealt statically --
    // round r of wave w is chunk w + r * W for the first staticRounds rounds -- and only the last eighth of the
    // queue is handed out through the atomic counter, which keeps the load balance of the dynamic scheme.
    const uint32_t waves = gridDim.x * (blockDim.x >> 6), waveId = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint32_t totalChunks = (count + kTraceChunk - 1) / kTraceChunk;
    const uint32_t staticRounds = (uint32_t)(((uint64_t)totalChunks * 7 / 8) / waves);
    uint32_t round = 0;
    bool have = false;
    uint32_t item = 0;
    // The wave's current chunk of the ray queue, staged in LDS (IO::kHasQueue): when the wave takes a chunk, two coalesced loads
    // bring its 128 queue entries in, and a refilling lane takes its entry from there by the rank the refill ballot gives it --
    // one dependent global load less at the head of the chain of every ray (queue entry -> ray -> root).  The first attempt
    // (round 2) lost 3 % to seven spilled registers at the 64-VGPR budget; without the SLP vectoriser the kernels have the room
    // (58 / 56 VGPRs): chess_like +0.9 %, street_like +1.8 %, atrium_like +0.6 % (two runs each, one call).
    __shared__ uint32_t s_stagedQueue[4][kTraceChunk]; // per wave of the block (256 threads)
    __attribute__((address_space(3))) uint32_t *stagedQueue = (__attribute__((address_space(3))) uint32_t *)&s_stagedQueue[threadIdx.x >> 6][0];
    uint32_t chunkStart = 0; // wave-uniform
    f3 o = F3s(0.0f), d = F3s(0.0f), id = F3s(0.0f);
    float tmin = 0.0f, tmax = 0.0f;
    Hit best;
    best.t = 0.0f; best.u = best.v = 0.0f; best.pair = best.prim = 0xffffffffu; best.slot = 0u;
    int ref = kRefDone;
    st.sp = 0;
    st.overflow = false;

    for (;;)
    {
        // ---- refill idle lanes from the wave's chunk
        uint64_t idleMask = __ballot(!have);
        // postponed while only a few lanes are idle and the others have work (wave-uniform decision)
        if ((uint32_t)__popcll(idleMask) < kRefillMin && idleMask != ~0ull && !(cursor == end && exhausted))
            idleMask = 0ull;
        if (idleMask)
        {
            if (cursor == end && !exhausted)
            {
                uint32_t c = 0;
                if (round < staticRounds)
                    c = (waveId + round++ * waves) * kTraceChunk;
                else
                {
                    if (lane == 0)
                        c = atomicAdd(chunkCounter, kTraceChunk);
                    c = (uint32_t)__builtin_amdgcn_readfirstlane((int)c) + staticRounds * waves * kTraceChunk;
                }
                if (c >= count)
                    exhausted = true;
                else
                {
                    cursor = c;
                    end = c + kTraceChunk < count ? c + kTraceChunk : count;
                }
                if (IO::kHasQueue && c < count)
                {
                    // (the wave's own LDS traffic is in order: no barrier, only the compiler must not move the reads up)
                    const uint32_t e0 = c + lane, e1 = c + 64u + lane;
                    const uint32_t q0 = e0 < count ? io.queueEntry(e0) : 0u, q1 = e1 < count ? io.queueEntry(e1) : 0u;
                    stagedQueue[lane] = q0;
                    stagedQueue[64u + lane] = q1;
                    chunkStart = c;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                }
            }
            if (cursor < end)
            {
                const uint32_t want = (uint32_t)__popcll(idleMask), avail = end - cursor;
                const uint32_t take = want < avail ? want : avail;
                const uint32_t rank = (uint32_t)__popcll(idleMask & below);
                if (!have && rank < take)
                {
                    item = cursor + rank;
                    if (IO::kHasQueue)
                        io.setEntry(stagedQueue[item - chunkStart]);
                    if (io.load(item, o, d, tmin, tmax))
                    {
                        have = true;
                        id = fastInverse(d);
                        best.t = PT_TMAX; // an any-hit query ends at its first hit: its limit stays tmax and best.t is not kept
                        best.pair = 0xffffffffu;
                        st.sp = 0;
                        ref = (sc.triCount && rayIsTraceable(o, d, PT_TMIN, PT_TMAX)) ? 0 : kRefDone;
                    }
                }
                cursor += take;
            }
            else if (exhausted && __ballot(have) == 0)
                break;
        }

        // ---- node phase: a few visits per lane; lanes at a leaf (ref < 0) wait for the leaf phase
        for (int step = 0; step < kNodeStepsPerRound; step++)
        {
            if (have && ref >= 0 && ref != kRefDone)
            {
                int r0, r1, r2, r3;
                const int h = visitNode(&sc.nodes[ref], o, id, PT_TMIN, ANY_HIT ? PT_TMAX : best.t * kCullSlack, r0, r1, r2, r3);
                if (h > 3) st.push((uint32_t)r3);
                if (h > 2) st.push((uint32_t)r2);
                if (h > 1) st.push((uint32_t)r1);
                ref = h > 0 ? r0 : (st.sp ? (int)st.pop() : kRefDone);
            }
        }

        // ---- leaf phase (single-triangle leaves: ~ref = slot, with kLeafNonOpaque where the any-hit stage runs)
        if (have && ref < 0)
        {
            const uint32_t leafBits = (uint32_t)~ref;
            const uint32_t leafSlot = ALPHA ? leafBits & kLeafSlotMask : leafBits;
            const Tri *tp = &sc.tris[leafSlot];
            const float4 ta = tp->a, tb = tp->b, tc = tp->c;
            // the any-hit record of a non-opaque triangle travels with the triangle: five independent loads, one wait
            float4 aa = make_float4(0.0f, 0.0f, 0.0f, 0.0f), ab = aa;
            const bool nonOpaque = ALPHA && (leafBits & kLeafNonOpaque) != 0u;
            if (nonOpaque)
            {
                aa = sc.alphaTris[leafSlot].a;
                ab = sc.alphaTris[leafSlot].b;
            }
            float t, u, v;
            ref = st.sp ? (int)st.pop() : kRefDone;
            bool candidate = intersectTri(F3(ta.x, ta.y, ta.z), F3(ta.w, tb.x, tb.y), F3(tb.z, tb.w, tc.x), o, d, PT_TMIN, PT_TMAX, t, u, v);
            uint32_t pair = __float_as_uint(tc.y), prim = __float_as_uint(tc.z);
            if (ALPHA && candidate && nonOpaque)
            {
                // the any-hit stage; the nearest ignored candidate of a closest ray (the decal) is the IO's business: it keeps
                // (distance, triangle) in the slot's record in memory, not in registers of every lane.  One behind the hit
                // found so far cannot matter.
                const float alpha = hitAlpha(sc, aa, ab, __float_as_uint(tc.w), u, v);
                candidate = ANY_HIT ? !(alpha < 1.0f) : !(alpha < 0.5f);
                if (!ANY_HIT && !candidate && t <= best.t)
                    io.ignored(t, u, v, leafSlot, sc);
                if (candidate && (!ANY_HIT || IO::kNeedsPrim))
                {
                    // the ids are read again behind the alpha fetch instead of being held in registers across it
                    const float4 again = sc.tris[fetchAgainAfter(leafSlot, alpha)].c;
                    pair = __float_as_uint(again.y);
                    prim = __float_as_uint(again.z);
                }
            }
            if (candidate)
            {
                // the prim id of the best hit is not carried in a register: an exact tie inside one (instance, mesh) pair
                // is rare enough to re-read it from the triangle record
                if (ANY_HIT)
                {
                    best.pair = IO::kNeedsPrim ? pair : 0u; // a shadow query only asks whether
                    best.slot = leafSlot;
                    ref = kRefDone;
                }
                else if (t < best.t || (t == best.t && (pair < best.pair || (pair == best.pair && prim < __float_as_uint(sc.tris[io.bestSlot(item)].c.z)))))
                {
                    // where the best hit lies on its triangle goes to the IO's record at once (write-through: a ray improves
                    // its hit two or three times): three registers less in every lane for the length of the walk.  (Keeping
                    // them in registers where the kernel has the room -- 57 instead of 54 VGPRs in the opaque closest kernel --
                    // measured -0.8 % / +0.2 % / +0.3 % on chess_like / street_like / atrium_like: the stores are not what it waits for.)
                    best.t = t;
                    best.pair = pair;
                    io.improve(item, t, u, v, leafSlot);
                }
            }
        }

        // ---- retire finished rays
        if (have && ref == kRefDone)
        {
            if (IO::kNeedsPrim)
                best.prim = best.pair != 0xffffffffu ? __float_as_uint(sc.tris[ANY_HIT ? best.slot : io.bestSlot(item)].c.z) : 0xffffffffu;
            io.store(item, best, best.pair != 0xffffffffu, ANY_HIT);
            have = false;
        }
    }
}

#undef PT_TMIN
#undef PT_TMAX

} // namespace ptd
