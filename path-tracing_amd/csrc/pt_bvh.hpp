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
//   BvhNode   64 B  both children's AABBs + child refs: one fetch = 4 x dwordx4
//   Tri       48 B  v0, e1, e2 (Moeller-Trumbore form) + (pair, prim) ids, in leaf order
// child ref >= 0: internal node index; < 0: leaf, triangle slot = ~ref.
#pragma once

#include "pt_device.hpp"

namespace ptd
{

struct BvhNode
{
    float4 a; // c0.lo.xyz, c0.hi.x
    float4 b; // c0.hi.yz, c1.lo.xy
    float4 c; // c1.lo.z, c1.hi.xyz
    int4 d;   // child0, child1, leaf triangle counts of child0 / child1
};
static_assert(sizeof(BvhNode) == 64, "BvhNode is 64 B");

struct Tri
{
    float4 a; // v0.xyz, e1.x
    float4 b; // e1.yz, e2.xy
    float4 c; // e2.z, pair (bits), prim (bits), unused
};
static_assert(sizeof(Tri) == 48, "Tri is 48 B");

struct Hit
{
    float t, u, v;
    uint32_t pair, prim; // pair == 0xffffffff: miss
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
                            float4 *__restrict__ boxHi, uint32_t *__restrict__ sceneBounds)
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
    const f3 e1 = w[1] - w[0], e2 = w[2] - w[0];
    Tri t;
    t.a = make_float4(w[0].x, w[0].y, w[0].z, e1.x);
    t.b = make_float4(e1.y, e1.z, e2.x, e2.y);
    t.c = make_float4(e2.z, __uint_as_float(p), __uint_as_float(prim), 0.0f);
    triTmp[g] = t;

    // bounds from the same p0, p0+e1, p0+e2 the intersection test sees, padded so the slab
    // test can never reject a ray the triangle test accepts
    float l[3], h[3];
    const float p0[3] = { w[0].x, w[0].y, w[0].z }, a1[3] = { e1.x, e1.y, e1.z }, a2[3] = { e2.x, e2.y, e2.z };
    for (int a = 0; a < 3; a++)
    {
        const float q1 = p0[a] + a1[a], q2 = p0[a] + a2[a];
        const float mn = fminf(p0[a], fminf(q1, q2)), mx = fmaxf(p0[a], fmaxf(q1, q2));
        const float pad = 1e-5f * fmaxf(fabsf(mn), fabsf(mx)) + 1e-7f;
        l[a] = mn - pad;
        h[a] = mx + pad;
    }
    boxLo[g] = make_float4(l[0], l[1], l[2], 0.0f);
    boxHi[g] = make_float4(h[0], h[1], h[2], 0.0f);
    for (int a = 0; a < 3; a++)
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

__global__ void k_morton(uint32_t n, const float4 *__restrict__ boxLo, const float4 *__restrict__ boxHi,
                         const uint32_t *__restrict__ sceneBounds, uint64_t *__restrict__ keys, uint32_t *__restrict__ vals)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n)
        return;
    const float4 lo = boxLo[g], hi = boxHi[g];
    const float c[3] = { 0.5f * (lo.x + hi.x), 0.5f * (lo.y + hi.y), 0.5f * (lo.z + hi.z) };
    uint32_t q[3];
    for (int a = 0; a < 3; a++)
    {
        const float mn = unorderedFloat(sceneBounds[a]), mx = unorderedFloat(sceneBounds[3 + a]);
        const float ext = mx - mn;
        float f = ext > 0.0f ? (c[a] - mn) / ext : 0.0f;
        f = f == f ? fminf(fmaxf(f, 0.0f), 1.0f) : 0.0f;
        const uint32_t v = (uint32_t)(f * 2097151.0f);
        q[a] = v > 2097151u ? 2097151u : v;
    }
    keys[g] = (expandBits21(q[0]) << 2) | (expandBits21(q[1]) << 1) | expandBits21(q[2]);
    vals[g] = g;
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
                         int *__restrict__ parentOfLeaf, int2 *__restrict__ ranges)
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
    ranges[i] = make_int2(lo, hi); // sorted-leaf range covered by this node (inclusive)
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

// Final layout: 64-B nodes holding both children's boxes; triangles in leaf order.  A
// child subtree of at most kMaxLeafTris triangles becomes ONE leaf (its triangles are
// contiguous in Morton order): ref = ~first, count in d.z / d.w.
constexpr int kDefaultLeafTris = 1; // measured on MI355X: 1 -> 591, 2 -> 573, 4 -> 518, 8 -> 437 Msamples/s (chess_like 1080p)

__global__ void k_emit(int n, const uint32_t *__restrict__ vals, const float4 *__restrict__ boxLo,
                       const float4 *__restrict__ boxHi, const int2 *__restrict__ children, const int2 *__restrict__ ranges,
                       const float4 *__restrict__ nodeLo, const float4 *__restrict__ nodeHi, const Tri *__restrict__ triTmp,
                       BvhNode *__restrict__ nodes, Tri *__restrict__ tris, int kMaxLeafTris)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        tris[i] = triTmp[vals[i]];
    if (i >= n - 1)
        return;
    const int2 ch = children[i];
    float4 l0, h0, l1, h1;
    if (ch.x < 0) { const uint32_t g = vals[~ch.x]; l0 = boxLo[g]; h0 = boxHi[g]; }
    else { l0 = nodeLo[ch.x]; h0 = nodeHi[ch.x]; }
    if (ch.y < 0) { const uint32_t g = vals[~ch.y]; l1 = boxLo[g]; h1 = boxHi[g]; }
    else { l1 = nodeLo[ch.y]; h1 = nodeHi[ch.y]; }
    int ref0 = ch.x, ref1 = ch.y, cnt0 = 1, cnt1 = 1;
    if (ref0 >= 0)
    {
        const int2 rg = ranges[ref0];
        if (rg.y - rg.x + 1 <= kMaxLeafTris) { cnt0 = rg.y - rg.x + 1; ref0 = ~rg.x; }
    }
    if (ref1 >= 0)
    {
        const int2 rg = ranges[ref1];
        if (rg.y - rg.x + 1 <= kMaxLeafTris) { cnt1 = rg.y - rg.x + 1; ref1 = ~rg.x; }
    }
    BvhNode nd;
    nd.a = make_float4(l0.x, l0.y, l0.z, h0.x);
    nd.b = make_float4(h0.y, h0.z, l1.x, l1.y);
    nd.c = make_float4(l1.z, h1.x, h1.y, h1.z);
    nd.d = make_int4(ref0, ref1, cnt0, cnt1);
    nodes[i] = nd;
}

// a one-triangle scene has no internal node: give it a root whose second child is empty
__global__ void k_single_leaf_root(const float4 *boxLo, const float4 *boxHi, const Tri *triTmp, BvhNode *nodes, Tri *tris)
{
    tris[0] = triTmp[0];
    BvhNode nd;
    nd.a = make_float4(boxLo[0].x, boxLo[0].y, boxLo[0].z, boxHi[0].x);
    nd.b = make_float4(boxHi[0].y, boxHi[0].z, 1e30f, 1e30f);
    nd.c = make_float4(1e30f, -1e30f, -1e30f, -1e30f);
    nd.d = make_int4(~0, ~0, 1, 0);
    nodes[0] = nd;
}

// ---------------------------------------------------------------------------------
// Traversal
// ---------------------------------------------------------------------------------

constexpr int kLdsStack = 24;   // entries per lane kept in LDS (lane-interleaved: no bank conflicts)
constexpr int kSpillStack = 72; // rest of the worst-case LBVH depth (63 Morton bits + 32 tie-break bits)
constexpr uint32_t kMaxNodeVisits = 1u << 20;

struct TraceScene
{
    const BvhNode *nodes;
    const Tri *tris;
    uint32_t triCount;
};

struct Stack
{
    uint32_t *lds; // &s_stack[0][lane]
    uint32_t stride;
    uint32_t spill[kSpillStack];
    int sp;
    PT_DEV void push(uint32_t v)
    {
        if (sp < kLdsStack)
            lds[sp * stride] = v;
        else if (sp - kLdsStack < kSpillStack)
            spill[sp - kLdsStack] = v;
        sp++;
    }
    PT_DEV uint32_t pop()
    {
        sp--;
        if (sp < kLdsStack)
            return lds[sp * stride];
        return (sp - kLdsStack < kSpillStack) ? spill[sp - kLdsStack] : 0u;
    }
};

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
    lo = fmaxf(lo, fminf(t0, t1));
    hi = fminf(hi, fmaxf(t0, t1));
    lo = fmaxf(lo, tmin);
    hi = fminf(hi, tmax);
    tn = lo;
    return lo <= hi * 1.0000004f;
}

// Closest hit = min t over all triangles the ray hits in (tmin, tmax); ties go to the
// smaller (pair, prim), i.e. the smaller global triangle id -- independent of tree shape.
template <bool ANY_HIT>
PT_DEV bool traceRay(const TraceScene &sc, f3 o, f3 d, float tmin, float tmax, Stack &st, Hit &best)
{
    best.t = tmax;
    best.u = best.v = 0.0f;
    best.pair = 0xffffffffu;
    best.prim = 0xffffffffu;
    if (sc.triCount == 0)
        return false;
    const f3 id = F3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    st.sp = 0;
    int node = 0;
    // a legitimate ray visits a few hundred nodes; the bound only turns a corrupted tree
    // into a wrong pixel instead of a hung GPU
    for (uint32_t visits = 0; visits < kMaxNodeVisits; visits++)
    {
        const BvhNode *np = &sc.nodes[node];
        const float4 na = np->a, nb = np->b, nc = np->c;
        const int4 nd = np->d;
        const float lim = best.t; // == tmax until something is hit
        float tn0, tn1;
        bool h0 = slab(na.x, na.y, na.z, na.w, nb.x, nb.y, o, id, tmin, lim, tn0);
        bool h1 = slab(nb.z, nb.w, nc.x, nc.y, nc.z, nc.w, o, id, tmin, lim, tn1);
#pragma unroll
        for (int k = 0; k < 2; k++)
        {
            const int ref = k ? nd.y : nd.x;
            const bool h = k ? h1 : h0;
            if (h && ref < 0)
            {
                const int cnt = k ? nd.w : nd.z;
                const Tri *tp = &sc.tris[~ref];
                for (int q = 0; q < cnt; q++, tp++)
                {
                    const float4 ta = tp->a, tb = tp->b, tc = tp->c;
                    float t, u, v;
                    if (intersectTri(F3(ta.x, ta.y, ta.z), F3(ta.w, tb.x, tb.y), F3(tb.z, tb.w, tc.x), o, d, tmin, tmax, t, u, v))
                    {
                        if (ANY_HIT)
                            return true;
                        const uint32_t pair = __float_as_uint(tc.y), prim = __float_as_uint(tc.z);
                        if (t < best.t || (t == best.t && (pair < best.pair || (pair == best.pair && prim < best.prim))))
                        {
                            best.t = t;
                            best.u = u;
                            best.v = v;
                            best.pair = pair;
                            best.prim = prim;
                        }
                    }
                }
            }
        }
        h0 = h0 && nd.x >= 0;
        h1 = h1 && nd.y >= 0;
        if (h0 && h1)
        {
            const bool firstIs0 = tn0 <= tn1;
            st.push((uint32_t)(firstIs0 ? nd.y : nd.x));
            node = firstIs0 ? nd.x : nd.y;
        }
        else if (h0)
            node = nd.x;
        else if (h1)
            node = nd.y;
        else
        {
            if (st.sp == 0)
                break;
            node = (int)st.pop();
        }
    }
    return best.pair != 0xffffffffu;
}

} // namespace ptd
