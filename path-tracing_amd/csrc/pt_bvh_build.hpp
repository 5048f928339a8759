// pt_bvh_build.hpp -- builder kernels of the software BVH (pt_bvh.hpp has the layout and the traversal): world-space triangle
// set-up and padded bounds, 63-bit Morton codes, a hand-written LSD radix sort, the binary topology (PLOC, or Karras 2012),
// bottom-up refit, the collapse into 4-wide quantised nodes, the breadth-first relayout, and the any-hit records.
//
// Stands in for the driver's VK_KHR_acceleration_structure build (Path-Tracing/Renderer/AccelerationStructure.cpp:64-301):
// the reference gives only the INPUT layout (one BLAS per Model, one geometry per Mesh with an optional baked mesh
// transform, one TLAS instance per ModelInstance); the algorithm is new.
#pragma once

#include "pt_bvh.hpp"

namespace ptd
{

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
    // centroid bounds of the scene for the Morton grid (a refit keeps the sorted order and does not need them): reduced over the
    // wave first -- six same-address atomics per TRIANGLE were most of this kernel's 4.4 ms at 4 M triangles
    if (refit)
        return;
    for (int a = 0; a < 3; a++)
    {
        const float c = 0.5f * (l[a] + h[a]);
        const bool ok = !isInert && c == c && fabsf(c) < 3.0e38f;
        uint32_t mn = ok ? orderedFloat(c) : 0xffffffffu, mx = ok ? orderedFloat(c) : 0u;
        const bool fullWave = __ballot(1) == ~0ull; // (the last wave of the launch may be partial: its lanes go one by one)
        if (fullWave)
            for (int o = 32; o > 0; o >>= 1)
            {
                mn = min(mn, (uint32_t)__shfl_xor((int)mn, o));
                mx = max(mx, (uint32_t)__shfl_xor((int)mx, o));
            }
        if (!fullWave || (threadIdx.x & 63u) == 0u)
        {
            if (mn != 0xffffffffu)
                atomicMin(&sceneBounds[a], mn);
            if (mx != 0u)
                atomicMax(&sceneBounds[3 + a], mx);
        }
    }
}

// ---- triangle pre-splitting (round 4) ------------------------------------------------------------------------------
// A triangle that is large against the cells of the tree around it -- a floor or wall made of two triangles, a long curtain
// strip, a diagonal beam -- has ONE box in a tree of single-triangle leaves: every ray that crosses that box pays a node visit
// up the whole chain of ancestors it inflates, and the test.  Splitting REFERENCES (Karras and Aila 2013, section 5; Ernst and
// Greiner 2007): before the Morton sort such a triangle is cut, along planes of the Morton grid, into up to kMaxSplitsPerTri
// pieces; each piece becomes a leaf of its own with the tight box of the triangle clipped to its cell, and all of them name the
// same triangle record.  What a ray hits does not change: a closest-hit walk that meets the triangle through two leaves finds the
// same (t, u, v, id) twice -- the second is no improvement --, an occlusion walk ends at the first, and the "nearest ignored
// candidate" of the any-hit stage is idempotent for a repeated candidate (tests/test_gpu_parity.py: brute force, oracle images).
//   priority  p = cbrt(2^-depth * (area of the box - smallest area its pieces could have)): depth = level of the coarsest
//             Morton plane that crosses the box (the earlier in the tree a plane, the more a straddling box costs), the ideal
//             area |n.x| + |n.y| + |n.z| of the edge cross product n = the triangle's three projections
//   budget    pieces(g) = 1 + floor(D p(g)), D chosen so that the extra references are `budget` x the triangle count
//   cut       at the coarsest Morton plane crossing the piece's tight box (midpoint where none does), the piece's count shared
//             between the halves by their widths; a piece's box = the triangle clipped to its cell, where the cell is widened by
//             the leaf padding of that axis (any point the triangle test accepts has a point of the triangle within the padding,
//             pt_bvh.hpp "Arithmetic"; that point lies in the widened cell of the side the accepted point is on), then padded
//             like every leaf box
// Static scenes only: a refit keeps topology and references, and the pieces of a moving triangle would have to be cut again.
constexpr uint32_t kMaxSplitsPerTri = 64;
constexpr double kSplitPriorityScale = 1048576.0; // priorities are summed as integers (order-independent: the build stays deterministic)

struct SplitGrid // the Morton grid of k_morton
{
    float mn[3], cell[3]; // cell = extent / 2^21 per axis (one value for all three with cubic cells)
};
PT_DEV SplitGrid splitGrid(const uint32_t *sceneBounds, int cubic)
{
    SplitGrid g;
    float widest = 0.0f;
    for (int a = 0; a < 3; a++)
        widest = fmaxf(widest, unorderedFloat(sceneBounds[3 + a]) - unorderedFloat(sceneBounds[a]));
    for (int a = 0; a < 3; a++)
    {
        g.mn[a] = unorderedFloat(sceneBounds[a]);
        const float ext = cubic ? widest : unorderedFloat(sceneBounds[3 + a]) - g.mn[a];
        g.cell[a] = ext > 0.0f ? ext / 2097152.0f : 0.0f;
    }
    return g;
}
// the coarsest grid plane strictly inside (lo, hi) on axis a: its depth in the tree of spatial medians (3 * level + axis;
// 1000 = none) and position
PT_DEV int coarsestPlane(const SplitGrid &g, int a, float lo, float hi, float &plane)
{
    if (!(g.cell[a] > 0.0f) || !(hi > lo))
        return 1000;
    const float fl = (lo - g.mn[a]) / g.cell[a], fh = (hi - g.mn[a]) / g.cell[a];
    const int cl = (int)fminf(fmaxf(fl, 0.0f), 2097151.0f), ch = (int)fminf(fmaxf(fh, 0.0f), 2097151.0f);
    if (cl == ch)
        return 1000;
    const int b = 31 - __clz(cl ^ ch); // highest bit in which the two cells differ
    const int first = (ch >> b) << b;  // first cell of the upper half
    plane = g.mn[a] + (float)first * g.cell[a];
    if (!(plane > lo && plane < hi))
        return 1000;
    return 3 * (20 - b) + a;
}

PT_DEV float boxArea(const float *lo, const float *hi)
{
    const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    return 2.0f * (dx * dy + dy * dz + dz * dx);
}

// per triangle: its priority (0 for an inert one), summed over the scene as an integer
__global__ void k_split_priority(uint32_t n, const Tri *__restrict__ triTmp, const float4 *__restrict__ boxLo, const float4 *__restrict__ boxHi,
                                 const uint8_t *__restrict__ inert, const uint32_t *__restrict__ sceneBounds, int cubic, float *__restrict__ priority,
                                 unsigned long long *__restrict__ sum)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    float p = 0.0f;
    if (g < n && !inert[g])
    {
        const SplitGrid grid = splitGrid(sceneBounds, cubic);
        const float4 l4 = boxLo[g], h4 = boxHi[g];
        const float lo[3] = { l4.x, l4.y, l4.z }, hi[3] = { h4.x, h4.y, h4.z };
        int depth = 1000;
        for (int a = 0; a < 3; a++)
        {
            float plane;
            const int d = coarsestPlane(grid, a, lo[a], hi[a], plane);
            depth = d < depth ? d : depth;
        }
        if (depth < 1000)
        {
            const Tri t = triTmp[g];
            const f3 nrm = cross(F3(t.a.w, t.b.x, t.b.y), F3(t.b.z, t.b.w, t.c.x));
            const float ideal = fabsf(nrm.x) + fabsf(nrm.y) + fabsf(nrm.z);
            const float excess = boxArea(lo, hi) - ideal;
            if (excess > 0.0f)
                p = cbrtf(exp2f(-(float)depth / 3.0f) * excess); // (depth counts the three axes of a level separately)
        }
        if (!(p == p) || p > 1.0e6f)
            p = 0.0f;
    }
    if (g < n)
        priority[g] = p;
    // one atomic per wave
    unsigned long long q = (unsigned long long)((double)p * kSplitPriorityScale);
    for (int off = 32; off > 0; off >>= 1)
        q += __shfl_down(q, off);
    if ((threadIdx.x & 63u) == 0 && q)
        atomicAdd(sum, q);
}

// pieces per triangle (as an array to be scanned): 1 + floor(D p), at most kMaxSplitsPerTri
__global__ void k_split_count(uint32_t n, const float *__restrict__ priority, float perPriority, uint32_t *__restrict__ count)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n)
        return;
    const float extra = floorf(perPriority * priority[g]);
    uint32_t c = 1u + (extra > 0.0f ? (uint32_t)fminf(extra, (float)(kMaxSplitsPerTri - 1)) : 0u);
    count[g] = c;
}

// the triangle (v0, v0 + e1, v0 + e2) clipped to a box: Sutherland-Hodgman against its six planes; returns the vertex count
// (0: nothing left) and the bounds of what is left
PT_DEV int clipTriToBox(const Tri &t, const float *cLo, const float *cHi, float *outLo, float *outHi)
{
    float pa[12][3], pb[12][3];
    int n = 3;
    pa[0][0] = t.a.x; pa[0][1] = t.a.y; pa[0][2] = t.a.z;
    pa[1][0] = t.a.x + t.a.w; pa[1][1] = t.a.y + t.b.x; pa[1][2] = t.a.z + t.b.y;
    pa[2][0] = t.a.x + t.b.z; pa[2][1] = t.a.y + t.b.w; pa[2][2] = t.a.z + t.c.x;
    for (int plane = 0; plane < 6 && n > 0; plane++)
    {
        const int a = plane >> 1;
        const bool upper = plane & 1; // keep x[a] <= bound, else x[a] >= bound
        const float bound = upper ? cHi[a] : cLo[a];
        int m = 0;
        for (int i = 0; i < n; i++)
        {
            const float *p = pa[i], *q = pa[i + 1 == n ? 0 : i + 1];
            const bool pin = upper ? p[a] <= bound : p[a] >= bound, qin = upper ? q[a] <= bound : q[a] >= bound;
            if (pin && m < 12)
            {
                pb[m][0] = p[0]; pb[m][1] = p[1]; pb[m][2] = p[2];
                m++;
            }
            if (pin != qin && m < 12)
            {
                const float w = (bound - p[a]) / (q[a] - p[a]);
                for (int k = 0; k < 3; k++)
                    pb[m][k] = p[k] + (q[k] - p[k]) * w;
                pb[m][a] = bound;
                m++;
            }
        }
        n = m;
        for (int i = 0; i < n; i++)
            for (int k = 0; k < 3; k++)
                pa[i][k] = pb[i][k];
    }
    for (int k = 0; k < 3; k++)
    {
        outLo[k] = 3.0e38f;
        outHi[k] = -3.0e38f;
    }
    for (int i = 0; i < n; i++)
        for (int k = 0; k < 3; k++)
        {
            outLo[k] = fminf(outLo[k], pa[i][k]);
            outHi[k] = fmaxf(outHi[k], pa[i][k]);
        }
    return n;
}

// the references of triangle g from refBase[g] on: its pieces' boxes (or its own box, when it is not split)
__global__ void k_split_write(uint32_t n, const Tri *__restrict__ triTmp, const float4 *__restrict__ boxLo, const float4 *__restrict__ boxHi,
                              const uint8_t *__restrict__ inert, const uint32_t *__restrict__ sceneBounds, int cubic, const uint32_t *__restrict__ refBase,
                              uint32_t total, float4 *__restrict__ refLo, float4 *__restrict__ refHi, uint32_t *__restrict__ refTri,
                              uint8_t *__restrict__ refInert)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n)
        return;
    const uint32_t base = refBase[g], pieces = (g + 1 < n ? refBase[g + 1] : total) - base;
    const float4 l4 = boxLo[g], h4 = boxHi[g];
    if (pieces <= 1u)
    {
        refLo[base] = l4;
        refHi[base] = h4;
        refTri[base] = g;
        refInert[base] = inert[g];
        return;
    }
    const SplitGrid grid = splitGrid(sceneBounds, cubic);
    const Tri t = triTmp[g];
    // the triangle's own leaf padding per axis (k_tri_setup), recovered from its padded and its exact bounds
    float pad[3], tLo[3], tHi[3];
    {
        const float p0[3] = { t.a.x, t.a.y, t.a.z }, a1[3] = { t.a.w, t.b.x, t.b.y }, a2[3] = { t.b.z, t.b.w, t.c.x };
        const float bl[3] = { l4.x, l4.y, l4.z };
        for (int a = 0; a < 3; a++)
        {
            const float q1 = p0[a] + a1[a], q2 = p0[a] + a2[a];
            tLo[a] = fminf(p0[a], fminf(q1, q2));
            tHi[a] = fmaxf(p0[a], fmaxf(q1, q2));
            pad[a] = tLo[a] - bl[a];
        }
    }
    // explicit stack of (cell, pieces): a cell is the triangle's exact bounds cut by the planes chosen so far
    float sLo[kMaxSplitsPerTri][3], sHi[kMaxSplitsPerTri][3];
    uint32_t sCount[kMaxSplitsPerTri];
    int sp = 0;
    for (int a = 0; a < 3; a++)
    {
        sLo[0][a] = tLo[a];
        sHi[0][a] = tHi[a];
    }
    sCount[0] = pieces;
    sp = 1;
    uint32_t out = base;
    const uint32_t end = base + pieces;
    while (sp > 0 && out < end)
    {
        sp--;
        float cLo[3], cHi[3], wLo[3], wHi[3], bLo[3], bHi[3];
        const uint32_t s = sCount[sp];
        for (int a = 0; a < 3; a++)
        {
            cLo[a] = sLo[sp][a];
            cHi[a] = sHi[sp][a];
            wLo[a] = cLo[a] - pad[a]; // the cell widened by the padding: see the header comment
            wHi[a] = cHi[a] + pad[a];
        }
        const int verts = clipTriToBox(t, wLo, wHi, bLo, bHi);
        if (verts == 0) // cannot happen for cells cut from tight bounds; keep the count with a box that is certainly conservative
            for (int a = 0; a < 3; a++)
            {
                bLo[a] = cLo[a];
                bHi[a] = cHi[a];
            }
        int axis = -1, bestDepth = 1000;
        float plane = 0.0f;
        if (s > 1u)
        {
            // cut inside the piece's tight bounds (within the cell): coarsest Morton plane, else the midpoint of the longest axis
            float lo[3], hi[3], longest = 0.0f;
            for (int a = 0; a < 3; a++)
            {
                lo[a] = fmaxf(bLo[a], cLo[a]);
                hi[a] = fminf(bHi[a], cHi[a]);
                longest = fmaxf(longest, hi[a] - lo[a]);
            }
            for (int a = 0; a < 3; a++)
            {
                float pl;
                // (not along an axis the piece is thin in: the halves would be as large as the piece)
                const int d = (hi[a] - lo[a]) >= 0.25f * longest ? coarsestPlane(grid, a, lo[a], hi[a], pl) : 1000;
                if (d < bestDepth)
                {
                    bestDepth = d;
                    axis = a;
                    plane = pl;
                }
            }
            if (axis < 0)
                for (int a = 0; a < 3; a++)
                    if (hi[a] - lo[a] == longest && longest > 0.0f)
                    {
                        const float mid = 0.5f * (lo[a] + hi[a]);
                        if (mid > lo[a] && mid < hi[a])
                        {
                            axis = a;
                            plane = mid;
                        }
                        break;
                    }
            if (axis >= 0 && sp + 2 <= (int)kMaxSplitsPerTri)
            {
                const float wl = plane - lo[axis], wr = hi[axis] - plane;
                uint32_t sl = (uint32_t)floorf((float)s * wl / (wl + wr) + 0.5f);
                sl = sl < 1u ? 1u : (sl > s - 1u ? s - 1u : sl);
                for (int a = 0; a < 3; a++)
                {
                    sLo[sp][a] = lo[a]; sHi[sp][a] = hi[a];
                    sLo[sp + 1][a] = lo[a]; sHi[sp + 1][a] = hi[a];
                }
                sHi[sp][axis] = plane; // lower half
                sCount[sp] = sl;
                sLo[sp + 1][axis] = plane; // upper half
                sCount[sp + 1] = s - sl;
                sp += 2;
                continue;
            }
        }
        // a leaf box: the clipped bounds, padded like every leaf box and kept inside the triangle's own padded box; a piece that
        // could not be cut stands for all of its count (repeated references are harmless)
        for (uint32_t k = 0; k < s && out < end; k++, out++)
        {
            refLo[out] = make_float4(fmaxf(bLo[0] - pad[0], l4.x), fmaxf(bLo[1] - pad[1], l4.y), fmaxf(bLo[2] - pad[2], l4.z), 0.0f);
            refHi[out] = make_float4(fminf(bHi[0] + pad[0], h4.x), fminf(bHi[1] + pad[1], h4.y), fminf(bHi[2] + pad[2], h4.z), 0.0f);
            refTri[out] = g;
            refInert[out] = 0;
        }
    }
    for (; out < end; out++) // (stack exhausted early: cannot happen; keep every slot defined)
    {
        refLo[out] = l4;
        refHi[out] = h4;
        refTri[out] = g;
        refInert[out] = 0;
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

// ---- level lists of the binary tree (round 5): bottom-up passes without fences ---------------------------------------
// k_refit above climbs from every leaf and hands a node to the second thread that arrives: two agent-scope fences and one
// atomic per node.  On this chip an agent-scope release writes the XCD's L2 back (eight L2s, not coherent with one another), so
// the pass cost 11 ms at 2 M triangles and 25 ms at 4 M -- more than a rendered frame -- and the reinsertion passes, the
// collapse's pricing and every animated frame's refit pay it.  Instead: the DEPTH of every node by pointer jumping over the parent
// links (log2(depth) gather passes: d[i] += d[anc[i]], anc[i] = anc[anc[i]]), the nodes sorted by depth with one or two passes of the
// radix sort above (stable, deterministic), and one launch per level from the deepest up -- a kernel boundary is all the ordering a
// level needs.  The lists follow the topology: recomputed after PLOC and after every reinsertion pass, kept for the refits of an
// animation (ptx_update_animation keeps topology and order).
constexpr uint32_t kMaxTreeLevels = 4096; // deeper trees (never seen: PLOC trees of 2-4 M triangles are 40-70 deep) take k_refit

__global__ void k_depth_init(int nodes, const int *__restrict__ parentOfNode, uint32_t *__restrict__ depth, int *__restrict__ anc)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nodes)
        return;
    const int p = parentOfNode[i];
    depth[i] = p >= 0 ? 1u : 0u;
    anc[i] = p;
}

// one pointer-jumping pass (ping-pong buffers); *pending becomes non-zero while some node has not reached the root
__global__ void k_depth_jump(int nodes, const uint32_t *__restrict__ dIn, const int *__restrict__ aIn, uint32_t *__restrict__ dOut,
                             int *__restrict__ aOut, uint32_t *__restrict__ pending)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nodes)
        return;
    int a = aIn[i];
    uint32_t d = dIn[i];
    if (a >= 0)
    {
        d += dIn[a];
        a = aIn[a];
    }
    dOut[i] = d;
    aOut[i] = a;
    if (__ballot(a >= 0) != 0ull && (threadIdx.x & 63u) == 0u)
        *pending = 1u;
}

__global__ void k_depth_keys(int nodes, const uint32_t *__restrict__ depth, uint64_t *__restrict__ keys, uint32_t *__restrict__ vals,
                             uint32_t *__restrict__ maxDepth)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t d = 0;
    if (i < nodes)
    {
        d = depth[i];
        keys[i] = d;
        vals[i] = (uint32_t)i;
    }
    // wave maximum, one atomic per wave
    for (int o = 32; o > 0; o >>= 1)
        d = max(d, (uint32_t)__shfl_xor((int)d, o));
    // (same-address atomics serialise at ~11 ns: 31 K of them were 0.35 ms of this kernel.  The maximum only grows, so a wave
    // whose value is not above what is already there has nothing to add)
    if ((threadIdx.x & 63u) == 0u && d > __builtin_nontemporal_load(maxDepth))
        atomicMax(maxDepth, d);
}

// first sorted position of every depth (levelStart[d]; the caller sets levelStart[maxDepth + 1] = nodes)
__global__ void k_level_starts(int nodes, const uint64_t *__restrict__ sortedKeys, uint32_t *__restrict__ levelStart)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nodes)
        return;
    const uint32_t d = (uint32_t)sortedKeys[i];
    if (i == 0 || (uint32_t)sortedKeys[i - 1] != d)
        levelStart[d] = (uint32_t)i;
}

// the boxes of one level: every child is a leaf or a node of the level below, written by the launch before this one
__global__ void k_refit_level(uint32_t first, uint32_t count, const uint32_t *__restrict__ order, const uint32_t *__restrict__ vals,
                              const float4 *__restrict__ boxLo, const float4 *__restrict__ boxHi, const int2 *__restrict__ children,
                              float4 *__restrict__ nodeLo, float4 *__restrict__ nodeHi)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count)
        return;
    const int node = (int)order[first + i];
    const int2 ch = children[node];
    float4 l0, h0, l1, h1;
    if (ch.x < 0) { const uint32_t g = vals[~ch.x]; l0 = boxLo[g]; h0 = boxHi[g]; }
    else { l0 = nodeLo[ch.x]; h0 = nodeHi[ch.x]; }
    if (ch.y < 0) { const uint32_t g = vals[~ch.y]; l1 = boxLo[g]; h1 = boxHi[g]; }
    else { l1 = nodeLo[ch.y]; h1 = nodeHi[ch.y]; }
    nodeLo[node] = make_float4(fminf(l0.x, l1.x), fminf(l0.y, l1.y), fminf(l0.z, l1.z), 0.0f);
    nodeHi[node] = make_float4(fmaxf(h0.x, h1.x), fmaxf(h0.y, h1.y), fmaxf(h0.z, h1.z), 0.0f);
}

// ---- reinsertion (round 4): the binary tree re-optimised before it is collapsed -----------------------------------
// PLOC decides every merge once, among neighbours in the Morton order; what it got wrong stays.  Parallel reinsertion (Meister
// and Bittner 2018) revisits it: every node x (a subtree or a leaf) looks for the place in the tree where it would cost least --
// removing x deletes its parent p (x's sibling s takes p's place) and shrinks the ancestors that held x only through p;
// inserting x beside a node t adds one node over (x, t) and grows t's ancestors below the common ancestor:
//   gain(t) = area(p) + sum over the ancestors a below the common one of (area(a) - area(a without x))
//             - area(x u t) - sum over t's ancestors u below the common one of (area(u u x) - area(u))
// The search walks up from p, level by level, and down the sibling subtree of each level with a stack, pruned by the best gain
// so far (the terms still to come are all losses).  Moves that gain are applied in parallel where they do not cross: a move
// claims every node on the path x .. common ancestor .. t with a 64-bit atomic max of (gain, x), and is applied only if it holds
// them all -- two moves that could knot the tree (each one's target inside the other's subtree) share a node of their paths.
// Then the boxes are recomputed bottom-up (k_refit) and the pass repeats.  Node ids, the root (node 0) and the arrays the
// collapse reads (children / parentOfNode / parentOfLeaf, leaf ref = ~sorted position) stay as PLOC made them.
struct ReinsertTree
{
    int n; // leaves
    int2 *children;
    int *parentOfNode, *parentOfLeaf;
    const float4 *nodeLo, *nodeHi;
    const uint32_t *vals;
    const float4 *leafLo, *leafHi;
};
PT_DEV int rtParent(const ReinsertTree &t, int ref) { return ref >= 0 ? t.parentOfNode[ref] : t.parentOfLeaf[~ref]; }
PT_DEV void rtSetParent(const ReinsertTree &t, int ref, int parent)
{
    if (ref >= 0)
        t.parentOfNode[ref] = parent;
    else
        t.parentOfLeaf[~ref] = parent;
}
PT_DEV void rtBox(const ReinsertTree &t, int ref, float *lo, float *hi)
{
    float4 l, h;
    if (ref >= 0) { l = t.nodeLo[ref]; h = t.nodeHi[ref]; }
    else { const uint32_t g = t.vals[~ref]; l = t.leafLo[g]; h = t.leafHi[g]; }
    lo[0] = l.x; lo[1] = l.y; lo[2] = l.z;
    hi[0] = h.x; hi[1] = h.y; hi[2] = h.z;
}
PT_DEV float rtArea(const float *lo, const float *hi)
{
    const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    return dx * dy + dy * dz + dz * dx;
}
PT_DEV float rtUnionArea(const float *lo, const float *hi, const float *lo2, const float *hi2)
{
    const float dx = fmaxf(hi[0], hi2[0]) - fminf(lo[0], lo2[0]), dy = fmaxf(hi[1], hi2[1]) - fminf(lo[1], lo2[1]),
                dz = fmaxf(hi[2], hi2[2]) - fminf(lo[2], lo2[2]);
    return dx * dy + dy * dz + dz * dx;
}
PT_DEV uint32_t rtIndex(const ReinsertTree &t, int ref) { return ref >= 0 ? (uint32_t)ref : (uint32_t)(t.n - 1) + (uint32_t)~ref; } // node or leaf -> lock slot
PT_DEV int rtRefOf(const ReinsertTree &t, uint32_t index) { return index < (uint32_t)(t.n - 1) ? (int)index : ~(int)(index - (uint32_t)(t.n - 1)); }

constexpr int kReinsertLevels = 16; // how far up the search goes
constexpr int kReinsertStack = 48;

// per x (every node but the root and the root's children, every leaf whose parent is not the root): best target, its gain, and
// the top of the path the move has to hold (the parent of the common ancestor, or the root)
__global__ void k_reinsert_find(ReinsertTree t, uint32_t stride, uint32_t phase, int *__restrict__ target, float *__restrict__ gain, int *__restrict__ top)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t index = k * stride + phase;
    if (index >= (uint32_t)(2 * t.n - 1))
        return;
    target[index] = kEmptyRef;
    gain[index] = 0.0f;
    const int x = rtRefOf(t, index);
    if (x == 0)
        return;
    const int p = rtParent(t, x);
    if (p <= 0) // the root's children stay: p would be deleted, and the root is node 0 by convention
        return;
    float xl[3], xh[3];
    rtBox(t, x, xl, xh);
    const float ax = rtArea(xl, xh);
    const int2 pc = t.children[p];
    const int s = pc.x == x ? pc.y : pc.x;
    float bl[3], bh[3]; // the box of what is left of the path's subtree once x is gone
    rtBox(t, s, bl, bh);
    float pl[3], ph[3];
    rtBox(t, p, pl, ph);
    float R = rtArea(pl, ph); // p is deleted
    float best = 0.0f;
    int bestT = kEmptyRef, bestTop = 0;
    int cur = p;
    int stRef[kReinsertStack];
    float stG[kReinsertStack];
    for (int level = 0; level < kReinsertLevels; level++)
    {
        // the sibling subtree of this level: at level 0 the subtree of s itself (x moves deeper into its sibling)
        const int anc = level == 0 ? p : rtParent(t, cur);
        if (anc < 0)
            break;
        int o;
        if (level == 0)
            o = s;
        else
        {
            const int2 ac = t.children[anc];
            o = ac.x == cur ? ac.y : ac.x;
        }
        int sp = 0;
        stRef[sp] = o;
        stG[sp] = 0.0f;
        sp++;
        while (sp > 0)
        {
            sp--;
            const int u = stRef[sp];
            const float g = stG[sp];
            float ul[3], uh[3];
            rtBox(t, u, ul, uh);
            const float au = rtUnionArea(ul, uh, xl, xh);
            const float d = R - g - au;
            if (d > best && !(level == 0 && u == s)) // (beside s: where it is now)
            {
                best = d;
                bestT = u;
                bestTop = level == 0 ? rtParent(t, p) : anc; // (p's parent gets s for p: it is part of every move)
            }
            if (u >= 0)
            {
                const float g2 = g + (au - rtArea(ul, uh));
                if (R - g2 - ax > best && sp + 2 <= kReinsertStack)
                {
                    const int2 uc = t.children[u];
                    stRef[sp] = uc.x; stG[sp] = g2; sp++;
                    stRef[sp] = uc.y; stG[sp] = g2; sp++;
                }
            }
        }
        if (level == 0)
            continue; // (next: the sibling of p under its parent; R and the path box are those of level 0)
        // one level up: `anc` loses x too and shrinks to (path box u o); beside the shrunken anc is a candidate as well
        float ol[3], oh[3], al[3], ah[3];
        rtBox(t, o, ol, oh);
        rtBox(t, anc, al, ah);
        for (int a = 0; a < 3; a++)
        {
            bl[a] = fminf(bl[a], ol[a]);
            bh[a] = fmaxf(bh[a], oh[a]);
        }
        R += rtArea(al, ah) - rtArea(bl, bh);
        cur = anc;
        if (anc != 0) // (beside the root there is no place)
        {
            const float d = R - rtUnionArea(bl, bh, xl, xh);
            if (d > best)
            {
                best = d;
                bestT = anc;
                const int up = rtParent(t, anc);
                bestTop = up >= 0 ? up : anc;
            }
        }
    }
    // (a gain below a millionth of the parent's area is rounding)
    if (bestT != kEmptyRef && best > 1e-6f * rtArea(pl, ph))
    {
        target[index] = bestT;
        gain[index] = best;
        top[index] = bestTop;
    }
}

PT_DEV unsigned long long rtKey(float g, uint32_t index) { return (unsigned long long)__float_as_uint(g) << 32 | index; }

// claim (mode 0) or check (mode 1) every node of the move's paths: x and its sibling, p and up to `top`, t and up to `top`
template <int MODE>
PT_DEV bool rtWalk(const ReinsertTree &t, uint32_t index, int target, int top, unsigned long long key, unsigned long long *lock)
{
    const int x = rtRefOf(t, index);
    const int p = rtParent(t, x);
    const int2 pc = t.children[p];
    const int s = pc.x == x ? pc.y : pc.x;
    bool ok = true;
    auto visit = [&](int ref) {
        if (MODE == 0)
            atomicMax(&lock[rtIndex(t, ref)], key);
        else
            ok = ok && lock[rtIndex(t, ref)] == key;
    };
    visit(x);
    visit(s);
    int guard = 0;
    for (int u = p; u >= 0 && guard < 4096; u = rtParent(t, u), guard++)
    {
        visit(u);
        if (u == top)
            break;
    }
    visit(target); // (a leaf's ref is negative: it is not a step of the loop below)
    for (int u = rtParent(t, target); u >= 0 && guard < 8192; u = rtParent(t, u), guard++)
    {
        visit(u);
        if (u == top)
            break;
    }
    return ok;
}

__global__ void k_reinsert_claim(ReinsertTree t, const int *__restrict__ target, const float *__restrict__ gain, const int *__restrict__ top,
                                 unsigned long long *__restrict__ lock)
{
    const uint32_t index = blockIdx.x * blockDim.x + threadIdx.x;
    if (index >= (uint32_t)(2 * t.n - 1) || target[index] == kEmptyRef)
        return;
    rtWalk<0>(t, index, target[index], top[index], rtKey(gain[index], index), lock);
}

// apply the moves that hold all their nodes; counts them
__global__ void k_reinsert_apply(ReinsertTree t, const int *__restrict__ target, const float *__restrict__ gain, const int *__restrict__ top,
                                 unsigned long long *__restrict__ lock, uint32_t *__restrict__ applied)
{
    const uint32_t index = blockIdx.x * blockDim.x + threadIdx.x;
    if (index >= (uint32_t)(2 * t.n - 1) || target[index] == kEmptyRef)
        return;
    const int tg = target[index];
    if (!rtWalk<1>(t, index, tg, top[index], rtKey(gain[index], index), lock))
        return;
    const int x = rtRefOf(t, index);
    const int p = rtParent(t, x);
    const int2 pc = t.children[p];
    const int s = pc.x == x ? pc.y : pc.x;
    const int a1 = rtParent(t, p);
    // remove: s takes p's place under a1
    {
        int2 c = t.children[a1];
        if (c.x == p) c.x = s; else c.y = s;
        t.children[a1] = c;
        rtSetParent(t, s, a1);
    }
    // insert: p becomes the node over (target, x), where the target stood
    const int tp = rtParent(t, tg);
    {
        int2 c = t.children[tp];
        if (c.x == tg) c.x = p; else c.y = p;
        t.children[tp] = c;
    }
    t.children[p] = make_int2(tg, x);
    rtSetParent(t, p, tp);
    rtSetParent(t, tg, p);
    rtSetParent(t, x, p);
    atomicAdd(applied, 1u);
}

// is this still a tree?  (verbose builds check after every reinsertion pass)  counts: [0] children whose parent pointer does not
// point back, [1] leaves that do not reach the root within 512 steps, [2] the longest leaf-to-root path
__global__ void k_tree_check(ReinsertTree t, uint32_t *__restrict__ counts)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < t.n - 1)
    {
        const int2 c = t.children[i];
        if (rtParent(t, c.x) != i)
            atomicAdd(&counts[0], 1u);
        if (rtParent(t, c.y) != i)
            atomicAdd(&counts[0], 1u);
    }
    if (i < t.n)
    {
        int u = t.parentOfLeaf[i], steps = 1;
        while (u > 0 && steps < 512)
        {
            u = t.parentOfNode[u];
            steps++;
        }
        if (u != 0)
            atomicAdd(&counts[1], 1u);
        atomicMax(&counts[2], (uint32_t)steps);
    }
}

// ---- cost-driven collapse (round 4) -------------------------------------------------------------------------------
// Which binary nodes become 4-wide nodes?  The greedy rule of round 1 -- start from a node's two children and, twice, replace
// the internal child with the largest surface area by its two children -- looks one step ahead.  This pass finds, for the given
// binary topology, the collapse that minimises the surface-area cost of the WIDE tree by dynamic programming over the binary
// tree (Ylitie, Karras, Laine 2017, section 4.1, with single-triangle leaves): a ray pays one node visit -- one fetch of 64
// bytes, four slab tests -- for every wide node whose box it enters, with probability proportional to that box's area, and the
// triangle tests are the same whatever node a triangle hangs from, so the objective is the summed area of the wide nodes.
//   T(x, j)   least cost of the subtree of binary node x when it may occupy at most j child slots of a wide node above it
//   leaf      T = the area of its box as the node above will store it (collapseCostOf), for every j
//   internal  T(x, 1) = area(x) + D(x, 4)                         x becomes a wide node: its children share four slots
//             T(x, j) = min(T(x, 1), D(x, j))        j = 2, 3, 4   or x is dissolved into the slots it was given
//             D(x, j) = min over k of T(left, k) + T(right, j - k)
// One thread per leaf walks up; the second thread to arrive at a node owns it (as k_refit).  Per node: the four costs and one
// byte of decisions -- keep_j (bit 2 + j: T(x, 1) <= D(x, j)), the best split of three slots (bit 0: left gets 2) and of four
// (bits 1..2: slots of the left child minus one) -- which k_emit follows from every node downwards.
constexpr float kCollapseTriCost = 1.0f; // a triangle test (three loads, the two-pass test) against a node visit (four loads, four slab tests)

// A child's T(., 1..4).  A leaf is not free after all: its box is stored in 8 bits per plane inside the box of the wide node it
// hangs from, i.e. grown by up to 1/255 of that node's extent per side -- a 6 cm triangle under a 10 m node is tested by every ray
// that passes within 4 cm of it.  The wide node is not known yet when the leaf's parent x is priced; x's own box is the smallest
// it can be (x dissolved into a larger node makes it worse), so the leaf costs the area of its box grown by extent(x) / 255.
PT_DEV float4 collapseCostOf(int ref, const float4 *cost, const uint32_t *vals, const float4 *boxLo, const float4 *boxHi, float gx, float gy, float gz)
{
    if (ref < 0)
    {
        const uint32_t g = vals[~ref];
        const float4 lo = boxLo[g], hi = boxHi[g];
        const float dx = (hi.x - lo.x) + 2.0f * gx, dy = (hi.y - lo.y) + 2.0f * gy, dz = (hi.z - lo.z) + 2.0f * gz;
        const float a = kCollapseTriCost * (dx * dy + dy * dz + dz * dx);
        return make_float4(a, a, a, a);
    }
    return loadUncached(&cost[ref]);
}

// T(node, 1..4) and the decision byte of one node from its children's (see above); UNCACHED: the children's costs come from other
// CUs of the same launch (k_collapse_cost) or from the launch before (k_collapse_cost_level)
template <bool UNCACHED>
PT_DEV void collapseNode(int node, const uint32_t *vals, const float4 *boxLo, const float4 *boxHi, const int2 *children, const float4 *nodeLo,
                         const float4 *nodeHi, float4 *cost, uint8_t *decide)
{
    const int2 ch = children[node];
    const float4 lo = nodeLo[node], hi = nodeHi[node];
    const float dx = hi.x - lo.x, dy = hi.y - lo.y, dz = hi.z - lo.z;
    const float q = 1.0f / 255.0f;
    const float4 l = ch.x >= 0 && !UNCACHED ? cost[ch.x] : collapseCostOf(ch.x, cost, vals, boxLo, boxHi, dx * q, dy * q, dz * q);
    const float4 r = ch.y >= 0 && !UNCACHED ? cost[ch.y] : collapseCostOf(ch.y, cost, vals, boxLo, boxHi, dx * q, dy * q, dz * q);
    const float area = dx * dy + dy * dz + dz * dx;
    const float d2 = l.x + r.x;
    const float d3a = l.x + r.y, d3b = l.y + r.x; // left 1 + right 2, left 2 + right 1
    const float d3 = fminf(d3a, d3b);
    const float d4a = l.x + r.z, d4b = l.y + r.y, d4c = l.z + r.x; // left 1, 2, 3
    const float d4 = fminf(d4a, fminf(d4b, d4c));
    const float t1 = area + d4;
    uint32_t bits = 0;
    bits |= d3b < d3a ? 1u : 0u;
    bits |= (d4b < d4a && d4b <= d4c) ? 2u : (d4c < d4a && d4c < d4b) ? 4u : 0u;
    bits |= t1 <= d2 ? 8u : 0u;
    bits |= t1 <= d3 ? 16u : 0u;
    bits |= t1 <= d4 ? 32u : 0u;
    decide[node] = (uint8_t)bits;
    cost[node] = make_float4(t1, fminf(t1, d2), fminf(t1, d3), fminf(t1, d4));
}

// the fence-and-atomic climb (kept for trees deeper than kMaxTreeLevels)
__global__ void k_collapse_cost(int n, const uint32_t *__restrict__ vals, const float4 *__restrict__ boxLo, const float4 *__restrict__ boxHi,
                                const int2 *__restrict__ children, const int *__restrict__ parentOfNode, const int *__restrict__ parentOfLeaf,
                                const float4 *__restrict__ nodeLo, const float4 *__restrict__ nodeHi, uint32_t *__restrict__ flags,
                                float4 *__restrict__ cost, uint8_t *__restrict__ decide)
{
    const int leaf = blockIdx.x * blockDim.x + threadIdx.x;
    if (leaf >= n)
        return;
    int node = parentOfLeaf[leaf];
    while (node >= 0)
    {
        __threadfence(); // release my child's costs / acquire the sibling's
        if (atomicAdd(&flags[node], 1u) == 0u)
            return;
        __threadfence();
        collapseNode<true>(node, vals, boxLo, boxHi, children, nodeLo, nodeHi, cost, decide);
        node = parentOfNode[node];
    }
}

// one level of the same pass over the level lists (k_refit_level's order): no fences
__global__ void k_collapse_cost_level(uint32_t first, uint32_t count, const uint32_t *__restrict__ order, const uint32_t *__restrict__ vals,
                                      const float4 *__restrict__ boxLo, const float4 *__restrict__ boxHi, const int2 *__restrict__ children,
                                      const float4 *__restrict__ nodeLo, const float4 *__restrict__ nodeHi, float4 *__restrict__ cost,
                                      uint8_t *__restrict__ decide)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count)
        collapseNode<false>((int)order[first + i], vals, boxLo, boxHi, children, nodeLo, nodeHi, cost, decide);
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
                       const uint32_t *__restrict__ indices, ShadeTri *__restrict__ shadeTris, const uint8_t *__restrict__ decide,
                       const uint32_t *__restrict__ refTri)
{
    // vals[i] = the leaf reference at sorted position i; refTri (null: the identity) names its triangle -- a triangle that was
    // split stands in several slots, record and all
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
    {
        const Tri t = triTmp[refTri ? refTri[vals[i]] : vals[i]];
        tris[i] = t;
        writeShadeTri(t, pairs, vertices, indices, &shadeTris[i]);
    }
    if (i >= n - 1)
        return;
    ChildBox c[kNodeWidth];
    int count = 2;
    if (decide)
    {
        // the cost-driven collapse (k_collapse_cost): node i shares its four slots between its children as decided there; a child
        // that was given j slots either stays one child (a leaf, j = 1, or keep_j) or hands its slots on to ITS children
        int ref[kNodeWidth] = { 0, 0, 0, 0 }, slots[kNodeWidth] = { 0, 0, 0, 0 }, sp = 0, out[kNodeWidth] = { 0, 0, 0, 0 };
        count = 0;
        {
            const int2 ch = children[i];
            const int k = 1 + (int)((decide[i] >> 1) & 3u);
            ref[0] = ch.y; slots[0] = 4 - k;
            ref[1] = ch.x; slots[1] = k;
            sp = 2;
        }
        while (sp > 0)
        {
            sp--;
            const int x = ref[sp], j = slots[sp];
            const uint32_t d = x >= 0 ? decide[x] : 0u;
            if (x < 0 || j == 1 || ((d >> (1 + j)) & 1u))
                out[count++] = x;
            else
            {
                const int2 ch = children[x];
                const int k = j == 2 ? 1 : j == 3 ? 1 + (int)(d & 1u) : 1 + (int)((d >> 1) & 3u);
                ref[sp] = ch.y; slots[sp] = j - k;
                ref[sp + 1] = ch.x; slots[sp + 1] = k;
                sp += 2;
            }
        }
        for (int k = 0; k < kNodeWidth; k++)
            if (k < count)
                fetchChild(out[k], vals, boxLo, boxHi, nodeLo, nodeHi, c[k]);
    }
    else
    {
        const int2 ch = children[i];
        fetchChild(ch.x, vals, boxLo, boxHi, nodeLo, nodeHi, c[0]);
        fetchChild(ch.y, vals, boxLo, boxHi, nodeLo, nodeHi, c[1]);
    }
    for (int round = 0; round < kNodeWidth - 2 && !decide; round++)
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

    uint32_t orderAxis = 0;
#ifdef PTX_EXP_AXIS_ORDER
    // experiment (docs/EXPERIMENTS.md, round 5): the children stored in DESCENDING order of their centroids along the axis on which
    // the centroids spread most; the closest-hit walk then enters the highest hit slot first for a ray that travels up that axis and
    // the lowest for one that travels down it -- a fixed build-time order instead of the sorting network
    {
        float lo3[3] = { 3.0e38f, 3.0e38f, 3.0e38f }, hi3[3] = { -3.0e38f, -3.0e38f, -3.0e38f };
        for (int k = 0; k < kNodeWidth; k++)
            if (k < count)
                for (int a = 0; a < 3; a++)
                {
                    const float m = c[k].lo[a] + c[k].hi[a];
                    lo3[a] = fminf(lo3[a], m);
                    hi3[a] = fmaxf(hi3[a], m);
                }
        const float sx = hi3[0] - lo3[0], sy = hi3[1] - lo3[1], sz = hi3[2] - lo3[2];
        orderAxis = sx >= sy && sx >= sz ? 0u : (sy >= sz ? 1u : 2u);
        float key[kNodeWidth];
        for (int k = 0; k < kNodeWidth; k++)
            key[k] = k < count ? (orderAxis == 0 ? c[k].lo[0] + c[k].hi[0] : orderAxis == 1 ? c[k].lo[1] + c[k].hi[1] : c[k].lo[2] + c[k].hi[2]) : -3.0e38f;
#define PT_ORDER_SWAP(a, b)                                                                                                \
        if (key[a] < key[b])                                                                                               \
        {                                                                                                                  \
            const float tk = key[a]; key[a] = key[b]; key[b] = tk;                                                         \
            const ChildBox tc = c[a]; c[a] = c[b]; c[b] = tc;                                                              \
        }
        PT_ORDER_SWAP(0, 1) PT_ORDER_SWAP(2, 3) PT_ORDER_SWAP(0, 2) PT_ORDER_SWAP(1, 3) PT_ORDER_SWAP(1, 2)
#undef PT_ORDER_SWAP
    }
#endif
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
    nd.a = make_float4(o[0], o[1], o[2], __uint_as_float(ebits[0] | (ebits[1] << 8) | (ebits[2] << 16) | (orderAxis << 24)));
    // a leaf of a non-opaque geometry says so in its ref: the traversal fetches its any-hit record beside the triangle
    for (int k = 0; k < count; k++)
        if (c[k].ref < 0 && (__float_as_uint(triTmp[refTri ? refTri[vals[~c[k].ref]] : vals[~c[k].ref]].c.w) & kTriNonOpaque))
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

// ---- subtree-contiguous (depth-first) order of the relaid-out nodes (round 4) ---------------------------------------
// The breadth-first array above keeps the top of the tree dense, but below it every level is an array of its own: a walk of
// eleven levels touches eleven regions of memory, and the deep levels of a 4 M-triangle tree are tens of megabytes each -- a
// 2 MB page or an L2 line brought in for one ray serves few others (atrium_like, 460 MB of tree against 256 MB of Infinity
// Cache: L2 hit rate 44 %, 2.4 M per-CU TLB misses per closest-hit launch, profiles/r03_mem_counters_atrium_like.txt).
// In depth-first pre-order every subtree is one contiguous range: the last six levels under a node (4,096 nodes, 256 KB) share
// a page, and rays that work in one part of the scene work in one part of the array.  Three passes over the levels the
// breadth-first pass found (their ranges are known on the host, so nothing is read back):
//   k_subtree_size   deepest level first: size[i] = 1 + the sizes of i's internal children
//   k_subtree_pos    root first: pos[child] = pos[i] + 1 + the sizes of the children in the slots before it
//   k_place_nodes    node i goes to pos[i], its child refs to pos[child]
// (refs of a breadth-first node: indices into the breadth-first array.)
__global__ void k_subtree_size(uint32_t lo, uint32_t hi, const BvhNode *__restrict__ nodes, uint32_t *__restrict__ size)
{
    const uint32_t i = lo + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= hi)
        return;
    const int4 r4 = nodes[i].refs;
    const int refs[kNodeWidth] = { r4.x, r4.y, r4.z, r4.w };
    uint32_t s = 1;
    for (int k = 0; k < kNodeWidth; k++)
        if (refs[k] >= 0 && refs[k] != kEmptyRef)
            s += size[refs[k]];
    size[i] = s;
}
__global__ void k_subtree_pos(uint32_t lo, uint32_t hi, const BvhNode *__restrict__ nodes, const uint32_t *__restrict__ size, uint32_t *__restrict__ pos)
{
    const uint32_t i = lo + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= hi)
        return;
    const int4 r4 = nodes[i].refs;
    const int refs[kNodeWidth] = { r4.x, r4.y, r4.z, r4.w };
    uint32_t next = (i == 0 ? 0u : pos[i]) + 1u;
    if (i == 0)
        pos[0] = 0u;
    for (int k = 0; k < kNodeWidth; k++)
        if (refs[k] >= 0 && refs[k] != kEmptyRef)
        {
            pos[refs[k]] = next;
            next += size[refs[k]];
        }
}
__global__ void k_place_nodes(uint32_t count, const BvhNode *__restrict__ nodes, const uint32_t *__restrict__ pos, BvhNode *__restrict__ out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count)
        return;
    BvhNode nd = nodes[i];
    int refs[kNodeWidth] = { nd.refs.x, nd.refs.y, nd.refs.z, nd.refs.w };
    for (int k = 0; k < kNodeWidth; k++)
        if (refs[k] >= 0 && refs[k] != kEmptyRef)
            refs[k] = (int)pos[refs[k]];
    nd.refs = make_int4(refs[0], refs[1], refs[2], refs[3]);
    out[pos[i]] = nd;
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

} // namespace ptd
