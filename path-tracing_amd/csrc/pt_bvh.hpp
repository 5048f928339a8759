// pt_bvh.hpp -- software BVH for gfx950: the node / triangle layout in HBM and the stack traversal used by
// the closest-hit and any-hit queries.  The builder kernels are in pt_bvh_build.hpp.
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
constexpr int kRefDone = 0x7fffffff; // traversal: the ray is finished
constexpr int kRefNone = 0x7ffffffd; // traversal: a node visit that hit no child (neither a node index nor a leaf nor kEmptyRef)

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
    // The same for a wave none of whose lanes leaves the LDS part in this step (the caller has asked: roomFor / allInLds, one
    // ballot and a scalar branch).  Instructions on a visit's dependent chain are what the traversal kernels pay for (round 4,
    // docs/EXPERIMENTS.md), and the two-level test of push() / pop() -- LDS, overflow region, overflow flag -- cost 18
    // instructions per push site, six sites per round: a tenth of the loop.
    PT_DEV void pushLds(uint32_t v)
    {
        lds[sp * stride] = v;
        sp++;
    }
    PT_DEV uint32_t popLds()
    {
        sp--;
        return lds[sp * stride];
    }
    PT_DEV bool roomFor(int entries) const { return __ballot(sp + entries > ldsDepth) == 0ull; } // of the lanes active here
    PT_DEV bool allInLds() const { return __ballot(sp > ldsDepth) == 0ull; }
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
//
// ORDERED = false (occlusion queries, raygen.rgen:31 gl_RayFlagsTerminateOnFirstHitEXT: any hit ends the ray, so the order in
// which the hit children are entered decides nothing): no sorting network; r0..r3 are the four refs in slot order and the
// return value is the MASK of the children hit (bit k = slot k); PT_ENTER_UNORDERED below walks them from the last slot down.
// Measured (round 4, launch alone, sorted -> unordered): k_trace_shadow -5 % on chess_like, -7 % on street_like, -17 % on
// atrium_like (whole frame +1 % / +1.5 % / +5.5 %).  Storing a node's children by surface area so that the largest (likeliest to
// hold an occluder) is entered first was measured too: street_like +1.5 %, chess_like flat, atrium_like -5 % (back to the
// sorted walk's time) -- not done.
template <bool ORDERED>
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
    if (!ORDERED)
    {
        r0 = refs.x; r1 = refs.y; r2 = refs.z; r3 = refs.w;
        return (h0 ? 1 : 0) | (h1 ? 2 : 0) | (h2 ? 4 : 0) | (h3 ? 8 : 0);
    }
#ifdef PTX_EXP_AXIS_ORDER
    {
        // experiment: k_emit stored the children in descending centroid order along the node's axis (bits 24..25 of the exponent
        // word); PT_ENTER_UNORDERED enters the highest hit slot first, which is the near end for a ray that travels up the axis --
        // one that travels down it sees the slots reversed
        const uint32_t axis = (eb >> 24) & 3u;
        const bool down = axis == 0u ? ngx : (axis == 1u ? ngy : ngz);
        r0 = down ? refs.w : refs.x; r1 = down ? refs.z : refs.y; r2 = down ? refs.y : refs.z; r3 = down ? refs.x : refs.w;
        const int fwd = (h0 ? 1 : 0) | (h1 ? 2 : 0) | (h2 ? 4 : 0) | (h3 ? 8 : 0), bwd = (h3 ? 1 : 0) | (h2 ? 2 : 0) | (h1 ? 4 : 0) | (h0 ? 8 : 0);
        return down ? bwd : fwd;
    }
#endif
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

// The walk of an occlusion query from a node: enter the hit child in the highest slot, keep the others on the stack.
#define PT_ENTER_UNORDERED(mask, r0, r1, r2, r3, st, ref, none, PUSH)                                                            \
    {                                                                                                                      \
        int next_ = (none);                                                                                                \
        bool have_ = false;                                                                                                \
        if ((mask) & 1) { next_ = r0; have_ = true; }                                                                      \
        if ((mask) & 2) { if (have_) st.PUSH((uint32_t)next_); next_ = r1; have_ = true; }                                 \
        if ((mask) & 4) { if (have_) st.PUSH((uint32_t)next_); next_ = r2; have_ = true; }                                 \
        if ((mask) & 8) { if (have_) st.PUSH((uint32_t)next_); next_ = r3; have_ = true; }                                 \
        ref = next_;                                                                                                       \
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
            const int h = visitNode<!ANY_HIT>(&sc.nodes[ref], o, id, tmin, ANY_HIT ? best.t : best.t * kCullSlack, r0, r1, r2, r3);
#ifdef PTX_EXP_AXIS_ORDER
            constexpr bool kMaskWalk = true; // (experiment: the ordered visit returns a mask in the node's build-time order)
#else
            constexpr bool kMaskWalk = ANY_HIT;
#endif
            if (kMaskWalk)
            {
                PT_ENTER_UNORDERED(h, r0, r1, r2, r3, st, ref, kRefNone, push)
            }
            else
            {
                if (h > 3) st.push((uint32_t)r3);
                if (h > 2) st.push((uint32_t)r2);
                if (h > 1) st.push((uint32_t)r1);
                ref = h > 0 ? r0 : kRefNone;
            }
            if (ref == kRefNone)
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
constexpr int kNodeStepsPerRound = 2;   // measured: 1 -> 1040, 2 -> 1064, 3 -> 1055, 4 -> 1034, 6 -> 975, 10 -> 872 Msamples/s; round 4, for the
                                        // (now unordered) occlusion walk alone, 1 / 2 / 3: chess_like 2,529 / 2,563 / 2,543, street_like 1,257 / 1,305 / 1,310,
                                        // atrium_like 892 / 896 / 892; refill threshold 8 / 16 / 24 again: 2,510 / 2,563 / 2,554, 1,275 / 1,305 / 1,291


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
#if defined(PTX_EXP_UNORDERED_CLOSEST)
                constexpr bool kUnordered = true, kVisitOrdered = false; // experiment: closest-hit rays enter the hit children in slot order too
#elif defined(PTX_EXP_AXIS_ORDER)
                constexpr bool kUnordered = true, kVisitOrdered = !ANY_HIT; // experiment: ... in the build-time order along the node's axis (a mask, not a count)
#else
                constexpr bool kUnordered = ANY_HIT, kVisitOrdered = !ANY_HIT;
#endif
                const int h = visitNode<kVisitOrdered>(&sc.nodes[ref], o, id, PT_TMIN, ANY_HIT ? PT_TMAX : best.t * kCullSlack, r0, r1, r2, r3);
                if (__builtin_expect(st.roomFor(kNodeWidth - 1), 1)) // every lane of this step stays in the LDS part of its stack
                {
                    if (kUnordered)
                    {
                        PT_ENTER_UNORDERED(h, r0, r1, r2, r3, st, ref, kRefNone, pushLds)
                    }
                    else
                    {
                        if (h > 3) st.pushLds((uint32_t)r3);
                        if (h > 2) st.pushLds((uint32_t)r2);
                        if (h > 1) st.pushLds((uint32_t)r1);
                        ref = h > 0 ? r0 : kRefNone;
                    }
                    if (ref == kRefNone)
                        ref = st.sp ? (int)st.popLds() : kRefDone;
                }
                else
                {
                    if (kUnordered)
                    {
                        PT_ENTER_UNORDERED(h, r0, r1, r2, r3, st, ref, kRefNone, push)
                    }
                    else
                    {
                        if (h > 3) st.push((uint32_t)r3);
                        if (h > 2) st.push((uint32_t)r2);
                        if (h > 1) st.push((uint32_t)r1);
                        ref = h > 0 ? r0 : kRefNone;
                    }
                    if (ref == kRefNone)
                        ref = st.sp ? (int)st.pop() : kRefDone;
                }
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
            if (__builtin_expect(st.allInLds(), 1))
                ref = st.sp ? (int)st.popLds() : kRefDone;
            else
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
