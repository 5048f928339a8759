// pt_runtime.hpp -- host side of the HIP library: the renderer object behind a PtxRenderer handle, its device buffers, scene
// upload, the tree build (kernels: pt_bvh_build.hpp), the bounce schedule of the wavefront backend, read-back, the output
// stage.  Functions here take a valid handle; include/ptx.h's entry points (ptx_capi.hip) are thin wrappers around them.
// No CPU fallback exists: without a HIP device createRenderer fails.
#pragma once

#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "pt_aux_kernels.hpp"
#include "pt_bvh_build.hpp"
#include "pt_post.hpp"

// =====================================================================================
// Host side: the renderer object behind the C-ABI
// =====================================================================================

// Owning device allocation: freed when it goes out of scope, so error returns (HIP_TRY / BUILD_TRY) and ptx_destroy
// release everything without a list of names to keep in step.
template <typename T> struct DevBuf
{
    T *p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    hipError_t alloc(size_t count)
    {
        if (count <= n && p)
            return hipSuccess;
        release();
        const hipError_t e = hipMalloc(reinterpret_cast<void **>(&p), (count ? count : 1) * sizeof(T));
        if (e == hipSuccess)
            n = count;
        else
            p = nullptr;
        return e;
    }
    void release()
    {
        if (p)
            (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    void swap(DevBuf &o)
    {
        std::swap(p, o.p);
        std::swap(n, o.n);
    }
};

// How ptx_build_accel builds the tree (the software stand-in for VkBuildAccelerationStructureFlags: the reference asks its driver
// for ePreferFastTrace, AccelerationStructure.cpp:319-324).  buildBestTree tries a few settings and keeps the cheapest tree.
struct TreeParams
{
    uint32_t plocRadius = kPlocRadius; // PLOC: clusters look for their merge partner this many positions to either side
    float plocShape = kPlocShape;      // PLOC: weight of the compactness term in the merge metric
    bool mortonCubic = false;          // Morton curve with cubic cells (one scale for the three axes)
    uint32_t collapse = 1;             // 4-wide collapse: 0 greedy by surface area (round 1), 1 cost-driven (k_collapse_cost)
    uint32_t reinsertPasses = 2;       // passes of parallel reinsertion over the binary tree before the collapse (k_reinsert_*): every
                                       // candidate tree gets these; a build that keeps its state for refits (animation) at most these
    uint32_t reinsertFinal = 32;       // ... and the candidate that wins is built once more with this many (15 ms per pass at 2 M triangles)
    float splitBudget = 0.0f;          // pre-splitting: extra leaf references as a fraction of the triangle count (0: none)
    uint32_t layout = 0;               // node order: 0 breadth-first levels, 1 depth-first (every subtree one contiguous range; measured flat)
};

// The environment switches of the library (INTEGRATION.md lists them): experiment and test knobs, read ONCE per handle when it
// is created -- no entry point calls getenv afterwards, and a handle keeps the values it was created with.
struct EnvSwitches
{
    // Who runs where (round 6; docs/EXPERIMENTS.md): the handle's streams are created with a CU mask (hipExtStreamCreateWithCUMask)
    //   0 none (any CU)   1 one XCD per handle (handle h -> XCD h % 8)   2 two groups of four XCDs (h % 2)
    //   3 main stream (closest + shade) on XCDs 0-5, auxiliary stream (shadow, tail) on XCDs 6-7   4 two XCDs per handle (h % 4)
    //   5 main stream on XCD h % 8, auxiliary stream on XCD (h + 4) % 8
    // With a mask the library creates the main stream itself even if the host passed one (the host's stream cannot be masked).
    uint32_t cuPartition = 0;
    bool singleStream = false; // PTX_SINGLE_STREAM=1: every handle as if created with PTX_DEVICE_SINGLE_STREAM
    bool verbose = false;        // PTX_VERBOSE: progress and statistics on stderr
    bool karrasBuilder = false;  // PTX_BUILDER=lbvh: Karras topology instead of PLOC
    bool plocFixed = false;      // PTX_PLOC_RADIUS / PTX_PLOC_SHAPE given: ONE tree with these parameters, no candidates
    uint32_t plocRadius = 0;     // PTX_PLOC_RADIUS
    float plocShape = 0.0f;      // PTX_PLOC_SHAPE
    int reinsertPasses = -1;     // PTX_REINSERT=N: N reinsertion passes for the winning tree, min(N, 2) for every candidate (-1: not given)
    float splitBudget = -1.0f;   // PTX_SPLIT_BUDGET: extra leaf references as a fraction of the triangle count (< 0: not given)
    int layout = -1;             // PTX_NODE_LAYOUT=0 / 1: breadth-first / depth-first node order (-1: not given)
    int collapse = -1;           // PTX_COLLAPSE=0 / 1: greedy / cost-driven 4-wide collapse (-1: not given; does not fix the other parameters)
    int shadeSort = -1;          // PTX_SHADE_SORT=0 / 1 overrides the scene's choice (-1: not given)
    long tailThreshold = -1;     // PTX_TAIL_THRESHOLD: live paths at or below which k_tail takes over (-1: the default)
    uint32_t framesPerWave = 8;  // PTX_FRAMES_PER_WAVE: samples of one pixel in neighbouring lanes, at most this many
    uint32_t raysPerThread = 0;  // PTX_RAYS_PER_THREAD (process-wide: the last handle created sets it)
    long residentCap = -1;       // PTX_RESIDENT_CAP=0: persistent grids are not capped at the resident block count
    uint32_t copyGroups = 0;     // PTX_COPY_GROUPS: workgroups of the read-back copy kernel (0: one, or 2 per rank of a tile shard, at most 16)
    bool snapshotMemcpy = false; // PTX_SNAPSHOT_MEMCPY=1: the read-back's device-side snapshot by hipMemcpyAsync (rounds 1-4) instead of a kernel
    bool fenceRefit = false;     // PTX_FENCE_REFIT=1: the round-4 bottom-up kernels (fence and atomic per node) instead of the level lists
    static EnvSwitches read()
    {
        EnvSwitches e;
        e.verbose = getenv("PTX_VERBOSE") != nullptr;
        if (const char *v = getenv("PTX_COPY_GROUPS"))
            e.copyGroups = (uint32_t)strtoul(v, nullptr, 10);
        e.snapshotMemcpy = getenv("PTX_SNAPSHOT_MEMCPY") != nullptr && std::strcmp(getenv("PTX_SNAPSHOT_MEMCPY"), "0") != 0;
        e.fenceRefit = getenv("PTX_FENCE_REFIT") != nullptr && std::strcmp(getenv("PTX_FENCE_REFIT"), "0") != 0;
        if (const char *v = getenv("PTX_BUILDER"))
            e.karrasBuilder = std::strcmp(v, "lbvh") == 0;
        if (const char *v = getenv("PTX_PLOC_SHAPE"))
        {
            e.plocFixed = true;
            e.plocShape = (float)atof(v);
        }
        if (const char *v = getenv("PTX_PLOC_RADIUS"))
        {
            e.plocFixed = true;
            e.plocRadius = std::max(1u, (uint32_t)strtoul(v, nullptr, 10));
        }
        if (const char *v = getenv("PTX_REINSERT"))
            e.reinsertPasses = atoi(v);
        if (const char *v = getenv("PTX_SPLIT_BUDGET"))
            e.splitBudget = (float)atof(v);
        if (const char *v = getenv("PTX_NODE_LAYOUT"))
            e.layout = atoi(v) ? 1 : 0;
        if (const char *v = getenv("PTX_COLLAPSE"))
            e.collapse = atoi(v) ? 1 : 0;
        if (const char *v = getenv("PTX_SHADE_SORT"))
            e.shadeSort = atoi(v) ? 1 : 0;
        if (const char *v = getenv("PTX_TAIL_THRESHOLD"))
            e.tailThreshold = (long)strtoul(v, nullptr, 10);
        if (const char *v = getenv("PTX_FRAMES_PER_WAVE"))
            e.framesPerWave = (uint32_t)atoi(v);
        if (const char *v = getenv("PTX_RAYS_PER_THREAD"))
            e.raysPerThread = std::max(1u, (uint32_t)strtoul(v, nullptr, 10));
        if (const char *v = getenv("PTX_RESIDENT_CAP"))
            e.residentCap = (long)strtoul(v, nullptr, 10);
        e.singleStream = getenv("PTX_SINGLE_STREAM") != nullptr && std::strcmp(getenv("PTX_SINGLE_STREAM"), "0") != 0;
        if (const char *v = getenv("PTX_CU_PARTITION"))
            e.cuPartition = (uint32_t)strtoul(v, nullptr, 10);
        return e;
    }
};

struct PtxRenderer
{
    EnvSwitches env;
    int device = 0;
    uint32_t backend = PTX_BACKEND_WAVEFRONT;
    hipStream_t stream = nullptr;
    bool ownStream = false;
    bool counted = false; // in g_liveHandles
    std::string error;

    // scene (HBM copies of the Scene getters)
    DevBuf<PtxVertex> vertices;
    DevBuf<uint32_t> indices;
    DevBuf<PtxMetallicRoughnessMaterial> mr;
    DevBuf<PtxSpecularGlossinessMaterial> sg;
    DevBuf<PtxPhongMaterial> phong;
    DevBuf<DevPair> pairs;
    DevBuf<uint32_t> pairFirst;
    DevBuf<PtxLightsUbo> lights;
    DevBuf<DevTexture> textures;
    DevBuf<uint32_t> texels8; // upload time: the pools of the image formats, in which mip chains are built; released after
    DevBuf<float4> texelsF;
    DevBuf<float> srgbLut;
    DevBuf<DevTexture> renderTextures; // what the render kernels sample: every texel decoded to four floats, one pool
    DevBuf<float4> renderTexels;
    // what the any-hit stages read (scenes with non-opaque geometry; pt_bvh.hpp, hitAlpha)
    DevBuf<AlphaTex> alphaTex;     // per colour texture
    DevBuf<uint32_t> alphaTexOf;   // scene texture -> entry of alphaTex
    DevBuf<float4> alphaQuads;     // 2 x 2 alpha footprints of their base levels
    DevBuf<AlphaTri> alphaTris;    // per triangle slot, written behind k_emit
    uint32_t textureCount = 0;
    uint32_t skyKind = PTX_SKYBOX_CLEAR_COLOR; // its images follow the scene textures in `textures`
    bool samplerNeeded = false; // some uploaded texture is not a 1x1 white placeholder
    // animation (row N3)
    std::vector<DevPair> hostPairs;            // to recompose pair transforms when instances move
    std::vector<uint32_t> pairInstance;        // pair -> instance
    std::vector<PtxTransform> pairMeshTransform; // pair -> baked mesh transform
    uint32_t instanceCount = 0, skinnedCount = 0, boneCount = 0;
    uint64_t staticVertexCount = 0;
    DevBuf<PtxAnimatedVertex> animatedVertices;
    DevBuf<uint32_t> skinSource;
    DevBuf<PtxTransform> bones;
    struct BuildState // what a refit reuses from the last full build: sorted order and the binary topology
    {
        DevBuf<Tri> triTmp;
        DevBuf<float4> boxLo, boxHi, nodeLo, nodeHi;
        DevBuf<uint32_t> sceneBounds, vals0, vals1, hist, histSums, flags;
        DevBuf<uint8_t> inert; // per flattened triangle: left out of the tree (zero area) by the last full build
        // leaf references of a build that splits triangles (k_split_*): boxes, triangle and zero-area flag per reference
        DevBuf<float4> refLo, refHi;
        DevBuf<uint32_t> refTri;
        DevBuf<uint8_t> refInert;
        uint32_t refCount = 0; // references of the last full build (= triangles unless it split some)
        uint32_t treeTris = 0; // triangles in the tree = the first treeTris entries of the sorted order
        DevBuf<uint64_t> keys0, keys1;
        DevBuf<int2> children;
        DevBuf<int> parentOfNode, parentOfLeaf;
        DevBuf<BvhNode> rawNodes; // k_emit's output, one slot per binary node; k_relayout_level compacts it into `nodes`
        DevBuf<float4> collapseCost; // k_collapse_cost: T(x, 1..4) per binary node
        DevBuf<uint8_t> collapseDecide;
        DevBuf<uint32_t> oldOf;   // [0] onwards: emitted index of every node of the compact array; the last entry is the level counter
        // level lists of the binary topology (pt_bvh_build.hpp, round 5): the nodes sorted by depth, for the bottom-up passes
        DevBuf<uint32_t> lvDepth0, lvDepth1, lvVals0, lvVals1, lvStartDev;
        DevBuf<int> lvAnc0, lvAnc1;
        const uint32_t *levelOrder = nullptr; // lvVals0 or lvVals1
        std::vector<uint32_t> levelStart;     // [d] first position of depth d in levelOrder, [maxDepth + 1] = nodes
        bool levelsValid = false;
        bool valid = false;
        void release()
        {
            triTmp.release(); boxLo.release(); boxHi.release(); nodeLo.release(); nodeHi.release(); sceneBounds.release();
            vals0.release(); vals1.release(); hist.release(); histSums.release(); flags.release(); inert.release(); keys0.release(); keys1.release();
            children.release(); parentOfNode.release(); parentOfLeaf.release(); rawNodes.release(); oldOf.release(); collapseCost.release(); collapseDecide.release();
            refLo.release(); refHi.release(); refTri.release(); refInert.release();
            lvDepth0.release(); lvDepth1.release(); lvVals0.release(); lvVals1.release(); lvStartDev.release(); lvAnc0.release(); lvAnc1.release();
            levelOrder = nullptr; levelStart.clear(); levelsValid = false;
            valid = false;
        }
    } build;
    bool anyNonOpaque = false;  // some instanced geometry lacks the opaque flag: any-hit stages run
    bool mixedMaterialTypes = false; // the instanced meshes use more than one material type (ShaderTypes.incl:143-145): k_shade sorts its queue
    bool mixedTextured = false;      // ... or materials with and without scene textures: the sampler runs for waves of textured hits only
    bool usePloc = true;        // PLOC topology instead of Karras (PTX_BUILDER=lbvh switches back)
    TreeParams tree;            // of the tree in use; the per-frame rebuilds of an animation build with them again
    bool reinsertBroken = false; // a reinsertion pass once left something that was not a tree (k_tree_check): off for this handle
    uint32_t residentClosest[2] = { 0, 0 }, residentShadow[2] = { 0, 0 }; // blocks the chip holds at once, per [ALPHA] variant
    DevBuf<float4> decal;
    DevBuf<float> decalT;
    size_t decalCapacity = 0;
    uint32_t pairCount = 0, triCount = 0, dxNormalTextures = 0;
    uint32_t treeTris = 0; // triCount minus the zero-area triangles, which are not in the tree
    bool sceneReady = false, accelReady = false;
    // ptx_share_scene: this renderer renders the scene and tree of `sceneOwner` instead of holding copies (frames in flight
    // share one scene, as the reference's per-frame resources do); the owner knows who borrows from it
    PtxRenderer *sceneOwner = nullptr;
    std::vector<PtxRenderer *> sceneSharers;

    // accel
    DevBuf<BvhNode> nodes;
    DevBuf<Tri> tris;
    DevBuf<ShadeTri> shadeTris; // deindexed vertices per triangle slot (leaf order), written by k_emit

    // frame
    uint32_t width = 0, height = 0;
    PtxTileShard shard = { 0, 1, 32 };
    DevBuf<float4> image;
    // output stage (row N4)
    DevBuf<float> postRgb, bloomRgb; // rgba16f-valued post-process image and bloom mip chain (3 floats per texel)
    DevBuf<float4> outLinear;        // tone-mapped image (OutputSaver's m_LinearImage)
    DevBuf<uint32_t> outSrgb8;
    bool outputReady = false;
    float4 *boundImage = nullptr; // external accumulation buffer, if bound
    float4 *boundShard = nullptr; // ... or the dense tile-major shard buffer the samples are accumulated in (ptx_bind_shard_accumulation)
    size_t boundShardBytes = 0;
    // device alias of the page-locked host frame last used by a read-back / unpack (hipHostGetDevicePointer once per buffer)
    const void *hostAliasOf = nullptr;
    size_t hostAliasBytes = 0;
    float4 *hostAlias = nullptr;
    // pipelined read-back (ptx_readback_begin / _end): snapshot of the image, copied out on its own stream
    DevBuf<float4> staging;
    hipStream_t copyStream = nullptr;
    hipEvent_t evSnapshot = nullptr, evCopied = nullptr;
    bool copyInFlight = false;

    // wavefront state
    size_t slotCapacity = 0;
    DevBuf<float4> rayO, rayD, thr, rad, hit, shO, shD, shC, slotRad;
    DevBuf<float4> diffs; // 3 x slotCapacity ray differentials, only for scenes with textures
    size_t diffCapacity = 0;
    DevBuf<uint4> meta;
    DevBuf<uint32_t> hitPair, queue0, queue1, shadowQueue, restartQueue, counters, spill;
    DevBuf<uint8_t> shadowResult;
    uint32_t *hostCounters = nullptr; // pinned

    DevBuf<float> testIn, testOut;
    DevBuf<PtxLightsUbo> testUbo;

    hipEvent_t evA = nullptr, evB = nullptr, evT0 = nullptr, evT1 = nullptr; // render / build span; ptx_trace_rays kernel span

    // bounce schedule of the wavefront backend (renderImpl): closest + shade on `stream`, shadow + tail on `auxStream`
    hipStream_t auxStream = nullptr;
    bool auxIsMain = false; // single-stream handle: auxStream is `stream` itself
    bool singleStream = false; // PtxDeviceDesc.flags & PTX_DEVICE_SINGLE_STREAM
    uint32_t handleSeq = 0, mainXcds = 0xffu, auxXcds = 0xffu; // PTX_CU_PARTITION: the XCDs this handle's streams may use
    DevBuf<uint32_t> spillAux; // traversal-stack overflow region of the kernels on auxStream
    struct BounceEvents
    {
        hipEvent_t t0 = nullptr, t1 = nullptr, t2 = nullptr; // stream: before closest, after closest, after shade
        hipEvent_t x0 = nullptr, x1 = nullptr, x2 = nullptr; // auxStream: before shadow, after shadow, after tail
    };
    std::vector<BounceEvents> bounceEvents; // [min(BounceCount, kMaxTimedBounces)], reused cyclically beyond
    // what the last launch left for ptx_get_stats / the next launch to pick up once the device is done
    bool statsPending = false;
    uint32_t pendingBounces = 0, pendingTailBelow = 0, pendingSlots = 0;
    uint64_t pendingEpoch = 0;
    uint32_t pendingDeadSlots = 0; // slots of ragged edge tiles outside the image: in the first queue, not rays
    bool pendingVerbose = false;
    std::vector<uint32_t> hintActive; // queue length per bounce of the last canonical launch: sizes the grids of the next one
    uint32_t hintSlots = 0, hintBounces = 0;
    uint64_t hintEpoch = 0;  // sceneEpoch of the scene the hint was learnt on
    uint64_t sceneEpoch = 1; // bumped by every upload and full build of THIS handle's scene (a refit keeps it: the poses of
                             // an animation differ little from frame to frame)
    PtxStats stats = {};
};

static int fail(PtxRenderer *r, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (r)
        r->error = buf;
    return code;
}

#define HIP_TRY(r, expr)                                                                                                   \
    do                                                                                                                     \
    {                                                                                                                      \
        const hipError_t e_ = (expr);                                                                                      \
        if (e_ != hipSuccess)                                                                                              \
            return fail(r, e_ == hipErrorOutOfMemory ? PTX_ERROR_OUT_OF_MEMORY : PTX_ERROR_DEVICE, "%s: %s", #expr,       \
                        hipGetErrorString(e_));                                                                            \
    } while (0)

// the renderer whose scene buffers, tree and scene flags `r` renders with
static const PtxRenderer *sceneOf(const PtxRenderer *r)
{
    return r->sceneOwner ? r->sceneOwner : r;
}

// May `r` trace and shade right now?  A borrower is only as ready as its owner: between the owner's ptx_scene_upload and
// its ptx_build_accel the owner's tree describes the OLD triangles while pairs, vertices and textures are the new ones.
static bool sceneUsable(const PtxRenderer *r)
{
    return r->accelReady && (!r->sceneOwner || (r->sceneOwner->sceneReady && r->sceneOwner->accelReady));
}

static void detachSharedScene(PtxRenderer *r)
{
    if (!r->sceneOwner)
        return;
    std::vector<PtxRenderer *> &v = r->sceneOwner->sceneSharers;
    for (size_t i = 0; i < v.size(); i++)
        if (v[i] == r)
        {
            v.erase(v.begin() + (long)i);
            break;
        }
    r->sceneOwner = nullptr;
    r->accelReady = false; // ptx_share_scene released the borrower's own scene: it needs ptx_scene_upload + ptx_build_accel again
}

// Before the owner of a shared scene changes it (upload, rebuild, animation step): the borrowers' frames in flight end.
static void quiesceSharers(PtxRenderer *r)
{
    for (PtxRenderer *sh : r->sceneSharers)
    {
        if (sh->stream)
            (void)hipStreamSynchronize(sh->stream);
        if (sh->auxStream)
            (void)hipStreamSynchronize(sh->auxStream);
    }
}

static float4 *imagePtr(PtxRenderer *r)
{
    return r->boundImage ? r->boundImage : r->image.p;
}
// where k_accumulate adds the samples: the row-major frame, or this rank's dense tile-major shard (ptx_bind_shard_accumulation)
static float4 *accumTarget(PtxRenderer *r)
{
    return r->boundShard ? r->boundShard : imagePtr(r);
}
static int frameIsElsewhere(PtxRenderer *r, const char *who)
{
    return fail(r, PTX_ERROR_NOT_READY, "%s: the accumulation of this renderer lives in a shard buffer (ptx_bind_shard_accumulation); the frame "
                                         "is composed by ptx_unpack_shards on the rank that gathers it", who);
}

static uint32_t gridFor(size_t n, uint32_t block = kBlock, uint32_t cap = 256 * 8)
{
    size_t g = (n + block - 1) / block;
    if (g < 1)
        g = 1;
    if (g > cap)
        g = cap;
    return static_cast<uint32_t>(g);
}

// Grid of a persistent traversal kernel.  A launch never finishes before its longest ray (hundreds of dependent node
// fetches), which is many times an average ray: giving every thread several rays of a SMALL launch costs nothing, and
// leaves compute units free for the kernel running beside it on the other stream.
// Rays per thread: PTX_RAYS_PER_THREAD fixes it; otherwise 4 while the process holds fewer than four renderers -- a frame alone
// on the machine wants every launch spread over all of it (one frame in flight: 9.6 ms per frame with 4, 10.6 with 8 / 12) -- and,
// with four or more (frames in flight share the machine), 8, or 12 for a launch below 3 M rays: a SMALL launch then does better
// with fewer, longer-lived waves -- a wave that works through eight chunks drains its stragglers once, not once per two chunks,
// and holds a quarter of the wave slots meanwhile.  Measured with eight frames in flight on 16 hardware queues, 4 / 6 / 8 / 12 rays
// per thread: a rank's tile shard of 8 (2.07 M slots) 1.158 / 1.100 / 1.085 / 1.074 ms per step, of 4: 1.912 / 1.903 / 1.886 /
// 1.846, of 2: 3.442 / 3.431 / 3.401 / 3.355; the whole frame, read-back included, three interleaved runs of 4 / this rule / 8:
// chess_like 2,439 / 2,437 / 2,444 Msamples/s, atrium_like 816 / 812 / 820; 16 and 32 lose on both.
static uint32_t g_raysPerThread = 0;
static std::atomic<uint32_t> g_liveHandles{0};
static uint32_t raysPerThreadFor(size_t n)
{
    if (g_raysPerThread)
        return g_raysPerThread;
    return g_liveHandles.load(std::memory_order_relaxed) < 4u ? 4u : (n < 3000000u ? 12u : 8u);
}
// A persistent kernel must not launch more blocks than the chip holds at once: with the static chunk schedule the chunks
// of a block that is not resident yet wait until a resident block has drained the whole queue, and then run on a mostly
// empty chip.  residentBlocks = occupancy (blocks per CU, from the kernel's VGPR / LDS use) x compute units.
static uint32_t g_residentCap = 1; // PTX_RESIDENT_CAP=0: the old fixed cap of 2048 blocks
static uint32_t traceGridFor(size_t n, uint32_t residentBlocks = 0)
{
    const uint32_t per = raysPerThreadFor(n);
    const uint32_t g = gridFor((n + per - 1) / per);
    return g_residentCap && residentBlocks && g > residentBlocks ? residentBlocks : g;
}

template <typename K>
static uint32_t residentBlocksOf(K kernel, int device)
{
    int perCu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, kernel, kBlock, 0) != hipSuccess || perCu < 1)
        return 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus < 1)
        return 0;
    return (uint32_t)perCu * (uint32_t)cus;
}

static LaunchParams makeParams(const PtxRenderer *r, const PtxRaygenUniformData *u, uint32_t firstFrame, uint32_t frames)
{
    LaunchParams p;
    std::memset(&p, 0, sizeof(p));
    if (u)
        p.u = *u;
    p.width = r->width;
    p.height = r->height;
    p.rank = r->shard.rank;
    p.worldSize = r->shard.worldSize;
    p.tileSize = r->shard.tileSize;
    p.tilesX = (r->width + p.tileSize - 1) / p.tileSize;
    const uint32_t tilesY = (r->height + p.tileSize - 1) / p.tileSize;
    p.numTiles = p.tilesX * tilesY;
    p.ownedTiles = p.numTiles > p.rank ? (p.numTiles - p.rank + p.worldSize - 1) / p.worldSize : 0;
    p.slotsPerFrame = p.ownedTiles * p.tileSize * p.tileSize;
    p.frames = frames;
    p.firstFrame = firstFrame;
    p.numSlots = p.slotsPerFrame * frames;
    const uint32_t maxPerWave = r->env.framesPerWave;
    p.framesPerWave = 1;
    while (p.framesPerWave < maxPerWave && p.framesPerWave < 8u && frames % (p.framesPerWave * 2u) == 0u)
        p.framesPerWave *= 2u;
    for (uint32_t t = p.rank; t < p.numTiles; t += p.worldSize)
    {
        const uint32_t x0 = (t % p.tilesX) * p.tileSize, y0 = (t / p.tilesX) * p.tileSize;
        const uint32_t w = r->width - x0 < p.tileSize ? r->width - x0 : p.tileSize;
        const uint32_t h = r->height - y0 < p.tileSize ? r->height - y0 : p.tileSize;
        p.ownedPixels += w * h;
    }
    return p;
}

// Every frame in flight owns two HIP streams (main + auxiliary), and the runtime multiplexes all streams of a process
// onto GPU_MAX_HW_QUEUES hardware queues -- 4 unless the environment says otherwise.  Streams that share a queue run
// one after the other, which is exactly what frames in flight are meant to avoid: measured on chess_like with 2 / 4 /
// 8 / 16 queues, whole frame (3 in flight) 10.7 / 8.50 / 8.17 / 8.16 ms per step, one rank's shard of 8 (12 in flight)
// - / 1.73 / 1.68 / 1.54.  The variable is read when the runtime initialises (the first HIP call of the process) and is
// the HOST's to set (the Python package and bench.py set it at import): a library that edits its host's
// environment at load time races with getenv in the host's other threads and cannot know whether HIP is up already.  What
// the library does instead is REPORT: PtxStats::hardwareQueues says how many queues the environment grants, and the
// handle whose streams no longer fit says so once (ptx_last_error after a successful ptx_create, stderr under PTX_VERBOSE).
static uint32_t hardwareQueuesGranted()
{
    const char *e = getenv("GPU_MAX_HW_QUEUES");
    const unsigned long v = e ? strtoul(e, nullptr, 10) : 0ul;
    return v ? (uint32_t)v : 4u; // the runtime's default
}
static std::atomic<bool> g_queueWarningGiven{false};

static void destroyRenderer(PtxRenderer *r);

// CU masks of PTX_CU_PARTITION.  Bit i of the mask is compute unit i in the driver's numbering, which deals the CUs of a
// multi-XCD device round-robin over the XCDs (bit i -> XCD i % 8, the amdgpu driver's mqd_symmetrically_map_cu_mask walks the mask
// with a stride of the XCD count): XCD x = bits x, x + 8, x + 16 ...
static std::atomic<uint32_t> g_handleSeq{0};
static void xcdMask(uint32_t xcdBits, uint32_t cus, std::vector<uint32_t> &mask)
{
    mask.assign((cus + 31) / 32, 0u);
    for (uint32_t i = 0; i < cus; i++)
        if (xcdBits & (1u << (i % 8u)))
            mask[i / 32] |= 1u << (i % 32);
}
// the XCDs of the handle's main / auxiliary stream under partition `mode`; 0xff = no mask
static void partitionOf(uint32_t mode, uint32_t h, uint32_t &mainXcds, uint32_t &auxXcds)
{
    mainXcds = auxXcds = 0xffu;
    switch (mode)
    {
    case 1: mainXcds = auxXcds = 1u << (h % 8u); break;
    case 2: mainXcds = auxXcds = (h % 2u) ? 0xf0u : 0x0fu; break;
    case 3: mainXcds = 0x3fu; auxXcds = 0xc0u; break;
    case 4: mainXcds = auxXcds = 0x3u << (2u * (h % 4u)); break;
    case 5: mainXcds = 1u << (h % 8u); auxXcds = 1u << ((h + 4u) % 8u); break;
    case 6: mainXcds = auxXcds = 0x1ffu; break; // control: hipExtStreamCreateWithCUMask with EVERY CU enabled
    case 7: mainXcds = auxXcds = 0x2ffu; break; // control: the library's own plain streams instead of the host's
    default: break;
    }
}
static hipError_t createStreamOn(hipStream_t *s, uint32_t xcds, int device)
{
    int cus = 0;
    if (xcds == 0xffu || xcds == 0x2ffu || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus < 8)
        return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
    std::vector<uint32_t> mask;
    xcdMask(xcds & 0xffu, (uint32_t)cus, mask);
    return hipExtStreamCreateWithCUMask(s, (uint32_t)mask.size(), mask.data());
}

static int createRenderer(const PtxDeviceDesc *desc, PtxRenderer **out)
{
    if (!out)
        return PTX_ERROR_INVALID_ARGUMENT;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return PTX_ERROR_NO_DEVICE; // no CPU fallback: the product path needs a HIP device
    PtxRenderer *r = new PtxRenderer;
    r->device = desc ? desc->deviceIndex : 0;
    r->backend = desc ? desc->backend : PTX_BACKEND_WAVEFRONT;
    r->singleStream = desc && (desc->flags & PTX_DEVICE_SINGLE_STREAM) != 0u;
    if (r->device < 0 || r->device >= n || hipSetDevice(r->device) != hipSuccess)
    {
        delete r;
        return PTX_ERROR_NO_DEVICE;
    }
    r->env = EnvSwitches::read(); // the only place the library reads its switches
    r->handleSeq = g_handleSeq++;
    partitionOf(r->env.cuPartition, r->handleSeq, r->mainXcds, r->auxXcds);
    if (desc && desc->stream && r->mainXcds == 0xffu)
        r->stream = static_cast<hipStream_t>(desc->stream);
    else
    {
        if (createStreamOn(&r->stream, r->mainXcds, r->device) != hipSuccess)
        {
            delete r;
            return PTX_ERROR_DEVICE;
        }
        r->ownStream = true;
    }
    r->usePloc = !r->env.karrasBuilder;
    if (r->env.collapse >= 0)
        r->tree.collapse = (uint32_t)r->env.collapse;
    if (r->env.layout >= 0)
        r->tree.layout = (uint32_t)r->env.layout;
    if (r->env.splitBudget >= 0.0f)
        r->tree.splitBudget = r->env.splitBudget;
    if (r->env.reinsertPasses >= 0)
    {
        r->tree.reinsertFinal = (uint32_t)r->env.reinsertPasses;
        r->tree.reinsertPasses = std::min(r->tree.reinsertPasses, r->tree.reinsertFinal);
    }
    if (r->env.plocFixed)
    {
        r->tree.plocShape = r->env.plocShape;
        if (r->env.plocRadius)
            r->tree.plocRadius = r->env.plocRadius;
    }
    if (r->env.raysPerThread)
        g_raysPerThread = r->env.raysPerThread;
    if (r->env.residentCap >= 0)
        g_residentCap = (uint32_t)r->env.residentCap;
    // (a masked stream holds fewer blocks at once: the persistent grids are sized to the XCDs the stream may use)
    const uint32_t mainShare = (uint32_t)__builtin_popcount(r->mainXcds & 0xffu), auxShare = (uint32_t)__builtin_popcount(r->auxXcds & 0xffu);
    r->residentClosest[0] = residentBlocksOf(k_trace_closest<false>, r->device) * mainShare / 8u;
    r->residentClosest[1] = residentBlocksOf(k_trace_closest<true>, r->device) * mainShare / 8u;
    r->residentShadow[0] = residentBlocksOf(k_trace_shadow<false>, r->device) * auxShare / 8u;
    r->residentShadow[1] = residentBlocksOf(k_trace_shadow<true>, r->device) * auxShare / 8u;
    if (r->env.verbose)
        fprintf(stderr, "[ptx] resident blocks: closest %u / %u, shadow %u / %u\n", r->residentClosest[0], r->residentClosest[1], r->residentShadow[0],
                r->residentShadow[1]);
    (void)hipEventCreate(&r->evA);
    (void)hipEventCreate(&r->evB);
    (void)hipEventCreate(&r->evT0);
    (void)hipEventCreate(&r->evT1);
    (void)hipHostMalloc(reinterpret_cast<void **>(&r->hostCounters), C_COUNT * sizeof(uint32_t), hipHostMallocDefault);
    if (r->counters.alloc(C_COUNT) != hipSuccess || r->lights.alloc(1) != hipSuccess || !r->hostCounters ||
        r->spill.alloc((size_t)kGlobalSpill * kMaxPersistentThreads) != hipSuccess)
    {
        destroyRenderer(r);
        return PTX_ERROR_OUT_OF_MEMORY;
    }
    r->stats.hardwareQueues = hardwareQueuesGranted();
    r->counted = true;
    const uint32_t live = ++g_liveHandles;
    const uint32_t streamsPerHandle = (r->singleStream || r->env.singleStream) ? 1u : 2u;
    if (streamsPerHandle * live > r->stats.hardwareQueues && live > 1 && !g_queueWarningGiven.exchange(true))
    {
        // not an error: the handle works, its frames just run behind the other handles' instead of beside them
        fail(r, PTX_OK, "%u handles (%u stream(s) each) share %u hardware queues: frames in flight will serialise; export GPU_MAX_HW_QUEUES=16 "
                        "(24 for a process that also gathers) before the process first uses HIP", live, streamsPerHandle, (uint32_t)r->stats.hardwareQueues);
        if (r->env.verbose)
            fprintf(stderr, "[ptx] %s\n", r->error.c_str());
    }
    *out = r;
    return PTX_OK;
}

static void destroyRenderer(PtxRenderer *r)
{
    if (!r)
        return;
    (void)hipSetDevice(r->device);
    if (r->counted)
        --g_liveHandles;
    if (r->stream)
        (void)hipStreamSynchronize(r->stream);
    detachSharedScene(r);
    for (PtxRenderer *sh : r->sceneSharers) // borrowers of this scene: wait for their frames, then they have no scene
    {
        if (sh->stream)
            (void)hipStreamSynchronize(sh->stream);
        if (sh->auxStream)
            (void)hipStreamSynchronize(sh->auxStream);
        sh->sceneOwner = nullptr;
        sh->accelReady = false;
    }
    r->sceneSharers.clear();
    // every DevBuf member (scene, tree, build state, wavefront state, animation, output stage) frees itself in `delete r`
    if (r->auxStream && !r->auxIsMain) { (void)hipStreamSynchronize(r->auxStream); (void)hipStreamDestroy(r->auxStream); }
    for (PtxRenderer::BounceEvents &e : r->bounceEvents)
        for (hipEvent_t ev : { e.t0, e.t1, e.t2, e.x0, e.x1, e.x2 })
            if (ev)
                (void)hipEventDestroy(ev);
    if (r->copyStream) { (void)hipStreamSynchronize(r->copyStream); (void)hipStreamDestroy(r->copyStream); }
    if (r->evSnapshot) (void)hipEventDestroy(r->evSnapshot);
    if (r->evCopied) (void)hipEventDestroy(r->evCopied);
    if (r->hostCounters)
        (void)hipHostFree(r->hostCounters);
    if (r->evA) (void)hipEventDestroy(r->evA);
    if (r->evB) (void)hipEventDestroy(r->evB);
    if (r->evT0) (void)hipEventDestroy(r->evT0);
    if (r->evT1) (void)hipEventDestroy(r->evT1);
    if (r->ownStream && r->stream)
        (void)hipStreamDestroy(r->stream);
    delete r;
}


// Does the material's branch of material.glsl:62-142 fetch a scene texture (an index at or past PTX_SCENE_TEXTURE_OFFSET inside
// the uploaded table) through any of its five slots?  The five indices sit at the same offsets in the three 96-byte structs.
static bool materialSamplesSceneTexture(const PtxSceneDesc *s, uint32_t materialId)
{
    const uint32_t type = materialId & 0xffu, index = materialId >> 8;
    const uint32_t *idx = nullptr;
    if (type == PTX_MATERIAL_TYPE_METALLIC_ROUGHNESS && index < s->metallicRoughnessMaterialCount)
        idx = &s->metallicRoughnessMaterials[index].EmissiveIdx;
    else if (type == PTX_MATERIAL_TYPE_SPECULAR_GLOSSINESS && index < s->specularGlossinessMaterialCount)
        idx = &s->specularGlossinessMaterials[index].EmissiveIdx;
    else if (type == PTX_MATERIAL_TYPE_PHONG && index < s->phongMaterialCount)
        idx = &s->phongMaterials[index].EmissiveIdx;
    if (!idx)
        return false;
    for (int k = 0; k < 5; k++)
        if (idx[k] >= PTX_SCENE_TEXTURE_OFFSET && idx[k] - PTX_SCENE_TEXTURE_OFFSET < s->textureCount)
            return true;
    return false;
}

// world = A_instance * A_mesh * x (sampling.glsl:7)
static void composeTransform(const float *Ai, const float *Am, float *M)
{
    for (int r = 0; r < 3; r++)
    {
        for (int c = 0; c < 3; c++)
            M[r * 4 + c] = (Ai[r * 4 + 0] * Am[0 * 4 + c] + Ai[r * 4 + 1] * Am[1 * 4 + c]) + Ai[r * 4 + 2] * Am[2 * 4 + c];
        M[r * 4 + 3] = ((Ai[r * 4 + 0] * Am[0 * 4 + 3] + Ai[r * 4 + 1] * Am[1 * 4 + 3]) + Ai[r * 4 + 2] * Am[2 * 4 + 3]) + Ai[r * 4 + 3];
    }
}

// inverse of the 3x3 linear part by cofactors * (1/det), columns out
static void inverseLinear(const float *M, float *Rinv)
{
    const float m00 = M[0], m01 = M[4], m02 = M[8]; // column 0 of the math matrix
    const float m10 = M[1], m11 = M[5], m12 = M[9];
    const float m20 = M[2], m21 = M[6], m22 = M[10];
    const float det = (m00 * (m11 * m22 - m21 * m12) - m10 * (m01 * m22 - m21 * m02)) + m20 * (m01 * m12 - m11 * m02);
    const float id = 1.0f / det;
    Rinv[0] = (m11 * m22 - m21 * m12) * id;
    Rinv[3] = -(m10 * m22 - m20 * m12) * id;
    Rinv[6] = (m10 * m21 - m20 * m11) * id;
    Rinv[1] = -(m01 * m22 - m21 * m02) * id;
    Rinv[4] = (m00 * m22 - m20 * m02) * id;
    Rinv[7] = -(m00 * m21 - m20 * m01) * id;
    Rinv[2] = (m01 * m12 - m11 * m02) * id;
    Rinv[5] = -(m00 * m12 - m10 * m02) * id;
    Rinv[8] = (m00 * m11 - m10 * m01) * id;
}

template <typename T> static int upload(PtxRenderer *r, DevBuf<T> &buf, const T *src, size_t count)
{
    HIP_TRY(r, buf.alloc(count));
    if (count)
        HIP_TRY(r, hipMemcpyAsync(buf.p, src, count * sizeof(T), hipMemcpyHostToDevice, r->stream));
    return PTX_OK;
}

static int shareScene(PtxRenderer *r, PtxRenderer *owner)
{
    if (!r || !owner || r == owner)
        return fail(r, PTX_ERROR_INVALID_ARGUMENT, "ptx_share_scene: need two different renderers");
    if (owner->sceneOwner || !r->sceneSharers.empty())
        return fail(r, PTX_ERROR_INVALID_ARGUMENT, "ptx_share_scene: the owner must hold its own scene, and a renderer others share from cannot borrow");
    if (owner->device != r->device)
        return fail(r, PTX_ERROR_INVALID_ARGUMENT, "ptx_share_scene: renderers on different devices (%d, %d)", r->device, owner->device);
    if (!owner->sceneReady || !owner->accelReady)
        return fail(r, PTX_ERROR_NOT_READY, "ptx_share_scene: the owner needs ptx_scene_upload and ptx_build_accel first");
    HIP_TRY(r, hipSetDevice(r->device));
    // the owner's uploads and build are enqueued on ITS stream: finished before any stream of the borrower reads them;
    // the borrower's own frames in flight end before its scene goes away
    HIP_TRY(r, hipStreamSynchronize(owner->stream));
    HIP_TRY(r, hipStreamSynchronize(r->stream));
    if (r->auxStream)
        HIP_TRY(r, hipStreamSynchronize(r->auxStream));
    detachSharedScene(r);
    // its own copies are not needed any more
    r->vertices.release(); r->indices.release(); r->mr.release(); r->sg.release(); r->phong.release(); r->pairs.release();
    r->pairFirst.release(); r->textures.release(); r->texels8.release(); r->texelsF.release(); r->srgbLut.release();
    r->renderTextures.release(); r->renderTexels.release();
    r->alphaTex.release(); r->alphaTexOf.release(); r->alphaQuads.release(); r->alphaTris.release();
    r->animatedVertices.release(); r->skinSource.release(); r->bones.release();
    r->nodes.release(); r->tris.release(); r->shadeTris.release();
    r->build.release();
    r->sceneReady = false;
    r->sceneOwner = owner;
    owner->sceneSharers.push_back(r);
    r->accelReady = true;
    r->hintSlots = 0u; // whatever this handle had learnt, it had learnt on another scene
    r->stats.triangles = owner->stats.triangles;
    r->stats.bvhNodes = owner->stats.bvhNodes;
    r->stats.treeTriangles = owner->stats.treeTriangles;
    r->stats.treeReferences = owner->stats.treeReferences;
    return PTX_OK;
}

static int sceneUpload(PtxRenderer *r, const PtxSceneDesc *s)
{
    if (!r || !s)
        return fail(r, PTX_ERROR_INVALID_ARGUMENT, "ptx_scene_upload: null argument");
    HIP_TRY(r, hipSetDevice(r->device));

    // validate indices the kernels will dereference (the reference trusts its importer); a description that is refused
    // here leaves the handle as it was -- its own scene, or the one it borrows
    for (uint32_t i = 0; i < s->instanceCount; i++)
        if (s->instances[i].ModelIndex >= s->modelCount)
            return fail(r, PTX_ERROR_INVALID_ARGUMENT, "instance %u: model index out of range", i);
    for (uint32_t m = 0; m < s->modelCount; m++)
        if ((uint64_t)s->models[m].MeshOffset + s->models[m].MeshCount > s->meshCount)
            return fail(r, PTX_ERROR_INVALID_ARGUMENT, "model %u: mesh range out of bounds", m);
    for (uint32_t k = 0; k < s->meshCount; k++)
    {
        const PtxMeshRecord &rec = s->meshes[k];
        if (rec.GeometryIndex >= s->geometryCount || rec.TransformIndex >= s->transformCount)
            return fail(r, PTX_ERROR_INVALID_ARGUMENT, "mesh %u: geometry/transform index out of range", k);
        const uint32_t type = rec.MaterialId & 0xffu, index = rec.MaterialId >> 8;
        const uint32_t limit = type == PTX_MATERIAL_TYPE_METALLIC_ROUGHNESS    ? s->metallicRoughnessMaterialCount
                               : type == PTX_MATERIAL_TYPE_SPECULAR_GLOSSINESS ? s->specularGlossinessMaterialCount
                               : type == PTX_MATERIAL_TYPE_PHONG               ? s->phongMaterialCount
                                                                               : 0xffffffffu;
        if (type <= PTX_MATERIAL_TYPE_PHONG && index >= limit)
            return fail(r, PTX_ERROR_INVALID_ARGUMENT, "mesh %u: material index out of range", k);
    }
    for (uint32_t g = 0; g < s->geometryCount; g++)
    {
        const PtxGeometry &geo = s->geometries[g];
        // an animated geometry addresses the animated vertex / index arrays (Renderer.cpp:280-312)
        const uint64_t vLimit = geo.IsAnimated ? (s->animatedVertices ? s->animatedVertexCount : 0) : s->vertexCount;
        const uint64_t iLimit = geo.IsAnimated ? (s->animatedIndices ? s->animatedIndexCount : 0) : s->indexCount;
        const uint32_t *idx = geo.IsAnimated ? s->animatedIndices : s->indices;
        if ((uint64_t)geo.VertexOffset + geo.VertexLength > vLimit || (uint64_t)geo.IndexOffset + geo.IndexLength > iLimit)
            return fail(r, PTX_ERROR_INVALID_ARGUMENT, "geometry %u: vertex/index range out of bounds", g);
        for (uint32_t k = 0; k < geo.IndexLength; k++)
            if (idx[geo.IndexOffset + k] >= geo.VertexLength)
                return fail(r, PTX_ERROR_INVALID_ARGUMENT, "geometry %u: index %u beyond its vertex range", g, k);
    }

    // from here on the old scene is gone, whatever happens
    detachSharedScene(r); // a renderer that was borrowing a scene gets its own again
    quiesceSharers(r);
    r->sceneReady = r->accelReady = false;
    r->sceneEpoch++;

    // (instance, mesh) pairs in instance-then-mesh order; global triangle id = running prim count
    // Device vertex buffer = scene vertices, then one skinned copy per instanced animated mesh in pair order
    // (OutAnimatedVertexBuffer, Renderer.cpp:296-303); device index buffer = scene indices, then the animated indices.
    std::vector<DevPair> pairs;
    std::vector<uint32_t> pairFirst;
    std::vector<uint32_t> skinSource; // output vertex -> animated vertex (AnimatedVertexMapBuffer)
    r->pairInstance.clear();
    r->pairMeshTransform.clear();
    uint64_t tri = 0;
    bool anyNonOpaque = false;
    for (uint32_t i = 0; i < s->instanceCount; i++)
    {
        const PtxModelInstance &inst = s->instances[i];
        const PtxModel &model = s->models[inst.ModelIndex];
        for (uint32_t k = 0; k < model.MeshCount; k++)
        {
            const PtxMeshRecord &rec = s->meshes[model.MeshOffset + k];
            const PtxGeometry &geo = s->geometries[rec.GeometryIndex];
            DevPair pr;
            composeTransform(inst.Transform.m, s->transforms[rec.TransformIndex].m, pr.M);
            inverseLinear(pr.M, pr.Rinv);
            pr.vertexOffset = geo.VertexOffset;
            pr.indexOffset = geo.IndexOffset;
            if (geo.IsAnimated)
            {
                if (s->vertexCount + skinSource.size() + geo.VertexLength > 0xffffffffull || s->indexCount + s->animatedIndexCount > 0xffffffffull)
                    return fail(r, PTX_ERROR_INVALID_ARGUMENT, "animated meshes exceed the 32-bit vertex / index space");
                pr.vertexOffset = static_cast<uint32_t>(s->vertexCount + skinSource.size());
                pr.indexOffset = static_cast<uint32_t>(s->indexCount + geo.IndexOffset);
                for (uint32_t v = 0; v < geo.VertexLength; v++)
                    skinSource.push_back(geo.VertexOffset + v);
            }
            r->pairInstance.push_back(i);
            r->pairMeshTransform.push_back(s->transforms[rec.TransformIndex]);
            pr.materialId = rec.MaterialId;
            pr.flags = (geo.IsOpaque ? 0u : kPairNonOpaque) | (materialSamplesSceneTexture(s, rec.MaterialId) ? kPairTextured : 0u);
            if (pr.flags & kPairNonOpaque)
                anyNonOpaque = true;
            pairs.push_back(pr);
            pairFirst.push_back(static_cast<uint32_t>(tri));
            tri += geo.IndexLength / 3;
        }
    }
    if (tri > kMaxTriangles)
        return fail(r, PTX_ERROR_INVALID_ARGUMENT, "scene has %llu triangles; limit is 2^30-1", (unsigned long long)tri);
    pairFirst.push_back(static_cast<uint32_t>(tri));
    {
        uint32_t typesSeen = 0; // bit per material type, unknown types share bit 3
        for (const DevPair &pr : pairs)
            typesSeen |= 1u << ((pr.materialId & 0xffu) <= PTX_MATERIAL_TYPE_PHONG ? (pr.materialId & 0xffu) : 3u);
        r->mixedMaterialTypes = (typesSeen & (typesSeen - 1u)) != 0u;
        bool textured = false, plain = false;
        for (const DevPair &pr : pairs)
            ((pr.flags & kPairTextured) ? textured : plain) = true;
        r->mixedTextured = textured && plain;
    }
    r->pairCount = static_cast<uint32_t>(pairs.size());
    r->triCount = static_cast<uint32_t>(tri);
    r->dxNormalTextures = s->dxNormalTextures;

    int rc;
    r->hostPairs = pairs;
    r->instanceCount = s->instanceCount;
    r->staticVertexCount = s->vertexCount;
    r->skinnedCount = static_cast<uint32_t>(skinSource.size());
    r->boneCount = 0;
    r->build.release();
    // bind pose of every skinned copy (OutBindPoseAnimatedVertices); the staging vectors live until the stream
    // synchronisation at the end of this function
    std::vector<PtxVertex> verts;
    std::vector<uint32_t> inds;
    {
        verts.assign(s->vertices, s->vertices + s->vertexCount);
        verts.reserve(verts.size() + skinSource.size());
        for (uint32_t src : skinSource)
        {
            const PtxAnimatedVertex &a = s->animatedVertices[src];
            PtxVertex v;
            std::memset(&v, 0, sizeof(v));
            std::memcpy(v.Position, a.Position, 12); std::memcpy(v.TexCoords, a.TexCoords, 8); std::memcpy(v.Normal, a.Normal, 12);
            std::memcpy(v.Tangent, a.Tangent, 12); std::memcpy(v.Bitangent, a.Bitangent, 12);
            verts.push_back(v);
        }
        inds.assign(s->indices, s->indices + s->indexCount);
        if (s->animatedIndices)
            inds.insert(inds.end(), s->animatedIndices, s->animatedIndices + s->animatedIndexCount);
        if ((rc = upload(r, r->vertices, verts.data(), verts.size())) != PTX_OK) return rc;
        if ((rc = upload(r, r->indices, inds.data(), inds.size())) != PTX_OK) return rc;
        if ((rc = upload(r, r->animatedVertices, s->animatedVertices, skinSource.empty() ? 0 : s->animatedVertexCount)) != PTX_OK) return rc;
        if ((rc = upload(r, r->skinSource, skinSource.data(), skinSource.size())) != PTX_OK) return rc;
    }
    if ((rc = upload(r, r->mr, s->metallicRoughnessMaterials, s->metallicRoughnessMaterialCount)) != PTX_OK) return rc;
    if ((rc = upload(r, r->sg, s->specularGlossinessMaterials, s->specularGlossinessMaterialCount)) != PTX_OK) return rc;
    if ((rc = upload(r, r->phong, s->phongMaterials, s->phongMaterialCount)) != PTX_OK) return rc;
    if ((rc = upload(r, r->pairs, pairs.data(), pairs.size())) != PTX_OK) return rc;
    if ((rc = upload(r, r->pairFirst, pairFirst.data(), pairFirst.size())) != PTX_OK) return rc;
    // textures (row N1): level 0 to the pools, then the mip chain level by level on the device
    {
        HIP_TRY(r, r->srgbLut.alloc(256));
        k_build_srgb_lut<<<1, 256, 0, r->stream>>>(r->srgbLut.p);
        r->textureCount = s->textures ? s->textureCount : 0;
        // the skybox images follow the scene textures in the table, one level each (TextureUploader.cpp:203-262)
        r->skyKind = s->skybox ? s->skyboxKind : (uint32_t)PTX_SKYBOX_CLEAR_COLOR;
        if (r->skyKind > PTX_SKYBOX_CUBE)
            return fail(r, PTX_ERROR_INVALID_ARGUMENT, "unknown skybox kind %u", r->skyKind);
        const uint32_t skyCount = r->skyKind == PTX_SKYBOX_2D ? 1u : r->skyKind == PTX_SKYBOX_CUBE ? 6u : 0u;
        const uint32_t total = r->textureCount + skyCount;
        auto descOf = [&](uint32_t i) -> const PtxTextureDesc & { return i < r->textureCount ? s->textures[i] : s->skybox[i - r->textureCount]; };
        if (r->skyKind == PTX_SKYBOX_CUBE)
            for (uint32_t f = 0; f < 6; f++)
                if (s->skybox[f].width != s->skybox[0].width || s->skybox[f].height != s->skybox[0].width || s->skybox[f].format != s->skybox[0].format)
                    return fail(r, PTX_ERROR_INVALID_ARGUMENT, "cube skybox: the six faces must be equal squares of one format");
        // TextureUploader::DetermineMaxTextureSizes (TextureUploader.cpp:551-569): the largest square extent whose full chain
        // fits the per-texture share of the budget (Config.h:63-64,162-163: min(80 % of the device memory, 1 GiB)), per format;
        // forceFullTextureSize keeps MaxTextureDataSize = 4096 (TextureUploader.h:74).  Block-compressed files arrive decoded
        // to RGBA8 and are budgeted as that.
        uint32_t maxExtent[3] = { 4096u, 4096u, 4096u };
        if (!s->forceFullTextureSize && r->textureCount && s->textureMemoryBudget != ~0ull)
        {
            uint64_t budget = s->textureMemoryBudget;
            if (!budget)
            {
                size_t freeB = 0, totalB = 0;
                HIP_TRY(r, hipMemGetInfo(&freeB, &totalB));
                budget = (uint64_t)totalB / 100u * 80u;
                if (budget > (1024ull << 20))
                    budget = 1024ull << 20;
            }
            const uint64_t perTexture = budget / r->textureCount;
            for (uint32_t f = 0; f <= PTX_TEXTURE_RGBA32F; f++)
                while (maxExtent[f] > 1u)
                {
                    uint64_t texels = 0;
                    for (uint32_t e = maxExtent[f]; e; e >>= 1)
                        texels += (uint64_t)e * e;
                    if (texels * (f == PTX_TEXTURE_RGBA32F ? 16u : 4u) <= perTexture)
                        break;
                    maxExtent[f] >>= 1;
                }
        }
        auto fullLevels = [](uint32_t w, uint32_t h) {
            uint32_t m = w > h ? w : h, levels = 1;
            while (m > 1) { m >>= 1; levels++; } // floor(log2(max)) + 1, Image.cpp:14-17
            return levels > 16u ? 16u : levels;
        };
        auto dim = [](uint32_t v, uint32_t l) { return v >> l ? v >> l : 1u; };
        // per texture: what lands in the table, and how its level 0 is produced
        struct Placement
        {
            uint32_t srcW, srcH;  // the file's level 0
            uint32_t fileLevels;  // levels in the caller's data
            uint32_t firstFile;   // file level that becomes level 0 when the file's own chain is used
            bool useFileChain;    // every level comes from the file (TextureUploader.cpp:440,492-501)
            uint32_t halvings;    // blits from the file's level 0 down towards the budgeted extent (:479-490)
            int temp;             // table entry of the scratch chain, or -1
        };
        std::vector<Placement> place(total);
        std::vector<DevTexture> table(total);
        size_t n8 = 0, nf = 0, scratch8 = 0, scratchF = 0;
        uint32_t scaled = 0;
        for (uint32_t i = 0; i < total; i++)
        {
            const PtxTextureDesc &d = descOf(i);
            DevTexture &t = table[i];
            Placement &pl = place[i];
            if (d.format > PTX_TEXTURE_RGBA32F)
                return fail(r, PTX_ERROR_INVALID_ARGUMENT, "texture %u: unknown format %u", i, d.format);
            pl.srcW = d.width ? d.width : 1;
            pl.srcH = d.height ? d.height : 1;
            pl.fileLevels = d.levels ? d.levels : 1u;
            pl.firstFile = 0;
            pl.useFileChain = false;
            pl.halvings = 0;
            pl.temp = -1;
            t.width = pl.srcW;
            t.height = pl.srcH;
            t.format = d.format;
            if (i < r->textureCount)
            {
                // TextureUploader::UploadTexture (:409-415): integer scale that brings both sides under the limit
                const uint32_t mx = maxExtent[d.format];
                const uint32_t scale = std::max((pl.srcW + mx - 1) / mx, (pl.srcH + mx - 1) / mx);
                t.width = std::max(pl.srcW / scale, 1u);
                t.height = std::max(pl.srcH / scale, 1u);
                t.levels = fullLevels(t.width, t.height);
                if (pl.fileLevels > fullLevels(pl.srcW, pl.srcH))
                    return fail(r, PTX_ERROR_INVALID_ARGUMENT, "texture %u: %u levels for a %u x %u image", i, pl.fileLevels, pl.srcW, pl.srcH);
                if (scale == 1)
                    pl.useFileChain = pl.fileLevels == t.levels && t.levels > 1;
                else
                {
                    // a file with its own chain: the levels from the budgeted extent down are taken as they are (:492-501)
                    const uint32_t skip = pl.fileLevels > t.levels ? pl.fileLevels - t.levels : 0u;
                    if (skip && dim(pl.srcW, skip) == t.width && dim(pl.srcH, skip) == t.height)
                    {
                        pl.useFileChain = true;
                        pl.firstFile = skip;
                    }
                    else
                    {
                        while (dim(pl.srcW, pl.halvings + 1) >= t.width && dim(pl.srcH, pl.halvings + 1) >= t.height &&
                               (dim(pl.srcW, pl.halvings) > t.width || dim(pl.srcH, pl.halvings) > t.height))
                            pl.halvings++;
                        pl.temp = (int)(total + scaled++);
                        size_t need = 0;
                        for (uint32_t l = 0; l <= pl.halvings; l++)
                            need += (size_t)dim(pl.srcW, l) * dim(pl.srcH, l);
                        size_t &sc = d.format == PTX_TEXTURE_RGBA32F ? scratchF : scratch8;
                        sc = std::max(sc, need);
                    }
                }
            }
            else
                t.levels = 1;
            size_t &cursor = t.format == PTX_TEXTURE_RGBA32F ? nf : n8;
            for (uint32_t l = 0; l < t.levels; l++)
            {
                t.levelOffset[l] = (uint32_t)cursor;
                cursor += (size_t)dim(t.width, l) * dim(t.height, l);
            }
        }
        // scratch chains of the textures that are scaled down: one region per pool behind the textures, used by one
        // texture after the other (stream order); their table entries follow the real ones
        table.resize(total + scaled);
        for (uint32_t i = 0; i < total; i++)
            if (place[i].temp >= 0)
            {
                DevTexture &t = table[(size_t)place[i].temp];
                t.width = place[i].srcW;
                t.height = place[i].srcH;
                t.format = table[i].format;
                t.levels = place[i].halvings + 1;
                size_t cursor = t.format == PTX_TEXTURE_RGBA32F ? nf : n8;
                for (uint32_t l = 0; l < t.levels; l++)
                {
                    t.levelOffset[l] = (uint32_t)cursor;
                    cursor += (size_t)dim(t.width, l) * dim(t.height, l);
                }
            }
        if (n8 + scratch8 > 0xffffffffull || nf + scratchF > 0xffffffffull)
            return fail(r, PTX_ERROR_INVALID_ARGUMENT, "texture pool exceeds 2^32 texels");
        HIP_TRY(r, r->textures.alloc(total + scaled));
        HIP_TRY(r, r->texels8.alloc(n8 + scratch8));
        HIP_TRY(r, r->texelsF.alloc(nf + scratchF));
        if (n8)
            HIP_TRY(r, hipMemsetAsync(r->texels8.p, 0, n8 * 4, r->stream)); // a texture without data reads as zeros
        if (nf)
            HIP_TRY(r, hipMemsetAsync(r->texelsF.p, 0, nf * 16, r->stream));
        // A 1x1 opaque-white 8-bit texture decodes to exactly (1,1,1,1) in both formats, which is what the
        // kernels without the sampler return for any index >= 9: only other content needs the TEX variants.
        r->samplerNeeded = false;
        r->anyNonOpaque = anyNonOpaque;
        for (uint32_t i = 0; i < r->textureCount; i++)
        {
            const PtxTextureDesc &d = s->textures[i];
            const bool whitePlaceholder = table[i].width == 1 && table[i].height == 1 && d.format != PTX_TEXTURE_RGBA32F && d.data &&
                                          *static_cast<const uint32_t *>(d.data) == 0xffffffffu && place[i].srcW == 1 && place[i].srcH == 1;
            if (!whitePlaceholder)
                r->samplerNeeded = true;
        }
        if (!table.empty())
            HIP_TRY(r, hipMemcpyAsync(r->textures.p, table.data(), table.size() * sizeof(DevTexture), hipMemcpyHostToDevice, r->stream));
        TextureView tv;
        tv.textures = r->textures.p; tv.textureCount = r->textureCount; tv.texels8 = r->texels8.p; tv.texelsF = r->texelsF.p;
        tv.srgbLut = r->srgbLut.p;
        for (uint32_t i = 0; i < total; i++)
        {
            const PtxTextureDesc &d = descOf(i);
            const DevTexture &t = table[i];
            const Placement &pl = place[i];
            const bool isFloat = t.format == PTX_TEXTURE_RGBA32F;
            const size_t texel = isFloat ? 16 : 4;
            auto poolAt = [&](uint32_t offset) -> void * { return isFloat ? (void *)(r->texelsF.p + offset) : (void *)(r->texels8.p + offset); };
            auto blit = [&](uint32_t src, uint32_t srcLevel, uint32_t dst, uint32_t dstLevel) {
                const uint32_t dw = dim(table[dst].width, dstLevel), dh = dim(table[dst].height, dstLevel);
                k_blit_level<<<(dw * dh + 255) / 256, 256, 0, r->stream>>>(tv, src, srcLevel, dst, dstLevel, r->texels8.p, r->texelsF.p);
            };
            if (d.data && pl.useFileChain)
            {
                // the file's own levels, from the one that has the budgeted extent
                const uint8_t *p = static_cast<const uint8_t *>(d.data);
                for (uint32_t l = 0; l < pl.firstFile; l++)
                    p += (size_t)dim(pl.srcW, l) * dim(pl.srcH, l) * texel;
                for (uint32_t l = 0; l < t.levels; l++)
                {
                    const size_t nl = (size_t)dim(t.width, l) * dim(t.height, l);
                    HIP_TRY(r, hipMemcpyAsync(poolAt(t.levelOffset[l]), p, nl * texel, hipMemcpyHostToDevice, r->stream));
                    p += nl * texel;
                }
                continue;
            }
            if (d.data && pl.temp >= 0)
            {
                // scaled down: the file's level 0 into the scratch chain, halved by linear blits, then into level 0
                const DevTexture &tt = table[(size_t)pl.temp];
                HIP_TRY(r, hipMemcpyAsync(poolAt(tt.levelOffset[0]), d.data, (size_t)pl.srcW * pl.srcH * texel, hipMemcpyHostToDevice, r->stream));
                for (uint32_t l = 1; l <= pl.halvings; l++)
                    blit((uint32_t)pl.temp, l - 1, (uint32_t)pl.temp, l);
                if (dim(pl.srcW, pl.halvings) == t.width && dim(pl.srcH, pl.halvings) == t.height)
                    HIP_TRY(r, hipMemcpyAsync(poolAt(t.levelOffset[0]), poolAt(tt.levelOffset[pl.halvings]), (size_t)t.width * t.height * texel,
                                              hipMemcpyDeviceToDevice, r->stream));
                else
                    blit((uint32_t)pl.temp, pl.halvings, i, 0);
            }
            else if (d.data)
                HIP_TRY(r, hipMemcpyAsync(poolAt(t.levelOffset[0]), d.data, (size_t)t.width * t.height * texel, hipMemcpyHostToDevice, r->stream));
            for (uint32_t l = 1; l < t.levels; l++)
                blit(i, l - 1, i, l);
        }
        // The pool the render kernels sample (pt_device.hpp, fetchTexel): every level of every texture decoded to four floats,
        // the RGBA32F pool first, the 8-bit textures behind it; `renderTextures` is the table with offsets into that pool.
        if ((uint64_t)nf + n8 > 0xffffffffull)
            return fail(r, PTX_ERROR_INVALID_ARGUMENT, "texture pool exceeds 2^32 texels");
        HIP_TRY(r, r->renderTexels.alloc(nf + n8));
        HIP_TRY(r, r->renderTextures.alloc(total));
        if (nf)
            HIP_TRY(r, hipMemcpyAsync(r->renderTexels.p, r->texelsF.p, nf * sizeof(float4), hipMemcpyDeviceToDevice, r->stream));
        std::vector<DevTexture> renderTable(table.begin(), table.begin() + total);
        for (uint32_t i = 0; i < total; i++)
        {
            DevTexture &t = renderTable[i];
            if (t.format == PTX_TEXTURE_RGBA32F)
                continue;
            size_t count = 0;
            for (uint32_t l = 0; l < t.levels; l++)
                count += (size_t)dim(t.width, l) * dim(t.height, l);
            const uint32_t first = t.levelOffset[0];
            k_decode_texels<<<(uint32_t)((count + 255) / 256), 256, 0, r->stream>>>(r->texels8.p, r->srgbLut.p, first, (uint32_t)count, t.format,
                                                                                  r->renderTexels.p + nf + first);
            for (uint32_t l = 0; l < t.levels; l++)
                t.levelOffset[l] += (uint32_t)nf;
        }
        if (total)
            HIP_TRY(r, hipMemcpyAsync(r->renderTextures.p, renderTable.data(), total * sizeof(DevTexture), hipMemcpyHostToDevice, r->stream));
        // any-hit data: the alpha footprints of every texture some material names as its colour texture
        std::vector<AlphaTex> alphaTex;
        std::vector<uint32_t> alphaTexOf(r->textureCount ? r->textureCount : 1u, kNoAlphaTex);
        if (anyNonOpaque)
        {
            size_t quads = 0;
            bool tooLarge = false; // the extent of an alpha texture rides in 15 + 15 bits of the triangle record
            auto mark = [&](uint32_t colorIdx) {
                if (colorIdx < PTX_SCENE_TEXTURE_OFFSET || colorIdx - PTX_SCENE_TEXTURE_OFFSET >= r->textureCount)
                    return;
                const uint32_t ti = colorIdx - PTX_SCENE_TEXTURE_OFFSET;
                if (alphaTexOf[ti] != kNoAlphaTex)
                    return;
                if (renderTable[ti].width > 32768u || renderTable[ti].height > 32768u)
                {
                    tooLarge = true;
                    return;
                }
                alphaTexOf[ti] = (uint32_t)alphaTex.size();
                alphaTex.push_back({ renderTable[ti].width, renderTable[ti].height, (uint32_t)quads, 0u });
                quads += (size_t)renderTable[ti].width * renderTable[ti].height;
            };
            for (uint32_t i = 0; i < s->metallicRoughnessMaterialCount; i++) mark(s->metallicRoughnessMaterials[i].ColorIdx);
            for (uint32_t i = 0; i < s->specularGlossinessMaterialCount; i++) mark(s->specularGlossinessMaterials[i].ColorIdx);
            for (uint32_t i = 0; i < s->phongMaterialCount; i++) mark(s->phongMaterials[i].ColorIdx);
            if (tooLarge)
                return fail(r, PTX_ERROR_INVALID_ARGUMENT, "a colour texture of a non-opaque geometry is larger than 32768 texels across");
            if (quads >= 0xffffffffull)
                return fail(r, PTX_ERROR_INVALID_ARGUMENT, "alpha footprints exceed 2^32 texels");
            HIP_TRY(r, r->alphaQuads.alloc(quads));
            for (uint32_t ti = 0; ti < r->textureCount; ti++)
                if (alphaTexOf[ti] != kNoAlphaTex)
                {
                    const AlphaTex &at = alphaTex[alphaTexOf[ti]];
                    k_alpha_quads<<<(at.width * at.height + 255) / 256, 256, 0, r->stream>>>(at.width, at.height, r->renderTexels.p + renderTable[ti].levelOffset[0],
                                                                                            r->alphaQuads.p + at.offset);
                }
        }
        if ((rc = upload(r, r->alphaTex, alphaTex.data(), alphaTex.size())) != PTX_OK) return rc;
        if ((rc = upload(r, r->alphaTexOf, alphaTexOf.data(), alphaTexOf.size())) != PTX_OK) return rc;
        HIP_TRY(r, hipStreamSynchronize(r->stream)); // `table` and the caller's texel arrays may go away
        HIP_TRY(r, hipGetLastError());
        // the pools of the upload formats have done their work (mip chains, scaling)
        r->texels8.release(); r->texelsF.release(); r->srgbLut.release(); r->textures.release();
    }
    HIP_TRY(r, hipStreamSynchronize(r->stream)); // the host vectors above go out of scope
    r->sceneReady = true;
    r->stats.triangles = tri;
    return PTX_OK;
}

static SceneView makeSceneView(const PtxRenderer *r);

// One 8-bit pass of the LSD radix sort of pt_bvh_build.hpp over `count` (key, value) pairs; hist / histSums sized by the caller.
static void radixPass(PtxRenderer *r, uint32_t count, const uint64_t *kin, const uint32_t *vin, uint64_t *kout, uint32_t *vout, uint32_t shift,
                      uint32_t *hist, uint32_t *histSums)
{
    const uint32_t numTiles = (count + kSortTile - 1) / kSortTile;
    const uint32_t histCount = 256 * numTiles, histBlocks = (histCount + kScan32Block - 1) / kScan32Block;
    k_sort_hist<<<numTiles, 64, 0, r->stream>>>(count, kin, shift, numTiles, hist);
    if (histBlocks > 1)
    {
        k_scan32_sums<<<histBlocks, 256, 0, r->stream>>>(histCount, hist, histSums);
        k_scan_exclusive<<<1, 1024, 0, r->stream>>>(histBlocks, histSums);
        k_scan32_apply<<<histBlocks, 256, 0, r->stream>>>(histCount, hist, histSums);
    }
    else
        k_scan_exclusive<<<1, 1024, 0, r->stream>>>(histCount, hist);
    k_sort_scatter<<<numTiles, 64, 0, r->stream>>>(count, kin, vin, kout, vout, shift, numTiles, hist);
}

// Level lists of the CURRENT binary topology over nv leaves (B.children / B.parentOfNode): B.levelOrder, B.levelStart.
// *usable = false: the tree is deeper than kMaxTreeLevels (or is not a tree) and the caller takes the fence-and-atomic kernels.
// Scratch: B.keys0 / keys1 (the Morton keys are done with), B.hist / histSums.
static hipError_t treeLevels(PtxRenderer *r, uint32_t nv, bool *usable)
{
    PtxRenderer::BuildState &B = r->build;
    B.levelsValid = false;
    *usable = false;
    if (nv < 2)
        return hipSuccess;
    const uint32_t nodes = nv - 1, blocks = (nodes + 255) / 256;
#define LV_TRY(expr) do { const hipError_t e_ = (expr); if (e_ != hipSuccess) return e_; } while (0)
    LV_TRY(B.lvDepth0.alloc(nodes)); LV_TRY(B.lvDepth1.alloc(nodes)); LV_TRY(B.lvAnc0.alloc(nodes)); LV_TRY(B.lvAnc1.alloc(nodes));
    LV_TRY(B.lvVals0.alloc(nodes)); LV_TRY(B.lvVals1.alloc(nodes)); LV_TRY(B.lvStartDev.alloc(kMaxTreeLevels + 2));
    uint32_t *d0 = B.lvDepth0.p, *d1 = B.lvDepth1.p, *flag = B.lvStartDev.p; // (flag: the first word, before the starts are written)
    int *a0 = B.lvAnc0.p, *a1 = B.lvAnc1.p;
    k_depth_init<<<blocks, 256, 0, r->stream>>>((int)nodes, B.parentOfNode.p, d0, a0);
    bool done = false;
    for (uint32_t pass = 0; pass < 24 && !done; pass++) // pass k covers paths of 2^(k + 1) links
    {
        LV_TRY(hipMemsetAsync(flag, 0, sizeof(uint32_t), r->stream));
        k_depth_jump<<<blocks, 256, 0, r->stream>>>((int)nodes, d0, a0, d1, a1, flag);
        std::swap(d0, d1);
        std::swap(a0, a1);
        if (pass >= 4) // (a tree of 64 or more leaves is at least six deep: no point in asking earlier)
        {
            uint32_t pending = 0;
            LV_TRY(hipMemcpyAsync(&pending, flag, sizeof(pending), hipMemcpyDeviceToHost, r->stream));
            LV_TRY(hipStreamSynchronize(r->stream));
            done = pending == 0;
        }
    }
    if (!done)
        return hipSuccess; // a parent chain longer than 2^24: not a tree the level passes can take
    uint32_t maxDepth = 0;
    LV_TRY(hipMemsetAsync(flag, 0, sizeof(uint32_t), r->stream));
    k_depth_keys<<<blocks, 256, 0, r->stream>>>((int)nodes, d0, B.keys0.p, B.lvVals0.p, flag);
    LV_TRY(hipMemcpyAsync(&maxDepth, flag, sizeof(maxDepth), hipMemcpyDeviceToHost, r->stream));
    LV_TRY(hipStreamSynchronize(r->stream));
    if (maxDepth >= kMaxTreeLevels)
        return hipSuccess;
    const uint64_t *sortedKeys = B.keys1.p;
    radixPass(r, nodes, B.keys0.p, B.lvVals0.p, B.keys1.p, B.lvVals1.p, 0, B.hist.p, B.histSums.p);
    B.levelOrder = B.lvVals1.p;
    if (maxDepth > 255)
    {
        radixPass(r, nodes, B.keys1.p, B.lvVals1.p, B.keys0.p, B.lvVals0.p, 8, B.hist.p, B.histSums.p);
        B.levelOrder = B.lvVals0.p;
        sortedKeys = B.keys0.p;
    }
    k_level_starts<<<blocks, 256, 0, r->stream>>>((int)nodes, sortedKeys, B.lvStartDev.p);
    B.levelStart.assign(maxDepth + 2, 0u);
    LV_TRY(hipMemcpyAsync(B.levelStart.data(), B.lvStartDev.p, (maxDepth + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost, r->stream));
    LV_TRY(hipStreamSynchronize(r->stream));
    B.levelStart[maxDepth + 1] = nodes;
    for (uint32_t d = 0; d <= maxDepth; d++) // every depth up to the deepest holds a node, in order
        if (B.levelStart[d] >= B.levelStart[d + 1])
            return hipSuccess;
#undef LV_TRY
    B.levelsValid = true;
    *usable = true;
    return hipSuccess;
}

// Bottom-up boxes of the binary tree over the level lists: one launch per level, deepest first.
static void refitLevels(PtxRenderer *r, const uint32_t *vin, const float4 *refLo, const float4 *refHi)
{
    PtxRenderer::BuildState &B = r->build;
    for (size_t d = B.levelStart.size() - 1; d-- > 0;)
    {
        const uint32_t first = B.levelStart[d], count = B.levelStart[d + 1] - first;
        k_refit_level<<<(count + 255) / 256, 256, 0, r->stream>>>(first, count, B.levelOrder, vin, refLo, refHi, B.children.p, B.nodeLo.p, B.nodeHi.p);
    }
}

// Full build (refit = false) or refit: new triangle records and leaf boxes, then the bottom-up box pass and the
// 4-wide emit over the KEPT Morton order and binary topology.  keepState leaves the temporaries allocated for
// later refits; a static scene frees them.
static int buildAccel(PtxRenderer *r, bool refit, bool keepState)
{
    HIP_TRY(r, hipSetDevice(r->device));
    const uint32_t nTri = r->triCount;
    if (nTri == 0)
    {
        if (!refit)
        {
            HIP_TRY(r, r->nodes.alloc(1));
            HIP_TRY(r, r->tris.alloc(1));
            HIP_TRY(r, r->shadeTris.alloc(1));
        }
        r->accelReady = true;
        r->treeTris = 0;
        r->stats.bvhNodes = 0;
        r->stats.treeTriangles = r->stats.treeReferences = 0;
        r->stats.lastBuildMs = 0.0;
        return PTX_OK;
    }
    PtxRenderer::BuildState &B = r->build;
#define BUILD_TRY(expr)                                                                                                    \
    do                                                                                                                     \
    {                                                                                                                      \
        const hipError_t e_ = (expr);                                                                                      \
        if (e_ != hipSuccess)                                                                                              \
        {                                                                                                                  \
            B.release();                                                                                                   \
            return fail(r, e_ == hipErrorOutOfMemory ? PTX_ERROR_OUT_OF_MEMORY : PTX_ERROR_DEVICE, "%s: %s", #expr,       \
                        hipGetErrorString(e_));                                                                            \
        }                                                                                                                  \
    } while (0)
    // ---- per triangle: world-space record, padded box, zero-area flag
    if (!refit)
    {
        B.valid = false;
        // the level lists describe the topology of the LAST build: a full build starts without them (a build whose reinsertion
        // passes do not run -- PTX_REINSERT=0, a broken pass, the rebuild after a revived triangle -- would otherwise price its
        // collapse in the order and over the node count of an older tree)
        B.levelsValid = false;
        B.levelOrder = nullptr;
        B.levelStart.clear();
        BUILD_TRY(B.triTmp.alloc(nTri)); BUILD_TRY(B.boxLo.alloc(nTri)); BUILD_TRY(B.boxHi.alloc(nTri)); BUILD_TRY(B.inert.alloc(nTri));
        BUILD_TRY(B.sceneBounds.alloc(8));
    }
    // [0..5] centroid bounds (ordered floats), [6] references in the tree (k_count_valid), [7] a refit found a revived triangle
    const uint32_t initBounds[8] = { 0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u, 0u, 0u };
    BUILD_TRY(hipMemcpyAsync(B.sceneBounds.p, initBounds, sizeof(initBounds), hipMemcpyHostToDevice, r->stream));
    BUILD_TRY(hipEventRecord(r->evA, r->stream));
    k_tri_setup<<<(nTri + 255) / 256, 256, 0, r->stream>>>(nTri, r->pairCount, r->pairFirst.p, r->pairs.p, r->vertices.p, r->indices.p, B.triTmp.p,
                                                          B.boxLo.p, B.boxHi.p, B.sceneBounds.p, B.inert.p, refit ? 1 : 0);

    // ---- leaf references: one per triangle, or the pieces of the triangles worth splitting (static scenes, full builds: a
    // refit keeps the references of the last full build, and a build that keeps its state for refits does not split)
    uint32_t n = refit ? B.refCount : nTri;
    const float4 *refLo = B.boxLo.p, *refHi = B.boxHi.p;
    const uint8_t *refInert = B.inert.p;
    const uint32_t *refTri = nullptr; // reference -> triangle (null: the identity)
    if (!refit && !keepState && r->tree.splitBudget > 0.0f && nTri > 1)
    {
        DevBuf<float> priority;
        DevBuf<uint32_t> count, sums;
        DevBuf<unsigned long long> sum;
        const uint32_t sb = (nTri + 1 + kScan32Block - 1) / kScan32Block;
        BUILD_TRY(priority.alloc(nTri)); BUILD_TRY(count.alloc((size_t)nTri + 1)); BUILD_TRY(sums.alloc(sb)); BUILD_TRY(sum.alloc(1));
        BUILD_TRY(hipMemsetAsync(sum.p, 0, sizeof(unsigned long long), r->stream));
        k_split_priority<<<(nTri + 255) / 256, 256, 0, r->stream>>>(nTri, B.triTmp.p, B.boxLo.p, B.boxHi.p, B.inert.p, B.sceneBounds.p,
                                                                   r->tree.mortonCubic ? 1 : 0, priority.p, sum.p);
        unsigned long long total = 0;
        BUILD_TRY(hipMemcpyAsync(&total, sum.p, sizeof(total), hipMemcpyDeviceToHost, r->stream));
        BUILD_TRY(hipStreamSynchronize(r->stream));
        if (total)
        {
            const float perPriority = (float)((double)r->tree.splitBudget * nTri / ((double)total / kSplitPriorityScale));
            BUILD_TRY(hipMemsetAsync(count.p + nTri, 0, sizeof(uint32_t), r->stream));
            k_split_count<<<(nTri + 255) / 256, 256, 0, r->stream>>>(nTri, priority.p, perPriority, count.p);
            if (sb > 1)
            {
                k_scan32_sums<<<sb, 256, 0, r->stream>>>(nTri + 1, count.p, sums.p);
                k_scan_exclusive<<<1, 1024, 0, r->stream>>>(sb, sums.p);
                k_scan32_apply<<<sb, 256, 0, r->stream>>>(nTri + 1, count.p, sums.p);
            }
            else
                k_scan_exclusive<<<1, 1024, 0, r->stream>>>(nTri + 1, count.p);
            uint32_t refs = 0;
            BUILD_TRY(hipMemcpyAsync(&refs, count.p + nTri, sizeof(refs), hipMemcpyDeviceToHost, r->stream));
            BUILD_TRY(hipStreamSynchronize(r->stream));
            if (refs > nTri && refs <= kMaxTriangles)
            {
                BUILD_TRY(B.refLo.alloc(refs)); BUILD_TRY(B.refHi.alloc(refs)); BUILD_TRY(B.refTri.alloc(refs)); BUILD_TRY(B.refInert.alloc(refs));
                k_split_write<<<(nTri + 255) / 256, 256, 0, r->stream>>>(nTri, B.triTmp.p, B.boxLo.p, B.boxHi.p, B.inert.p, B.sceneBounds.p,
                                                                        r->tree.mortonCubic ? 1 : 0, count.p, refs, B.refLo.p, B.refHi.p, B.refTri.p, B.refInert.p);
                BUILD_TRY(hipStreamSynchronize(r->stream)); // (count and priority go out of scope)
                n = refs;
                refLo = B.refLo.p; refHi = B.refHi.p; refInert = B.refInert.p; refTri = B.refTri.p;
            }
        }
    }
    if (!refit)
        B.refCount = n;

    // ---- per reference: everything from the Morton sort on
    const uint32_t numTiles = (n + kSortTile - 1) / kSortTile;
    const uint32_t histCount = 256 * numTiles, histBlocks = (histCount + kScan32Block - 1) / kScan32Block;
    if (!refit)
    {
        BUILD_TRY(r->nodes.alloc(n)); // (as many as the emitted array: the two change places in the depth-first relayout)
        BUILD_TRY(r->tris.alloc(n));
        BUILD_TRY(r->shadeTris.alloc(n));
        BUILD_TRY(B.nodeLo.alloc(n)); BUILD_TRY(B.nodeHi.alloc(n)); BUILD_TRY(B.vals0.alloc(n)); BUILD_TRY(B.vals1.alloc(n));
        BUILD_TRY(B.hist.alloc(histCount)); BUILD_TRY(B.histSums.alloc(histBlocks)); BUILD_TRY(B.flags.alloc(n));
        BUILD_TRY(B.keys0.alloc(n)); BUILD_TRY(B.keys1.alloc(n));
        BUILD_TRY(B.children.alloc(n)); BUILD_TRY(B.parentOfNode.alloc(n)); BUILD_TRY(B.parentOfLeaf.alloc(n));
        BUILD_TRY(B.rawNodes.alloc(n)); BUILD_TRY(B.oldOf.alloc((size_t)n + 1)); BUILD_TRY(B.collapseCost.alloc(n)); BUILD_TRY(B.collapseDecide.alloc(n));
    }

    // PLOC temporaries: two cluster sequences, neighbour indices, scan flags (sized for all n; freed when this returns)
    DevBuf<int> cl0, cl1;
    DevBuf<float4> lo0, hi0, lo1, hi1;
    DevBuf<uint32_t> nn;
    DevBuf<unsigned long long> flags, sums, total;
    if (!refit && r->usePloc && n > 1)
    {
        BUILD_TRY(cl0.alloc(n)); BUILD_TRY(cl1.alloc(n)); BUILD_TRY(lo0.alloc(n)); BUILD_TRY(hi0.alloc(n)); BUILD_TRY(lo1.alloc(n));
        BUILD_TRY(hi1.alloc(n)); BUILD_TRY(nn.alloc(n)); BUILD_TRY(flags.alloc(n)); BUILD_TRY(sums.alloc((n + kScanBlock - 1) / kScanBlock));
        BUILD_TRY(total.alloc(1));
    }
    BUILD_TRY(hipMemsetAsync(B.flags.p, 0, (size_t)n * 4, r->stream));

    const uint32_t blocks = (n + 255) / 256;
    // 8 radix passes ping-pong the buffers an even number of times: the sorted order ends in keys0 / vals0
    uint64_t *kin = B.keys0.p, *kout = B.keys1.p;
    uint32_t *vin = B.vals0.p, *vout = B.vals1.p;
    uint32_t nv = B.treeTris; // triangles in the tree: all but the zero-area ones, which sort to the end
    if (!refit)
    {
        k_morton<<<blocks, 256, 0, r->stream>>>(n, refLo, refHi, B.sceneBounds.p, refInert, B.keys0.p, B.vals0.p, r->tree.mortonCubic ? 1 : 0);
        for (uint32_t shift = 0; shift < 64; shift += 8) // 63-bit keys + the all-ones sentinel of inert triangles: 8 passes
        {
            k_sort_hist<<<numTiles, 64, 0, r->stream>>>(n, kin, shift, numTiles, B.hist.p);
            if (histBlocks > 1)
            {
                k_scan32_sums<<<histBlocks, 256, 0, r->stream>>>(histCount, B.hist.p, B.histSums.p);
                k_scan_exclusive<<<1, 1024, 0, r->stream>>>(histBlocks, B.histSums.p);
                k_scan32_apply<<<histBlocks, 256, 0, r->stream>>>(histCount, B.hist.p, B.histSums.p);
            }
            else
                k_scan_exclusive<<<1, 1024, 0, r->stream>>>(histCount, B.hist.p);
            k_sort_scatter<<<numTiles, 64, 0, r->stream>>>(n, kin, vin, kout, vout, shift, numTiles, B.hist.p);
            std::swap(kin, kout);
            std::swap(vin, vout);
        }
        k_count_valid<<<1, 1, 0, r->stream>>>(n, kin, &B.sceneBounds.p[6]);
        BUILD_TRY(hipMemcpyAsync(&nv, &B.sceneBounds.p[6], sizeof(nv), hipMemcpyDeviceToHost, r->stream));
        BUILD_TRY(hipStreamSynchronize(r->stream));
        B.treeTris = nv;
    }
    r->treeTris = nv;
    r->stats.bvhNodes = nv > 1 ? nv - 1 : (nv ? 1 : 0);
    r->stats.treeReferences = nv;
    r->stats.treeTriangles = nTri - (n - nv); // a zero-area triangle has exactly one reference, and they are the ones left out
    const uint32_t vblocks = (nv + 255) / 256;
    if (nv == 1)
        k_single_leaf_root<<<1, 1, 0, r->stream>>>(vin, refLo, refHi, B.triTmp.p, r->nodes.p, r->tris.p, r->pairs.p, r->vertices.p,
                                                   r->indices.p, r->shadeTris.p);
    else if (nv > 1)
    {
        bool boxesDone = false;
        if (!refit && r->usePloc)
        {
            // PLOC over the sorted leaves (temporaries allocated above, outside the timed span)
            k_ploc_init<<<vblocks, 256, 0, r->stream>>>(nv, vin, refLo, refHi, cl0.p, lo0.p, hi0.p);
            int *cIn = cl0.p, *cOut = cl1.p;
            float4 *lIn = lo0.p, *hIn = hi0.p, *lOut = lo1.p, *hOut = hi1.p;
            uint32_t count = nv;
            int nextId = (int)nv - 2;
            uint32_t iterations = 0;
            while (count > 1)
            {
                const uint32_t cb = (count + 255) / 256, sb = (count + kScanBlock - 1) / kScanBlock;
                k_ploc_nearest<<<cb, 256, 0, r->stream>>>(count, r->tree.plocRadius, r->tree.plocShape, lIn, hIn, nn.p);
                k_ploc_flags<<<cb, 256, 0, r->stream>>>(count, nn.p, flags.p);
                k_scan64_sums<<<sb, 256, 0, r->stream>>>(count, flags.p, sums.p);
                k_scan64_top<<<1, 1024, 0, r->stream>>>(sb, sums.p, total.p);
                k_scan64_apply<<<sb, 256, 0, r->stream>>>(count, flags.p, sums.p);
                k_ploc_merge<<<cb, 256, 0, r->stream>>>(count, cIn, lIn, hIn, nn.p, flags.p, nextId, cOut, lOut, hOut, B.children.p, B.parentOfNode.p,
                                                        B.parentOfLeaf.p, B.nodeLo.p, B.nodeHi.p);
                unsigned long long t = 0;
                BUILD_TRY(hipMemcpyAsync(&t, total.p, sizeof(t), hipMemcpyDeviceToHost, r->stream));
                BUILD_TRY(hipStreamSynchronize(r->stream));
                const uint32_t kept = (uint32_t)t, merged = (uint32_t)(t >> 32);
                if (merged == 0 || kept + merged != count)
                {
                    B.release();
                    return fail(r, PTX_ERROR_DEVICE, "ptx_build_accel: PLOC made no progress (%u clusters, %u kept, %u merged)", count, kept, merged);
                }
                nextId -= (int)merged;
                count = kept;
                std::swap(cIn, cOut);
                std::swap(lIn, lOut);
                std::swap(hIn, hOut);
                iterations++;
            }
            if (r->env.verbose)
                std::fprintf(stderr, "[ptx] PLOC: %u triangles (%u inert left out), %u iterations\n", nv, n - nv, iterations);
            boxesDone = true;
        }
        else if (!refit)
            k_karras<<<vblocks, 256, 0, r->stream>>>((int)nv, kin, B.children.p, B.parentOfNode.p, B.parentOfLeaf.p);
        // bottom-up boxes: over the level lists (kept from the last full build for a refit); the fence-and-atomic climb only for a
        // tree the lists cannot take
        const bool fenceKernels = r->env.fenceRefit;
        bool haveLevels = false;
        if (!boxesDone)
        {
            if (refit)
                haveLevels = B.levelsValid && !fenceKernels;
            else if (!fenceKernels)
                BUILD_TRY(treeLevels(r, nv, &haveLevels));
            if (haveLevels)
                refitLevels(r, vin, refLo, refHi);
            else
                k_refit<<<vblocks, 256, 0, r->stream>>>((int)nv, vin, refLo, refHi, B.children.p, B.parentOfNode.p, B.parentOfLeaf.p,
                                                   B.nodeLo.p, B.nodeHi.p, B.flags.p);
        }
        const uint32_t reinsertPasses = (refit || r->reinsertBroken) ? 0u : r->tree.reinsertPasses;
        if (reinsertPasses && nv > 3)
        {
            // parallel reinsertion over the binary tree (k_reinsert_find / _claim / _apply), boxes recomputed after every pass
            const uint32_t slots = 2 * nv - 1, sblocks = (slots + 255) / 256;
            DevBuf<int> target, top;
            DevBuf<float> gain;
            DevBuf<unsigned long long> lock;
            DevBuf<uint32_t> applied;
            BUILD_TRY(target.alloc(slots)); BUILD_TRY(top.alloc(slots)); BUILD_TRY(gain.alloc(slots)); BUILD_TRY(lock.alloc(slots)); BUILD_TRY(applied.alloc(1));
            const ReinsertTree rt = { (int)nv, B.children.p, B.parentOfNode.p, B.parentOfLeaf.p, B.nodeLo.p, B.nodeHi.p, vin, refLo, refHi };
            for (uint32_t pass = 0; pass < reinsertPasses; pass++)
            {
                BUILD_TRY(hipMemsetAsync(lock.p, 0, (size_t)slots * sizeof(unsigned long long), r->stream));
                BUILD_TRY(hipMemsetAsync(applied.p, 0, sizeof(uint32_t), r->stream));
                k_reinsert_find<<<sblocks, 256, 0, r->stream>>>(rt, 1u, 0u, target.p, gain.p, top.p);
                k_reinsert_claim<<<sblocks, 256, 0, r->stream>>>(rt, target.p, gain.p, top.p, lock.p);
                k_reinsert_apply<<<sblocks, 256, 0, r->stream>>>(rt, target.p, gain.p, top.p, lock.p, applied.p);
                // still a tree?  (a knot would hang k_refit: checked BEFORE the boxes are recomputed)
                uint32_t moved = 0, check[3] = { 0, 0, 0 };
                {
                    DevBuf<uint32_t> counts;
                    BUILD_TRY(counts.alloc(3));
                    BUILD_TRY(hipMemsetAsync(counts.p, 0, 3 * sizeof(uint32_t), r->stream));
                    k_tree_check<<<vblocks, 256, 0, r->stream>>>(rt, counts.p);
                    BUILD_TRY(hipMemcpyAsync(check, counts.p, sizeof(check), hipMemcpyDeviceToHost, r->stream));
                    BUILD_TRY(hipMemcpyAsync(&moved, applied.p, sizeof(moved), hipMemcpyDeviceToHost, r->stream));
                    BUILD_TRY(hipStreamSynchronize(r->stream));
                }
                if (r->env.verbose)
                    std::fprintf(stderr, "[ptx] reinsertion pass %u: %u moves; check: %u bad parent links, %u leaves off the root, longest path %u\n", pass,
                                 moved, check[0], check[1], check[2]);
                if (check[0] || check[1])
                {
                    // Not a tree any more (never seen since the path locks were completed, but the moves of a pass race by design):
                    // this handle builds without reinsertion from now on, starting with this tree again.
                    r->reinsertBroken = true;
                    fail(r, PTX_OK, "ptx_build_accel: reinsertion pass %u left %u bad parent links, %u leaves off the root: rebuilt without reinsertion",
                         pass, check[0], check[1]);
                    if (r->env.verbose)
                        std::fprintf(stderr, "[ptx] %s\n", r->error.c_str());
                    BUILD_TRY(hipStreamSynchronize(r->stream));
                    return buildAccel(r, false, keepState);
                }
                haveLevels = false;
                if (!fenceKernels)
                    BUILD_TRY(treeLevels(r, nv, &haveLevels)); // the pass changed the topology
                if (haveLevels)
                    refitLevels(r, vin, refLo, refHi);
                else
                {
                    BUILD_TRY(hipMemsetAsync(B.flags.p, 0, (size_t)nv * 4, r->stream));
                    k_refit<<<vblocks, 256, 0, r->stream>>>((int)nv, vin, refLo, refHi, B.children.p, B.parentOfNode.p, B.parentOfLeaf.p, B.nodeLo.p, B.nodeHi.p,
                                                       B.flags.p);
                }
            }
            BUILD_TRY(hipStreamSynchronize(r->stream)); // (the pass's buffers go out of scope)
        }
        if (r->tree.collapse)
        {
            if (!refit && !B.levelsValid && !fenceKernels) // (PLOC computes its boxes itself: no pass above has asked for the lists yet)
                BUILD_TRY(treeLevels(r, nv, &haveLevels));
            if (B.levelsValid && !fenceKernels)
                for (size_t d = B.levelStart.size() - 1; d-- > 0;)
                {
                    const uint32_t first = B.levelStart[d], count = B.levelStart[d + 1] - first;
                    k_collapse_cost_level<<<(count + 255) / 256, 256, 0, r->stream>>>(first, count, B.levelOrder, vin, refLo, refHi, B.children.p, B.nodeLo.p,
                                                                                     B.nodeHi.p, B.collapseCost.p, B.collapseDecide.p);
                }
            else
            {
                BUILD_TRY(hipMemsetAsync(B.flags.p, 0, (size_t)nv * 4, r->stream)); // (the arrival flags of k_refit: done with)
                k_collapse_cost<<<vblocks, 256, 0, r->stream>>>((int)nv, vin, refLo, refHi, B.children.p, B.parentOfNode.p, B.parentOfLeaf.p, B.nodeLo.p, B.nodeHi.p,
                                                           B.flags.p, B.collapseCost.p, B.collapseDecide.p);
            }
        }
        k_emit<<<vblocks, 256, 0, r->stream>>>((int)nv, vin, refLo, refHi, B.children.p, B.nodeLo.p, B.nodeHi.p, B.triTmp.p,
                                              B.rawNodes.p, r->tris.p, r->pairs.p, r->vertices.p, r->indices.p, r->shadeTris.p,
                                              r->tree.collapse ? B.collapseDecide.p : nullptr, refTri);
        // breadth-first relayout into the compact array (k_relayout_level): the host reads the level's end after each launch
        uint32_t *nextFree = B.oldOf.p + n;
        const uint32_t first[1] = { 0u }, one = 1u;
        BUILD_TRY(hipMemcpyAsync(B.oldOf.p, first, sizeof(first), hipMemcpyHostToDevice, r->stream)); // the root stays node 0
        BUILD_TRY(hipMemcpyAsync(nextFree, &one, sizeof(one), hipMemcpyHostToDevice, r->stream));
        uint32_t lo = 0, hi = 1, levels = 0;
        std::vector<uint32_t> levelStart; // of the breadth-first array, plus its end
        while (lo < hi)
        {
            levelStart.push_back(lo);
            k_relayout_level<<<(hi - lo + 255) / 256, 256, 0, r->stream>>>(lo, hi, B.rawNodes.p, B.oldOf.p, nextFree, r->nodes.p);
            uint32_t end = 0;
            BUILD_TRY(hipMemcpyAsync(&end, nextFree, sizeof(end), hipMemcpyDeviceToHost, r->stream));
            BUILD_TRY(hipStreamSynchronize(r->stream));
            if (end < hi || end > nv - 1)
            {
                B.release();
                return fail(r, PTX_ERROR_DEVICE, "ptx_build_accel: relayout placed %u nodes of at most %u", end, nv - 1);
            }
            lo = hi;
            hi = end;
            levels++;
        }
        r->stats.bvhNodes = hi;
        if (r->tree.layout == 1 && hi > 1)
        {
            // depth-first order (k_subtree_size / _pos / k_place_nodes); scratch: the build's flag and neighbour arrays are done with
            levelStart.push_back(hi);
            uint32_t *size = B.flags.p, *pos = B.vals1.p == vin ? B.vals0.p : B.vals1.p; // (the sorted order sits in the other one)
            for (uint32_t l = levels; l-- > 0;)
                k_subtree_size<<<(levelStart[l + 1] - levelStart[l] + 255) / 256, 256, 0, r->stream>>>(levelStart[l], levelStart[l + 1], r->nodes.p, size);
            for (uint32_t l = 0; l < levels; l++)
                k_subtree_pos<<<(levelStart[l + 1] - levelStart[l] + 255) / 256, 256, 0, r->stream>>>(levelStart[l], levelStart[l + 1], r->nodes.p, size, pos);
            k_place_nodes<<<(hi + 255) / 256, 256, 0, r->stream>>>(hi, r->nodes.p, pos, B.rawNodes.p);
            r->nodes.swap(B.rawNodes); // the emitted nodes are not needed again before the next k_emit, which rewrites them all
        }
        if (r->env.verbose)
            std::fprintf(stderr, "[ptx] relayout: %u of %u emitted nodes are live, %u levels\n", hi, nv - 1, levels);
    }
    if (r->anyNonOpaque && nv) // the any-hit records of the slots k_emit has just written
    {
        BUILD_TRY(r->alphaTris.alloc(n));
        k_alpha_tris<<<vblocks, 256, 0, r->stream>>>(nv, r->tris.p, r->shadeTris.p, makeSceneView(r), r->alphaTexOf.p, r->alphaTex.p, r->alphaTris.p);
    }
    uint32_t revived = 0;
    if (refit)
        BUILD_TRY(hipMemcpyAsync(&revived, &B.sceneBounds.p[7], sizeof(revived), hipMemcpyDeviceToHost, r->stream));
    BUILD_TRY(hipEventRecord(r->evB, r->stream));
    BUILD_TRY(hipStreamSynchronize(r->stream));
    BUILD_TRY(hipGetLastError());
    if (revived) // a triangle the last full build left out has an area now: it is not in the kept topology
        return buildAccel(r, false, keepState);
    float ms = 0.0f;
    (void)hipEventElapsedTime(&ms, r->evA, r->evB);
    r->stats.lastBuildMs = ms;
    if (keepState)
        B.valid = true;
    else
        B.release();
#undef BUILD_TRY
    r->accelReady = true;
    return PTX_OK;
}

static TraceScene makeTraceScene(const PtxRenderer *r);

// The price of the tree just built on the sampled segments (k_sample_tree_cost): mean visits + tests per ray, the tail, and
// the figure the candidates are compared by.  The tail term: a persistent traversal launch ends with its longest ray, and in the
// thin launches of late bounces and of small tile shards that ray IS the launch -- a tree that saves 1 % on the mean and grows
// its longest walks by a third is not cheaper.
struct TreeCost
{
    double mean = 0.0;   // visits + tests per ray, without the top 0.1 % of the rays (robust against the odd ray that skims a surface)
    uint32_t p999 = 0;   // 99.9th percentile
    uint32_t worst = 0;
    double figure() const { return mean + kTreeTailWeight * (double)p999; }
    static constexpr double kTreeTailWeight = 0.02; // a p99.9 five times the mean adds 10 % to the figure
};
constexpr uint32_t kTreeSampleRays = 65536;

static int sampleTreeCost(PtxRenderer *r, DevBuf<float4> &segments, bool drawSegments, TreeCost *cost)
{
    DevBuf<uint32_t> d;
    HIP_TRY(r, d.alloc(kTreeSampleRays));
    HIP_TRY(r, segments.alloc(2 * (size_t)kTreeSampleRays));
    const TraceScene sc = makeTraceScene(r);
    if (drawSegments)
        k_sample_segments<<<kTreeSampleRays / kBlock, kBlock, 0, r->stream>>>(sc, kTreeSampleRays, segments.p);
    if (r->anyNonOpaque)
        k_sample_tree_cost<true><<<kTreeSampleRays / kBlock, kBlock, 0, r->stream>>>(sc, segments.p, kTreeSampleRays, r->spill.p, d.p);
    else
        k_sample_tree_cost<false><<<kTreeSampleRays / kBlock, kBlock, 0, r->stream>>>(sc, segments.p, kTreeSampleRays, r->spill.p, d.p);
    std::vector<uint32_t> h(kTreeSampleRays);
    HIP_TRY(r, hipMemcpyAsync(h.data(), d.p, kTreeSampleRays * sizeof(uint32_t), hipMemcpyDeviceToHost, r->stream));
    HIP_TRY(r, hipStreamSynchronize(r->stream));
    HIP_TRY(r, hipGetLastError());
    std::sort(h.begin(), h.end());
    const uint32_t kept = kTreeSampleRays - kTreeSampleRays / 1000;
    unsigned long long sum = 0;
    for (uint32_t k = 0; k < kept; k++)
        sum += h[k];
    cost->mean = (double)sum / kept;
    cost->p999 = h[kept - 1];
    cost->worst = h.back();
    return PTX_OK;
}

static int buildBestTree(PtxRenderer *r)
{
    if (r && r->sceneOwner)
        return fail(r, PTX_ERROR_INVALID_ARGUMENT, "ptx_build_accel: this renderer shares another renderer's scene (ptx_share_scene)");
    if (!r || !r->sceneReady)
        return fail(r, PTX_ERROR_NOT_READY, "ptx_build_accel: no scene uploaded");
    quiesceSharers(r);
    r->sceneEpoch++; // schedules learnt on the old tree's scene are not this one's (ptx_scene_upload without a build in between cannot render)
    // Which tree?  Build a few candidates, price each on the same sampled surface-to-surface segments, keep the cheapest (its
    // buffers are swapped aside while the others are built; lastBuildMs is the time of everything).  Parameters given in the
    // environment, the Karras builder and small scenes skip the comparison; the per-frame rebuilds of an animation use the
    // parameters chosen here.
    struct Candidate { uint32_t radius; float shape; bool cubic; };
    // (round 4, on the cosine-ray sampler, twelve settings tried per stand-in: these seven hold every scene's best or come within
    // 0.3 % of it -- chess_like (64, 0.25), atrium_like (4, 0.25), street_like (8, 1, cubic), temple_like (16, 0.25); the spread
    // between best and worst setting of a scene is 5-10 %)
    static const Candidate kTreeCandidates[] = { { 8u, 0.0f, false }, { 16u, 0.0f, false }, { 16u, 0.25f, false }, { 32u, 1.0f, false }, { 8u, 1.0f, true },
                                                 { 64u, 0.25f, false }, { 4u, 0.25f, false } };
    constexpr uint32_t kCandidates = sizeof(kTreeCandidates) / sizeof(kTreeCandidates[0]);
    if (!r->usePloc || r->env.plocFixed || r->triCount < 4096u)
    {
        // ONE tree: it gets the full number of reinsertion passes at once
        const TreeParams keep = r->tree;
        r->tree.reinsertPasses = std::max(keep.reinsertPasses, keep.reinsertFinal);
        const int rc = buildAccel(r, false, false);
        r->tree = keep;
        return rc;
    }
    DevBuf<BvhNode> bestNodes;
    DevBuf<Tri> bestTris;
    DevBuf<ShadeTri> bestShadeTris;
    DevBuf<AlphaTri> bestAlphaTris;
    DevBuf<float4> segments;
    auto swapTree = [&]() { r->nodes.swap(bestNodes); r->tris.swap(bestTris); r->shadeTris.swap(bestShadeTris); r->alphaTris.swap(bestAlphaTris); };
    TreeCost cost[kCandidates];
    uint64_t bestNodeCount = 0, bestReferences = 0;
    uint32_t bestTreeTris = 0; // (leaf slots: with pre-splitting the candidates can differ -- cubic cells move the cut planes)
    double totalMs = 0.0;
    uint32_t best = 0, built = 0;
    const TreeParams given = r->tree; // what the candidates do not vary (the collapse)
    // While candidates are built the renderer's buffers hold whichever tree was built last and the best one sits in the locals
    // above: nothing may render (or borrow the scene) until the final swap.  A candidate that fails (out of memory, a device
    // error) does not take the scene down with it when an earlier one succeeded: that tree, its parameters and its node count
    // are put back and the build succeeds with it.
    r->accelReady = false;
    for (uint32_t k = 0; k < kCandidates; k++)
    {
        r->tree = given;
        r->tree.plocRadius = kTreeCandidates[k].radius;
        r->tree.plocShape = kTreeCandidates[k].shape;
        r->tree.mortonCubic = kTreeCandidates[k].cubic;
        int rc = buildAccel(r, false, false);
        totalMs += r->stats.lastBuildMs;
        if (rc != PTX_OK || (rc = sampleTreeCost(r, segments, k == 0, &cost[k])) != PTX_OK)
        {
            r->accelReady = false;
            if (k == 0)
                return rc; // no tree at all: the error stands (ptx_last_error has the text)
            if (r->env.verbose)
                std::fprintf(stderr, "[ptx] tree candidate %u failed (%s): keeping candidate %u\n", k, r->error.c_str(), best);
            break;
        }
        built = k + 1;
        if (k == 0 || cost[k].figure() < cost[best].figure())
        {
            best = k;
            bestNodeCount = r->stats.bvhNodes;
            bestReferences = r->stats.treeReferences;
            bestTreeTris = r->treeTris;
            swapTree(); // the renderer's buffers now hold the previous best (or nothing): the next candidate is built over them
        }
    }
    // The winner once more, with the full number of reinsertion passes (the candidates had a few: the ranking is the same with 2
    // as with 64, the cost keeps falling for dozens of passes).  Built over the renderer's buffers -- they hold a loser --, priced
    // on the same rays, and kept only if it is no worse; if it fails, the candidate stands.
    TreeCost finalCost;
    bool haveFinal = false;
    if (given.reinsertFinal > given.reinsertPasses && built > 0)
    {
        r->tree = given;
        r->tree.plocRadius = kTreeCandidates[best].radius;
        r->tree.plocShape = kTreeCandidates[best].shape;
        r->tree.mortonCubic = kTreeCandidates[best].cubic;
        r->tree.reinsertPasses = given.reinsertFinal;
        int rc = buildAccel(r, false, false);
        totalMs += r->stats.lastBuildMs;
        if (rc == PTX_OK && sampleTreeCost(r, segments, false, &finalCost) == PTX_OK && finalCost.figure() <= cost[best].figure())
        {
            haveFinal = true;
            bestNodeCount = r->stats.bvhNodes;
            bestReferences = r->stats.treeReferences;
            bestTreeTris = r->treeTris;
        }
        else if (r->env.verbose)
            std::fprintf(stderr, "[ptx] the fully re-optimised tree was not kept (%s)\n", rc == PTX_OK ? "no cheaper" : r->error.c_str());
        r->accelReady = false;
    }
    if (!haveFinal)
        swapTree();
    r->stats.bvhNodes = bestNodeCount;
    r->stats.treeReferences = bestReferences;
    r->treeTris = bestTreeTris;
    r->tree = given;
    r->tree.plocRadius = kTreeCandidates[best].radius;
    r->tree.plocShape = kTreeCandidates[best].shape;
    r->tree.mortonCubic = kTreeCandidates[best].cubic;
    r->stats.lastBuildMs = totalMs;
    r->accelReady = true;
    if (r->env.verbose)
    {
        std::fprintf(stderr, "[ptx] tree cost on %u sampled rays, mean (lowest 99.9 %%) / p99.9 / max visits + tests per ray:", kTreeSampleRays);
        for (uint32_t k = 0; k < built; k++)
            std::fprintf(stderr, " (radius %u, shape %.2f%s) %.2f / %u / %u%s", kTreeCandidates[k].radius, kTreeCandidates[k].shape,
                         kTreeCandidates[k].cubic ? ", cubic cells" : "", cost[k].mean, cost[k].p999, cost[k].worst, k == best ? " <- kept" : "");
        if (haveFinal)
            std::fprintf(stderr, "; with %u reinsertion passes %.2f / %u / %u", given.reinsertFinal, finalCost.mean, finalCost.p999, finalCost.worst);
        std::fprintf(stderr, "; collapse %s, %u reinsertion passes per candidate; %.1f ms\n", given.collapse ? "cost-driven" : "greedy", given.reinsertPasses, totalMs);
    }
    return PTX_OK;
}

// Renderer.cpp:1750-1754 (+ RecordSkinningCommands :854-890, AccelerationStructure::Update :48-57)
static int updateAnimation(PtxRenderer *r, const PtxTransform *instanceTransforms, uint32_t instanceCount, const PtxTransform *boneTransforms, uint32_t boneCount, uint32_t accelUpdate)
{
    if (!r || accelUpdate > PTX_ACCEL_REBUILD)
        return fail(r, PTX_ERROR_INVALID_ARGUMENT, "ptx_update_animation: bad argument");
    if (r->sceneOwner)
        return fail(r, PTX_ERROR_INVALID_ARGUMENT, "ptx_update_animation: this renderer shares another renderer's scene (ptx_share_scene)");
    if (!r->sceneReady)
        return fail(r, PTX_ERROR_NOT_READY, "ptx_update_animation: no scene uploaded");
    quiesceSharers(r);
    if (instanceTransforms && instanceCount != r->instanceCount)
        return fail(r, PTX_ERROR_INVALID_ARGUMENT, "ptx_update_animation: %u instance transforms for a scene of %u instances", instanceCount,
                    r->instanceCount);
    HIP_TRY(r, hipSetDevice(r->device));
    if (instanceTransforms)
    {
        for (size_t p = 0; p < r->hostPairs.size(); p++)
        {
            DevPair &pr = r->hostPairs[p];
            composeTransform(instanceTransforms[r->pairInstance[p]].m, r->pairMeshTransform[p].m, pr.M);
            inverseLinear(pr.M, pr.Rinv);
        }
        if (!r->hostPairs.empty())
            HIP_TRY(r, hipMemcpyAsync(r->pairs.p, r->hostPairs.data(), r->hostPairs.size() * sizeof(DevPair), hipMemcpyHostToDevice, r->stream));
    }
    if (boneTransforms && r->skinnedCount)
    {
        if (boneCount > r->bones.n)
            HIP_TRY(r, r->bones.alloc(boneCount));
        r->boneCount = boneCount;
        if (boneCount)
            HIP_TRY(r, hipMemcpyAsync(r->bones.p, boneTransforms, (size_t)boneCount * sizeof(PtxTransform), hipMemcpyHostToDevice, r->stream));
        k_skin<<<(r->skinnedCount + 255) / 256, 256, 0, r->stream>>>(r->animatedVertices.p, r->skinSource.p, r->skinnedCount, r->bones.p, boneCount,
                                                                  r->vertices.p + r->staticVertexCount);
    }
    HIP_TRY(r, hipStreamSynchronize(r->stream)); // the caller's arrays may go away
    const bool refit = accelUpdate == PTX_ACCEL_REFIT && r->build.valid && r->accelReady;
    return buildAccel(r, refit, true);
}


// Kernel variant of the uploaded scene: 0 = opaque geometry with the fixed 1x1 textures only, 1 = ray
// differentials + software sampler, 2 = 1 + the any-hit stages (alpha test, decals).
static int kernelMode(const PtxRenderer *r)
{
    const PtxRenderer *s = sceneOf(r);
    return s->anyNonOpaque ? 2 : (s->samplerNeeded ? 1 : 0);
}

static SceneView makeSceneView(const PtxRenderer *r)
{
    const PtxRenderer *s = sceneOf(r);
    SceneView sv;
    sv.shadeTris = s->shadeTris.p;
    sv.vertices = s->vertices.p; sv.indices = s->indices.p; sv.mr = s->mr.p; sv.sg = s->sg.p; sv.phong = s->phong.p;
    sv.pairs = s->pairs.p; sv.dxNormalTextures = s->dxNormalTextures;
    sv.lights = r->lights.p; // the lights come with every launch: each frame in flight has its own
    sv.tex.textures = s->renderTextures.p; sv.tex.textureCount = s->textureCount; sv.tex.texels8 = nullptr; sv.tex.texelsF = s->renderTexels.p;
    sv.tex.srgbLut = nullptr;
    sv.skyKind = s->skyKind;
    return sv;
}

static TraceScene makeTraceScene(const PtxRenderer *r)
{
    const PtxRenderer *s = sceneOf(r);
    TraceScene sc;
    sc.nodes = s->nodes.p; sc.tris = s->tris.p; sc.triCount = s->treeTris;
    sc.alphaTris = s->alphaTris.p; sc.alphaQuads = s->alphaQuads.p;
    return sc;
}

static int ensureSlots(PtxRenderer *r, size_t slots)
{
    if (kernelMode(r) >= 1 && r->diffCapacity < std::max(slots, r->slotCapacity))
    {
        HIP_TRY(r, r->diffs.alloc(3 * std::max(slots, r->slotCapacity)));
        r->diffCapacity = std::max(slots, r->slotCapacity);
    }
    if (kernelMode(r) == 2 && r->decalCapacity < std::max(slots, r->slotCapacity))
    {
        HIP_TRY(r, r->decal.alloc(std::max(slots, r->slotCapacity)));
        HIP_TRY(r, r->decalT.alloc(std::max(slots, r->slotCapacity)));
        r->decalCapacity = std::max(slots, r->slotCapacity);
    }
    if (slots <= r->slotCapacity)
        return PTX_OK;
    HIP_TRY(r, r->slotRad.alloc(slots));
    HIP_TRY(r, r->rayO.alloc(slots)); HIP_TRY(r, r->rayD.alloc(slots)); HIP_TRY(r, r->thr.alloc(slots)); HIP_TRY(r, r->rad.alloc(slots));
    HIP_TRY(r, r->hit.alloc(slots)); HIP_TRY(r, r->shO.alloc(slots)); HIP_TRY(r, r->shD.alloc(slots)); HIP_TRY(r, r->shC.alloc(slots));
    HIP_TRY(r, r->meta.alloc(slots)); HIP_TRY(r, r->hitPair.alloc(slots));
    HIP_TRY(r, r->queue0.alloc(slots)); HIP_TRY(r, r->queue1.alloc(slots)); HIP_TRY(r, r->shadowQueue.alloc(slots)); HIP_TRY(r, r->shadowResult.alloc(slots));
    HIP_TRY(r, r->restartQueue.alloc(slots));
    r->slotCapacity = slots;
    return PTX_OK;
}

__global__ void k_upload_lights(PtxLightsUbo lights, PtxLightsUbo *dst) // 3,120 bytes as a kernel argument: no staging buffer to keep alive
{
    const uint32_t *src = reinterpret_cast<const uint32_t *>(&lights);
    uint32_t *d = reinterpret_cast<uint32_t *>(dst);
    for (uint32_t i = threadIdx.x; i < sizeof(PtxLightsUbo) / 4; i += blockDim.x)
        d[i] = src[i];
}

static int ensureRenderResources(PtxRenderer *r, uint32_t bounces)
{
    if (!r->auxStream)
    {
    {
        // PTX_DEVICE_SINGLE_STREAM (or PTX_SINGLE_STREAM=1 in the environment): the shadow and tail kernels ride on the main stream
        // too -- no overlap inside a frame, one hardware queue per frame in flight instead of two (twice the frames on the same
        // queues; what a rank's thin tile shard of an N-GPU job wants, include/ptx.h)
        if (r->env.singleStream || r->singleStream)
        {
            r->auxStream = r->stream;
            r->auxIsMain = true;
        }
        else
            HIP_TRY(r, createStreamOn(&r->auxStream, r->auxXcds, r->device));
    }
        HIP_TRY(r, r->spillAux.alloc((size_t)kGlobalSpill * kMaxPersistentThreads));
    }
    const size_t want = bounces < (uint32_t)kMaxTimedBounces ? bounces : (uint32_t)kMaxTimedBounces;
    while (r->bounceEvents.size() < want)
    {
        PtxRenderer::BounceEvents e;
        for (hipEvent_t *ev : { &e.t0, &e.t1, &e.t2, &e.x0, &e.x1, &e.x2 })
            HIP_TRY(r, hipEventCreate(ev));
        r->bounceEvents.push_back(e);
    }
    return PTX_OK;
}

// What the kernels of one launch share.
struct RenderPlan
{
    LaunchParams p;
    SceneView sv;
    TraceScene sc;
    int mode;           // kernelMode()
    uint32_t bounces;   // BounceCount
    uint32_t tailBelow; // queues of at most this many paths are finished by k_tail
    uint32_t sortShade; // material-sorted shade queue (scenes that mix material types; PTX_SHADE_SORT=0 / 1 overrides)
    Wavefront wf, wfAux;
};

// One BOUNCE of the wavefront over queue `qin` (length in the counter block, at most `est`), enqueued without waiting
// for the device: every kernel reads its queue length from the counter block (BounceCtl).
//
//   stream     P(b)  closest(b)  [wait aux(b-1)]  shade(b)                      P(b+1) closest(b+1) ...
//   auxStream                                     [wait shade(b)] shadow(b) [tail(b)]
//
// shadow(b) only adds into rad[slot], which shade(b + 1) reads -- not closest(b + 1) -- so it runs beside the next
// traversal; k_tail, where the schedule has one, follows it in stream order (the NEE adds it continues from have
// landed): the last shadow query before the tail needs no event of its own.
// tail: 0 = none, 1 = k_tail takes the queue shade(b) filled if it holds at most pl.tailBelow paths, 2 = takes it whatever
// its length (nothing is enqueued behind this bounce).
static int enqueueBounce(PtxRenderer *r, const RenderPlan &pl, uint32_t b, int qin, uint32_t est, uint32_t skipBelow, int tail)
{
    const bool textured = pl.mode >= 1, alpha = pl.mode == 2;
    hipStream_t S = r->stream, X = r->auxStream;
    PtxRenderer::BounceEvents &ev = r->bounceEvents[(b - 1) % r->bounceEvents.size()];
    const BounceCtl ctl = { b, skipBelow, pl.sortShade };
    const int qout = qin ^ 1;
    k_prologue<<<1, 1, 0, S>>>(pl.wf, qin, ctl);
    HIP_TRY(r, hipEventRecord(ev.t0, S));
    if (alpha)
        k_trace_closest<true><<<traceGridFor(est, r->residentClosest[1]), kBlock, 0, S>>>(pl.sc, pl.wf, qin, ctl);
    else
        k_trace_closest<false><<<traceGridFor(est, r->residentClosest[0]), kBlock, 0, S>>>(pl.sc, pl.wf, qin, ctl);
    HIP_TRY(r, hipEventRecord(ev.t1, S));
    if (b > 1) // shade reads rad[slot]: the previous bounce's shadow adds must have landed
        HIP_TRY(r, hipStreamWaitEvent(S, r->bounceEvents[(b - 2) % r->bounceEvents.size()].x2, 0));
    const uint32_t shadeGrid = gridFor((est + kShadeItems - 1) / kShadeItems);
    if (textured)
        k_shade<true><<<shadeGrid, kBlock, 0, S>>>(pl.p, pl.sv, pl.wf, qin, ctl);
    else
        k_shade<false><<<shadeGrid, kBlock, 0, S>>>(pl.p, pl.sv, pl.wf, qin, ctl);
    HIP_TRY(r, hipEventRecord(ev.t2, S));
    HIP_TRY(r, hipStreamWaitEvent(X, ev.t2, 0));
    HIP_TRY(r, hipEventRecord(ev.x0, X));
    if (alpha)
        k_trace_shadow<true><<<traceGridFor(est, r->residentShadow[1]), kBlock, 0, X>>>(pl.p, pl.sc, pl.wfAux, qout, (int)(b & 1u));
    else
        k_trace_shadow<false><<<traceGridFor(est, r->residentShadow[0]), kBlock, 0, X>>>(pl.p, pl.sc, pl.wfAux, qout, (int)(b & 1u));
    k_apply_shadow<<<gridFor(est, kBlock, 4096u), kBlock, 0, X>>>(pl.p, pl.wfAux, (int)(b & 1u));
    HIP_TRY(r, hipEventRecord(ev.x1, X));
    if (tail)
    {
        // grid-stride loop; the spill region holds kMaxPersistentThreads.  A hinted schedule (tail == 2) hands the tail
        // whatever is left, and the hint is last frame's: a view that keeps four times the paths alive still finds a thread
        // per path (blocks beyond the queue return at once), anything beyond that strides.
        const uint32_t room = tail == 2 ? 4u * pl.tailBelow : pl.tailBelow;
        const uint32_t most = est < room ? est : room;
        const BounceCtl tctl = { b, tail == 2 ? 0xffffffffu : pl.tailBelow, 0u };
        const dim3 grid(gridFor(most, kBlock, kMaxPersistentThreads / kBlock));
        if (pl.mode == 2)
            k_tail<2><<<grid, kBlock, 0, X>>>(pl.p, pl.sv, pl.sc, pl.wfAux, qout, tctl);
        else if (pl.mode == 1)
            k_tail<1><<<grid, kBlock, 0, X>>>(pl.p, pl.sv, pl.sc, pl.wfAux, qout, tctl);
        else
            k_tail<0><<<grid, kBlock, 0, X>>>(pl.p, pl.sv, pl.sc, pl.wfAux, qout, tctl);
    }
    HIP_TRY(r, hipEventRecord(ev.x2, X));
    return PTX_OK;
}

// One ROUND: the slots listed in queue 0 (ACTIVE0 set by the caller, at most `upperBound`) start at bounce 0 of a sample
// and are advanced BounceCount times, or until the queue is short enough for k_tail to finish them in one launch.
//
// With a hint (queue lengths of the previous canonical launch of this shape) the whole round is enqueued at once: the
// bounces that ran as wavefront kernels last time, then k_tail for whatever is left -- results do not depend on who
// finishes a path, only the time does, and collectRender drops a hint that turned out wrong.  No kernel is launched just
// to find its queue empty, and the host does not wait for the device.
// Without one (first launch of a shape, rounds of a multi-sample launch) the round is driven bounce by bounce: the host
// reads the counter block after every shade kernel and decides -- which is also how the hint is learned.
static int enqueueRound(PtxRenderer *r, const RenderPlan &pl, uint32_t upperBound, const uint32_t *hint)
{
    hipStream_t S = r->stream;
    uint32_t last = 0;
    int qin = 0;
    if (hint)
    {
        last = pl.bounces;
        if (pl.tailBelow)
            for (uint32_t b = 1; b < pl.bounces && b < (uint32_t)kMaxTimedBounces; b++)
                if (hint[b + 1] <= pl.tailBelow) // k_tail took the queue of bounce b (or nothing was left of it)
                {
                    last = b;
                    break;
                }
        for (uint32_t b = 1; b <= last; b++)
        {
            uint32_t est = upperBound; // exact for the first bounce; later ones shrink
            if (b > 1 && b <= (uint32_t)kMaxTimedBounces)
            {
                const uint64_t e = (uint64_t)hint[b] + hint[b] / 4 + 4096;
                est = e < upperBound ? (uint32_t)e : upperBound;
            }
            const int rcb = enqueueBounce(r, pl, b, qin, est, 0u, (b == last && last < pl.bounces) ? 2 : 0);
            if (rcb != PTX_OK)
                return rcb;
            qin ^= 1;
        }
    }
    else
    {
        uint32_t est = upperBound;
        for (uint32_t b = 1; b <= pl.bounces; b++)
        {
            const int rcb = enqueueBounce(r, pl, b, qin, est, 0u, 0);
            if (rcb != PTX_OK)
                return rcb;
            last = b;
            qin ^= 1;
            HIP_TRY(r, hipMemcpyAsync(r->hostCounters, r->counters.p, C_COUNT * sizeof(uint32_t), hipMemcpyDeviceToHost, S));
            HIP_TRY(r, hipStreamSynchronize(S));
            if (r->hostCounters[C_OVERFLOW])
                return fail(r, PTX_ERROR_DEVICE, "ptx_render: traversal stack overflow (tree deeper than %d levels)", kLdsStack + kGlobalSpill);
            est = r->hostCounters[qin ? C_ACTIVE1 : C_ACTIVE0];
            if (est == 0u)
                break;
            if (b < pl.bounces && est <= pl.tailBelow)
            {
                // the queue is short: k_tail finishes it, behind the shadow kernel of this bounce on its stream
                PtxRenderer::BounceEvents &ev = r->bounceEvents[(b - 1) % r->bounceEvents.size()];
                const BounceCtl tctl = { b, 0xffffffffu, 0u };
                const dim3 grid(gridFor(est, kBlock, kMaxPersistentThreads / kBlock));
                if (pl.mode == 2)
                    k_tail<2><<<grid, kBlock, 0, r->auxStream>>>(pl.p, pl.sv, pl.sc, pl.wfAux, qin, tctl);
                else if (pl.mode == 1)
                    k_tail<1><<<grid, kBlock, 0, r->auxStream>>>(pl.p, pl.sv, pl.sc, pl.wfAux, qin, tctl);
                else
                    k_tail<0><<<grid, kBlock, 0, r->auxStream>>>(pl.p, pl.sv, pl.sc, pl.wfAux, qin, tctl);
                HIP_TRY(r, hipEventRecord(ev.x2, r->auxStream)); // re-recorded behind the tail: what the stream waits for below
                break;
            }
        }
    }
    if (last)
        HIP_TRY(r, hipStreamWaitEvent(S, r->bounceEvents[(last - 1) % r->bounceEvents.size()].x2, 0));
    HIP_TRY(r, hipGetLastError());
    return PTX_OK;
}

// Statistics and errors of the last wavefront launch, once the device is done with it (blocks until then).
static int collectRender(PtxRenderer *r)
{
    if (!r->statsPending)
        return PTX_OK;
    r->statsPending = false;
    HIP_TRY(r, hipEventSynchronize(r->evB));
    const uint32_t *h = r->hostCounters;
    if (h[C_OVERFLOW])
        return fail(r, PTX_ERROR_DEVICE, "ptx_render: traversal stack overflow (tree deeper than %d levels)", kLdsStack + kGlobalSpill);
    if (h[C_OVERFLOW + 1])
        return fail(r, PTX_ERROR_DEVICE, "ptx_render: %u paths never produced a finite sample in %u attempts", h[C_OVERFLOW + 1], kMaxSampleRetries);
    unsigned long long waveSegments = 0;
    std::memcpy(&waveSegments, &h[C_WAVE_SEGMENTS], sizeof(waveSegments));
    waveSegments -= r->pendingDeadSlots; // k_prologue counted the whole first queue
    r->stats.segments = waveSegments + h[C_SEGMENTS];
    r->stats.tracedRays = waveSegments;
    r->stats.shadowRays = h[C_HITS];
    r->stats.pathSamples = h[C_SAMPLES];
    r->stats.retries = h[C_RETRIES];
    // kernel times of the bounces that ran (the others returned at once), from the events around every launch
    const uint32_t timed = r->pendingBounces < (uint32_t)kMaxTimedBounces ? r->pendingBounces : (uint32_t)kMaxTimedBounces;
    const uint32_t tailPaths = h[C_TAIL_PATHS], tailBounce = h[C_TAIL_PATHS + 1]; // k_tail took the queue shade(tailBounce) filled
    for (uint32_t b = 1; b <= timed; b++)
    {
        const PtxRenderer::BounceEvents &ev = r->bounceEvents[b - 1];
        const uint32_t active = h[C_BOUNCE_ACTIVE + b];
        const bool ran = active != 0u && !(tailPaths && b > tailBounce);
        if (!ran)
            continue;
        float closestMs = 0.0f, shadeMs = 0.0f, shadowMs = 0.0f, tailMs = 0.0f;
        (void)hipEventElapsedTime(&closestMs, ev.t0, ev.t1);
        (void)hipEventElapsedTime(&shadeMs, ev.t1, ev.t2);
        (void)hipEventElapsedTime(&shadowMs, ev.x0, ev.x1);
        r->stats.lastTraceMs += closestMs;
        r->stats.lastShadeMs += shadeMs;
        r->stats.lastShadowMs += shadowMs;
        r->stats.traceLaunches += 2;
        if (tailPaths && tailBounce == b)
        {
            (void)hipEventElapsedTime(&tailMs, ev.x1, ev.x2);
            r->stats.lastTailMs += tailMs;
        }
        if (r->pendingVerbose)
        {
            fprintf(stderr, "[ptx] bounce %u: %u rays closest %.3f ms (%.2f Grays/s) | shade (incl. wait for the previous shadow kernel) %.3f ms | shadow %.3f ms\n",
                    b, active, closestMs, active / closestMs / 1e6, shadeMs, shadowMs);
            if (tailMs > 0.0f)
                fprintf(stderr, "[ptx] tail: %u paths, %u segments, %.3f ms\n", tailPaths, h[C_SEGMENTS], tailMs);
        }
    }
    if (r->pendingVerbose && h[C_RETRIES])
        fprintf(stderr, "[ptx] %u NaN / Inf sample restarts\n", h[C_RETRIES]);
    // grid / schedule hints for the next launch of this shape.  The queue k_tail took over is part of them (a truncated
    // schedule has no prologue behind its last bounce to record it); a tail that had to take more than its threshold
    // means the hints were off: forget them, the next launch runs the full schedule and learns again.
    r->hintActive.assign(h + C_BOUNCE_ACTIVE, h + C_BOUNCE_ACTIVE + kMaxTimedBounces + 1);
    if (tailPaths && tailBounce + 1 <= (uint32_t)kMaxTimedBounces)
        r->hintActive[tailBounce + 1] = tailPaths;
    r->hintSlots = tailPaths > r->pendingTailBelow ? 0u : r->pendingSlots;
    r->hintBounces = r->pendingBounces;
    r->hintEpoch = r->pendingEpoch;
    return PTX_OK;
}

static int renderImpl(PtxRenderer *r, const PtxRaygenUniformData *uniform, const PtxLightsUbo *lights, uint32_t firstFrame, uint32_t frames)
{
    if (!r || !uniform || !lights)
        return fail(r, PTX_ERROR_INVALID_ARGUMENT, "ptx_render: null argument");
    if (!sceneUsable(r) || !imagePtr(r))
        return fail(r, PTX_ERROR_NOT_READY, "ptx_render: need ptx_scene_upload (or ptx_share_scene), ptx_build_accel and ptx_resize first");
    if (uniform->SampleCount == 0 || uniform->SampleCount > 0xffffu || uniform->BounceCount > 0xffffu || frames == 0)
        return fail(r, PTX_ERROR_INVALID_ARGUMENT, "ptx_render: SampleCount must be in [1, 65535], BounceCount <= 65535");
    if (lights->LightCount > PTX_MAX_LIGHT_COUNT)
        return fail(r, PTX_ERROR_INVALID_ARGUMENT, "ptx_render: LightCount %u exceeds MaxLightCount", lights->LightCount);
    HIP_TRY(r, hipSetDevice(r->device));
    {
        const int rcPrev = collectRender(r); // statistics / errors of the previous launch; the counter block is reused below
        if (rcPrev != PTX_OK)
            return rcPrev;
    }

    const LaunchParams p = makeParams(r, uniform, firstFrame, frames);
    if ((uint64_t)p.slotsPerFrame * frames > 0x7fffffffull)
        return fail(r, PTX_ERROR_INVALID_ARGUMENT, "ptx_render: too many path slots in one batch");
    const int rc = ensureSlots(r, p.numSlots);
    if (rc != PTX_OK)
        return rc;

    k_upload_lights<<<1, 256, 0, r->stream>>>(*lights, r->lights.p);
    HIP_TRY(r, hipMemsetAsync(r->counters.p, 0, C_COUNT * sizeof(uint32_t), r->stream));

    RenderPlan pl;
    pl.p = p;
    pl.sv = makeSceneView(r);
    pl.mode = kernelMode(r);
    pl.sc = makeTraceScene(r);
    pl.bounces = uniform->BounceCount;
    const SceneView &sv = pl.sv;
    const TraceScene &sc = pl.sc;
    const int mode = pl.mode;

    r->stats.pathSamples = r->stats.segments = r->stats.shadowRays = r->stats.retries = 0;
    r->stats.traceLaunches = 0;
    r->stats.tracedRays = 0;
    r->stats.lastTraceMs = 0.0;
    r->stats.lastShadeMs = r->stats.lastShadowMs = r->stats.lastTailMs = 0.0;
    HIP_TRY(r, hipEventRecord(r->evA, r->stream));

    if (p.numSlots == 0)
    {
        HIP_TRY(r, hipEventRecord(r->evB, r->stream));
        return PTX_OK;
    }

    if (uniform->BounceCount == 0 && r->backend != PTX_BACKEND_MEGAKERNEL)
    {
        // raygen.rgen:62: the bounce loop never runs, every sample ends with radiance 0 -- nothing is generated, traced or
        // shaded (the wavefront kernels test the bounce limit only AFTER a bounce); the image still gets its alpha
        HIP_TRY(r, hipMemsetAsync(r->slotRad.p, 0, (size_t)p.numSlots * sizeof(float4), r->stream));
        k_accumulate<<<gridFor((size_t)p.slotsPerFrame * p.framesPerWave), kBlock, 0, r->stream>>>(p, r->slotRad.p, accumTarget(r), nullptr, r->boundShard ? 1u : 0u);
        HIP_TRY(r, hipEventRecord(r->evB, r->stream));
        HIP_TRY(r, hipGetLastError());
        r->stats.pathSamples = (uint64_t)p.ownedPixels * frames * uniform->SampleCount;
        return PTX_OK;
    }

    if (r->backend == PTX_BACKEND_MEGAKERNEL)
    {
        const dim3 grid((p.numSlots + kBlock - 1) / kBlock);
        if (mode == 2)
            k_megakernel<2><<<grid, kBlock, 0, r->stream>>>(p, sv, sc, r->slotRad.p, r->counters.p);
        else if (mode == 1)
            k_megakernel<1><<<grid, kBlock, 0, r->stream>>>(p, sv, sc, r->slotRad.p, r->counters.p);
        else
            k_megakernel<0><<<grid, kBlock, 0, r->stream>>>(p, sv, sc, r->slotRad.p, r->counters.p);
        k_accumulate<<<gridFor((size_t)p.slotsPerFrame * p.framesPerWave), kBlock, 0, r->stream>>>(p, r->slotRad.p, accumTarget(r), nullptr, r->boundShard ? 1u : 0u);
        HIP_TRY(r, hipMemcpyAsync(r->hostCounters, r->counters.p, C_COUNT * sizeof(uint32_t), hipMemcpyDeviceToHost, r->stream));
        HIP_TRY(r, hipEventRecord(r->evB, r->stream));
        HIP_TRY(r, hipStreamSynchronize(r->stream));
        HIP_TRY(r, hipGetLastError());
        if (r->hostCounters[C_OVERFLOW])
            return fail(r, PTX_ERROR_DEVICE, "ptx_render: traversal stack overflow in the megakernel (depth > %d)", kLdsStackMega);
        if (r->hostCounters[C_OVERFLOW + 1])
            return fail(r, PTX_ERROR_DEVICE, "ptx_render: %u paths never produced a finite sample in %u attempts", r->hostCounters[C_OVERFLOW + 1],
                        kMaxSampleRetries);
        r->stats.segments = r->hostCounters[C_SEGMENTS];
        r->stats.shadowRays = r->hostCounters[C_HITS];
        r->stats.pathSamples = r->hostCounters[C_SAMPLES];
        r->stats.retries = r->hostCounters[C_RETRIES];
        return PTX_OK;
    }

    // ---- wavefront.  The whole launch is enqueued without waiting for the device (enqueueRound): the kernels take their
    // queue lengths from the counter block, k_tail decides for itself when to take a queue over, and the rare NaN / Inf
    // restarts of a canonical launch are finished on the device too (k_finish_restarts).  The host reads ONE counter
    // block per launch, after the fact (collectRender: statistics, errors, grid hints).  A step of the benchmark used
    // to carry 26 host round trips (0.33 ms of idle GPU per 10 ms step, 6 % of a 1/8 tile shard's step).
    //
    // Measured and dropped along the way: sub-batches of one call as interleaved state machines
    // (PTX_BATCHES: 1 -> 1204, 2 -> 1017, 3 -> 1006, 4 -> 733 Msamples/s -- they pass through their throughput- and
    // latency-bound phases in lockstep), staggered sub-batches (the tail kernel starves beside full-size kernels), ONE
    // traversal launch per bounce carrying closest(b + 1) and shadow(b) (11.4 vs 11.1 ms), a one-entry software pipeline
    // in k_shade (3.50 vs 3.44 ms).
    {
        const int rcr = ensureRenderResources(r, pl.bounces);
        if (rcr != PTX_OK)
            return rcr;
    }
    pl.sortShade = (sceneOf(r)->mixedMaterialTypes || sceneOf(r)->mixedTextured) ? 1u : 0u;
    if (r->env.shadeSort >= 0)
        pl.sortShade = (uint32_t)r->env.shadeSort;
    // measured with 16 hardware queues (chess_like, ms per step at 25 / 50 / 75 / 100 / 200 / 400 K live paths): whole frame 8.04 / 7.82 /
    // 7.85 / 7.80 / 8.14 / 8.13, a rank's tile shard of 8: 1.44 / 1.44 / 1.39 / 1.39 / 1.54 / 1.55, of 4: 2.29 / 2.24 / 2.24 / 2.33 / 2.34 /
    // 2.77, of 2: 3.93 / 3.89 / 3.91 / 3.98 / 4.06 / 4.41; the other scenes are flat from 50 K to 200 K
    pl.tailBelow = 75000;
    if (r->env.tailThreshold >= 0)
        pl.tailBelow = (uint32_t)r->env.tailThreshold;
    Wavefront &wf = pl.wf;
    wf.rayO = r->rayO.p; wf.rayD = r->rayD.p; wf.thr = r->thr.p; wf.rad = r->rad.p;
    wf.meta = r->meta.p; wf.hit = r->hit.p; wf.hitPair = r->hitPair.p;
    wf.shO = r->shO.p; wf.shD = r->shD.p; wf.shC = r->shC.p; wf.slotRad = r->slotRad.p;
    wf.queue[0] = r->queue0.p; wf.queue[1] = r->queue1.p; wf.shadowQueue = r->shadowQueue.p; wf.shadowResult = r->shadowResult.p;
    wf.restartQueue = r->restartQueue.p;
    for (int k = 0; k < 3; k++)
        wf.diff[k] = mode >= 1 ? r->diffs.p + (size_t)k * r->diffCapacity : nullptr;
    wf.decal = mode == 2 ? r->decal.p : nullptr;
    wf.decalT = mode == 2 ? r->decalT.p : nullptr;
    wf.counters = r->counters.p;
    wf.spill = r->spill.p;
    pl.wfAux = wf;
    pl.wfAux.spill = r->spillAux.p;

    const bool canonical = uniform->SampleCount == 1;
    // the learnt schedule belongs to (shape of the launch, scene it was learnt on); a camera or light change inside one scene
    // keeps it -- a hint that is off costs time, never results, and the tail's grid leaves room for that (enqueueBounce)
    const uint32_t *hint = (canonical && r->hintSlots == p.numSlots && r->hintBounces == pl.bounces && r->hintEpoch == sceneOf(r)->sceneEpoch && !r->hintActive.empty())
                               ? r->hintActive.data() : nullptr;
    k_generate<<<gridFor(p.numSlots), kBlock, 0, r->stream>>>(p, wf);
    HIP_TRY(r, hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(&r->counters.p[C_ACTIVE0]), (int)p.numSlots, 1, r->stream));
    int rcq = enqueueRound(r, pl, p.numSlots, hint);
    if (rcq != PTX_OK)
        return rcq;
    if (canonical)
    {
        const dim3 grid(64);
        if (mode == 2)
            k_finish_restarts<2><<<grid, kBlock, 0, r->stream>>>(p, sv, sc, wf);
        else if (mode == 1)
            k_finish_restarts<1><<<grid, kBlock, 0, r->stream>>>(p, sv, sc, wf);
        else
            k_finish_restarts<0><<<grid, kBlock, 0, r->stream>>>(p, sv, sc, wf);
    }
    else
    {
        // multi-sample launch: every slot comes back through the restart queue once per extra sample (and per NaN / Inf
        // restart, raygen.rgen:99-112), one round each; a path that never yields a finite sample would go round for ever
        // (it hangs the GPU in the reference): give up instead
        const uint64_t maxRounds = (uint64_t)uniform->SampleCount * 64 + 64;
        for (uint64_t round = 1;; round++)
        {
            HIP_TRY(r, hipMemcpyAsync(r->hostCounters, r->counters.p, C_COUNT * sizeof(uint32_t), hipMemcpyDeviceToHost, r->stream));
            HIP_TRY(r, hipStreamSynchronize(r->stream));
            if (r->hostCounters[C_OVERFLOW])
                return fail(r, PTX_ERROR_DEVICE, "ptx_render: traversal stack overflow (tree deeper than %d levels)", kLdsStack + kGlobalSpill);
            const uint32_t restarts = r->hostCounters[C_RESTART];
            if (!restarts)
                break;
            if (round > maxRounds)
                return fail(r, PTX_ERROR_DEVICE, "ptx_render: %u paths still active after %llu rounds", restarts, (unsigned long long)maxRounds);
            k_restart<<<gridFor(restarts), kBlock, 0, r->stream>>>(p, wf, restarts); // next primary ray of every re-queued slot
            HIP_TRY(r, hipMemcpyAsync(wf.queue[0], wf.restartQueue, (size_t)restarts * sizeof(uint32_t), hipMemcpyDeviceToDevice, r->stream));
            HIP_TRY(r, hipMemsetAsync(&r->counters.p[C_RESTART], 0, sizeof(uint32_t), r->stream));
            HIP_TRY(r, hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(&r->counters.p[C_ACTIVE0]), (int)restarts, 1, r->stream));
            rcq = enqueueRound(r, pl, restarts, nullptr);
            if (rcq != PTX_OK)
                return rcq;
        }
    }
    k_accumulate<<<gridFor((size_t)p.slotsPerFrame * p.framesPerWave), kBlock, 0, r->stream>>>(p, r->slotRad.p, accumTarget(r), nullptr, r->boundShard ? 1u : 0u);
    HIP_TRY(r, hipMemcpyAsync(r->hostCounters, r->counters.p, C_COUNT * sizeof(uint32_t), hipMemcpyDeviceToHost, r->stream));
    HIP_TRY(r, hipEventRecord(r->evB, r->stream));
    HIP_TRY(r, hipGetLastError());
    r->statsPending = true;
    r->pendingBounces = pl.bounces;
    r->pendingTailBelow = pl.tailBelow;
    r->pendingSlots = canonical ? p.numSlots : 0u;
    r->pendingEpoch = sceneOf(r)->sceneEpoch;
    r->pendingDeadSlots = p.numSlots - p.ownedPixels * frames;
    r->pendingVerbose = r->env.verbose;
    return PTX_OK;
}

// Device alias of a page-locked host frame (hipHostMalloc / hipHostRegister memory), looked up once per (pointer, size): the
// owner of a gathered frame passes the same few buffers step after step.  nullptr: the device cannot address the buffer.
static float4 *hostFrameAlias(PtxRenderer *r, const void *pinnedHost, size_t bytes)
{
    if (r->hostAlias && r->hostAliasOf == pinnedHost && r->hostAliasBytes == bytes)
        return r->hostAlias;
    void *dp = nullptr;
    if (hipHostGetDevicePointer(&dp, const_cast<void *>(pinnedHost), 0) != hipSuccess || !dp)
    {
        (void)hipGetLastError();
        return nullptr;
    }
    r->hostAliasOf = pinnedHost;
    r->hostAliasBytes = bytes;
    r->hostAlias = static_cast<float4 *>(dp);
    return r->hostAlias;
}

// Read-back that overlaps the next launches: a device-to-device snapshot of the image on the render stream (33 MB at
// 1080p: ~20 us), then the PCIe copy on a second stream while the render stream goes on.  The reference reads its
// output back the same way, a frame late (OutputSaver.cpp:120-199).
static int readbackBegin(PtxRenderer *r, float *pinnedHost, size_t bytes)
{
    if (!r || !pinnedHost || !imagePtr(r) || bytes != (size_t)r->width * r->height * sizeof(float4))
        return fail(r, PTX_ERROR_INVALID_ARGUMENT, "ptx_readback_begin: buffer must be width*height*16 bytes");
    if (r->boundShard)
        return frameIsElsewhere(r, "ptx_readback_begin");
    HIP_TRY(r, hipSetDevice(r->device));
    if (!r->evSnapshot)
    {
        HIP_TRY(r, hipEventCreateWithFlags(&r->evSnapshot, hipEventDisableTiming));
        // The copy may be a kernel storing to host memory (k_copy_out): the event the host waits on must release those stores to
        // the system scope, also for page-locked memory that is not host-coherent.
        HIP_TRY(r, hipEventCreateWithFlags(&r->evCopied, hipEventDisableTiming | hipEventReleaseToSystem));
    }
    // The copy to the host rides on the renderer's auxiliary stream -- idle once the frame's shadow and tail kernels are done,
    // and not needed again before this renderer's next frame -- instead of a third stream per frame in flight: the streams of a
    // process share GPU_MAX_HW_QUEUES hardware queues, and streams on one queue run one after the other.
    if (!r->auxStream && !r->copyStream)
        HIP_TRY(r, hipStreamCreateWithFlags(&r->copyStream, hipStreamNonBlocking));
    const hipStream_t copyOn = r->auxStream ? r->auxStream : r->copyStream;
    HIP_TRY(r, r->staging.alloc((size_t)r->width * r->height));
    if (r->copyInFlight) // the previous copy still reads the staging image
        HIP_TRY(r, hipStreamWaitEvent(r->stream, r->evCopied, 0));
    // the snapshot by a copy KERNEL (33 MB at the memory's rate: ~20 us), not hipMemcpyAsync: the runtime's device-to-device copy
    // took 0.5 ms per 1080p image and the copies of the frames in flight queue behind one another -- a floor of 0.5 ms per step
    // under every renderer that reads back, half the step of a 1 / 8 tile shard (tools/experiments/gather_cost.sh, round 5)
    if (r->env.snapshotMemcpy)
        HIP_TRY(r, hipMemcpyAsync(r->staging.p, imagePtr(r), bytes, hipMemcpyDeviceToDevice, r->stream));
    else
    {
        k_copy_out<<<1024, kBlock, 0, r->stream>>>(imagePtr(r), r->staging.p, (uint32_t)(bytes / sizeof(float4)));
        HIP_TRY(r, hipGetLastError()); // (a failed launch would hand the host a stale staging image)
    }
    HIP_TRY(r, hipEventRecord(r->evSnapshot, r->stream));
    HIP_TRY(r, hipStreamWaitEvent(copyOn, r->evSnapshot, 0));
    // The snapshot leaves through ONE workgroup writing to the page-locked buffer (posted writes over PCIe, 33 MB in a few ms)
    // rather than through hipMemcpyAsync: a DMA burst at the link's full rate delays the completion signals and packet fetches
    // of every other frame in flight for its 1.2 ms -- measured on chess_like with 8 frames in flight: no read-back 2,600
    // Msamples/s, hipMemcpyAsync (SDMA) 2,416 / 2,422, copy kernel with 256 / 64 / 8 / 4 / 2 / 1 workgroups 2,359 / 2,395 / 2,445
    // / 2,441 / 2,465 / 2,483-2,494.  Host memory the device cannot address (not page-locked) takes the runtime's copy.
    float4 *const hostOnDevice = hostFrameAlias(r, pinnedHost, bytes);
    if (hostOnDevice)
    {
        // ... for a whole frame on one GPU.  Rank 0 of an N-GPU job renders 1 / N of the frame per step and still reads ALL of it
        // back: there the link, not the rendering, sets the pace, and one workgroup's 8 GB/s (4.1 ms per 1080p image) made a 1 / 8
        // step of chess_like 2.2 ms instead of 0.96 (tools/experiments/gather_cost.sh) -- more workgroups with more ranks.
        const uint32_t groups = r->env.copyGroups ? r->env.copyGroups : (r->shard.worldSize > 1 ? std::min(16u, 2u * r->shard.worldSize) : 1u);
        k_copy_out<<<groups, kBlock, 0, copyOn>>>(r->staging.p, hostOnDevice, (uint32_t)(bytes / sizeof(float4)));
        HIP_TRY(r, hipGetLastError());
    }
    else
        HIP_TRY(r, hipMemcpyAsync(pinnedHost, r->staging.p, bytes, hipMemcpyDeviceToHost, copyOn));
    HIP_TRY(r, hipEventRecord(r->evCopied, copyOn));
    r->copyInFlight = true;
    return PTX_OK;
}

static int unpackShard(PtxRenderer *r, uint32_t rank, const void *devSrc, float *pinnedHost = nullptr, size_t hostBytes = 0)
{
    if (!r || !devSrc || !imagePtr(r) || rank >= r->shard.worldSize)
        return fail(r, PTX_ERROR_INVALID_ARGUMENT, "ptx_unpack_shard: bad argument");
    float4 *hostOnDevice = nullptr;
    if (pinnedHost)
    {
        if (hostBytes != (size_t)r->width * r->height * sizeof(float4))
            return fail(r, PTX_ERROR_INVALID_ARGUMENT, "ptx_unpack_shard_host: buffer must be width*height*16 bytes");
        HIP_TRY(r, hipSetDevice(r->device));
        hostOnDevice = hostFrameAlias(r, pinnedHost, hostBytes);
        if (!hostOnDevice)
            return fail(r, PTX_ERROR_INVALID_ARGUMENT, "ptx_unpack_shard_host: the buffer is not page-locked memory the device can address");
        if (!r->evSnapshot)
        {
            HIP_TRY(r, hipEventCreateWithFlags(&r->evSnapshot, hipEventDisableTiming));
            HIP_TRY(r, hipEventCreateWithFlags(&r->evCopied, hipEventDisableTiming | hipEventReleaseToSystem));
        }
    }
    PtxRenderer tmp;
    tmp.width = r->width;
    tmp.height = r->height;
    tmp.shard = r->shard;
    tmp.shard.rank = rank;
    const LaunchParams p = makeParams(&tmp, nullptr, 0, 1);
    if (p.slotsPerFrame)
    {
        // towards the host a FEW workgroups (PTX_COPY_GROUPS; default 4 per shard): what the link carries in a burst, the command
        // processor's own traffic over it waits behind -- 2 / 4 / 8 / 16 / 64 / 2,048 workgroups: 1.45 / 1.38 / 1.42 / 1.47 / 1.50 /
        // 1.51 ms per 1 / 8 step of chess_like (profiles/r05_unpack_groups.txt)
        uint32_t grid = gridFor(p.slotsPerFrame);
        if (hostOnDevice)
            grid = std::min(grid, r->env.copyGroups ? r->env.copyGroups : 4u);
        k_unpack_shard<<<grid, kBlock, 0, r->stream>>>(p, static_cast<const float4 *>(devSrc), imagePtr(r), hostOnDevice);
    }
    HIP_TRY(r, hipGetLastError());
    if (hostOnDevice) // ptx_readback_end waits for the LAST of these: the stores of every unpack before it are released with it
    {
        HIP_TRY(r, hipEventRecord(r->evCopied, r->stream));
        r->copyInFlight = true;
    }
    return PTX_OK;
}

// The whole gathered frame in ONE launch (k_gather_frame): `devSrc` holds the shards of ranks 0 .. worldSize-1, `strideBytes`
// apart, each in ptx_pack_shard's layout.  Targets: the device image (toDeviceImage), the host's page-locked frame, or both --
// a rank that only hands the frame to the host (OutputSaver's role, OutputSaver.cpp:120-199) never rewrites its device image.
static int unpackShards(PtxRenderer *r, const void *devSrc, size_t strideBytes, int toDeviceImage, float *pinnedHost, size_t hostBytes)
{
    if (!r || !devSrc || !r->width || (!toDeviceImage && !pinnedHost))
        return fail(r, PTX_ERROR_INVALID_ARGUMENT, "ptx_unpack_shards: need the gathered shards and at least one target");
    if (toDeviceImage && !imagePtr(r))
        return fail(r, PTX_ERROR_NOT_READY, "ptx_unpack_shards: no accumulation image (call ptx_resize)");
    size_t largest = 0;
    for (uint32_t k = 0; k < r->shard.worldSize; k++)
        largest = std::max(largest, ptx_shard_bytes(r, k));
    if (strideBytes < largest || strideBytes % sizeof(float4) != 0 || strideBytes / sizeof(float4) > 0xffffffffull)
        return fail(r, PTX_ERROR_INVALID_ARGUMENT, "ptx_unpack_shards: the stride must be a multiple of 16 bytes and at least the largest shard (%zu bytes)", largest);
    HIP_TRY(r, hipSetDevice(r->device));
    float4 *hostOnDevice = nullptr;
    if (pinnedHost)
    {
        if (hostBytes != (size_t)r->width * r->height * sizeof(float4))
            return fail(r, PTX_ERROR_INVALID_ARGUMENT, "ptx_unpack_shards: the host buffer must be width*height*16 bytes");
        hostOnDevice = hostFrameAlias(r, pinnedHost, hostBytes);
        if (!hostOnDevice)
            return fail(r, PTX_ERROR_INVALID_ARGUMENT, "ptx_unpack_shards: the host buffer is not page-locked memory the device can address");
        if (!r->evSnapshot)
        {
            HIP_TRY(r, hipEventCreateWithFlags(&r->evSnapshot, hipEventDisableTiming));
            HIP_TRY(r, hipEventCreateWithFlags(&r->evCopied, hipEventDisableTiming | hipEventReleaseToSystem));
        }
    }
    GatherParams g;
    g.width = r->width;
    g.height = r->height;
    g.tileSize = r->shard.tileSize;
    g.tilesX = (r->width + g.tileSize - 1) / g.tileSize;
    g.worldSize = r->shard.worldSize;
    g.strideSlots = (uint32_t)(strideBytes / sizeof(float4));
    // towards the host a FEW workgroups (PTX_COPY_GROUPS): what the link carries in a burst, the command processor's own traffic
    // over it waits behind (profiles/r05_unpack_groups.txt); device-only: the memory's rate
    uint32_t grid = gridFor((size_t)r->width * r->height);
    if (hostOnDevice)
        grid = std::min(grid, r->env.copyGroups ? r->env.copyGroups : 16u);
    k_gather_frame<<<grid, kBlock, 0, r->stream>>>(g, static_cast<const float4 *>(devSrc), toDeviceImage ? imagePtr(r) : nullptr, hostOnDevice);
    HIP_TRY(r, hipGetLastError());
    if (hostOnDevice) // ptx_readback_end waits for it
    {
        HIP_TRY(r, hipEventRecord(r->evCopied, r->stream));
        r->copyInFlight = true;
    }
    return PTX_OK;
}

// ptx_bind_shard_accumulation: the samples of a tile-sharded renderer are accumulated IN the dense tile-major buffer that is the
// message of the gather (k_accumulate's shard-major target) -- no ptx_pack_shard pass, no row-major frame on a rank that is not
// the frame's owner.  The buffer must hold this rank's shard (ptx_shard_bytes); entries of ragged tiles outside the image stay 0.
static int bindShardAccumulation(PtxRenderer *r, void *devShard, size_t bytes)
{
    if (!r || !r->width)
        return fail(r, PTX_ERROR_NOT_READY, "ptx_bind_shard_accumulation: call ptx_resize and ptx_set_tile_shard first");
    if (devShard && bytes < ptx_shard_bytes(r, r->shard.rank))
        return fail(r, PTX_ERROR_INVALID_ARGUMENT, "ptx_bind_shard_accumulation: the buffer must hold ptx_shard_bytes() = %zu bytes", ptx_shard_bytes(r, r->shard.rank));
    r->boundShard = static_cast<float4 *>(devShard);
    r->boundShardBytes = devShard ? ptx_shard_bytes(r, r->shard.rank) : 0;
    return PTX_OK;
}

// Renderer::RecordPostProcessCommands + RecordSaveOutputCommands (Renderer.cpp:928-1085, :1204-1246)
static int postprocess(PtxRenderer *r, const PtxPostProcessingUniformData *uniform, uint32_t toneMappingMode)
{
    if (!r || !uniform || toneMappingMode > PTX_TONE_MAPPING_HDR)
        return fail(r, PTX_ERROR_INVALID_ARGUMENT, "ptx_postprocess: bad argument");
    if (!imagePtr(r))
        return fail(r, PTX_ERROR_NOT_READY, "ptx_postprocess: no accumulation image (call ptx_resize)");
    if (r->boundShard)
        return frameIsElsewhere(r, "ptx_postprocess");
    HIP_TRY(r, hipSetDevice(r->device));
    const uint32_t W = r->width, H = r->height, n = W * H;
    uint32_t levels = 1;
    for (uint32_t m = W > H ? W : H; m > 1; m >>= 1)
        levels++;
    // mips 0 .. maxMipLevel-1 take part, maxMipLevel = min(levels - 3, MaxBloomMipmapLevel) (Renderer.cpp:955-956)
    uint32_t used = levels >= 5 ? (levels - 3 < 12 ? levels - 3 : 12) : 1;
    BloomLevel L[13];
    size_t total = 0;
    for (uint32_t l = 0; l < used; l++)
    {
        L[l].w = (W >> l) ? (W >> l) : 1;
        L[l].h = (H >> l) ? (H >> l) : 1;
        total += (size_t)L[l].w * L[l].h * 3;
    }
    HIP_TRY(r, r->postRgb.alloc((size_t)n * 3));
    HIP_TRY(r, r->bloomRgb.alloc(total));
    HIP_TRY(r, r->outLinear.alloc(n));
    size_t off = 0;
    for (uint32_t l = 0; l < used; l++)
    {
        L[l].rgb = r->bloomRgb.p + off;
        off += (size_t)L[l].w * L[l].h * 3;
    }
    k_postprocess<<<gridFor(n), kBlock, 0, r->stream>>>(imagePtr(r), n, *uniform, r->postRgb.p, L[0].rgb);
    for (uint32_t i = 0; i + 1 < used; i++)
        k_bloom_downsample<<<gridFor((size_t)L[i + 1].w * L[i + 1].h), kBlock, 0, r->stream>>>(L[i], L[i + 1]);
    for (uint32_t i = used - 1; i > 0; i--)
        k_bloom_upsample<<<gridFor((size_t)L[i - 1].w * L[i - 1].h), kBlock, 0, r->stream>>>(L[i], L[i - 1]);
    k_compose_tonemap<<<gridFor(n), kBlock, 0, r->stream>>>(r->postRgb.p, L[0].rgb, n, *uniform, toneMappingMode, r->outLinear.p);
    HIP_TRY(r, hipGetLastError());
    r->outputReady = true;
    return PTX_OK;
}

// OutputSaver: blit of the tone-mapped image into its output image + readback (OutputSaver.cpp:64-86, :120-199)
static int readOutput(PtxRenderer *r, uint32_t outputFormat, void *host, size_t bytes)
{
    if (!r || !host || outputFormat > PTX_OUTPUT_RGBA32F)
        return fail(r, PTX_ERROR_INVALID_ARGUMENT, "ptx_read_output: bad argument");
    if (!r->outputReady)
        return fail(r, PTX_ERROR_NOT_READY, "ptx_read_output: call ptx_postprocess first");
    const uint32_t n = r->width * r->height;
    const size_t want = (size_t)n * (outputFormat == PTX_OUTPUT_RGBA32F ? 16 : 4);
    if (bytes != want)
        return fail(r, PTX_ERROR_INVALID_ARGUMENT, "ptx_read_output: buffer must be %zu bytes", want);
    HIP_TRY(r, hipSetDevice(r->device));
    if (outputFormat == PTX_OUTPUT_RGBA32F)
        HIP_TRY(r, hipMemcpyAsync(host, r->outLinear.p, bytes, hipMemcpyDeviceToHost, r->stream));
    else
    {
        HIP_TRY(r, r->outSrgb8.alloc(n));
        k_encode_srgb8<<<gridFor(n), kBlock, 0, r->stream>>>(r->outLinear.p, n, r->outSrgb8.p);
        HIP_TRY(r, hipMemcpyAsync(host, r->outSrgb8.p, bytes, hipMemcpyDeviceToHost, r->stream));
    }
    HIP_TRY(r, hipStreamSynchronize(r->stream));
    HIP_TRY(r, hipGetLastError());
    return PTX_OK;
}

static int getStats(PtxRenderer *r, PtxStats *stats)
{
    if (!r || !stats)
        return PTX_ERROR_INVALID_ARGUMENT;
    HIP_TRY(r, hipStreamSynchronize(r->stream));
    {
        const int rcc = collectRender(r);
        if (rcc != PTX_OK)
            return rcc;
    }
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, r->evA, r->evB) == hipSuccess)
        r->stats.lastRenderMs = ms;
    *stats = r->stats;
    return PTX_OK;
}

static int testEval(PtxRenderer *r, uint32_t fn, const float *in, float *out, uint32_t n)
{
    if (!r || fn >= PTX_FN_COUNT || !in || !out)
        return fail(r, PTX_ERROR_INVALID_ARGUMENT, "ptx_test_eval: bad argument");
    if (!n)
        return PTX_OK;
    HIP_TRY(r, hipSetDevice(r->device));
    const size_t ni = (size_t)n * h_inStride[fn], no = (size_t)n * h_outStride[fn];
    HIP_TRY(r, r->testIn.alloc(ni));
    HIP_TRY(r, r->testOut.alloc(no));
    if (fn == PTX_FN_SAMPLE_LIGHT)
        HIP_TRY(r, r->testUbo.alloc(n));
    HIP_TRY(r, hipMemcpyAsync(r->testIn.p, in, ni * 4, hipMemcpyHostToDevice, r->stream));
    HIP_TRY(r, hipMemsetAsync(r->testOut.p, 0, no * 4, r->stream));
    k_test_eval<<<(n + 63) / 64, 64, 0, r->stream>>>(fn, r->testIn.p, r->testOut.p, n, r->testUbo.p);
    HIP_TRY(r, hipMemcpyAsync(out, r->testOut.p, no * 4, hipMemcpyDeviceToHost, r->stream));
    HIP_TRY(r, hipStreamSynchronize(r->stream));
    HIP_TRY(r, hipGetLastError());
    return PTX_OK;
}

static int testTexture(PtxRenderer *r, const float *in, float *out, uint32_t n, int implicitLod)
{
    if (!r || !in || !out)
        return fail(r, PTX_ERROR_INVALID_ARGUMENT, "ptx_test_texture: null argument");
    if (!r->sceneReady)
        return fail(r, PTX_ERROR_NOT_READY, "ptx_test_texture: no scene uploaded");
    if (!n)
        return PTX_OK;
    HIP_TRY(r, hipSetDevice(r->device));
    HIP_TRY(r, r->testIn.alloc((size_t)n * 7));
    HIP_TRY(r, r->testOut.alloc((size_t)n * 4));
    HIP_TRY(r, hipMemcpyAsync(r->testIn.p, in, (size_t)n * 28, hipMemcpyHostToDevice, r->stream));
    TextureView tv;
    tv.textures = r->renderTextures.p; tv.textureCount = r->textureCount; tv.texels8 = nullptr; tv.texelsF = r->renderTexels.p;
    tv.srgbLut = nullptr;
    k_test_texture<<<(n + 63) / 64, 64, 0, r->stream>>>(tv, r->testIn.p, r->testOut.p, n, implicitLod);
    HIP_TRY(r, hipMemcpyAsync(out, r->testOut.p, (size_t)n * 16, hipMemcpyDeviceToHost, r->stream));
    HIP_TRY(r, hipStreamSynchronize(r->stream));
    HIP_TRY(r, hipGetLastError());
    return PTX_OK;
}

static int traceRays(PtxRenderer *r, const float *rays, uint32_t n, int anyHit, float *hits, uint32_t *ids)
{
    if (!r || !rays || !hits || !ids)
        return fail(r, PTX_ERROR_INVALID_ARGUMENT, "ptx_trace_rays: null argument");
    if (!sceneUsable(r))
        return fail(r, PTX_ERROR_NOT_READY, "ptx_trace_rays: no acceleration structure (of a shared scene: the owner's is being replaced)");
    if (!n)
        return PTX_OK;
    HIP_TRY(r, hipSetDevice(r->device));
    DevBuf<float4> dRays, dHits;
    DevBuf<uint2> dIds;
    HIP_TRY(r, dRays.alloc((size_t)n * 2));
    HIP_TRY(r, dHits.alloc(n));
    HIP_TRY(r, dIds.alloc(n));
    TraceScene sc;
    sc = makeTraceScene(r);
    hipError_t e = hipMemcpyAsync(dRays.p, rays, (size_t)n * 32, hipMemcpyHostToDevice, r->stream);
    if (e == hipSuccess)
    {
        (void)hipEventRecord(r->evT0, r->stream);
        (void)hipMemsetAsync(&r->counters.p[C_CHUNK], 0, sizeof(uint32_t), r->stream);
        if (sceneOf(r)->anyNonOpaque)
            k_trace_rays<true><<<gridFor(n), kBlock, 0, r->stream>>>(sc, dRays.p, n, anyHit, dHits.p, dIds.p, &r->counters.p[C_CHUNK], r->spill.p);
        else
            k_trace_rays<false><<<gridFor(n), kBlock, 0, r->stream>>>(sc, dRays.p, n, anyHit, dHits.p, dIds.p, &r->counters.p[C_CHUNK], r->spill.p);
        (void)hipEventRecord(r->evT1, r->stream);
        e = hipMemcpyAsync(hits, dHits.p, (size_t)n * 16, hipMemcpyDeviceToHost, r->stream);
    }
    if (e == hipSuccess)
        e = hipMemcpyAsync(ids, dIds.p, (size_t)n * 8, hipMemcpyDeviceToHost, r->stream);
    if (e == hipSuccess)
        e = hipStreamSynchronize(r->stream);
    if (e == hipSuccess)
        e = hipGetLastError();
    float ms = 0.0f;
    (void)hipEventElapsedTime(&ms, r->evT0, r->evT1);
    r->stats.lastTraceMs = ms;
    dRays.release(); dHits.release(); dIds.release();
    if (e != hipSuccess)
        return fail(r, PTX_ERROR_DEVICE, "ptx_trace_rays: %s", hipGetErrorString(e));
    return PTX_OK;
}
