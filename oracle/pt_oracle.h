/*
 * pt_oracle.h -- CPU ORACLE for the path-tracing hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library.  The product (path-tracing_amd/) never includes, links or calls anything
 * in oracle/.
 *
 * It is a plain-C restatement of the reference's GLSL ray-tracing stages
 *   Path-Tracing/Shaders/{raygen.rgen, closestHit.rchit, miss.rmiss, occlusion.rmiss,
 *   common.glsl, shading.glsl, bsdf.glsl, sampling.glsl, ray.glsl, material.glsl,
 *   tracing.glsl}
 * with a brute-force / binned-SAH BVH closest-hit query standing in for the driver's
 * traceRayEXT.  Every function cites the file:line it follows.
 *
 * PARITY PIN: the pure-math functions are pinned to the reference by golden vectors
 * generated from the reference's own GLSL sources (tools/gen_golden.py compiles
 * shading.glsl / bsdf.glsl / common.glsl / ray.glsl / sampling.glsl / tracing.glsl /
 * material.glsl:55-171 as C++ through a builtin shim; vectors in tests/golden/).  The
 * shim's bit-exact mode takes pow / sin / cos from pt_oracle_math.h, so for those three
 * builtins the pin is structural; golden_libm.json (glibc) checks the kernels
 * independently.  Two stage bodies are pinned the same way (golden_stage_*.json): the
 * main() of raygen.rgen with its trace calls scripted (pto_test_raygen) and the main() of
 * closestHit.rchit over a one-triangle scene (pto_test_closest_hit), and the two any-hit
 * mains for one candidate (pto_test_any_hit), the miss main (pto_test_miss).  What the
 * Vulkan driver and sampler do (tree, ray / triangle test, textureGrad) is restated by reading and checked by the closed forms of
 * tests/test_analytic.py.  Image-level parity is UNPINNED by the reference -- "parity
 * unpinned": it holds no reference image, its tests assert finiteness only, and it
 * cannot be built here (Vulkan RT + 11 absent submodules) -- see DESIGN.md section 2.
 *
 * Arithmetic conventions (GLSL leaves these implementation-defined; both this oracle
 * and the HIP kernels fix them identically so images can be compared bit-for-bit):
 *   - IEEE binary32, round-to-nearest, no contraction (-ffp-contract=off), fma only
 *     where the GLSL says fma(); sqrt correctly rounded;
 *   - every division of the shader path is a * rcp(b) with rcp the correctly rounded
 *     reciprocal on [2^-126, 2^126] and flushed outside (pto_rcp, pt_oracle_math.h: inside
 *     GLSL's 2.5-ULP latitude; round 5 -- it was the correctly rounded quotient before);
 *     a vector / scalar is one reciprocal and three products;
 *   - dot(a,b) = (a.x*b.x + a.y*b.y) + a.z*b.z;  length = sqrt(dot);
 *     normalize(v) = v * rcp(sqrt(dot(v,v)));  cross, reflect, refract, mix, clamp as
 *     in the GLSL 4.60 spec 8.5; mat3*vec3 = (c0*x + c1*y) + c2*z; inverse(mat3) by
 *     cofactors * rcp(det);
 *   - pow(x,2) = x*x, pow(x,5) = x2*x2*x; general pow/sin/cos by the fixed polynomial
 *     kernels pto_powf / pto_sincosf below.
 */
#ifndef PT_ORACLE_H
#define PT_ORACLE_H

#include "../include/ptx.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct PtoScene PtoScene;

typedef struct PtoStats {
    uint64_t pathSamples;
    uint64_t segments;
    uint64_t shadowRays;
    uint64_t retries;
    uint64_t triangles;
    uint64_t nodesVisited;
    uint64_t trisTested;
} PtoStats;

typedef struct PtoHit {
    float t, u, v;
    uint32_t tri; /* global flattened triangle id, 0xffffffff = miss */
} PtoHit;

/* Flatten instances x meshes x primitives to world space (the stand-in for the
 * BLAS/TLAS build, AccelerationStructure.cpp:64-301); buildBvh != 0 adds a binned-SAH
 * BVH, otherwise queries are brute force. */
PTX_API PtoScene *pto_scene_create(const PtxSceneDesc *desc, int buildBvh);
/* The same with the state of one animated frame: instance transforms and bone matrices as Scene::Update left
 * them (NULL = the desc's transforms / bind pose); animated meshes are skinned on the CPU (skinning.comp). */
PTX_API PtoScene *pto_scene_create_posed(const PtxSceneDesc *desc, const PtxTransform *instanceTransforms, const PtxTransform *bones,
                                         uint32_t boneCount, int buildBvh);
PTX_API void pto_scene_destroy(PtoScene *s);
PTX_API uint64_t pto_scene_triangle_count(const PtoScene *s);

/* One launch of raygen.rgen over pixels [x0,x1) x [y0,y1) of a W x H image; adds into
 * accum (W*H*4 floats, alpha set to 1).  tileShard may be NULL (whole region). */
PTX_API int pto_render(const PtoScene *s, const PtxRaygenUniformData *u, const PtxLightsUbo *lights, uint32_t W,
                       uint32_t H, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, const PtxTileShard *tileShard,
                       float *accum, int threads, int bruteForce, PtoStats *stats);

/* traceRayEXT stand-ins over rays packed as (ox,oy,oz,tmin, dx,dy,dz,tmax). */
PTX_API void pto_trace_closest(const PtoScene *s, const float *rays, uint32_t n, PtoHit *hits, int bruteForce);
PTX_API void pto_trace_any(const PtoScene *s, const float *rays, uint32_t n, uint32_t *occluded, int bruteForce);

/* Sampler entry, same packing as ptx_test_texture (include/ptx.h). */
PTX_API int pto_test_texture(const PtoScene *s, const float *in, float *out, uint32_t n, int implicitLod);

/* Output stage (pt_oracle_post.c): ptx_postprocess / ptx_read_output on host arrays. */
PTX_API int pto_postprocess(const float *accum, uint32_t W, uint32_t H, const PtxPostProcessingUniformData *u, uint32_t toneMode,
                            float *outLinear);
PTX_API int pto_encode_output(const float *linear, uint32_t W, uint32_t H, uint32_t format, void *out);
PTX_API int pto_test_post(uint32_t which, const float *in, float *out, uint32_t n);

/* pto_rsq on the positive normal range against the EXACT correctly rounded 1 / sqrt(x): all 2^24 (mantissa, exponent parity)
 * classes, x * m^2 compared with 1 at the two midpoints m next to the result in 128-bit integer arithmetic; and the device's
 * algorithm (pt_device.hpp rsq_) run in float on the CPU from the seeds RN - 1 ULP, RN, RN + 1 ULP.  Returns the number of classes
 * where either differs from the exact answer (0), *seedDependent = classes where the three seeds disagree. */
PTX_API uint64_t pto_rsq_selfcheck(uint64_t *seedDependent);
/* Function-level entry, same packing as ptx_test_eval (include/ptx.h). */
PTX_API int pto_test_eval(uint32_t fn, const float *in, float *out, uint32_t n);
/* Stage level: raygen.rgen main() for one pixel with scripted trace calls (layout at the definition); checked against
 * tests/golden/golden_stage_*.json, which tools/gen_golden.py produces from the reference's raygen.rgen text. */
PTX_API int pto_test_raygen(const uint32_t *in, uint32_t *out, uint32_t n);
/* closestHit.rchit main() on triangle 0 of a scene, for given hits and incoming payloads (layout at the definition) */
/* miss.rmiss main() for ray directions */
PTX_API int pto_test_miss(const PtoScene *s, const uint32_t *in, uint32_t *out, uint32_t n);
/* anyhit.rahit / occlusionAnyhit.rahit main() for candidate hits on triangle 0 (layout at the definition) */
PTX_API int pto_test_any_hit(const PtoScene *s, const uint32_t *in, uint32_t *out, uint32_t n);
PTX_API int pto_test_closest_hit(const PtoScene *s, const PtxLightsUbo *lights, const uint32_t *in, uint32_t *out, uint32_t n);

#ifdef __cplusplus
}
#endif
#endif
