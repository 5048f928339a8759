// ptx_capi.hip -- the C-ABI of include/ptx.h (the one translation unit of libptx_hip.so).  Every entry point cites the
// Renderer member it replaces in include/ptx.h; the implementations are in pt_runtime.hpp (host side) and pt_wavefront.hpp /
// pt_bvh.hpp / pt_bvh_build.hpp / pt_device.hpp (device side).  No exceptions cross this boundary: status codes + ptx_last_error.
#include "pt_runtime.hpp"

extern "C" {

uint32_t ptx_abi_version(void)
{
    return PTX_ABI_VERSION;
}

int ptx_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
        return 0;
    return n;
}

int ptx_create(const PtxDeviceDesc *desc, PtxRenderer **out)
{
    return createRenderer(desc, out);
}

void ptx_destroy(PtxRenderer *r)
{
    destroyRenderer(r);
}

const char *ptx_last_error(const PtxRenderer *r)
{
    return r ? r->error.c_str() : "null renderer";
}

int ptx_set_backend(PtxRenderer *r, uint32_t backend)
{
    if (!r || backend > PTX_BACKEND_MEGAKERNEL)
        return fail(r, PTX_ERROR_INVALID_ARGUMENT, "ptx_set_backend: bad backend %u", backend);
    if (r->backend != backend)
        r->hintSlots = 0u; // the learnt bounce schedule is the wavefront backend's
    r->backend = backend;
    return PTX_OK;
}

int ptx_share_scene(PtxRenderer *r, PtxRenderer *owner)
{
    return shareScene(r, owner);
}

int ptx_scene_upload(PtxRenderer *r, const PtxSceneDesc *s)
{
    return sceneUpload(r, s);
}

int ptx_build_accel(PtxRenderer *r)
{
    return buildBestTree(r);
}

int ptx_update_animation(PtxRenderer *r, const PtxTransform *instanceTransforms, uint32_t instanceCount, const PtxTransform *boneTransforms, uint32_t boneCount, uint32_t accelUpdate)
{
    return updateAnimation(r, instanceTransforms, instanceCount, boneTransforms, boneCount, accelUpdate);
}

int ptx_resize(PtxRenderer *r, uint32_t width, uint32_t height)
{
    if (!r || !width || !height || (uint64_t)width * height > 0x7fffffffull)
        return fail(r, PTX_ERROR_INVALID_ARGUMENT, "ptx_resize: bad extent %ux%u", width, height);
    HIP_TRY(r, hipSetDevice(r->device));
    r->width = width;
    r->height = height;
    r->outputReady = false;
    r->boundImage = nullptr;
    r->boundShard = nullptr;
    r->boundShardBytes = 0;
    r->hostAlias = nullptr; // (a frame buffer of the new size is another registration)
    HIP_TRY(r, r->image.alloc((size_t)width * height));
    return ptx_reset_accumulation(r);
}

int ptx_set_tile_shard(PtxRenderer *r, const PtxTileShard *s)
{
    if (!r || !s || !s->worldSize || s->rank >= s->worldSize || !s->tileSize || (s->tileSize % 8) != 0 || s->tileSize > 1024)
        return fail(r, PTX_ERROR_INVALID_ARGUMENT, "ptx_set_tile_shard: need rank < worldSize and tileSize a multiple of 8");
    if (r->boundShard && (r->shard.rank != s->rank || r->shard.worldSize != s->worldSize || r->shard.tileSize != s->tileSize))
    {
        r->boundShard = nullptr; // the bound buffer was laid out for the previous shard
        r->boundShardBytes = 0;
    }
    r->shard = *s;
    return PTX_OK;
}

int ptx_reset_accumulation(PtxRenderer *r)
{
    if (!r || !imagePtr(r))
        return fail(r, PTX_ERROR_NOT_READY, "ptx_reset_accumulation: no accumulation image (call ptx_resize)");
    if (r->boundShard)
        HIP_TRY(r, hipMemsetAsync(r->boundShard, 0, r->boundShardBytes, r->stream));
    else
        HIP_TRY(r, hipMemsetAsync(imagePtr(r), 0, (size_t)r->width * r->height * sizeof(float4), r->stream));
    return PTX_OK;
}

int ptx_render(PtxRenderer *r, const PtxRaygenUniformData *uniform, const PtxLightsUbo *lights)
{
    return renderImpl(r, uniform, lights, uniform ? uniform->TotalSamples : 0, 1);
}

int ptx_render_frames(PtxRenderer *r, const PtxRaygenUniformData *uniform, const PtxLightsUbo *lights, uint32_t firstFrame, uint32_t frames)
{
    if (!uniform)
        return fail(r, PTX_ERROR_INVALID_ARGUMENT, "ptx_render_frames: null uniform");
    PtxRaygenUniformData u = *uniform;
    u.SampleCount = 1; // canonical schedule: one sample per launch, RNG frame = launch index
    u.TotalSamples = firstFrame;
    return renderImpl(r, &u, lights, firstFrame, frames);
}

int ptx_synchronize(PtxRenderer *r)
{
    if (!r)
        return PTX_ERROR_INVALID_ARGUMENT;
    HIP_TRY(r, hipStreamSynchronize(r->stream));
    return collectRender(r); // errors of an asynchronous launch surface here
}

int ptx_readback(PtxRenderer *r, float *rgba, size_t bytes)
{
    if (!r || !rgba || !imagePtr(r) || bytes != (size_t)r->width * r->height * sizeof(float4))
        return fail(r, PTX_ERROR_INVALID_ARGUMENT, "ptx_readback: buffer must be width*height*16 bytes");
    if (r->boundShard)
        return frameIsElsewhere(r, "ptx_readback");
    HIP_TRY(r, hipMemcpyAsync(rgba, imagePtr(r), bytes, hipMemcpyDeviceToHost, r->stream));
    HIP_TRY(r, hipStreamSynchronize(r->stream));
    return collectRender(r); // an error of the launch that produced the image surfaces with it
}

int ptx_readback_begin(PtxRenderer *r, float *pinnedHost, size_t bytes)
{
    return readbackBegin(r, pinnedHost, bytes);
}

int ptx_readback_end(PtxRenderer *r)
{
    if (!r)
        return PTX_ERROR_INVALID_ARGUMENT;
    if (r->copyInFlight)
    {
        HIP_TRY(r, hipEventSynchronize(r->evCopied));
        r->copyInFlight = false;
    }
    return PTX_OK;
}

void *ptx_device_accum_ptr(PtxRenderer *r)
{
    return r ? imagePtr(r) : nullptr;
}

size_t ptx_accum_bytes(const PtxRenderer *r)
{
    return r ? (size_t)r->width * r->height * sizeof(float4) : 0;
}

size_t ptx_shard_bytes(const PtxRenderer *r, uint32_t rank)
{
    if (!r || !r->width || rank >= r->shard.worldSize)
        return 0;
    PtxRenderer tmp;
    tmp.width = r->width;
    tmp.height = r->height;
    tmp.shard = r->shard;
    tmp.shard.rank = rank;
    const LaunchParams p = makeParams(&tmp, nullptr, 0, 1);
    return (size_t)p.slotsPerFrame * sizeof(float4);
}

int ptx_pack_shard(PtxRenderer *r, void *devDst)
{
    if (!r || !devDst || !imagePtr(r))
        return fail(r, PTX_ERROR_INVALID_ARGUMENT, "ptx_pack_shard: null argument");
    const LaunchParams p = makeParams(r, nullptr, 0, 1);
    if (r->boundShard) // the accumulation already IS the packed shard (ptx_bind_shard_accumulation)
    {
        if (devDst != r->boundShard && p.slotsPerFrame)
            HIP_TRY(r, hipMemcpyAsync(devDst, r->boundShard, (size_t)p.slotsPerFrame * sizeof(float4), hipMemcpyDeviceToDevice, r->stream));
        return PTX_OK;
    }
    if (p.slotsPerFrame)
        k_pack_shard<<<gridFor(p.slotsPerFrame), kBlock, 0, r->stream>>>(p, imagePtr(r), static_cast<float4 *>(devDst));
    HIP_TRY(r, hipGetLastError());
    return PTX_OK;
}

int ptx_unpack_shard(PtxRenderer *r, uint32_t rank, const void *devSrc)
{
    return unpackShard(r, rank, devSrc);
}

int ptx_unpack_shard_host(PtxRenderer *r, uint32_t rank, const void *devSrc, float *pinnedHost, size_t bytes)
{
    if (!pinnedHost)
        return fail(r, PTX_ERROR_INVALID_ARGUMENT, "ptx_unpack_shard_host: null host buffer");
    return unpackShard(r, rank, devSrc, pinnedHost, bytes);
}

int ptx_unpack_shards(PtxRenderer *r, const void *devSrc, size_t strideBytes, int toDeviceImage, float *pinnedHost, size_t bytes)
{
    return unpackShards(r, devSrc, strideBytes, toDeviceImage, pinnedHost, bytes);
}

int ptx_bind_shard_accumulation(PtxRenderer *r, void *devShard, size_t bytes)
{
    return bindShardAccumulation(r, devShard, bytes);
}

int ptx_postprocess(PtxRenderer *r, const PtxPostProcessingUniformData *uniform, uint32_t toneMappingMode)
{
    return postprocess(r, uniform, toneMappingMode);
}

int ptx_read_output(PtxRenderer *r, uint32_t outputFormat, void *host, size_t bytes)
{
    return readOutput(r, outputFormat, host, bytes);
}

int ptx_write_accumulation(PtxRenderer *r, const float *rgba, size_t bytes)
{
    if (!r || !rgba || !imagePtr(r) || bytes != (size_t)r->width * r->height * sizeof(float4))
        return fail(r, PTX_ERROR_INVALID_ARGUMENT, "ptx_write_accumulation: buffer must be width*height*16 bytes");
    if (r->boundShard)
        return frameIsElsewhere(r, "ptx_write_accumulation");
    HIP_TRY(r, hipMemcpyAsync(imagePtr(r), rgba, bytes, hipMemcpyHostToDevice, r->stream));
    HIP_TRY(r, hipStreamSynchronize(r->stream));
    return PTX_OK;
}

int ptx_get_stats(PtxRenderer *r, PtxStats *stats)
{
    return getStats(r, stats);
}

int ptx_test_input_stride(uint32_t fn)
{
    return fn < PTX_FN_COUNT ? h_inStride[fn] : -1;
}

int ptx_test_output_stride(uint32_t fn)
{
    return fn < PTX_FN_COUNT ? h_outStride[fn] : -1;
}

int ptx_test_eval(PtxRenderer *r, uint32_t fn, const float *in, float *out, uint32_t n)
{
    return testEval(r, fn, in, out, n);
}

int ptx_test_texture(PtxRenderer *r, const float *in, float *out, uint32_t n, int implicitLod)
{
    return testTexture(r, in, out, n, implicitLod);
}

int ptx_trace_rays(PtxRenderer *r, const float *rays, uint32_t n, int anyHit, float *hits, uint32_t *ids)
{
    return traceRays(r, rays, n, anyHit, hits, ids);
}

int ptx_bind_accumulation(PtxRenderer *r, void *devPtr, size_t bytes)
{
    if (!r || !r->width)
        return fail(r, PTX_ERROR_NOT_READY, "ptx_bind_accumulation: call ptx_resize first");
    if (devPtr && bytes != (size_t)r->width * r->height * sizeof(float4))
        return fail(r, PTX_ERROR_INVALID_ARGUMENT, "ptx_bind_accumulation: buffer must be width*height*16 bytes");
    r->boundImage = static_cast<float4 *>(devPtr);
    return PTX_OK;
}

} // extern "C"
