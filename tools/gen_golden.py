#!/usr/bin/env python3
"""Generate tests/golden/*.json from the REFERENCE's own GLSL sources.

Runs only in the build container (needs /root/reference).  The reference cannot be
built (Vulkan RT + absent submodules), but its pure-math shader headers compile as C++
once the GLSL builtins are supplied by tools/glsl_shim.hpp.  This script

  1. reads the listed line ranges of Path-Tracing/Shaders/*.glsl / *.incl,
  2. applies five mechanical rewrites (plus wrappers: the straight-line bodies of miss.rmiss:20-25,
     postprocess.comp:22-36, composition.comp:22 and toneMapping.comp:19-21 become functions of the values
     their main() reads) (out/inout -> references, float-literal suffix,
     swizzle -> method, drop #include/#version, braces around rand() argument lists so
     C++ keeps GLSL's left-to-right evaluation),
  3. writes the result into a TEMP directory (reference text is never copied into the
     repo), appends tools/golden_main.inc and compiles with g++ -ffp-contract=off,
  4. runs it twice -- libm transcendentals (golden_libm.json) and the fixed polynomial
     kernels (-DSHIM_FIXED, golden_fixed.json) -- and stores the vectors; each build also
     emits the stage-level cases (golden_stage_*.json): raygen.rgen's main() with scripted
     trace calls and closestHit.rchit's main() over a one-triangle scene.

Only inputs/outputs (uint32 bit patterns) are committed.
"""
import os
import re
import subprocess
import sys
import tempfile

REF = "/root/reference/Path-Tracing/Shaders"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "tests", "golden")
if "--out" in sys.argv:  # tests/test_oracle_golden.py regenerates into a scratch directory and compares with the committed archives
    OUT = sys.argv[sys.argv.index("--out") + 1]

# (file, [(first_line, last_line), ...])  1-based inclusive
SOURCES = [
    ("common.glsl", [(3, 25), (102, 122), (133, 202)]),
    ("shading.glsl", [(3, 129)]),
    ("bsdf.glsl", [(4, 132)]),
    ("ray.glsl", [(3, 131)]),
    ("sampling.glsl", [(3, 3), (17, 56)]),
    ("tracing.glsl", [(1, 161)]),
]
STRUCTS = [
    ("ShaderTypes.incl", ["Camera", "Vertex", "DirectionalLight", "PointLight"]),
    ("ShaderRendererTypes.incl", ["MaterialSample"]),
]

FLOAT_LIT = re.compile(r"(?<![A-Za-z_0-9.])((?:\d+\.\d*|\.\d+)(?:[eE][+-]?\d+)?|\d+[eE][+-]?\d+)(?![fF0-9A-Za-z_.])")


FLOAT_LIT_F = re.compile(r"(?<![A-Za-z_0-9.])((?:\d+\.\d*|\.\d+)(?:[eE][+-]?\d+)?|\d+[eE][+-]?\d+)[fF](?![0-9A-Za-z_.])")


def rewrite(text: str) -> str:
    text = re.sub(r"^\s*#(include|version|extension).*$", "", text, flags=re.M)
    text = re.sub(r"\binout\s+(\w+)\s+(\w+)", r"\1& \2", text)
    text = re.sub(r"\bout\s+(\w+)\s+(\w+)", r"\1& \2", text)
    text = FLOAT_LIT_F.sub(r"float(\1F)", text)  # literals the shader text already suffixes
    text = FLOAT_LIT.sub(r"float(\1f)", text)  # `float` is glsl::Float (glsl_shim.hpp): a literal and a variable have ONE type in ?:
    text = re.sub(r"\.(xyz|yzw|xy|yz|zw|rgb)\b", r".\1()", text)
    # GLSL evaluates call arguments left to right (GLSL 4.60 6.1.1); C++ only does so
    # for braced initialisers, and the draw order of rand() is part of the contract.
    text = re.sub(r"\bvec([23])\(((?:rand\(rngState\)(?:,\s*)?)+)\)", r"vec\1{\2}", text)
    return text


def lines(path, ranges):
    src = open(path).read().split("\n")
    out = []
    for a, b in ranges:
        out += src[a - 1:b]
    return "\n".join(out)


def extract_struct(path, name):
    src = open(path).read()
    m = re.search(r"struct\s+%s\s*\{.*?\};" % name, src, flags=re.S)
    if not m:
        raise SystemExit(f"struct {name} not found in {path}")
    return m.group(0)


def write_gz(dst, data):
    """gzip without a timestamp or file name in the header: the same vectors give the same bytes."""
    import gzip

    with open(dst, "wb") as raw, gzip.GzipFile(filename="", mode="wb", fileobj=raw, mtime=0) as f:
        f.write(data)


def main():
    if not os.path.isdir(REF):
        raise SystemExit("reference not present; golden vectors can only be regenerated in the build container")
    os.makedirs(OUT, exist_ok=True)
    with tempfile.TemporaryDirectory(prefix="ptgolden_") as tmp:
        parts = ['#include "%s/glsl_shim.hpp"' % HERE, "namespace glsl {", "const uint MaxLightCount = 64u;"]
        for f, names in STRUCTS:
            for n in names:
                parts.append(extract_struct(os.path.join(REF, f), n))
        parts.append("uint u_LightCount; DirectionalLight u_DirectionalLight; PointLight u_Lights[MaxLightCount];")
        for f, ranges in SOURCES:
            parts.append("// ---- %s" % f)
            parts.append(rewrite(lines(os.path.join(REF, f), ranges)))
        # miss.rmiss:20-25 is straight-line code inside main(): wrapped as a function of the ray direction
        parts.append("// ---- miss.rmiss")
        parts.append("vec2 missSkyboxTexCoords(vec3 gl_WorldRayDirectionEXT)\n{\n" +
                     rewrite(lines(os.path.join(REF, "miss.rmiss"), [(20, 25)])) + "\n    return texCoords;\n}")
        # output stage: the straight-line bodies of postprocess.comp:22-36, composition.comp:22 and
        # toneMapping.comp:19-21, each wrapped as a function of the values its imageLoads return
        parts.append("// ---- postprocess.comp / composition.comp / toneMapping.comp")
        parts.append(extract_struct(os.path.join(REF, "ShaderRendererTypes.incl"), "PostProcessingUniformData"))
        parts.append("PostProcessingUniformData mainUniform; const uint ToneMappingModeSDR = 0u, ToneMappingModeHDR = 1u, s_ToneMappingMode = ToneMappingModeSDR;")
        parts.append("void postprocessPixel(vec3 accColor, vec3& colorOut, vec3& bloomOut)\n{\n" +
                     rewrite(lines(os.path.join(REF, "postprocess.comp"), [(22, 36)])) + "\n    colorOut = color; bloomOut = bloomColor;\n}")
        parts.append("vec3 compositionPixel(vec3 postProcessColor, vec3 bloomColor)\n{\n" +
                     rewrite(lines(os.path.join(REF, "composition.comp"), [(22, 22)])) + "\n    return color;\n}")
        parts.append("vec3 toneMapPixel(vec3 color)\n{\n" +
                     rewrite(lines(os.path.join(REF, "toneMapping.comp"), [(19, 21)])) + "\n    return outColor;\n}")
        # material.glsl:4-23 (sampleValue) and :55-171 (ReconstructNormalFromXY, the three sampleMaterial overloads and the
        # dispatcher with flipNormalY) compile once `textures[]` / textureGrad and the three material buffers exist:
        # a texture is ONE texel here (what textureGrad returns is the input of the arithmetic under test)
        parts.append("// ---- material.glsl")
        st = os.path.join(REF, "ShaderTypes.incl")
        for n in ("MetallicRoughnessMaterial", "SpecularGlossinessMaterial", "PhongMaterial"):
            parts.append(extract_struct(st, n))
        parts.append(rewrite(lines(st, [(18, 19), (143, 145), (164, 168)])))
        parts.append(rewrite(lines(os.path.join(REF, "Debug", "DebugShaderTypes.incl"), [(33, 39)])))
        parts.append("struct Sampler2D { vec4 texel; }; Sampler2D textures[8];\n"
                     "inline vec4 textureGrad(const Sampler2D &s, vec2, vec2, vec2) { return s.texel; }\n"
                     "MetallicRoughnessMaterial metallicRoughnessMaterials[1]; SpecularGlossinessMaterial specularGlossinessMaterials[1]; "
                     "PhongMaterial phongMaterials[1];")
        parts.append(rewrite(lines(os.path.join(REF, "material.glsl"), [(4, 23), (55, 171)])))
        # ---- stage level: raygen.rgen:22-118 (checkOccluded + main) with the trace calls scripted.  The launch built-ins, the
        # uniform block, the payload and the image become globals of a nested namespace; traceRayEXT hands out the records
        # of a script (golden_main.inc) -- so what is pinned is the loop itself: RNG draws, the order of the radiance /
        # throughput updates, roulette, the NaN / inf restart
        parts.append("// ---- raygen.rgen")
        parts.append("namespace stage_raygen {")
        sr = os.path.join(REF, "ShaderRendererTypes.incl")
        parts.append(extract_struct(sr, "RaygenUniformData"))
        parts.append(extract_struct(sr, "Payload"))
        parts.append(rewrite(lines(sr, [(124, 127)])))
        parts.append("RaygenUniformData mainUniform; Payload payload; bool isOccluded; uvec2 gl_LaunchIDEXT, gl_LaunchSizeEXT;\n"
                     "const int u_TopLevelAS = 0, gl_RayFlagsTerminateOnFirstHitEXT = 4, gl_RayFlagsNoneEXT = 0;\n"
                     "struct Image { vec4 previous, stored; } u_Image;\n"
                     "inline vec4 imageLoad(Image &i, ivec2) { return i.previous; }\n"
                     "inline void imageStore(Image &i, ivec2, vec4 v) { i.stored = v; }\n"
                     "void traceRayEXT(int, int, int, uint, int, uint, vec3 origin, float tmin, vec3 direction, float tmax, int payloadLocation);")
        parts.append(rewrite(lines(os.path.join(REF, "raygen.rgen"), [(22, 118)])).replace("void main()", "void raygenMain()"))
        parts.append("} // namespace stage_raygen")
        # ---- stage level: closestHit.rchit:52-161 main().  Buffers, the shader record, the hit attributes and the ray
        # built-ins become globals; the vertex fetch (common.glsl:27-46, 124-130) and transform() (sampling.glsl:5-15) are
        # the reference's text over a one-triangle vertex / index buffer; textures are one texel each, as for sampleMaterial
        parts.append("// ---- closestHit.rchit")
        parts.append("namespace stage_hit {")
        parts.append(extract_struct(sr, "Payload"))
        parts.append(extract_struct(sr, "SBTBuffer"))
        parts.append(rewrite(lines(sr, [(97, 99)])))  # HitFlags*
        parts.append("struct VertexBuffer { const vec2 *v; }; struct IndexBuffer { const uint *v; };\n"
                     "struct Geometry { VertexBuffer Vertices; IndexBuffer Indices; };\n"
                     "Geometry geometries[1]; mat3x4 transforms[1]; mat3x4 gl_ObjectToWorld3x4EXT; SBTBuffer sbt; Payload payload; vec3 attribs;\n"
                     "uint s_HitFlags; int gl_PrimitiveID; vec3 gl_WorldRayOriginEXT, gl_WorldRayDirectionEXT; float gl_RayTmaxEXT;")
        parts.append(rewrite(lines(os.path.join(REF, "common.glsl"), [(27, 46), (124, 130)])))
        parts.append(rewrite(lines(os.path.join(REF, "sampling.glsl"), [(5, 15)])))
        parts.append(rewrite(lines(os.path.join(REF, "closestHit.rchit"), [(52, 161)])).replace("void main()", "void closestHitMain()"))
        # the two any-hit stages (anyhit.rahit:36-64, occlusionAnyhit.rahit:35-53) over the same globals; material.glsl:25-54
        # supplies getColorTextureIdx / getColorFactor; ignoreIntersectionEXT sets a flag
        parts.append("bool g_ignored;\n#define ignoreIntersectionEXT g_ignored = true\n"
                     "inline vec4 texture(const Sampler2D &s, vec2) { return s.texel; }")
        parts.append(rewrite(lines(os.path.join(REF, "material.glsl"), [(25, 54)])))
        parts.append(rewrite(lines(os.path.join(REF, "anyhit.rahit"), [(36, 65)])).replace("void main()", "void anyHitMain()"))
        parts.append(rewrite(lines(os.path.join(REF, "occlusionAnyhit.rahit"), [(35, 54)])).replace("void main()", "void occlusionAnyHitMain()"))
        parts.append("#undef ignoreIntersectionEXT")
        # miss.rmiss:16-39 main(): the sky constant, or a one-texel 2-D / cube sky (the lookup itself is the sampler's)
        parts.append(rewrite(lines(sr, [(92, 95)])))  # MissFlags*
        parts.append("uint s_MissFlags; Sampler2D skybox2D; struct SamplerCube { vec4 texel; } skyboxCube;\n"
                     "inline vec4 texture(const SamplerCube &s, vec3) { return s.texel; }")
        parts.append(rewrite(lines(os.path.join(REF, "miss.rmiss"), [(16, 39)])).replace("void main()", "void missMain()"))
        parts.append("} // namespace stage_hit")
        parts.append("} // namespace glsl")
        parts.append('#include "%s/golden_main.inc"' % HERE)
        cpp = os.path.join(tmp, "golden.cpp")
        open(cpp, "w").write("\n".join(parts))
        if os.environ.get("PT_GOLDEN_KEEP"):  # debugging the shim: a copy of the generated file OUTSIDE the repository
            open(os.environ["PT_GOLDEN_KEEP"], "w").write("\n".join(parts))
        for mode, flag in (("libm", []), ("fixed", ["-DSHIM_FIXED"])):
            exe = os.path.join(tmp, "golden_" + mode)
            cmd = ["g++", "-std=c++20", "-O1", "-ffp-contract=off", "-fno-fast-math", "-w"] + flag + [cpp, "-o", exe, "-lm"]
            subprocess.check_call(cmd)
            data = subprocess.check_output([exe])
            dst = os.path.join(OUT, "golden_%s.json.gz" % mode)
            write_gz(dst, data)
            print("wrote", os.path.relpath(dst), len(data), "bytes")
            data = subprocess.check_output([exe, "stage"])  # the stage-level cases go into their own file
            dst = os.path.join(OUT, "golden_stage_%s.json.gz" % mode)
            write_gz(dst, data)
            print("wrote", os.path.relpath(dst), len(data), "bytes")


if __name__ == "__main__":
    sys.exit(main())
