"""Row N2: the glTF 2.0 importer (host mirror of SceneImporter.cpp).  Assets are written by tests/gltf_util.py."""
import ctypes as C
import math

import numpy as np
import pytest

import util
from gltf_util import GltfWriter, cube, quad


class _Tex(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("format", C.c_uint32), ("levels", C.c_uint32), ("data", C.c_void_p)]


def _mr(desc, i):
    return np.frombuffer(C.string_at(desc.metallicRoughnessMaterials + 96 * i, 96), np.float32), \
        np.frombuffer(C.string_at(desc.metallicRoughnessMaterials + 96 * i, 96), np.uint32)


def _png_bytes(pkg, tmp_path, img, name="t.png"):
    pkg.write_image(tmp_path / name, img, pkg.OUTPUT_PNG)
    return (tmp_path / name).read_bytes()


def _static_asset(pkg, tmp_path, tex_mode, container):
    w = GltfWriter()
    tex = np.zeros((8, 4, 4), np.uint8)
    tex[..., 0] = 250
    tex[:4, :, 1] = 200   # top half of the image is yellowish (v = 0 at the top in glTF)
    tex[..., 3] = 255
    tex[7, 3, 3] = 0      # one fully transparent texel: 4 channels -> hasTransparency, rgb zeroed
    t = w.image_texture(_png_bytes(pkg, tmp_path, tex), tex_mode, tmp_path)
    floor_mat = w.material(name="Floor", pbrMetallicRoughness={"baseColorTexture": {"index": t}, "baseColorFactor": [0.5, 1, 1, 1],
                                                               "metallicFactor": 0.0, "roughnessFactor": 0.7})
    glass = w.material(name="Glass", pbrMetallicRoughness={"baseColorFactor": [1, 1, 1, 1], "metallicFactor": 0.0, "roughnessFactor": 0.05},
                       emissiveFactor=[0.1, 0.2, 0.3],
                       extensions={"KHR_materials_transmission": {"transmissionFactor": 0.9}, "KHR_materials_ior": {"ior": 1.33},
                                   "KHR_materials_volume": {"attenuationColor": [0.8, 0.9, 1.0], "attenuationDistance": 2.5},
                                   "KHR_materials_emissive_strength": {"emissiveStrength": 4.0}})
    pos, nrm, uv, idx = quad(4.0)
    floor = w.mesh([w.primitive(pos, idx, nrm, uv, material=floor_mat)])
    cp, cn, ci = cube(0.5)
    shared = w.primitive(cp, ci, cn, material=glass)
    no_material = {"attributes": dict(shared["attributes"]), "indices": shared["indices"]}  # same geometry, default material
    box = w.mesh([shared])
    box2 = w.mesh([no_material])
    w.node(mesh=floor)
    group = w.node(translation=[1.0, 0.5, 0.0], rotation=[0.0, math.sin(math.pi / 8), 0.0, math.cos(math.pi / 8)])  # 45 degrees about y
    w.node(parent=group, mesh=box, scale=[1.0, 2.0, 1.0])
    w.node(parent=group, mesh=box2, translation=[0.0, 0.0, 2.0])
    w.doc["extensions"] = {"KHR_lights_punctual": {"lights": [{"type": "point", "color": [1.0, 0.5, 0.25], "intensity": 8.0},
                                                               {"type": "directional", "color": [1, 1, 1], "intensity": 2.0}]}}
    w.node(translation=[0.0, 3.0, 1.0], extensions={"KHR_lights_punctual": {"light": 0}})
    w.node(rotation=[-math.sin(math.pi / 4), 0, 0, math.cos(math.pi / 4)], extensions={"KHR_lights_punctual": {"light": 1}})  # -z -> -y
    w.doc["cameras"] = [{"type": "perspective", "perspective": {"yfov": 0.8, "znear": 0.05, "zfar": 200.0, "aspectRatio": 1.5}}]
    w.node(translation=[0.0, 2.0, 6.0], camera=0)
    path = tmp_path / ("scene.glb" if container == "glb" else "scene.gltf")
    if container == "glb":
        w.write_glb(path)
    else:
        w.write_gltf(path, external_bin=(container == "gltf+bin"))
    return path


@pytest.mark.parametrize("tex_mode,container", [("uri", "gltf+bin"), ("data", "gltf"), ("view", "glb")])
def test_static_gltf_import(pkg, orc, tmp_path, tex_mode, container):
    path = _static_asset(pkg, tmp_path, tex_mode, container)
    s = pkg.Scene("file:" + str(path))
    d = s.desc
    a = util.desc_arrays(d)
    # everything below the root is static: ONE model instance with baked mesh transforms (SceneImporter.cpp:708-837)
    assert d.instanceCount == 1 and d.modelCount == 1 and d.meshCount == 3
    # the cube geometry is shared by the two primitives that differ only in material (:402-413)
    assert d.geometryCount == 2 and len({int(m["GeometryIndex"]) for m in a["meshes"]}) == 2
    assert sorted(int(g["IndexLength"]) for g in a["geometries"]) == [6, 36]
    # materials: Floor, Glass, + the default material for the primitive without one
    assert d.metallicRoughnessMaterialCount == 3
    by_tris = {int(a["geometries"][m["GeometryIndex"]]["IndexLength"]): m for m in a["meshes"]}
    floor_f, floor_u = _mr(d, int(by_tris[6]["MaterialId"]) >> 8)
    assert np.allclose(floor_f[4:8], [0.5, 1, 1, 1]) and np.isclose(floor_f[8], 0.7) and floor_f[9] == 0.0
    assert floor_u[20] == 9 and floor_u[21] == 1 and floor_u[19] == 4  # ColorIdx = first scene texture, NormalIdx = the default normal texture
    glass_ids = sorted({int(m["MaterialId"]) >> 8 for m in a["meshes"] if int(a["geometries"][m["GeometryIndex"]]["IndexLength"]) == 36})
    gf, _ = _mr(d, glass_ids[0])
    assert np.allclose(gf[0:3], [0.1, 0.2, 0.3]) and gf[3] == 4.0 and np.isclose(gf[10], 1.33) and np.isclose(gf[11], 0.9)
    assert np.allclose(gf[12:15], [0.8, 0.9, 1.0]) and np.isclose(gf[15], 2.5)
    df, _ = _mr(d, glass_ids[1])
    assert np.allclose(df[4:8], 1.0) and df[8] == 1.0 and df[9] == 1.0 and np.isclose(df[10], 1.5) and df[15] > 1e30
    # texture: decoded PNG, sRGB colour format, the alpha-0 texel zeroed; 4 channels -> the floor geometry is non-opaque
    t = (_Tex * d.textureCount).from_address(d.textures)
    assert d.textureCount == 1 and (t[0].width, t[0].height, t[0].format) == (4, 8, 1)
    px = np.frombuffer(C.string_at(t[0].data, 4 * 8 * 4), np.uint8).reshape(8, 4, 4)
    assert (px[0, 0] == [250, 200, 0, 255]).all() and (px[7, 3] == 0).all()
    floor_geo = a["geometries"][by_tris[6]["GeometryIndex"]]
    assert floor_geo["IsOpaque"] == 0 and all(g["IsOpaque"] == 1 for g in a["geometries"] if g["IndexLength"] == 36)
    # aiProcess_FlipUVs: v -> 1 - v
    fv = a["vertices"][floor_geo["VertexOffset"]:floor_geo["VertexOffset"] + 4]
    assert np.allclose(fv[:, 3:5], [[0, 1], [1, 1], [1, 0], [0, 0]])
    # geometry in world space through the oracle: rays from above hit the floor at y = 0 and the scaled cube at y = 1.5
    osc = orc.OracleScene(d, build_bvh=False)
    rays = np.float32([[-3, 5, -3, 1e-5, 0, -1, 0, 1e4], [1.0, 5, 0.0, 1e-5, 0, -1, 0, 1e4]])
    hit = osc.trace_closest(rays)
    assert np.allclose(hit["t"], [5.0, 3.5], atol=1e-5)
    c, sn = math.cos(math.pi / 4), math.sin(math.pi / 4)  # the child cube sits at group * (0, 0, 2) = (1 + 2 sin45, 0.5, 2 cos45)
    hit = osc.trace_closest(np.float32([[1 + 2 * sn, 5, 2 * c, 1e-5, 0, -1, 0, 1e4]]))
    assert np.allclose(hit["t"], [4.0], atol=1e-5)
    # lights: the point light sits at its node, colour * intensity, attenuation (0, 0, 1); the directional light points down
    L = s.lights
    assert L.LightCount == 1 and np.allclose(list(L.Lights[0].Position), [0, 3, 1]) and np.allclose(list(L.Lights[0].Color), [8, 4, 2])
    assert (L.Lights[0].AttenuationConstant, L.Lights[0].AttenuationLinear, L.Lights[0].AttenuationQuadratic) == (0, 0, 1)
    assert np.allclose(list(L.Directional.Direction), [0, -1, 0], atol=1e-6) and np.allclose(list(L.Directional.Color), [2, 2, 2])
    # the file's camera becomes the active scene camera: the floor is in view, the image is finite
    u = s.uniform(48, 32, bounces=3, sample_count=2)
    eye = np.array(u.ViewInverse, np.float32).reshape(4, 4)[3, :3]
    assert np.allclose(eye, [0, 2, 6], atol=1e-5)
    img, st = osc.render(u, L, 48, 32)
    assert np.isfinite(img).all() and st.shadowRays > 0 and img[24:, :, :3].mean() > img[:4, :, :3].mean() * 0.1


def test_animated_and_skinned_glb_import(pkg, orc, tmp_path):
    w = GltfWriter()
    mat = w.material(pbrMetallicRoughness={"baseColorFactor": [0.8, 0.3, 0.2, 1], "metallicFactor": 0.0})
    pos, nrm, uv, idx = quad(3.0)
    w.node(mesh=w.mesh([w.primitive(pos, idx, nrm, uv, material=mat)]))
    cp, cn, ci = cube(0.25)
    mover = w.node(translation=[0.0, 1.0, 0.0], mesh=w.mesh([w.primitive(cp, ci, cn, material=mat)]))
    # a two-bone strip along +y: 3 rings of 2 vertices, the middle ring half / half
    sp = np.float32([[-0.1, 0, 0], [0.1, 0, 0], [-0.1, 1, 0], [0.1, 1, 0], [-0.1, 2, 0], [0.1, 2, 0]]) + np.float32([2, 0, 0])
    sn = np.float32([[0, 0, 1]] * 6)
    si = np.uint16([0, 1, 3, 0, 3, 2, 2, 3, 5, 2, 5, 4])
    joints = np.uint16([[0, 0, 0, 0]] * 2 + [[0, 1, 0, 0]] * 2 + [[1, 0, 0, 0]] * 2)
    weights = np.float32([[1, 0, 0, 0]] * 2 + [[0.5, 0.5, 0, 0]] * 2 + [[1, 0, 0, 0]] * 2)
    strip = w.mesh([w.primitive(sp, si, sn, material=mat, joints=joints, weights=weights)])
    rig = w.node()
    j0 = w.node(parent=rig, translation=[2.0, 0.0, 0.0])
    j1 = w.node(parent=j0, translation=[0.0, 1.0, 0.0])
    ibm = np.zeros((2, 16), np.float32)
    for k, t in enumerate(([2.0, 0.0, 0.0], [2.0, 1.0, 0.0])):  # inverse bind = translate(-joint position), column-major
        m = np.eye(4, dtype=np.float32)
        m[:3, 3] = -np.float32(t)
        ibm[k] = m.T.reshape(-1)
    w.doc["skins"] = [{"joints": [j0, j1], "inverseBindMatrices": w.accessor(ibm)}]
    w.node(parent=rig, mesh=strip, skin=0)
    times = w.accessor(np.float32([0.0, 1.0, 2.0]), minmax=True)
    trans = w.accessor(np.float32([[0, 1, 0], [2, 1, 0], [0, 1, 0]]))
    rots = w.accessor(np.float32([[0, 0, 0, 1], [0, 0, math.sin(math.pi / 4), math.cos(math.pi / 4)], [0, 0, 0, 1]]))  # j1: 90 degrees about z
    w.doc["animations"] = [{"samplers": [{"input": times, "output": trans, "interpolation": "LINEAR"},
                                          {"input": times, "output": rots, "interpolation": "LINEAR"}],
                            "channels": [{"sampler": 0, "target": {"node": mover, "path": "translation"}},
                                         {"sampler": 1, "target": {"node": j1, "path": "rotation"}}]}]
    path = tmp_path / "anim.glb"
    w.write_glb(path)

    s = pkg.Scene("file:" + str(path))
    d = s.desc
    a = util.desc_arrays(d)
    # instances: the static root model (floor), the animated node's model (cube), the skinned strip (:708-837)
    assert d.instanceCount == 3 and d.animatedVertexCount == 6 and d.animatedIndexCount == 12
    assert sorted(int(g["IsAnimated"]) for g in a["geometries"]) == [0, 0, 1]
    assert s.lib.pth_scene_bone_count(s.handle) == 2
    assert s.update(0.0)
    it0, bn0 = s.animation_state()
    cube_instance = [i for i in range(3) if np.allclose(it0[i].reshape(3, 4)[:, 3], [0, 1, 0])]
    assert len(cube_instance) == 1
    assert np.allclose(bn0[0].reshape(3, 4), np.eye(4)[:3], atol=1e-6) and np.allclose(bn0[1].reshape(3, 4), np.eye(4)[:3], atol=1e-6)
    s.update(0.5)  # key times are milliseconds at 1000 ticks / s: 0.5 s = half way to the second key
    it1, bn1 = s.animation_state()
    assert np.allclose(it1[cube_instance[0]].reshape(3, 4)[:, 3], [1, 1, 0], atol=1e-5)
    # j1 rotated by 45 degrees about z around its own origin (2, 1, 0): bone 1 = T(joint) R T(-joint)
    c = math.cos(math.pi / 4)
    assert np.allclose(bn1[1].reshape(3, 4)[:, :3], [[c, -c, 0], [c, c, 0], [0, 0, 1]], atol=1e-5)
    assert np.allclose(bn1[1].reshape(3, 4) @ [2, 1, 0, 1], [2, 1, 0], atol=1e-5)
    # the posed oracle scene: the strip's top ring has swung to the left of x = 2
    osc = orc.OracleScene(d, build_bvh=False, instance_transforms=it1, bones=bn1)
    rays = np.float32([[2.0 - c * 0.9, 1.0 + c * 0.9, 5.0, 1e-5, 0, 0, -1, 1e4],   # on the rotated upper segment
                       [2.0, 1.9, 5.0, 1e-5, 0, 0, -1, 1e4]])                        # where the bind-pose tip used to be
    hit = osc.trace_closest(rays)
    first = util.pair_first(d)
    assert hit["tri"][0] != 0xFFFFFFFF and hit["tri"][1] == 0xFFFFFFFF and np.isclose(hit["t"][0], 5.0, atol=1e-4)
    s.update(1.5)  # the loop closes after 2 s
    it2, bn2 = s.animation_state()
    assert np.allclose(it2, it0, atol=1e-5) and np.allclose(bn2, bn0, atol=1e-5) and len(first) == 4


def test_importer_errors(pkg, tmp_path):
    (tmp_path / "bad.gltf").write_text('{"asset": {"version": "1.0"}}')
    with pytest.raises(pkg.PtxError):
        pkg.Scene("file:" + str(tmp_path / "bad.gltf"))
    (tmp_path / "broken.gltf").write_text('{"asset": {"version": "2.0"}, "scenes": [{"nodes": [0]}], "nodes": [{"mesh": 0}], '
                                          '"meshes": [{"primitives": [{"attributes": {"POSITION": 5}}]}]}')
    with pytest.raises(pkg.PtxError):
        pkg.Scene("file:" + str(tmp_path / "broken.gltf"))
    with pytest.raises(pkg.PtxError):
        pkg.Scene("file:" + str(tmp_path / "missing.glb"))
    # crafted offsets / lengths / strides: a span before or past its buffer must be refused, not read
    import json

    def variant(name, edit):
        w = GltfWriter()
        pos, nrm, uv, idx = quad(1.0)
        w.node(mesh=w.mesh([w.primitive(pos, idx, nrm, uv)]))
        p = tmp_path / name
        w.write_gltf(p, external_bin=False)
        doc = json.loads(p.read_text())
        edit(doc)
        p.write_text(json.dumps(doc))
        return p

    ok = variant("ok.gltf", lambda d: None)
    assert pkg.Scene("file:" + str(ok)).triangle_count == 2
    edits = {
        "neg_offset": lambda d: d["bufferViews"][0].update(byteOffset=-16, byteLength=64),
        "neg_length": lambda d: d["bufferViews"][0].update(byteOffset=16, byteLength=-8),
        "huge_length": lambda d: d["bufferViews"][0].update(byteLength=2**62),
        "offset_past_end": lambda d: d["bufferViews"][0].update(byteOffset=2**40, byteLength=-(2**40) + 8),
        "neg_accessor_offset": lambda d: d["accessors"][0].update(byteOffset=-4),
        "huge_accessor_offset": lambda d: d["accessors"][0].update(byteOffset=2**63 - 1),
        "neg_stride": lambda d: d["bufferViews"][0].update(byteStride=-12),
        "huge_count": lambda d: d["accessors"][0].update(count=2**40),
    }
    for name, edit in edits.items():
        with pytest.raises(pkg.PtxError):
            pkg.Scene("file:" + str(variant(name + ".gltf", edit)))


@pytest.mark.gpu
def test_imported_scene_renders_like_the_oracle(pkg, orc, tmp_path):
    path = _static_asset(pkg, tmp_path, "view", "glb")
    scene = pkg.Scene("file:" + str(path))
    W, H = 128, 72
    r = pkg.Renderer()
    r.upload(scene)
    r.resize(W, H)
    u = scene.uniform(W, H, bounces=5, sample_count=2)
    r.render(u, scene.lights)
    ref, st = orc.OracleScene(scene.desc).render(u, scene.lights, W, H)
    assert r.stats().segments == st.segments and (r.readback().view(np.uint32) == ref.view(np.uint32)).all()
    r.close()


# ---------------------------------------------------------------------------------------
# scene descriptions (the aggregates of ExampleScenes.cpp:87-236): several components, a skybox file, flags
# ---------------------------------------------------------------------------------------
def _component(pkg, tmp_path, name, build):
    w = GltfWriter()
    build(w)
    path = tmp_path / name
    if name.endswith(".glb"):
        w.write_glb(path)
    else:
        w.write_gltf(path, external_bin=False)
    return path


def _description_assets(pkg, tmp_path):
    """Three components in the style of Intel Sponza (main + two add-ons), one of them with a
    KHR_materials_pbrSpecularGlossiness material and textures, and an .hdr sky."""
    rng = np.random.default_rng(3)
    diffuse = np.zeros((16, 16, 4), np.uint8)
    diffuse[..., :3] = rng.integers(60, 255, (16, 16, 3))
    diffuse[..., 3] = 255
    spec_gloss = np.zeros((8, 8, 4), np.uint8)
    spec_gloss[..., :3] = 200
    spec_gloss[..., 3] = np.linspace(40, 250, 8, dtype=np.uint8)[None, :]   # glossiness in alpha (material.glsl:107)
    nrm = np.zeros((8, 8, 4), np.uint8)
    nrm[..., 0], nrm[..., 1], nrm[..., 2], nrm[..., 3] = 140, 110, 255, 255

    def main(w):
        pos, n, uv, idx = quad(5.0)
        w.node(mesh=w.mesh([w.primitive(pos, idx, n, uv, material=w.material(name="Main Floor", pbrMetallicRoughness={"baseColorFactor": [0.7, 0.7, 0.75, 1], "metallicFactor": 0.0, "roughnessFactor": 0.8}))]))
        w.doc["extensions"] = {"KHR_lights_punctual": {"lights": [{"type": "directional", "color": [1, 1, 1], "intensity": 2.5}]}}
        w.node(rotation=[-math.sin(math.pi / 4), 0, 0, math.cos(math.pi / 4)], extensions={"KHR_lights_punctual": {"light": 0}})
        w.doc["cameras"] = [{"type": "perspective", "perspective": {"yfov": 0.8, "znear": 0.05, "zfar": 200.0, "aspectRatio": 1.5}}]
        w.node(translation=[0.0, 2.0, 6.0], camera=0)

    def curtains(w):
        td = w.image_texture(_png_bytes(pkg, tmp_path, diffuse, "d.png"), "view", tmp_path, "d.png")
        ts = w.image_texture(_png_bytes(pkg, tmp_path, spec_gloss, "s.png"), "view", tmp_path, "s.png")
        tn = w.image_texture(_png_bytes(pkg, tmp_path, nrm, "n.png"), "view", tmp_path, "n.png")
        sg = w.material(name="Curtain SG", normalTexture={"index": tn},
                        extensions={"KHR_materials_pbrSpecularGlossiness": {"diffuseFactor": [0.9, 0.8, 0.7, 1.0], "diffuseTexture": {"index": td},
                                                                            "specularFactor": [0.8, 0.7, 0.6], "glossinessFactor": 0.9,
                                                                            "specularGlossinessTexture": {"index": ts}}})
        cp, cn, ci = cube(0.6)
        uv = np.float32([[(k % 4) in (1, 2), (k % 4) in (2, 3)] for k in range(len(cp))])
        w.node(mesh=w.mesh([w.primitive(cp, ci, cn, uv, material=sg)]), translation=[-1.2, 0.6, 0.0])

    def ivy(w):
        cp, cn, ci = cube(0.4)
        w.node(mesh=w.mesh([w.primitive(cp, ci, cn, material=w.material(name="Ivy", pbrMetallicRoughness={"baseColorFactor": [0.2, 0.6, 0.25, 1], "metallicFactor": 0.0}))]),
               translation=[1.3, 0.4, 0.5])

    paths = [_component(pkg, tmp_path, "main.gltf", main), _component(pkg, tmp_path, "curtains.glb", curtains), _component(pkg, tmp_path, "ivy.gltf", ivy)]
    sky = np.zeros((16, 32, 4), np.float32)
    sky[..., 0], sky[..., 1], sky[..., 2], sky[..., 3] = 0.3, 0.5, np.linspace(0.2, 1.4, 16, dtype=np.float32)[:, None], 1.0
    pkg.write_image(tmp_path / "sky.hdr", sky, pkg.OUTPUT_HDR)
    return paths


def _describe(tmp_path, **fields):
    import json

    (tmp_path / "scene.json").write_text(json.dumps(fields))
    return "description:@" + str(tmp_path / "scene.json")


def test_scene_description_combines_components_sky_and_flags(pkg, orc, tmp_path):
    paths = _description_assets(pkg, tmp_path)
    parts = [pkg.Scene("file:" + str(p)).triangle_count for p in paths]
    name = _describe(tmp_path, components=[p.name for p in paths], skybox="sky.hdr", dxNormalTextures=True, forceFullTextureSize=True)
    s = pkg.Scene(name)
    d = s.desc
    assert s.triangle_count == sum(parts) == 2 + 12 + 12
    assert d.skyboxKind == 1 and d.dxNormalTextures == 1 and d.forceFullTextureSize == 1   # Skybox2D (equirectangular .hdr)
    assert d.specularGlossinessMaterialCount == 1 and d.metallicRoughnessMaterialCount >= 2 and d.textureCount == 3
    sg = np.frombuffer((C.c_uint8 * 96).from_address(d.specularGlossinessMaterials), np.uint32)
    assert (sg[18:23] == [4, 9, 10, 11, 11]).all()   # emissive default; diffuse, normal; specular and glossiness share the SG texture
    assert np.allclose(sg.view(np.float32)[4:12], [0.9, 0.8, 0.7, 1.0, 0.8, 0.7, 0.6, 0.9])
    # one light, one camera came with the main component; a finite render whose misses see the sky file
    W, H = 96, 64
    s.set_active_camera(0)
    img, st = orc.OracleScene(d).render(s.uniform(W, H, bounces=3, sample_count=4), s.lights, W, H)
    assert np.isfinite(img).all() and st.retries == 0
    top = img[:6, :, :3].mean(axis=(0, 1)) / 4
    assert top[2] > 0.1 and abs(top[0] / top[1] - 0.6) < 0.15   # hdrToLdr keeps the hue of the (0.3, 0.5, z) sky
    # same description without flags: they are off; a missing component is dropped, a missing sky too
    s2 = pkg.Scene(_describe(tmp_path, components=[paths[0].name, "not_there.gltf", paths[2].name], skybox="no_sky.hdr"))
    assert s2.triangle_count == parts[0] + parts[2] and s2.desc.skyboxKind == 0 and s2.desc.dxNormalTextures == 0 and s2.desc.forceFullTextureSize == 0
    with pytest.raises(pkg.PtxError):
        pkg.Scene(_describe(tmp_path, components=["a.gltf", "b.gltf"]))   # "Entire scene not found"
    with pytest.raises(pkg.PtxError):
        pkg.Scene(_describe(tmp_path, components=[paths[0].name], mapping="bogus"))
    # the ORCA slot remap (ExampleScenes.cpp:113-118): every material becomes MetallicRoughness, roughness and metalness
    # both read the texture assimp files under "specular" -- for a glTF SG material its specularGlossinessTexture
    s3 = pkg.Scene(_describe(tmp_path, components=[paths[1].name], mapping="orca"))
    d3 = s3.desc
    assert d3.specularGlossinessMaterialCount == 0
    mr = np.frombuffer((C.c_uint8 * (96 * d3.metallicRoughnessMaterialCount)).from_address(d3.metallicRoughnessMaterials), np.uint32).reshape(-1, 24)
    curtain = mr[[r[22] >= 9 for r in mr]][0]   # dwords 19..23: emissive, colour, normal, roughness, metallic
    assert curtain[22] == curtain[23] and curtain[21] >= 9 and curtain[21] != curtain[22]


@pytest.mark.gpu
def test_described_scene_with_specular_glossiness_renders_like_the_oracle(pkg, orc, tmp_path):
    paths = _description_assets(pkg, tmp_path)
    scene = pkg.Scene(_describe(tmp_path, components=[p.name for p in paths], skybox="sky.hdr", dxNormalTextures=True, forceFullTextureSize=True))
    scene.set_active_camera(0)
    W, H = 192, 128
    r = pkg.Renderer()
    r.upload(scene)
    r.resize(W, H)
    ref = np.zeros((H, W, 4), np.float32)
    osc = orc.OracleScene(scene.desc)
    for f in range(3):
        u = scene.uniform(W, H, bounces=5, total_samples=f)
        r.render(u, scene.lights)
        _, ost = osc.render(u, scene.lights, W, H, accum=ref)
        st = r.stats()
        assert (st.segments, st.shadowRays) == (ost.segments, ost.shadowRays)
    img = r.readback()
    r.close()
    assert (img.view(np.uint32) == ref.view(np.uint32)).all()
