"""Row N2: binary FBX through the host importer (FbxReader.cpp + the SceneImporter pipeline).  The files are written by
tests/fbx_util.py; what assimp's FBX converter would report is restated in the expectations."""
import ctypes as C
import json
import math

import numpy as np
import pytest

import fbx_util as F
import util

CUBE_POINTS = np.float64([[-1, -1, -1], [1, -1, -1], [1, 1, -1], [-1, 1, -1], [-1, -1, 1], [1, -1, 1], [1, 1, 1], [-1, 1, 1]]) * 0.5
CUBE_QUADS = [(0, 3, 2, 1), (4, 5, 6, 7), (0, 1, 5, 4), (2, 3, 7, 6), (1, 2, 6, 5), (0, 4, 7, 3)]  # outward winding
CUBE_NORMALS = np.float64([[0, 0, -1], [0, 0, 1], [0, -1, 0], [0, 1, 0], [1, 0, 0], [-1, 0, 0]])


def _cube_geometry(oid, by_polygon_materials):
    poly = np.int32([i if k < 3 else ~i for q in CUBE_QUADS for k, i in enumerate(q)])
    normals = np.repeat(CUBE_NORMALS, 4, axis=0).reshape(-1)
    uv = np.float64([[0, 0], [1, 0], [1, 1], [0, 1]]).reshape(-1)
    uv_index = np.int32([0, 1, 2, 3] * 6)
    return F.obj("Geometry", oid, "BoxGeo", "Geometry", "Mesh", [
        F.node("Vertices", [F.Z(CUBE_POINTS.reshape(-1))]),
        F.node("PolygonVertexIndex", [poly]),
        F.node("LayerElementNormal", [0], [F.node("MappingInformationType", ["ByPolygonVertex"]), F.node("ReferenceInformationType", ["Direct"]),
                                           F.node("Normals", [normals])]),
        F.node("LayerElementUV", [0], [F.node("MappingInformationType", ["ByPolygonVertex"]), F.node("ReferenceInformationType", ["IndexToDirect"]),
                                       F.node("UV", [uv]), F.node("UVIndex", [F.Z(uv_index)])]),
        F.node("LayerElementMaterial", [0], [F.node("MappingInformationType", ["ByPolygon" if by_polygon_materials else "AllSame"]),
                                             F.node("Materials", [np.int32([0, 0, 0, 1, 1, 1]) if by_polygon_materials else np.int32([0])])]),
    ])


def _floor_geometry(oid):
    pts = np.float64([[-4, 0, 4], [4, 0, 4], [4, 0, -4], [-4, 0, -4]])  # one quad, normal +y, control-point normals and UVs
    return F.obj("Geometry", oid, "FloorGeo", "Geometry", "Mesh", [
        F.node("Vertices", [pts.reshape(-1)]),
        F.node("PolygonVertexIndex", [np.int32([0, 1, 2, ~3])]),
        F.node("LayerElementNormal", [0], [F.node("MappingInformationType", ["ByVertice"]), F.node("ReferenceInformationType", ["Direct"]),
                                           F.node("Normals", [np.float64([[0, 1, 0]] * 4).reshape(-1)])]),
        F.node("LayerElementUV", [0], [F.node("MappingInformationType", ["ByVertice"]), F.node("ReferenceInformationType", ["Direct"]),
                                       F.node("UV", [np.float64([[0, 0], [1, 0], [1, 1], [0, 1]]).reshape(-1)])]),
        F.node("LayerElementMaterial", [0], [F.node("MappingInformationType", ["AllSame"]), F.node("Materials", [np.int32([0])])]),
    ])


def _write_scene(pkg, tmp_path, version=7400, name="scene.fbx", animated=False):
    for fn, rgb in (("floor_d.png", (200, 180, 60)), ("floor_s.png", (0, 120, 30)), ("floor_n.png", (128, 128, 255))):
        img = np.zeros((4, 4, 4), np.uint8)
        img[..., :3], img[..., 3] = rgb, 255
        pkg.write_image(tmp_path / fn, img, pkg.OUTPUT_PNG)
    ROOT, BOX, FLOOR, LAMP, SUN, CAM = 100, 101, 102, 103, 104, 105
    objects = F.node("Objects", [], [
        F.obj("Model", ROOT, "Group", "Model", "Null", [F.p70(Lcl_Translation=(1.0, 0.5, 0.0), Lcl_Rotation=(0.0, 45.0, 0.0))]),
        F.obj("Model", BOX, "Box", "Model", "Mesh", [F.p70(Lcl_Scaling=(1.0, 2.0, 1.0), GeometricTranslation=(0.0, 0.25, 0.0))]),
        F.obj("Model", FLOOR, "Floor", "Model", "Mesh", [F.p70(PreRotation=(0.0, 0.0, 0.0))]),
        F.obj("Model", LAMP, "Lamp", "Model", "Light", [F.p70(Lcl_Translation=(0.0, 3.0, 1.0))]),
        F.obj("Model", SUN, "Sun", "Model", "Light", [F.p70(Lcl_Rotation=(0.0, 0.0, 0.0))]),
        F.obj("Model", CAM, "Cam", "Model", "Camera", [F.p70(Lcl_Translation=(-6.0, 2.0, 0.0))]),
        _cube_geometry(200, True),
        _floor_geometry(201),
        F.obj("Material", 300, "Floor Phong", "Material", "", [F.p70(DiffuseColor=(0.5, 0.6, 0.7), ShininessExponent=20.0, SpecularFactor=0.5,
                                                                       EmissiveColor=(0.1, 0.2, 0.3), EmissiveFactor=2.0)]),
        F.obj("Material", 301, "Box A", "Material", "", [F.p70(DiffuseColor=(1.0, 0.0, 0.0), Shininess=5.0)]),
        F.obj("Material", 302, "Box B", "Material", "", [F.p70(DiffuseColor=(0.0, 1.0, 0.0), Shininess=5.0)]),
        F.obj("Texture", 400, "d", "Texture", "", [F.node("RelativeFilename", ["floor_d.png"])]),
        F.obj("Texture", 401, "s", "Texture", "", [F.node("RelativeFilename", ["sub\\..\\floor_s.png"])]),
        F.obj("Texture", 402, "n", "Texture", "", [F.node("FileName", ["floor_n.png"])]),
        F.obj("NodeAttribute", 500, "LampAttr", "NodeAttribute", "Light", [F.p70(LightType=0, Color=(1.0, 0.5, 0.25), Intensity=800.0)]),
        F.obj("NodeAttribute", 501, "SunAttr", "NodeAttribute", "Light", [F.p70(LightType=1, Color=(1.0, 1.0, 1.0), Intensity=200.0)]),
        F.obj("NodeAttribute", 502, "CamAttr", "NodeAttribute", "Camera", [F.p70(FieldOfView=60.0, AspectWidth=3.0, AspectHeight=2.0, NearPlane=0.1, FarPlane=500.0)]),
    ])
    extra_objects, extra_links = [], []
    if animated:
        # one stack: the group turns about y from 45 to 135 degrees and slides along x from 1 to 3, over two seconds
        second = 46186158000
        extra_objects = [
            F.obj("AnimationStack", 600, "Take", "AnimStack", ""), F.obj("AnimationLayer", 601, "Base", "AnimLayer", ""),
            F.obj("AnimationCurveNode", 610, "R", "AnimCurveNode", ""), F.obj("AnimationCurveNode", 611, "T", "AnimCurveNode", ""),
            F.obj("AnimationCurve", 620, "", "AnimCurve", "", [F.node("KeyTime", [np.int64([0, 2 * second])]), F.node("KeyValueFloat", [np.float32([45.0, 135.0])])]),
            F.obj("AnimationCurve", 621, "", "AnimCurve", "", [F.node("KeyTime", [np.int64([0, second, 2 * second])]), F.node("KeyValueFloat", [np.float32([1.0, 2.0, 3.0])])]),
        ]
        extra_links = [F.oo(601, 600), F.oo(610, 601), F.oo(611, 601), F.op(610, ROOT, "Lcl Rotation"), F.op(611, ROOT, "Lcl Translation"),
                       F.op(620, 610, "d|Y"), F.op(621, 611, "d|X")]
    objects[2].extend(extra_objects)
    connections = F.node("Connections", [], extra_links + [
        F.oo(ROOT, 0), F.oo(FLOOR, 0), F.oo(LAMP, 0), F.oo(SUN, 0), F.oo(CAM, 0), F.oo(BOX, ROOT),
        F.oo(200, BOX), F.oo(301, BOX), F.oo(302, BOX), F.oo(201, FLOOR), F.oo(300, FLOOR),
        F.op(400, 300, "DiffuseColor"), F.op(401, 300, "SpecularColor"), F.op(402, 300, "NormalMap"),
        F.oo(500, LAMP), F.oo(501, SUN), F.oo(502, CAM),
    ])
    header = F.node("FBXHeaderExtension", [], [F.node("FBXVersion", [version])])
    F.write(tmp_path / name, [header, objects, connections], version)
    return tmp_path / name


def _describe(tmp_path, **fields):
    (tmp_path / "scene.json").write_text(json.dumps(fields))
    return "description:@" + str(tmp_path / "scene.json")


@pytest.mark.parametrize("version", [7400, 7500])
def test_binary_fbx_import(pkg, orc, tmp_path, version):
    path = _write_scene(pkg, tmp_path, version)
    # every classic FBX material carries a shininess -> assimp's Phong model -> the reference's `case Phong:` runs into
    # `default: throw` (SceneImporter.cpp:390-393): such a file loads only under a mapping that forces another model
    with pytest.raises(pkg.PtxError, match="Unsupported material type"):
        pkg.Scene("file:" + str(path))
    s = pkg.Scene(_describe(tmp_path, components=[path.name], mapping="orca"))  # ExampleScenes.cpp:113-141
    d = s.desc
    a = util.desc_arrays(d)
    assert s.triangle_count == 2 + 12
    assert d.instanceCount == 1 and d.meshCount == 3  # the floor, and the box split by its two materials
    assert sorted(int(g["IndexLength"]) for g in a["geometries"]) == [6, 18, 18]
    assert d.metallicRoughnessMaterialCount >= 3 and d.specularGlossinessMaterialCount == 0 and d.phongMaterialCount == 0
    mr = np.frombuffer((C.c_uint8 * (96 * d.metallicRoughnessMaterialCount)).from_address(d.metallicRoughnessMaterials), np.uint32).reshape(-1, 24)
    mrf = mr.view(np.float32)
    floor = int([m for m in a["meshes"] if a["geometries"][m["GeometryIndex"]]["IndexLength"] == 6][0]["MaterialId"]) >> 8
    # LoadMetallicRoughnessMaterial on an FBX aiMaterial: no base colour / factors -> white, 1, 1; emissive = colour x factor;
    # textures: colour <- DIFFUSE, normal <- NORMALS, roughness and metalness <- SPECULAR (the ORCA remap)
    assert np.allclose(mrf[floor, 4:8], 1.0) and mrf[floor, 8] == 1.0 and mrf[floor, 9] == 1.0
    assert np.allclose(mrf[floor, 0:3], [0.2, 0.4, 0.6]) and mrf[floor, 3] == 1.0
    ids = mr[floor, 19:24]
    assert ids[0] == 4 and ids[1] >= 9 and ids[2] >= 9 and ids[3] == ids[4] >= 9 and len({int(ids[1]), int(ids[2]), int(ids[3])}) == 3
    assert d.textureCount == 3
    # geometry in world space through the oracle
    osc = orc.OracleScene(d, build_bvh=False)
    hit = osc.trace_closest(np.float32([[-3, 5, -3, 1e-5, 0, -1, 0, 1e4], [1.0, 5, 0.0, 1e-5, 0, -1, 0, 1e4]]))
    # the box: half height 0.5, shifted up by the geometric 0.25, scaled x2 in y, on a group at y = 0.5 -> top at 0.5 + 2 * 0.75 = 2
    assert np.allclose(hit["t"], [5.0, 3.0], atol=1e-5)
    c = math.cos(math.pi / 4)
    edge = osc.trace_closest(np.float32([[1 + 0.5 * c * 2 - 0.02, 5, 0.0, 1e-5, 0, -1, 0, 1e4], [1 + 0.5 * c * 2 + 0.02, 5, 0.0, 1e-5, 0, -1, 0, 1e4]]))
    assert np.allclose(edge["t"], [3.0, 5.0], atol=1e-5), "the box is turned by 45 degrees: its corner points along +x"
    # aiProcess_FlipUVs on the floor's control-point UVs
    fg = [g for g in a["geometries"] if g["IndexLength"] == 6][0]
    fv = a["vertices"][fg["VertexOffset"]:fg["VertexOffset"] + 4]
    assert np.allclose(fv[:, 0:3], [[-4, 0, 4], [4, 0, 4], [4, 0, -4], [-4, 0, -4]]) and np.allclose(fv[:, 3:5], [[0, 1], [1, 1], [1, 0], [0, 0]])
    assert np.allclose(fv[:, 5:8], [0, 1, 0])
    # aiProcess_CalcTangentSpace on those coordinates: u grows along +x, the flipped v along +z
    assert np.allclose(fv[:, 8:11], [1, 0, 0], atol=1e-6) and np.allclose(fv[:, 11:14], [0, 0, 1], atol=1e-6)
    # lights: Intensity is a percentage; an FBX light shines along its local -Y
    L = s.lights
    assert L.LightCount == 1 and np.allclose(list(L.Lights[0].Position), [0, 3, 1]) and np.allclose(list(L.Lights[0].Color), [8, 4, 2])
    assert np.allclose(list(L.Directional.Direction), [0, -1, 0], atol=1e-6) and np.allclose(list(L.Directional.Color), [2, 2, 2])
    # the camera: at its node, looking along the node's +X (towards the scene), vertical angle from the horizontal 60 degrees at 3:2
    s.set_active_camera(0)
    u = s.uniform(48, 32, bounces=3, sample_count=2)
    view = np.array(u.ViewInverse, np.float32).reshape(4, 4)
    assert np.allclose(view[3, :3], [-6, 2, 0], atol=1e-5) and np.allclose(view[2, :3], [1, 0, 0], atol=1e-5)  # LookAtLH: the view z axis is the forward direction
    proj = np.array(u.ProjInverse, np.float32).reshape(4, 4)
    assert np.isclose(abs(proj[1, 1]), math.tan(math.radians(60) / 2) / 1.5, rtol=1e-4)
    img, st = osc.render(u, L, 48, 32)
    assert np.isfinite(img).all() and st.shadowRays > 0 and st.segments > 48 * 32 * 2


def test_fbx_reader_rejects_malformed_files(pkg, tmp_path):
    path = _write_scene(pkg, tmp_path)
    good = path.read_bytes()
    name = _describe(tmp_path, components=["bad.fbx"], mapping="orca")
    cases = {
        "ascii": b"; FBX 7.4.0 project file\n",
        "truncated": good[: len(good) // 2],
        "end offset past the file": good[:27] + (2**31).to_bytes(4, "little") + good[31:],
        "property list longer than the record": good[:35] + (2**30).to_bytes(4, "little") + good[39:],
    }
    for label, data in cases.items():
        (tmp_path / "bad.fbx").write_bytes(data)
        with pytest.raises(pkg.PtxError):
            pkg.Scene(name)
    # random corruption never crashes: it either loads or raises
    rng = np.random.default_rng(5)
    for _ in range(60):
        b = bytearray(good)
        for k in rng.integers(27, len(b), 6):
            b[k] = rng.integers(0, 256)
        (tmp_path / "bad.fbx").write_bytes(bytes(b))
        try:
            pkg.Scene(name)
        except pkg.PtxError:
            pass


@pytest.mark.gpu
def test_fbx_scene_renders_like_the_oracle(pkg, orc, tmp_path, gpu_renderer):
    path = _write_scene(pkg, tmp_path)
    s = pkg.Scene(_describe(tmp_path, components=[path.name], mapping="orca", dxNormalTextures=True))
    s.set_camera_pose((0.5, 2.5, 6.0), (0.05, -0.3, -1.0))
    W, H = 128, 80
    gpu_renderer.upload(s)
    gpu_renderer.resize(W, H)
    gpu_renderer.reset()
    ref = np.zeros((H, W, 4), np.float32)
    osc = orc.OracleScene(s.desc)
    for f in range(3):
        u = s.uniform(W, H, bounces=4, total_samples=f)
        gpu_renderer.render(u, s.lights)
        osc.render(u, s.lights, W, H, accum=ref)
    img = gpu_renderer.readback()
    assert (img.view(np.uint32) == ref.view(np.uint32)).all() and img[..., :3].max() > 0.05


def test_wavefront_obj_import(pkg, orc, tmp_path):
    """OBJ + MTL through the same pipeline.  assimp's OBJ importer gives every material a shininess, so -- like FBX -- a file
    loads only under a mapping that forces another model than Phong (the reference's missing `break`)."""
    img = np.zeros((4, 4, 4), np.uint8)
    img[..., :3], img[..., 3] = (90, 160, 220), 255
    pkg.write_image(tmp_path / "kd.png", img, pkg.OUTPUT_PNG)
    (tmp_path / "textures").mkdir()
    pkg.write_image(tmp_path / "textures" / "kd.png", img, pkg.OUTPUT_PNG)
    (tmp_path / "scene.mtl").write_text(
        "newmtl Floor\nKd 0.5 0.6 0.7\nKe 0.1 0.2 0.3\nNs 40\nmap_Kd -bm 1.0 kd.png\nmap_Ks textures\\kd.png\nmap_bump kd.png\n"
        "newmtl Lid\nKd 1 0 0\nd 0.5\n")
    (tmp_path / "scene.obj").write_text(
        "# a floor quad with uvs and normals, and a box top referenced by negative indices\n"
        "mtllib scene.mtl\n"
        "v -4 0 4\nv 4 0 4\nv 4 0 -4\nv -4 0 -4\n"
        "vt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\n"
        "vn 0 1 0\n"
        "o Floor\nusemtl Floor\nf 1/1/1 2/2/1 3/3/1 4/4/1\n"
        "o Lid\nusemtl Lid\n"
        "v -1 1 1\nv 1 1 1\nv 1 1 -1\nv -1 1 -1\nv 0 2 0\n"
        "f -5 -4 -1\nf -4 -3 -1\nf -3 -2 \\\n -1\nf -2 -5 -1\n"
        "usemtl Missing\nf 5//1 6//1 7//1\n")
    with pytest.raises(pkg.PtxError, match="Unsupported material type"):
        pkg.Scene("file:" + str(tmp_path / "scene.obj"))
    s = pkg.Scene(_describe(tmp_path, components=["scene.obj"], mapping="orca"))
    d = s.desc
    a = util.desc_arrays(d)
    assert s.triangle_count == 2 + 4 + 1 and d.meshCount == 3
    assert d.phongMaterialCount == 0 and d.specularGlossinessMaterialCount == 0
    mr = np.frombuffer((C.c_uint8 * (96 * d.metallicRoughnessMaterialCount)).from_address(d.metallicRoughnessMaterials), np.uint32).reshape(-1, 24)
    floor = int([m for m in a["meshes"] if a["geometries"][m["GeometryIndex"]]["IndexLength"] == 6][0]["MaterialId"]) >> 8
    f = mr.view(np.float32)[floor]
    assert np.allclose(f[0:3], [0.1, 0.2, 0.3]) and np.allclose(f[4:8], 1.0)   # MetallicRoughness under the ORCA mapping: white base colour
    ids = mr[floor, 19:24]
    assert ids[1] >= 9 and ids[3] == ids[4] >= 9 and ids[2] == 1   # map_Kd, map_Ks (backslash path); map_bump is HEIGHT: no slot reads it
    fg = [g for g in a["geometries"] if g["IndexLength"] == 6][0]
    fv = a["vertices"][fg["VertexOffset"]:fg["VertexOffset"] + 4]
    assert np.allclose(fv[:, 3:5], [[0, 1], [1, 1], [1, 0], [0, 0]]) and np.allclose(fv[:, 5:8], [0, 1, 0])   # FlipUVs
    osc = orc.OracleScene(d, build_bvh=False)
    hit = osc.trace_closest(np.float32([[-3, 5, -3, 1e-5, 0, -1, 0, 1e4], [0.0, 5, 0.0, 1e-5, 0, -1, 0, 1e4], [0.5, 5, 0.25, 1e-5, 0, -1, 0, 1e4]]))
    assert np.allclose(hit["t"], [5.0, 3.0, 3.5], atol=1e-5)   # the floor, the pyramid's apex (0, 2, 0), its slope at x = 0.5
    for bad in ("f 1 2 99\n", "f 0 1 2\n", "v 1 2 3\n"):   # index past the end, zero index, no faces at all
        (tmp_path / "bad.obj").write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\n" + bad if bad.startswith("f") else bad)
        with pytest.raises(pkg.PtxError):
            pkg.Scene(_describe(tmp_path, components=["bad.obj"], mapping="orca"))


def test_fbx_animation_stack(pkg, orc, tmp_path):
    """Curves on Lcl Rotation / Lcl Translation of a model: the animated node becomes an instance root (LoadModels), its keys
    arrive as milliseconds at 1000 ticks per second, components without a curve keep the model's own value."""
    path = _write_scene(pkg, tmp_path, animated=True, name="anim.fbx")
    s = pkg.Scene(_describe(tmp_path, components=[path.name], mapping="orca"))
    d = s.desc
    assert d.instanceCount == 2 and s.triangle_count == 2 + 12   # the static floor; the group with the box below it
    assert s.update(0.0)
    it0, _ = s.animation_state()
    box = [i for i in range(2) if abs(it0[i].reshape(3, 4)[1, 3] - 0.5) < 1e-5]
    assert len(box) == 1
    m0 = it0[box[0]].reshape(3, 4)
    c = math.cos(math.pi / 4)
    assert np.allclose(m0[:, 3], [1, 0.5, 0], atol=1e-5) and np.allclose(m0[:, :3], [[c, 0, c], [0, 1, 0], [-c, 0, c]], atol=1e-5)
    s.update(1.0)   # half way: 90 degrees about y, x = 2; y and z keep the model's own translation
    it1, _ = s.animation_state()
    m1 = it1[box[0]].reshape(3, 4)
    assert np.allclose(m1[:, 3], [2, 0.5, 0], atol=1e-4) and np.allclose(m1[:, :3], [[0, 0, 1], [0, 1, 0], [-1, 0, 0]], atol=1e-4)
    # the posed scene through the oracle: the box (2 high on the group) now stands at x = 2
    osc = orc.OracleScene(d, build_bvh=False, instance_transforms=it1)
    hit = osc.trace_closest(np.float32([[2.0, 5, 0.0, 1e-5, 0, -1, 0, 1e4], [0.2, 5, 0.0, 1e-5, 0, -1, 0, 1e4]]))
    assert np.allclose(hit["t"], [3.0, 5.0], atol=1e-4)


def test_fbx_skin_deformer(pkg, orc, tmp_path):
    """Skin / Cluster deformers: bones from the clusters' Models, offset matrices from TransformLink, per control point the
    four largest weights; a curve on a bone bends the strip (the same rig as the glTF test of tests/test_importer.py)."""
    second = 46186158000
    pts = np.float64([[-0.1, 0, 0], [0.1, 0, 0], [-0.1, 1, 0], [0.1, 1, 0], [-0.1, 2, 0], [0.1, 2, 0]]) + np.float64([2, 0, 0])
    geo = F.obj("Geometry", 200, "StripGeo", "Geometry", "Mesh", [
        F.node("Vertices", [pts.reshape(-1)]),
        F.node("PolygonVertexIndex", [np.int32([0, 1, 3, ~2, 2, 3, 5, ~4])]),
        F.node("LayerElementNormal", [0], [F.node("MappingInformationType", ["ByVertice"]), F.node("ReferenceInformationType", ["Direct"]),
                                           F.node("Normals", [np.float64([[0, 0, 1]] * 6).reshape(-1)])]),
    ])

    def bind(t):
        m = np.eye(4)
        m[:3, 3] = t
        return m.T.reshape(-1).copy()   # column-major, translation in elements 12..14

    objects = F.node("Objects", [], [
        F.obj("Model", 100, "Strip", "Model", "Mesh"),
        F.obj("Model", 101, "J0", "Model", "LimbNode", [F.p70(Lcl_Translation=(2.0, 0.0, 0.0))]),
        F.obj("Model", 102, "J1", "Model", "LimbNode", [F.p70(Lcl_Translation=(0.0, 1.0, 0.0))]),
        geo,
        F.obj("Material", 300, "Skin", "Material", "", [F.p70(DiffuseColor=(0.8, 0.3, 0.2))]),
        F.obj("Deformer", 400, "", "Deformer", "Skin"),
        F.obj("Deformer", 401, "", "SubDeformer", "Cluster", [F.node("Indexes", [np.int32([0, 1, 2, 3])]), F.node("Weights", [np.float64([1, 1, 0.5, 0.5])]),
                                                               F.node("Transform", [bind([-2, 0, 0])]), F.node("TransformLink", [bind([2, 0, 0])])]),
        F.obj("Deformer", 402, "", "SubDeformer", "Cluster", [F.node("Indexes", [np.int32([2, 3, 4, 5])]), F.node("Weights", [np.float64([0.5, 0.5, 1, 1])]),
                                                               F.node("Transform", [bind([-2, -1, 0])]), F.node("TransformLink", [bind([2, 1, 0])])]),
        F.obj("AnimationStack", 600, "Take", "AnimStack", ""), F.obj("AnimationLayer", 601, "Base", "AnimLayer", ""),
        F.obj("AnimationCurveNode", 610, "R", "AnimCurveNode", ""),
        F.obj("AnimationCurve", 620, "", "AnimCurve", "", [F.node("KeyTime", [np.int64([0, second, 2 * second])]), F.node("KeyValueFloat", [np.float32([0.0, 90.0, 0.0])])]),
    ])
    connections = F.node("Connections", [], [
        F.oo(100, 0), F.oo(101, 0), F.oo(102, 101), F.oo(200, 100), F.oo(300, 100),
        F.oo(400, 200), F.oo(401, 400), F.oo(402, 400), F.oo(101, 401), F.oo(102, 402),
        F.oo(601, 600), F.oo(610, 601), F.op(610, 102, "Lcl Rotation"), F.op(620, 610, "d|Z"),
    ])
    F.write(tmp_path / "skin.fbx", [objects, connections], 7400)
    s = pkg.Scene(_describe(tmp_path, components=["skin.fbx"], mapping="orca"))
    d = s.desc
    a = util.desc_arrays(d)
    assert d.animatedVertexCount == 6 and d.animatedIndexCount == 12 and s.lib.pth_scene_bone_count(s.handle) == 2
    assert [int(g["IsAnimated"]) for g in a["geometries"]] == [1]
    assert s.update(0.0)
    it0, bn0 = s.animation_state()
    assert np.allclose(bn0[0].reshape(3, 4), np.eye(4)[:3], atol=1e-6) and np.allclose(bn0[1].reshape(3, 4), np.eye(4)[:3], atol=1e-6)
    s.update(0.5)   # J1 at 45 degrees about z, around its own origin (2, 1, 0)
    it1, bn1 = s.animation_state()
    c = math.cos(math.pi / 4)
    assert np.allclose(bn1[1].reshape(3, 4)[:, :3], [[c, -c, 0], [c, c, 0], [0, 0, 1]], atol=1e-5)
    assert np.allclose(bn1[1].reshape(3, 4) @ [2, 1, 0, 1], [2, 1, 0], atol=1e-5)
    osc = orc.OracleScene(d, build_bvh=False, instance_transforms=it1, bones=bn1)
    hit = osc.trace_closest(np.float32([[2.0 - c * 0.9, 1.0 + c * 0.9, 5.0, 1e-5, 0, 0, -1, 1e4], [2.0, 1.9, 5.0, 1e-5, 0, 0, -1, 1e4]]))
    assert hit["tri"][0] != 0xFFFFFFFF and hit["tri"][1] == 0xFFFFFFFF and np.isclose(hit["t"][0], 5.0, atol=1e-4)
