"""Row N4: the output stage -- postprocess.comp -> bloom chain -> composition.comp -> toneMapping.comp ->
sRGB8 / RGBA32F output image -> PNG / TGA / HDR files, plus checkpoint / resume of the running sum.
The per-pixel shader arithmetic is pinned by golden vectors (test_oracle_golden.py / test_gpu_parity.py
pick the three functions up automatically); here: properties of the oracle, HIP == oracle, file round trips."""
import struct
import zlib

import numpy as np
import pytest


def _synthetic_sum(h, w, samples, seed=0):
    rng = np.random.default_rng(seed)
    img = np.zeros((h, w, 4), np.float32)
    yy, xx = np.mgrid[0:h, 0:w]
    img[..., 0] = 0.2 + 0.6 * xx / w
    img[..., 1] = 0.1 + 0.5 * yy / h
    img[..., 2] = 0.3
    img[h // 3:h // 3 + 6, w // 2:w // 2 + 6, :3] = 40.0  # a hot spot that blooms
    img[..., :3] += rng.uniform(0, 0.02, (h, w, 3))
    img[2, 3, 0] = np.nan   # -> (5000, 0, 0) marker, postprocess.comp:24-25
    img[4, 5, 1] = np.inf   # -> (0, 5000, 0) marker, :26-27
    img[..., :3] *= samples
    img[..., 3] = 1.0
    return img


def _decode_png(data: bytes):
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, w, h = 8, b"", 0, 0
    while pos < len(data):
        n, typ = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])[0] == (zlib.crc32(typ + body) & 0xFFFFFFFF), typ
        if typ == b"IHDR":
            w, h, depth, ctype = struct.unpack(">IIBB", body[:10])
            assert (depth, ctype) == (8, 6)
        elif typ == b"IDAT":
            idat += body
        pos += 12 + n
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, w * 4 + 1)
    assert (raw[:, 0] == 1).all()  # Sub filter
    rows = raw[:, 1:].reshape(h, w, 4).astype(np.uint32)
    return (np.cumsum(rows, axis=1) & 255).astype(np.uint8)


def test_oracle_output_stage_properties(orc):
    acc = _synthetic_sum(72, 128, samples=16)
    lin = orc.postprocess(acc, 16, exposure=1.0, bloom_threshold=1.0, bloom_intensity=1.0)
    assert lin.shape == acc.shape and np.isfinite(lin).all() and (lin[..., 3] == 1).all()
    assert lin[..., :3].min() >= 0 and lin[..., :3].max() <= 1  # 1 - exp(-c)
    assert (lin.astype(np.float16).astype(np.float32) == lin).all()  # every value is a binary16 value
    # markers: NaN -> red, Inf -> green (saturated after tone mapping; without bloom they stay pure)
    off = orc.postprocess(acc, 16, bloom_intensity=0.0)
    assert off[2, 3, 0] == 1 and off[2, 3, 1] == 0 and off[4, 5, 1] == 1 and off[4, 5, 0] == 0
    # bloom: the hot spot leaks light into its neighbourhood only when the intensity is non-zero
    hdr_on = orc.postprocess(acc, 16, bloom_intensity=1.0, tone_mapping=1)
    hdr_off = orc.postprocess(acc, 16, bloom_intensity=0.0, tone_mapping=1)
    ring = (slice(72 // 3 - 6, 72 // 3 - 2), slice(128 // 2, 128 // 2 + 6))
    assert (hdr_on[ring][..., :3] > hdr_off[ring][..., :3]).all()
    far = (slice(60, 70), slice(0, 20))
    assert np.abs(hdr_on[far][..., :3] - hdr_off[far][..., :3]).max() < 0.05
    assert (lin[..., :3] >= off[..., :3]).all()
    # HDR mode passes the composited colour through: without bloom it is sum / N * exposure (in binary16)
    want = (acc[..., :3] / np.float32(16) * np.float32(2.0)).astype(np.float32)
    got = orc.postprocess(acc, 16, exposure=2.0, bloom_intensity=0.0, tone_mapping=1)
    ok = np.isfinite(want).all(axis=-1)
    with np.errstate(over="ignore"):
        assert (got[ok][:, :3] == want[ok].astype(np.float16).astype(np.float32)).all()
    # sRGB8 encode: monotone, 0 -> 0, 1 -> 255
    ramp = np.zeros((1, 256, 4), np.float32)
    ramp[0, :, :3] = (np.arange(256) / 255.0)[:, None]
    ramp[..., 3] = 1
    enc = orc.encode_output(ramp, 0)
    assert enc[0, 0, 0] == 0 and enc[0, 255, 0] == 255 and (np.diff(enc[0, :, 0].astype(int)) >= 0).all() and (enc[..., 3] == 255).all()
    # tiny images skip the bloom chain instead of underflowing the mip count (Renderer.cpp:955-956)
    small = orc.postprocess(_synthetic_sum(6, 7, 4)[:, :, :], 4)
    assert np.isfinite(small).all()


def test_image_writers_round_trip(pkg, tmp_path):
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (37, 53, 4), dtype=np.uint8)
    img[5:20, 5:40] = [10, 200, 30, 255]  # a flat run for the LZ77 matcher
    p = tmp_path / "a.png"
    pkg.write_image(p, img, pkg.OUTPUT_PNG)
    assert (_decode_png(p.read_bytes()) == img).all()
    smooth = np.zeros((64, 64, 4), np.uint8)
    smooth[..., 0] = np.arange(64)[None, :] * 3
    smooth[..., 3] = 255
    pkg.write_image(p, smooth, pkg.OUTPUT_PNG)
    assert (_decode_png(p.read_bytes()) == smooth).all() and p.stat().st_size < smooth.nbytes // 8  # it does compress
    t = tmp_path / "a.tga"
    pkg.write_image(t, img, pkg.OUTPUT_TGA)
    raw = t.read_bytes()
    assert raw[2] == 2 and struct.unpack("<HH", raw[12:16]) == (53, 37) and raw[16] == 32 and raw[17] == 0x28
    assert (np.frombuffer(raw[18:], np.uint8).reshape(37, 53, 4)[..., [2, 1, 0, 3]] == img).all()
    f = rng.uniform(0, 20, (9, 11, 4)).astype(np.float32)
    f[0, 0, :3] = 0
    h = tmp_path / "a.hdr"
    pkg.write_image(h, f, pkg.OUTPUT_HDR)
    raw = h.read_bytes()
    head, body = raw.split(b"\n\n-Y 9 +X 11\n")
    assert head.startswith(b"#?RADIANCE") and b"FORMAT=32-bit_rle_rgbe" in head
    rgbe = np.frombuffer(body, np.uint8).reshape(9, 11, 4).astype(np.float32)
    dec = rgbe[..., :3] * (2.0 ** (rgbe[..., 3:] - 136))
    assert (dec[0, 0] == 0).all() and np.abs(dec - f[..., :3]).max() <= f[..., :3].max() / 128
    # JPEG: lossy -- a smooth image with an edge comes back within a few levels (quality 90, 4:2:0), a noise image decodes
    # to the right size; the file is baseline (SOF0) with its own Huffman tables
    yy, xx = np.mgrid[0:70, 0:93]
    pic = np.zeros((70, 93, 4), np.uint8)
    pic[..., 0] = 40 + xx * 2
    pic[..., 1] = 200 - yy * 2
    pic[..., 2] = np.where(xx > 45, 220, 30)
    pic[..., 3] = 255
    j = tmp_path / "a.jpg"
    pkg.write_image(j, pic, pkg.OUTPUT_JPG)
    raw = j.read_bytes()
    assert raw[:2] == b"\xff\xd8" and raw[-2:] == b"\xff\xd9" and b"\xff\xc0" in raw and raw.count(b"\xff\xc4") >= 4
    dec, ch = pkg.decode_image(raw)
    err = dec[..., :3].astype(np.float64) - pic[..., :3]
    psnr = 10 * np.log10(255.0 ** 2 / np.mean(err ** 2))
    assert dec.shape == pic.shape and ch == 3 and psnr > 32, psnr
    assert len(raw) < pic.nbytes // 6
    pkg.write_image(j, img, pkg.OUTPUT_JPG)
    dec, _ = pkg.decode_image(j.read_bytes())
    assert dec.shape == img.shape
    # checkpoint
    acc = rng.uniform(0, 5, (12, 7, 4)).astype(np.float32)
    c = tmp_path / "sum.ptxacc"
    pkg.save_checkpoint(c, acc, 37)
    back, n = pkg.load_checkpoint(c)
    assert n == 37 and (back.view(np.uint32) == acc.view(np.uint32)).all()
    with pytest.raises(pkg.PtxError):
        pkg.load_checkpoint(t)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(72, 128), (270, 480), (9, 13), (1, 1), (1080, 1920), (2160, 3840)])  # the last two: 8 and 9 bloom levels
def test_output_stage_matches_oracle(pkg, orc, shape):
    h, w = shape
    acc = _synthetic_sum(h, w, samples=8, seed=h) if h > 8 else np.random.default_rng(0).uniform(0, 30, (h, w, 4)).astype(np.float32)
    r = pkg.Renderer()
    r.resize(w, h)
    r.write_accumulation(acc)
    assert (r.readback().view(np.uint32) == acc.view(np.uint32)).all()
    for tone, bi, exposure in ((0, 1.0, 1.0), (1, 0.35, 2.5), (0, 0.0, 0.7)):
        r.postprocess(8, exposure=exposure, bloom_threshold=0.8, bloom_intensity=bi, tone_mapping=tone)
        ref = orc.postprocess(acc, 8, exposure=exposure, bloom_threshold=0.8, bloom_intensity=bi, tone_mapping=tone)
        got = r.read_output(pkg.OUTPUT_RGBA32F)
        assert (got.view(np.uint32) == ref.view(np.uint32)).all(), (tone, bi)
        assert (r.read_output(pkg.OUTPUT_RGBA8_SRGB) == orc.encode_output(ref, 0)).all()
    r.close()


@pytest.mark.gpu
def test_render_to_png_and_resume_from_checkpoint(pkg, orc, tmp_path):
    scene = pkg.Scene("default")
    W, H = 160, 90
    a = pkg.Renderer()
    a.upload(scene)
    a.resize(W, H)
    a.render_frames(scene.uniform(W, H, bounces=4), scene.lights, 0, 4)
    pkg.save_checkpoint(tmp_path / "half.ptxacc", a.readback(), 4)
    a.render_frames(scene.uniform(W, H, bounces=4), scene.lights, 4, 4)
    full = a.readback()
    # resume in a fresh renderer: restore the sum, continue at TotalSamples = 4
    b = pkg.Renderer()
    b.upload(scene)
    b.resize(W, H)
    acc, n = pkg.load_checkpoint(tmp_path / "half.ptxacc")
    b.write_accumulation(acc)
    b.render_frames(scene.uniform(W, H, bounces=4), scene.lights, n, 4)
    assert (b.readback().view(np.uint32) == full.view(np.uint32)).all()
    b.postprocess(8)
    png = tmp_path / "default.png"
    pkg.write_image(png, b.read_output(), pkg.OUTPUT_PNG)
    img = _decode_png(png.read_bytes())
    assert img.shape == (H, W, 4) and (img == orc.encode_output(orc.postprocess(full, 8), 0)).all()
    assert 20 < img[..., :3].mean() < 235  # a plausible exposure, not black / white
    a.close()
    b.close()
