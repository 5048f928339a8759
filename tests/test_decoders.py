"""TextureImporter mirror (rows N1 / N2): the image decoders the reference takes from stb_image / gli, written
out here.  Files are synthesised in the test (zlib-compressed PNGs of every colour type and filter, a
hand-assembled baseline JPEG, TGA, RGBE, BCn blocks) so that no asset is needed."""
import struct
import zlib

import numpy as np
import pytest


def _png(w, h, ctype, depth, rows, palette=None, trns=None, filters=None, level=6, split=1):
    """rows: list of raw (unfiltered) scanline bytes; filters: per-row PNG filter type to apply."""
    samples = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    bpp = max(1, samples * depth // 8)

    def chunk(t, b):
        return struct.pack(">I", len(b)) + t + b + struct.pack(">I", zlib.crc32(t + b) & 0xFFFFFFFF)

    raw = bytearray()
    prev = bytes(len(rows[0]))
    for y, row in enumerate(rows):
        ft = filters[y % len(filters)] if filters else 0
        out = bytearray(len(row))
        for i in range(len(row)):
            a = row[i - bpp] if i >= bpp else 0
            b = prev[i]
            c = prev[i - bpp] if i >= bpp else 0
            if ft == 0:
                p = 0
            elif ft == 1:
                p = a
            elif ft == 2:
                p = b
            elif ft == 3:
                p = (a + b) >> 1
            else:
                q = a + b - c
                pa, pb, pc = abs(q - a), abs(q - b), abs(q - c)
                p = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
            out[i] = (row[i] - p) & 255
        raw.append(ft)
        raw += out
        prev = row
    z = zlib.compress(bytes(raw), level)
    data = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0))
    if palette is not None:
        data += chunk(b"PLTE", bytes(palette))
    if trns is not None:
        data += chunk(b"tRNS", bytes(trns))
    step = (len(z) + split - 1) // split
    for k in range(0, len(z), step):
        data += chunk(b"IDAT", z[k:k + step])
    return data + chunk(b"IEND", b"")


def _adam7_png(w, h, ctype, depth, pixels, filters=None):
    """An interlaced PNG: pixels is (h, w, samples) uint8 for 8-bit types, or (h, w) of 0 / 1 for 1-bit greyscale."""
    def chunk(t, b):
        return struct.pack(">I", len(b)) + t + b + struct.pack(">I", zlib.crc32(t + b) & 0xFFFFFFFF)

    samples = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    bpp = max(1, samples * depth // 8)
    raw = bytearray()
    n = 0
    for x0, y0, dx, dy in ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)):
        sub = pixels[y0::dy, x0::dx]
        if sub.shape[0] == 0 or sub.shape[1] == 0:
            continue
        prev = None
        for line in sub:
            row = bytes(np.packbits(line)) if depth == 1 else line.tobytes()
            ft = filters[n % len(filters)] if filters else 0
            n += 1
            up = prev if prev is not None else bytes(len(row))
            out = bytearray(len(row))
            for i in range(len(row)):
                a = row[i - bpp] if i >= bpp else 0
                b = up[i]
                c = up[i - bpp] if i >= bpp else 0
                q = a + b - c
                pa, pb, pc = abs(q - a), abs(q - b), abs(q - c)
                p = [0, a, b, (a + b) >> 1, a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)][ft]
                out[i] = (row[i] - p) & 255
            raw.append(ft)
            raw += out
            prev = row
    return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 1)) + chunk(b"IDAT", zlib.compress(bytes(raw))) + chunk(b"IEND", b"")


def test_inflate_and_png_variants(pkg):
    rng = np.random.default_rng(0)
    w, h = 37, 23
    rgba = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    rgba[5:15, 3:30] = [9, 200, 77, 128]  # runs -> LZ77 matches, dynamic Huffman blocks
    for level, split in ((0, 1), (1, 3), (9, 1)):  # stored blocks, fixed / dynamic Huffman, split IDAT
        img, ch = pkg.decode_image(_png(w, h, 6, 8, [rgba[y].tobytes() for y in range(h)], filters=[0, 1, 2, 3, 4], level=level, split=split))
        assert ch == 4 and (img == rgba).all()
    rgb = rgba[..., :3].copy()
    img, ch = pkg.decode_image(_png(w, h, 2, 8, [rgb[y].tobytes() for y in range(h)], filters=[4, 3]))
    assert ch == 3 and (img[..., :3] == rgb).all() and (img[..., 3] == 255).all()
    grey = rgba[..., 0].copy()
    img, ch = pkg.decode_image(_png(w, h, 0, 8, [grey[y].tobytes() for y in range(h)], filters=[2]))
    assert ch == 1 and (img[..., 0] == grey).all() and (img[..., 1] == grey).all()
    ga = rgba[..., :2].copy()
    img, ch = pkg.decode_image(_png(w, h, 4, 8, [ga[y].tobytes() for y in range(h)]))
    assert ch == 2 and (img[..., 0] == ga[..., 0]).all() and (img[..., 3] == ga[..., 1]).all()
    # 16 bit -> the high byte; colour key -> alpha 0
    rgb16 = rng.integers(0, 65536, (h, w, 3)).astype(">u2")
    rgb16[2, 2] = [0x1234, 0x5678, 0x9ABC]
    img, ch = pkg.decode_image(_png(w, h, 2, 16, [rgb16[y].tobytes() for y in range(h)], trns=struct.pack(">HHH", 0x1234, 0x5678, 0x9ABC)))
    assert ch == 4 and (img[..., :3] == (rgb16.astype(np.uint16) >> 8)).all() and img[2, 2, 3] == 0 and img[0, 0, 3] == 255
    # palette with transparency, 4 bits per index
    pal = rng.integers(0, 256, (16, 3), dtype=np.uint8)
    idx = rng.integers(0, 16, (h, 38))
    rows = [bytes((idx[y, 0::2] << 4 | idx[y, 1::2]).astype(np.uint8)) for y in range(h)]
    img, ch = pkg.decode_image(_png(38, h, 3, 4, rows, palette=pal.reshape(-1), trns=[0, 128]))
    assert ch == 4 and (img[..., :3] == pal[idx]).all() and (img[..., 3] == np.where(idx == 0, 0, np.where(idx == 1, 128, 255))).all()
    # 1-bit greyscale
    bits = rng.integers(0, 2, (h, 40))
    rows = [bytes(np.packbits(bits[y].astype(np.uint8))) for y in range(h)]
    img, _ = pkg.decode_image(_png(40, h, 0, 1, rows))
    assert (img[..., 0] == bits * 255).all()
    # our own encoder's files decode too (OutputSaver::EncodePng)
    import tempfile, os
    with tempfile.TemporaryDirectory() as d:
        pkg.write_image(os.path.join(d, "x.png"), rgba, pkg.OUTPUT_PNG)
        img, _ = pkg.decode_image(open(os.path.join(d, "x.png"), "rb").read())
        assert (img == rgba).all()
        pkg.write_image(os.path.join(d, "x.tga"), rgba, pkg.OUTPUT_TGA)
        img, ch = pkg.decode_image(open(os.path.join(d, "x.tga"), "rb").read())
        assert ch == 4 and (img == rgba).all()
        f = rng.uniform(0, 50, (9, 12, 4)).astype(np.float32)
        pkg.write_image(os.path.join(d, "x.hdr"), f, pkg.OUTPUT_HDR)
        img, ch = pkg.decode_image(open(os.path.join(d, "x.hdr"), "rb").read())
        assert img.dtype == np.float32 and ch == 3 and np.abs(img[..., :3] - f[..., :3]).max() <= f[..., :3].max() / 100 and (img[..., 3] == 1).all()
    # Adam7: the seven reduced images, each filtered on its own width (8-bit RGBA and 1-bit grey, sizes that leave passes empty)
    for iw, ih in ((37, 23), (1, 1), (2, 5), (9, 3)):
        px = rng.integers(0, 256, (ih, iw, 4), dtype=np.uint8)
        img, ch = pkg.decode_image(_adam7_png(iw, ih, 6, 8, px, filters=[4, 1, 0, 2, 3]))
        assert ch == 4 and (img == px).all(), (iw, ih)
    bw = rng.integers(0, 2, (11, 19)).astype(np.uint8)
    img, _ = pkg.decode_image(_adam7_png(19, 11, 0, 1, bw))
    assert (img[..., 0] == bw * 255).all()
    # a header that declares 16384 x 16384 over a few bytes of data is refused BEFORE the 1 GiB pixel buffer is asked for
    import resource
    import time
    tiny = _adam7_png(3, 3, 6, 8, rng.integers(0, 256, (3, 3, 4), dtype=np.uint8))
    ihdr = struct.pack(">IIBBBBB", 16384, 16384, 8, 6, 0, 0, 1)
    huge = tiny[:12] + b"IHDR" + ihdr + struct.pack(">I", zlib.crc32(b"IHDR" + ihdr) & 0xFFFFFFFF) + tiny[33:]
    for lie in (huge, huge.replace(ihdr, ihdr[:-1] + b"\x00")):  # interlaced and not
        before, t0 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss, time.time()
        with pytest.raises(pkg.PtxError):
            pkg.decode_image(lie)
        assert resource.getrusage(resource.RUSAGE_SELF).ru_maxrss - before < 64 * 1024 and time.time() - t0 < 1.0  # KiB
    # errors: truncated stream, unknown interlace method, garbage
    good = _png(w, h, 6, 8, [rgba[y].tobytes() for y in range(h)])
    for bad in (good[:100], good.replace(b"IHDR" + good[16:28] + b"\x00", b"IHDR" + good[16:28] + b"\x02"), b"not an image at all" * 4):
        with pytest.raises(pkg.PtxError):
            pkg.decode_image(bad)


def test_hdr_rle_and_tga_rle(pkg):
    w, h = 40, 3
    rgbe = np.zeros((h, w, 4), np.uint8)
    rgbe[..., 0] = 128
    rgbe[..., 1] = np.arange(w)[None, :] * 3
    rgbe[..., 2] = 7
    rgbe[..., 3] = 130
    body = bytearray()
    for y in range(h):
        body += bytes([2, 2, w >> 8, w & 255])
        for c in range(4):
            line = rgbe[y, :, c]
            x = 0
            while x < w:
                run = 1
                while x + run < w and run < 127 and line[x + run] == line[x]:
                    run += 1
                if run > 2:
                    body += bytes([128 + run, int(line[x])])
                    x += run
                else:
                    n = min(w - x, 5)
                    body += bytes([n]) + bytes(line[x:x + n])
                    x += n
    data = b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n" % (h, w) + bytes(body)
    img, ch = pkg.decode_image(data)
    want = rgbe[..., :3].astype(np.float32) * np.float32(2.0 ** (130 - 136))
    assert img.shape == (h, w, 4) and (img[..., :3] == want).all()
    # TGA: RLE, bottom-up, 24 bit
    px = np.zeros((4, 6, 3), np.uint8)
    px[..., 0] = 10
    px[..., 1] = np.arange(6)[None, :] * 20
    px[1] = [1, 2, 3]
    tga = bytearray(18)
    tga[2] = 10
    tga[12:14] = struct.pack("<H", 6)
    tga[14:16] = struct.pack("<H", 4)
    tga[16] = 24
    for y in (3, 2, 1, 0):  # bottom-up
        if y == 1:
            tga += bytes([0x80 + 5, 3, 2, 1])
        else:
            tga += bytes([5]) + b"".join(bytes([int(p[2]), int(p[1]), int(p[0])]) for p in px[y])
    img, ch = pkg.decode_image(bytes(tga))
    assert ch == 3 and (img[..., :3] == px).all() and (img[..., 3] == 255).all()


def test_block_compressed_dds(pkg):
    def header(w, h, fourcc):
        hd = bytearray(128)
        hd[0:4] = b"DDS "
        struct.pack_into("<IIII", hd, 4, 124, 0x1007, h, w)
        struct.pack_into("<II", hd, 76, 32, 4)
        hd[84:88] = fourcc
        return bytes(hd)

    # BC1: c0 > c1 -> 4-colour mode; indices pick the two end points and the two interpolants
    c0, c1 = 0xF800, 0x001F  # red, blue
    block = struct.pack("<HHI", c0, c1, 0b11100100_11100100_11100100_11100100)
    img, ch = pkg.decode_image(header(4, 4, b"DXT1") + block)
    assert (img[0, 0] == [255, 0, 0, 255]).all() and (img[0, 1] == [0, 0, 255, 255]).all()
    assert (img[0, 2] == [170, 0, 85, 255]).all() and (img[0, 3] == [85, 0, 170, 255]).all()
    # c0 <= c1 -> 3 colours + transparent black
    block = struct.pack("<HHI", c1, c0, 0xFFFFFFFF)
    img, _ = pkg.decode_image(header(4, 4, b"DXT1") + block)
    assert (img == 0).all()
    # BC3: interpolated alpha block + colour block (no punch-through); 5x3 image = partial blocks
    alpha = bytes([200, 100]) + (0b001_000_001_000_001_000_001_000_001_000_001_000_001_000_001_000).to_bytes(6, "little")
    color = struct.pack("<HHI", 0x07E0, 0x07E0, 0)
    img, ch = pkg.decode_image(header(5, 3, b"DXT5") + (alpha + color) * 2)
    assert img.shape == (3, 5, 4) and ch == 4 and (img[..., 1] == 255).all() and set(np.unique(img[..., 3])) == {100, 200}
    # BC5: two BC4 channels -> R, G
    r = bytes([10, 250]) + (0).to_bytes(6, "little")       # a0 <= a1: 6-value mode, index 0 = a0
    g = bytes([250, 10]) + (0o77777777_77777777 & ((1 << 48) - 1)).to_bytes(6, "little")  # index 7 = last interpolant
    img, ch = pkg.decode_image(header(4, 4, b"ATI2") + r + g)
    assert ch == 2 and (img[..., 0] == 10).all() and (img[..., 1] == (1 * 250 + 6 * 10) // 7).all() and (img[..., 2] == 0).all()


def test_dds_carries_its_own_mip_chain(pkg):
    """TextureInfo::Levels: the levels stored in the file arrive as they are (the reference uploads them instead of
    regenerating the chain, TextureUploader.cpp:440,492-501)."""
    def header(w, h, levels):
        hd = bytearray(128)
        hd[0:4] = b"DDS "
        struct.pack_into("<IIII", hd, 4, 124, 0x1007 | (0x20000 if levels else 0), h, w)
        struct.pack_into("<I", hd, 28, levels)
        struct.pack_into("<II", hd, 76, 32, 4)
        hd[84:88] = b"DXT1"
        return bytes(hd)

    def solid(c565, blocks):
        return struct.pack("<HHI", c565, c565, 0) * blocks

    colours = [0xF800, 0x07E0, 0x001F, 0xFFFF]  # red 8x8, green 4x4, blue 2x2, white 1x1
    data = header(8, 8, 4) + solid(colours[0], 4) + solid(colours[1], 1) + solid(colours[2], 1) + solid(colours[3], 1)
    levels = pkg.decode_image_levels(data)
    assert [l.shape for l in levels] == [(8, 8, 4), (4, 4, 4), (2, 2, 4), (1, 1, 4)]
    for l, rgb in zip(levels, ([255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 255])):
        assert (l[..., :3] == rgb).all() and (l[..., 3] == 255).all()
    img, ch = pkg.decode_image(data)  # the level-0 view of the same file
    assert img.shape == (8, 8, 4) and (img[..., 0] == 255).all()
    assert len(pkg.decode_image_levels(header(8, 8, 0) + solid(colours[0], 4))) == 1
    for bad in (header(8, 8, 4) + solid(colours[0], 5), header(8, 8, 5) + solid(colours[0], 9)):  # truncated chain; too many levels
        with pytest.raises(pkg.PtxError):
            pkg.decode_image_levels(bad)


def _bits(v, n):
    return [(v >> (n - 1 - i)) & 1 for i in range(n)]


def test_baseline_jpeg_dc_only(pkg):
    """A hand-assembled 4:2:0 baseline JPEG whose blocks carry only DC terms: every 8x8 luma block decodes to a flat
    value, one chroma sample covers 16x16 pixels, and the JFIF matrix maps it to RGB."""
    W, H = 32, 16  # 2 MCUs of 16x16
    # Huffman tables: DC categories 0..8 as 4-bit codes 0000..1000 (code = category); AC: EOB only ('0')
    dc_bits = [0] * 16
    dc_bits[3] = 9
    dc_vals = list(range(9))
    ac_bits = [0] * 16
    ac_bits[0] = 1
    ac_vals = [0x00]

    def seg(marker, body):
        return bytes([0xFF, marker]) + struct.pack(">H", len(body) + 2) + bytes(body)

    dqt = bytes([0]) + bytes([8] * 64)  # one table, every step 8 -> DC value 8 * coeff, pixel = coeff + 128
    sof = bytes([8]) + struct.pack(">HH", H, W) + bytes([3, 1, 0x22, 0, 2, 0x11, 0, 3, 0x11, 0])
    dht = bytes([0x00] + dc_bits + dc_vals + [0x10] + ac_bits + ac_vals)
    sos = bytes([3, 1, 0x00, 2, 0x00, 3, 0x00, 0, 63, 0])
    # per MCU: 4 luma blocks, Cb, Cr.  Luma levels (pixel - 128) and chroma offsets:
    luma = [[-100, -50, 0, 50], [60, 70, 80, 90]]
    cb, cr = [40, -30], [-20, 64]
    bits = []
    pred = [0, 0, 0]

    def put_dc(comp, value):
        diff = value - pred[comp]
        pred[comp] = value
        cat = 0 if diff == 0 else int(abs(diff)).bit_length()
        bits.extend(_bits(cat, 4))
        if cat:
            bits.extend(_bits(diff if diff > 0 else diff + (1 << cat) - 1, cat))
        bits.append(0)  # EOB

    for m in range(2):
        for k in range(4):
            put_dc(0, luma[m][k])
        put_dc(1, cb[m])
        put_dc(2, cr[m])
    while len(bits) % 8:
        bits.append(1)
    scan = bytearray()
    for i in range(0, len(bits), 8):
        b = int("".join(map(str, bits[i:i + 8])), 2)
        scan.append(b)
        if b == 0xFF:
            scan.append(0)
    jpg = b"\xff\xd8" + seg(0xE0, b"JFIF\0\1\1\0\0\1\0\1\0\0") + seg(0xDB, dqt) + seg(0xC0, sof) + seg(0xC4, dht) + seg(0xDA, sos) + bytes(scan) + b"\xff\xd9"
    img, ch = pkg.decode_image(jpg)
    assert img.shape == (H, W, 4) and ch == 3 and (img[..., 3] == 255).all()
    for m in range(2):
        for k in range(4):
            y0, x0 = (k // 2) * 8, m * 16 + (k % 2) * 8
            Y, b, r = luma[m][k] + 128, cb[m], cr[m]
            want = np.clip(np.floor(np.array([Y + 1.402 * r, Y - 0.344136 * b - 0.714136 * r, Y + 1.772 * b]) + 0.5), 0, 255)
            blk = img[y0 + 2:y0 + 6, x0 + 2:x0 + 6, :3]  # the interior: chroma is interpolated across the MCU boundary
            assert (blk == blk[0, 0]).all() and np.abs(blk[0, 0].astype(int) - want).max() <= 1, (m, k)
    # restart intervals: the same scan cut after every MCU
    # (covered end to end by the encoder round trip in tests/test_output.py for AC coefficients and real images)
    # progressive files are refused
    with pytest.raises(pkg.PtxError):
        pkg.decode_image(jpg.replace(b"\xff\xc0", b"\xff\xc2"))
    # segments too short for what they declare: a 2-byte DRI (no interval), an SOS that ends inside its component list
    head = b"\xff\xd8" + seg(0xDB, dqt) + seg(0xC0, sof) + seg(0xC4, dht)
    for bad in (head + b"\xff\xdd\x00\x02" + seg(0xDA, sos) + bytes(scan) + b"\xff\xd9",
                head + seg(0xDA, sos[:3]) + bytes(scan) + b"\xff\xd9",
                head + b"\xff\xda\x00\x02",
                head + seg(0xC0, sof) + seg(0xDA, sos) + bytes(scan) + b"\xff\xd9"):
        with pytest.raises(pkg.PtxError):
            pkg.decode_image(bad)


def test_against_pillow(pkg, tmp_path):
    """Independent encoders / decoders where the image has Pillow: its files through our decoders, our files through its."""
    Image = pytest.importorskip("PIL.Image")
    import io

    rng = np.random.default_rng(11)
    yy, xx = np.mgrid[0:83, 0:125]
    pic = np.zeros((83, 125, 3), np.uint8)
    pic[..., 0] = (xx * 2) % 256
    pic[..., 1] = (yy * 3) % 256
    pic[..., 2] = ((xx // 9 + yy // 7) % 2) * 200 + rng.integers(0, 30, xx.shape)
    im = Image.fromarray(pic)

    def save(image, fmt, **kw):
        b = io.BytesIO()
        image.save(b, fmt, **kw)
        return b.getvalue()

    # JPEG written by libjpeg: 4:2:0, 4:4:4, 4:2:2, greyscale, optimised tables, restart markers
    for kw in ({"quality": 85, "subsampling": 2}, {"quality": 95, "subsampling": 0}, {"quality": 75, "subsampling": 1, "optimize": True},
               {"quality": 90, "subsampling": 2, "restart_marker_blocks": 3}):
        data = save(im, "JPEG", **kw)
        ours, ch = pkg.decode_image(data)
        theirs = np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))
        assert ch == 3 and ours.shape[:2] == theirs.shape[:2]
        diff = np.abs(ours[..., :3].astype(int) - theirs.astype(int))
        # same bit stream; libjpeg's integer IDCT and its smooth ("fancy") chroma upsampling differ from ours in rounding
        assert np.mean(diff) < 2.5 and np.percentile(diff, 99) <= 24, (kw, float(np.mean(diff)), int(diff.max()))
    grey = save(im.convert("L"), "JPEG", quality=90)
    ours, ch = pkg.decode_image(grey)
    assert ch == 1 and np.abs(ours[..., 0].astype(int) - np.asarray(Image.open(io.BytesIO(grey))).astype(int)).max() <= 2
    # progressive files (SOF2): spectral selection + successive approximation, the scan script libjpeg writes by default
    for kw in ({"quality": 85, "subsampling": 2}, {"quality": 95, "subsampling": 0}, {"quality": 60, "subsampling": 1, "optimize": True},
               {"quality": 90, "subsampling": 2, "restart_marker_blocks": 5}):
        data = save(im, "JPEG", progressive=True, **kw)
        assert b"\xff\xc2" in data
        ours, ch = pkg.decode_image(data)
        theirs = np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))
        diff = np.abs(ours[..., :3].astype(int) - theirs.astype(int))
        assert ch == 3 and np.mean(diff) < 2.5 and np.percentile(diff, 99) <= 24, (kw, float(np.mean(diff)), int(diff.max()))
        # the same picture as a baseline file decodes to (nearly) the same pixels: the coefficients are identical
        base, _ = pkg.decode_image(save(im, "JPEG", **kw))
        assert np.abs(ours.astype(int) - base.astype(int)).max() <= 1, kw
    pgrey = save(im.convert("L"), "JPEG", quality=90, progressive=True)
    ours, ch = pkg.decode_image(pgrey)
    assert ch == 1 and np.abs(ours[..., 0].astype(int) - np.asarray(Image.open(io.BytesIO(pgrey))).astype(int)).max() <= 2
    for cut in (len(pgrey) // 3, len(pgrey) // 2):   # a truncated progressive file still decodes (coarser), or raises: never crashes
        try:
            pkg.decode_image(pgrey[:cut])
        except pkg.PtxError:
            pass
    # PNG written by libpng / zlib: RGB, RGBA, palette, grey + alpha, 16-bit grey
    rgba = np.dstack([pic, rng.integers(0, 256, xx.shape, dtype=np.uint8)])
    for image, expect_ch in ((im, 3), (Image.fromarray(rgba), 4), (im.convert("P", palette=Image.ADAPTIVE, colors=64), 3),
                             (Image.fromarray(rgba).convert("LA"), 2)):
        data = save(image, "PNG", optimize=True)
        ours, ch = pkg.decode_image(data)
        assert ch == expect_ch and (ours == np.asarray(image.convert("RGBA"))).all()
    g16 = Image.fromarray((xx * 500 + yy).astype(np.uint16))
    ours, ch = pkg.decode_image(save(g16, "PNG"))
    assert ch == 1 and (ours[..., 0] == (np.asarray(g16) >> 8)).all()
    # TGA with RLE
    data = save(Image.fromarray(rgba), "TGA", compression="tga_rle")
    ours, ch = pkg.decode_image(data)
    assert ch == 4 and (ours == rgba).all()
    # and the other way round: Pillow reads what OutputSaver writes
    pkg.write_image(tmp_path / "o.png", rgba, pkg.OUTPUT_PNG)
    assert (np.asarray(Image.open(tmp_path / "o.png")) == rgba).all()
    pkg.write_image(tmp_path / "o.tga", rgba, pkg.OUTPUT_TGA)
    assert (np.asarray(Image.open(tmp_path / "o.tga")) == rgba).all()
    pkg.write_image(tmp_path / "o.jpg", rgba, pkg.OUTPUT_JPG)
    back = np.asarray(Image.open(tmp_path / "o.jpg").convert("RGB")).astype(np.float64)
    mine, _ = pkg.decode_image((tmp_path / "o.jpg").read_bytes())
    assert np.abs(back - mine[..., :3]).mean() < 1.0  # both decoders read the same picture out of our file
    ref = np.asarray(Image.open(io.BytesIO(save(im, "JPEG", quality=90, subsampling=2)))).astype(np.float64)
    ours_err, libjpeg_err = np.mean((back - pic) ** 2), np.mean((ref - pic) ** 2)
    assert ours_err < 1.25 * libjpeg_err  # as faithful as libjpeg at the same quality / subsampling (4:2:0 costs this busy image a lot)
    f = rng.uniform(0, 20, (9, 12, 4)).astype(np.float32)
    pkg.write_image(tmp_path / "o.hdr", f, pkg.OUTPUT_HDR)
    ours, _ = pkg.decode_image((tmp_path / "o.hdr").read_bytes())
    assert np.abs(ours[..., :3] - f[..., :3]).max() <= f[..., :3].max() / 100
