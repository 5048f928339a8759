#!/usr/bin/env python3
"""Mutation fuzzing of the host decoders and importers (CPU only), meant to run against a sanitizer build of the host library:

    cd path-tracing_amd/host && g++ -std=c++20 -O1 -g -fPIC -shared -fsanitize=address,undefined -fno-omit-frame-pointer \
        -fvisibility=hidden -I../../include -o /tmp/libptx_host_asan.so *.cpp        (RendererHip.cpp left out)
    PTX_HOST_LIB=/tmp/libptx_host_asan.so ASAN_OPTIONS=detect_leaks=0 \
        LD_PRELOAD="$(gcc -print-file-name=libasan.so) /usr/lib/x86_64-linux-gnu/libstdc++.so.6" python tools/fuzz_host.py

Seeds: a baseline and two progressive JPEGs and a PNG written by Pillow, and the animated binary FBX of tests/test_fbx.py.
Every mutant must either decode / load or raise PtxError; the sanitizers report anything else.  (Round 2: this found an
allocation sized by a corrupt PNG header, fopen() succeeding on a directory, and an accessor count used before its bounds
check.)"""
import sys, io, os, struct, pathlib, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import importlib
pkg = importlib.import_module('path-tracing_amd')
from PIL import Image
rng = np.random.default_rng(1)
yy, xx = np.mgrid[0:43, 0:61]
pic = np.stack([(xx*4)%256, (yy*5)%256, ((xx//5+yy//3)%2)*200], -1).astype(np.uint8)
im = Image.fromarray(pic)
def save(image, fmt, **kw):
    b = io.BytesIO(); image.save(b, fmt, **kw); return b.getvalue()
seeds = [save(im,"JPEG",progressive=True,quality=80), save(im,"JPEG",quality=80,subsampling=2), save(im,"PNG"), save(im.convert("L"),"JPEG",progressive=True)]
n=0
for data in seeds:
    for it in range(400):
        b = bytearray(data)
        for k in rng.integers(2, len(b), rng.integers(1, 8)):
            b[k] = rng.integers(0,256)
        if it % 7 == 0: b = b[:rng.integers(10, len(b))]
        try:
            pkg.decode_image(bytes(b)); n+=1
        except pkg.PtxError:
            pass
print("decoded", n)
# FBX / OBJ fuzz through the scene loader
import test_fbx as T
tmp = pathlib.Path(tempfile.mkdtemp())
path = T._write_scene(pkg, tmp, animated=True)
good = path.read_bytes()
name = T._describe(tmp, components=["bad.fbx"], mapping="orca")
ok=0
for it in range(300):
    b = bytearray(good)
    for k in rng.integers(27, len(b), rng.integers(1, 10)):
        b[k] = rng.integers(0,256)
    (tmp/"bad.fbx").write_bytes(bytes(b))
    try:
        pkg.Scene(name); ok+=1
    except pkg.PtxError:
        pass
print("fbx loaded", ok)
