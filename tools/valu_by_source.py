#!/usr/bin/env python3
"""Static VALU instruction count of one kernel by source function and line (task 4).

Usage: tools/valu_by_source.py [kernel-substring]   (default: the opaque shade kernel, 7k_shadeILb0)
Compiles the HIP translation unit to gfx950 assembly with line tables (-gline-tables-only, device only, the package's flags) and
attributes every v_* instruction between the kernel's label and its s_endpgm to the .loc in force: per source FUNCTION (the
enclosing `PT_DEV ... name(` of the line, found by scanning the source upwards) and per line.  Static counts: a line inside a loop
or under a branch weighs what the compiler emitted, not what a wave executes -- k_shade is nearly straight-line code per material
branch, so the ranking holds; the dynamic total is SQ_INSTS_VALU in profiles/*_sq.txt."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402


def main():
    pat = sys.argv[1] if len(sys.argv) > 1 else "7k_shadeILb0"
    pkg = graft.load_package()
    flags = [f for f in pkg.HIPCC_FLAGS if f not in ("-shared", "-fPIC")]
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, "k.s")
        subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + flags + ["--cuda-device-only", "-S", "-gline-tables-only", "-o", asm,
                               pkg.hip_sources()[0]], stderr=subprocess.DEVNULL)
        lines = open(asm).read().split("\n")
    files = {}
    for l in lines:
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"\s+"([^"]*)"', l)
        if m:
            files[int(m.group(1))] = os.path.join(m.group(2), m.group(3)) if not os.path.isabs(m.group(3)) else m.group(3)
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*:", l) and pat in l)
    per_line, cur = collections.Counter(), (0, 0)
    total = 0
    for l in lines[start + 1:]:
        t = l.strip()
        if t.startswith("s_endpgm"):
            break
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", t)
        if m:
            cur = (int(m.group(1)), int(m.group(2)))
            continue
        if t.startswith("v_"):
            per_line[cur] += 1
            total += 1
    src_cache = {}

    def function_of(fid, line):
        path = files.get(fid, "?")
        path = path if os.path.isabs(path) else os.path.join(ROOT, path)
        if path not in src_cache:
            try:
                src_cache[path] = open(path).read().split("\n")
            except OSError:
                src_cache[path] = []
        src = src_cache[path]
        for k in range(min(line, len(src)) - 1, -1, -1):
            m = re.match(r"^(?:template\s*<[^>]*>\s*)?(?:PT_DEV|__global__|static|inline|__device__)[^;(]*?\b([A-Za-z_][A-Za-z_0-9]*)\s*\(", src[k])
            if m and not src[k].startswith(" "):
                return os.path.basename(path) + ":" + m.group(1)
        return os.path.basename(path) + ":?"

    per_fn = collections.Counter()
    for (fid, line), c in per_line.items():
        per_fn[function_of(fid, line)] += c
    print(f"{lines[start].split(':')[0]}: {total} VALU instructions (static)")
    print("by function:")
    for fn, c in per_fn.most_common(30):
        print(f"  {c:6d}  {100.0 * c / total:5.1f} %  {fn}")
    print("by line (top 25):")
    for (fid, line), c in per_line.most_common(25):
        path = files.get(fid, "?")
        path = path if os.path.isabs(path) else os.path.join(ROOT, path)
        text = src_cache.get(path, [])
        print(f"  {c:6d}  {os.path.basename(path)}:{line}: {text[line - 1].strip()[:110] if 0 < line <= len(text) else ''}")


if __name__ == "__main__":
    main()
