"""The oracle against the golden vectors generated from the REFERENCE's own GLSL
(tools/gen_golden.py; inputs include the grids of Path-Tracing-Tests/TestData.h)."""
import numpy as np
import pytest

import util

# functions whose reference body calls sin / cos / pow / atan / asin / exp: the libm-generated vectors differ from
# the fixed polynomial kernels by a few ULP of the transcendental (amplified where the GLSL
# subtracts nearly equal numbers, e.g. sqrt(1 - x^2 - y^2) at the rim of the disk)
TRANSCENDENTAL = {"SchlickFresnel": 1e-6, "SampleGGX": 2e-6, "evaluateBSDF": 1e-6, "sampleBSDF": 5e-6,
                  "sampleUniformDiskConcentric": 1e-6, "sampleCosineHemisphere": 2e-4, "constructPrimaryRayLens": 1e-6,
                  "sampleLight": 1e-6, "missSkyboxTexCoords": 2e-7, "toneMapPixel": 2e-7}


def test_oracle_bitexact_against_reference_glsl(orc):
    """Every restated function reproduces the reference's GLSL body bit for bit when the GLSL
    builtins follow the same arithmetic conventions (golden_fixed)."""
    for name, (fn, inp, exp) in util.load_golden("fixed").items():
        out = orc.test_eval(fn, inp, exp.shape[1])
        ok = util.bits_equal_or_both_nan(out, exp)
        assert ok.all(), f"{name}: {int((~ok).sum())} outputs differ"


def test_oracle_against_reference_glsl_with_libm(orc):
    """Same GLSL bodies with glibc's sinf/cosf/powf: exact where no transcendental is involved,
    within a small absolute tolerance otherwise (independent check of the polynomial kernels)."""
    for name, (fn, inp, exp) in util.load_golden("libm").items():
        out = orc.test_eval(fn, inp, exp.shape[1])
        if name not in TRANSCENDENTAL:
            assert util.bits_equal_or_both_nan(out, exp).all(), name
            continue
        of, ef = out.view(np.float32), exp.view(np.float32)
        if name == "sampleBSDF":  # last column is the RNG state (integer): must be exact
            assert (out[:, 7] == exp[:, 7]).all()
            of, ef = of[:, :7], ef[:, :7]
        finite = np.isfinite(ef) & np.isfinite(of)
        assert (np.isfinite(ef) == np.isfinite(of)).mean() > 0.99
        err = np.abs(of[finite].astype(np.float64) - ef[finite]) / np.maximum(1.0, np.abs(ef[finite]))
        assert err.max() <= TRANSCENDENTAL[name], f"{name}: {err.max()}"


def test_reference_test_properties(orc):
    """The three properties the reference's own tests assert (ShadingTest.cpp: finite outputs on
    the TestData.h grids; BsdfTest.cpp:34-40: lobe weights sum to 1 within 4 ULP)."""
    g = util.load_golden("fixed")
    grids = {"GGXDistribution": 6, "Lambda": 6, "GGXSmith": 6, "DielectricFresnel": 4, "SchlickFresnel": 2,
             "EvaluateReflection": 54, "EvaluateRefraction": 108, "SampleGGX": 24}
    for name, n in grids.items():
        fn, inp, exp = g[name]
        out = orc.test_eval(fn, inp[:n], exp.shape[1]).view(np.float32)
        assert np.isfinite(out).all(), name
    fn, inp, exp = g["sampleLobePdfs"]
    assert inp.shape[0] == 125
    s = orc.test_eval(fn, inp, 4).view(np.float32).sum(axis=1, dtype=np.float32)
    assert np.all(np.abs(s - 1.0) <= 4 * np.finfo(np.float32).eps)


def test_rng_known_answers(orc, pkg):
    """jenkinsHash / initRng / xorshift / uintToFloat computed by hand from common.glsl:133-165."""
    def jenkins(x):
        x = (x + (x << 10)) & 0xFFFFFFFF
        x ^= x >> 6
        x = (x + (x << 3)) & 0xFFFFFFFF
        x ^= x >> 11
        x = (x + (x << 15)) & 0xFFFFFFFF
        return x

    cases = [(0, 0, 1920, 0), (1919, 1079, 1920, 7), (3839, 2159, 3840, 1023), (17, 5, 512, 3)]
    inp = np.array(cases, dtype=np.uint32)
    out = orc.test_eval(pkg.FN["rng"], inp, 5)
    for (px, py, w, frame), row in zip(cases, out):
        st = jenkins((px + py * w) ^ jenkins(frame))
        assert row[0] == st
        for k in range(4):
            st ^= (st << 13) & 0xFFFFFFFF
            st ^= st >> 17
            st ^= (st << 5) & 0xFFFFFFFF
            f = np.array([0x3F800000 | (st >> 9)], dtype=np.uint32).view(np.float32)[0] - np.float32(1.0)
            assert row[1 + k] == np.array([f], np.float32).view(np.uint32)[0]
            assert 0.0 <= f < 1.0


def test_analytic_checks(orc, pkg):
    """Checks the reference lacks: sampled-direction pdf consistency and normalised lobes."""
    rng = np.random.default_rng(3)
    n = 2000
    # sampleBSDF's pdf must equal evaluateBSDF's pdf for the direction it returned
    mat = np.zeros((n, 8), np.float32)
    mat[:, 0:3] = rng.uniform(0.1, 1.0, (n, 3))
    mat[:, 3] = rng.uniform(0.05, 1.0, n)
    mat[:, 4] = rng.choice([0.0, 1.0, 0.5], n)
    mat[:, 5] = rng.choice([0.0, 1.0, 0.3], n)
    mat[:, 6] = rng.choice([1.5, 1 / 1.5], n).astype(np.float32)
    V = rng.normal(size=(n, 3)).astype(np.float32)
    V[:, 2] = np.abs(V[:, 2]) + 0.05
    V /= np.linalg.norm(V, axis=1, keepdims=True)
    seeds = rng.integers(1, 2**32 - 1, n, dtype=np.uint32).view(np.float32).reshape(-1, 1)
    sb = orc.test_eval(pkg.FN["sampleBSDF"], np.concatenate([mat, V, seeds], axis=1), 8).view(np.float32)
    L = sb[:, 0:3]
    ok = np.isfinite(L).all(axis=1)
    ev = orc.test_eval(pkg.FN["evaluateBSDF"], np.concatenate([mat, V, L], axis=1)[ok], 4).view(np.float32)
    assert (ev[:, 3].view(np.uint32) == sb[ok, 3].view(np.uint32)).all()
    assert (ev[:, 0:3].view(np.uint32) == sb[ok, 4:7].view(np.uint32)).all()
    # cosine-hemisphere samples lie on the unit hemisphere; disk samples inside the unit disk
    u = rng.uniform(0, 1, (n, 2)).astype(np.float32)
    d = orc.test_eval(pkg.FN["sampleCosineHemisphere"], u, 3).view(np.float32)
    assert np.abs(np.linalg.norm(d, axis=1) - 1).max() < 2e-3 and (d[:, 2] >= 0).all()
    k = orc.test_eval(pkg.FN["sampleUniformDiskConcentric"], u, 2).view(np.float32)
    assert (np.linalg.norm(k, axis=1) <= 1 + 1e-6).all()
    # tangent frames are orthonormal
    nrm = rng.normal(size=(n, 3)).astype(np.float32)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    t = orc.test_eval(pkg.FN["computeTangentSpace"], nrm, 9).view(np.float32).reshape(-1, 3, 3)
    gram = np.einsum("nij,nkj->nik", t, t)
    assert np.abs(gram - np.eye(3)).max() < 1e-5
