"""CPU check of the arithmetic claims behind the fp16 three-product correlation kernel (csrc/pss_xcorr_f16.hip) and the
division k_collapse_arm2 uses (csrc/pss_xcorr.hip):
(1) a float scaled by a power of two into [512, 1024) is hi + lo (two fp16) to 2^-22 of the scale's magnitude;
(2) a product of two fp16 values is exact in fp32, so xh*th + xh*tl + xl*th accumulated in fp32 differs from the fp64
    correlation only by the dropped xl*tl term and fp32 accumulation rounding -- far below the 1e-5 parity bar;
(3) x * RN(1/5) corrected once through the exact remainder equals the correctly rounded x / 5 (the exhaustive check over
    all 2^31 non-negative floats takes 50 s and was run once; here: every exponent x 2^14 mantissas, compiled C, fmaf).
numpy restatement of f16_split / f16_scale_exp and of the kernel's term order."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def scale_exp(vmax):
    """power of two that brings vmax into [2^9, 2^10) (F16_TARGET_EXP = 9)"""
    _, e = np.frexp(np.float32(vmax))          # vmax = m * 2^e, m in [0.5, 1)
    return 10 - int(e)


def split(v):
    v = np.asarray(v, np.float32)
    hi = v.astype(np.float16)
    lo = (v - hi.astype(np.float32)).astype(np.float16)
    return hi, lo


def test_hi_lo_split_keeps_22_bits():
    rng = np.random.default_rng(3)
    x = (rng.standard_normal(200000) * rng.choice([1e-3, 1.0, 250.0], 200000)).astype(np.float32)
    k = scale_exp(np.abs(x).max())
    xs = np.ldexp(x, k).astype(np.float32)       # exact: power of two
    assert 512 <= np.abs(xs).max() < 1024
    assert np.array_equal(np.ldexp(xs, -k).astype(np.float32), x)
    hi, lo = split(xs)
    assert np.all(np.isfinite(hi.astype(np.float32))) and np.all(np.isfinite(lo.astype(np.float32)))
    err = np.abs(xs.astype(np.float64) - hi.astype(np.float64) - lo.astype(np.float64))
    # hi carries 11 bits, lo 11 more of the remainder (or all of it once it is below fp16's resolution near zero)
    assert err.max() <= 1024 * 2.0 ** -22


def test_three_products_in_fp32_match_the_fp64_correlation():
    import oracle as O
    rng = np.random.default_rng(4)
    n = 3000
    t = O.pss_td(2)
    t = (np.conj(t) / 137).astype(np.complex64)               # the correlation's taps (searcher.cpp:136)
    x = (rng.standard_normal(n + 137) + 1j * rng.standard_normal(n + 137)).astype(np.complex64) * np.float32(0.37)
    x[700:837] += (3.0 * O.pss_td(2)).astype(np.complex64)    # lags with a peak and lags with noise only
    kx = scale_exp(max(np.abs(x.real).max(), np.abs(x.imag).max()))
    kt = scale_exp(max(np.abs(t.real).max(), np.abs(t.imag).max()))
    xr_h, xr_l = split(np.ldexp(x.real, kx)); xi_h, xi_l = split(np.ldexp(x.imag, kx))
    tr_h, tr_l = split(np.ldexp(t.real, kt)); ti_h, ti_l = split(np.ldexp(t.imag, kt))
    f = lambda a: a.astype(np.float32)
    # every fp16 x fp16 product is exact in fp32 (22-bit significand)
    p32 = f(xr_h[:137]) * f(tr_h)
    assert np.array_equal(p32.astype(np.float64), xr_h[:137].astype(np.float64) * tr_h.astype(np.float64))
    got, want = [], []
    for lag in list(range(0, n, 37)) + [700]:
        sl = slice(lag, lag + 137)
        acc_r = np.float32(0); acc_i = np.float32(0)
        # kernel order: per 16-tap block (one MFMA: K = 32 exact products summed inside the instruction, one rounding into
        # the fp32 accumulator) the hi*hi, hi*lo and lo*hi terms go into the same accumulator
        d = lambda a: a.astype(np.float64)
        for b0 in range(0, 144, 16):
            tb = slice(b0, min(b0 + 16, 137))
            xb = slice(lag + b0, lag + min(b0 + 16, 137))
            for (ar, ai, br, bi) in ((xr_h, xi_h, tr_h, ti_h), (xr_h, xi_h, tr_l, ti_l), (xr_l, xi_l, tr_h, ti_h)):
                acc_r = np.float32(np.float64(acc_r) + np.sum(d(ar[xb]) * d(br[tb]) - d(ai[xb]) * d(bi[tb])))
                acc_i = np.float32(np.float64(acc_i) + np.sum(d(ar[xb]) * d(bi[tb]) + d(ai[xb]) * d(br[tb])))
        p = (np.float64(acc_r) ** 2 + np.float64(acc_i) ** 2) * 2.0 ** (-2 * (kx + kt))
        ref = np.abs(np.sum(x[sl].astype(np.complex128) * t.astype(np.complex128))) ** 2
        got.append(p); want.append(ref)
    got, want = np.array(got), np.array(want)
    # a lag where the 548 products nearly cancel has no relative accuracy to lose in ANY fp32-accumulating form (the fp32
    # MFMA kernel and the reference's own complex<float> store included): errors are measured against the larger of the
    # lag's power and the median power, as the parity tests' 1e-5 bar on the 15-window sums effectively does
    err = np.abs(got - want) / np.maximum(want, np.median(want))
    assert err.max() < 1e-6, err.max()
    assert abs(got[-1] - want[-1]) / want[-1] < 2e-7          # the peak itself


DIV5_SRC = r"""
#include <math.h>
#include <stdint.h>
#include <string.h>
long long div5_mismatches(int mant_bits) {
  long long bad = 0;
  const uint32_t step = 1u << (23 - mant_bits);
  for (uint32_t e = 0; e < 255; ++e)
    for (uint32_t m = 0; m < (1u << 23); m += step) {
      for (int edge = 0; edge < 2; ++edge) {
        uint32_t b = (e << 23) | (edge ? ((1u << 23) - 1 - m) : m);
        float x; memcpy(&x, &b, 4);
        const float q = x * 0.2f;
        const float r = fmaf(fmaf(-5.0f, q, x), 0.2f, q);
        if (r != x / 5.0f) ++bad;
      }
    }
  return bad;
}
"""


def test_corrected_multiply_is_the_correctly_rounded_quotient(tmp_path):
    src = tmp_path / "div5.c"
    src.write_text(DIV5_SRC)
    lib = tmp_path / "libdiv5.so"
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-o", str(lib), str(src), "-lm"])
    L = C.CDLL(str(lib))
    L.div5_mismatches.restype = C.c_longlong
    assert L.div5_mismatches(14) == 0          # 255 exponents (denormals included) x 2^14 mantissas from both ends
