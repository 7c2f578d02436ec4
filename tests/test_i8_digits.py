"""CPU check of the arithmetic claims behind the int8 correlation kernel (csrc/pss_xcorr_i8.hip):
(1) 127 - u8 fits int8 for all 256 codes; (2) a template tap quantised to |T_int| <= 8.3e6 splits exactly
into three balanced base-256 digits in [-128, 127]; (3) the worst-case int32 accumulators cannot overflow;
(4) the quantisation changes a PSS correlation magnitude by far less than the 1e-5 parity bar.
numpy restatement of the device helpers k_i8_scales / digits3 / the kernel's recombination."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))

QMAX = 8300000.0          # I8_QMAX


def digits3(v):
    v = np.asarray(v, np.int64)
    d0 = ((v + 128) & 255) - 128
    v1 = (v - d0) >> 8
    d1 = ((v1 + 128) & 255) - 128
    d2 = (v1 - d1) >> 8
    return d0, d1, d2


def test_sample_codes_fit_int8():
    s = 127 - np.arange(256)
    assert s.min() == -128 and s.max() == 127
    # the sample value is -(s)/128: the sign is absorbed in the template operand, the 1/128 in the output scale
    assert np.array_equal(-s / 128.0, (np.arange(256) - 127.0) / 128.0)


def test_three_balanced_digits_are_exact_over_the_whole_range():
    rng = np.random.default_rng(1)
    v = np.concatenate([rng.integers(-8300000, 8300001, 1_000_000), [0, 1, -1, 127, 128, -128, -129, 8300000, -8300000, 32767, 32768, -32768, -32769]])
    d0, d1, d2 = digits3(v)
    for d in (d0, d1, d2):
        assert d.min() >= -128 and d.max() <= 127
    assert np.array_equal(d0 + 256 * d1 + 65536 * d2, v)


def test_int32_accumulators_cannot_overflow():
    # one output = sum over <= 137 taps of two products (re*re, im*im), |sample| <= 128, |digit| <= 128
    worst_pass = 137 * 2 * 128 * 128
    assert worst_pass < 2 ** 31
    # digits 1 and 0 share an accumulator: (S1 << 8) + S0
    assert 256 * worst_pass + worst_pass < 2 ** 31


def test_quantised_template_correlation_is_within_the_parity_bar():
    import oracle as O
    rng = np.random.default_rng(2)
    n = 4096
    u8 = rng.integers(0, 256, (n + 137, 2))
    # plant a PSS so that some lags carry a real peak and others only noise
    t = O.pss_td(1)
    x = (u8 - 127.0) / 128.0
    xc = x[:, 0] + 1j * x[:, 1]
    xc[1000:1137] += 0.5 * t / np.abs(t).max()
    u8 = np.clip(np.round(np.stack([xc.real, xc.imag], 1) * 128 + 127), 0, 255).astype(np.int64)
    xs = (u8[:, 0] - 127.0) / 128.0 + 1j * (u8[:, 1] - 127.0) / 128.0
    # template as the device holds it: conj(fshift(pss_td))/137 rounded to fp32
    f_off, fs = 35e3, 1.92e6
    T = (np.conj(t * np.exp(2j * np.pi * f_off / fs * np.arange(137))) / 137).astype(np.complex64)
    tr, ti = T.real.astype(np.float64), T.imag.astype(np.float64)
    q = QMAX / max(np.abs(tr).max(), np.abs(ti).max())
    Tr, Ti = np.rint(tr * q).astype(np.int64), np.rint(ti * q).astype(np.int64)
    sc = np.float32(1.0 / (128.0 * q))
    a_r, a_i = 127 - u8[:, 0], 127 - u8[:, 1]                    # int8 sample codes (= -128 * sample)
    exact = np.empty(n)
    got = np.empty(n, np.float32)
    for k in range(n):
        seg = slice(k, k + 137)
        z = np.sum(T.astype(np.complex128) * xs[seg])
        exact[k] = z.real ** 2 + z.imag ** 2
        # the kernel: per digit pass integer dot products with operands (tr, -ti) / (ti, tr), digit 2 alone,
        # digits 1 and 0 sharing an accumulator, fp32 recombination, squared magnitude, scale at the end
        acc = []
        for (br, bi) in ((Tr, -Ti), (Ti, Tr)):
            d0r, d1r, d2r = digits3(br)
            d0i, d1i, d2i = digits3(bi)
            s2 = np.sum(a_r[seg] * d2r + a_i[seg] * d2i)
            lo = (np.sum(a_r[seg] * d1r + a_i[seg] * d1i) << 8) + np.sum(a_r[seg] * d0r + a_i[seg] * d0i)
            assert abs(int(s2)) < 2 ** 31 and abs(int(lo)) < 2 ** 31
            assert int(s2) * 65536 + int(lo) == int(np.sum(a_r[seg] * br + a_i[seg] * bi))      # digits recombine exactly
            acc.append(np.float32(np.float32(s2) * np.float32(65536.0) + np.float32(lo)))        # fmaf: one rounding; two here is an upper bound
        p = np.float32(acc[0] * acc[0] + acc[1] * acc[1])
        got[k] = p * (sc * sc)
    rel = np.abs(got.astype(np.float64) - exact) / exact.max()
    assert rel.max() < 1e-6, rel.max()
    big = exact > 1e-3 * exact.max()
    assert (np.abs(got[big].astype(np.float64) - exact[big]) / exact[big]).max() < 1e-5
