"""CPU check of the arithmetic claim behind the bf16 correlation kernel (csrc/pss_xcorr_bf16.hip):
RTL-SDR samples are exact in bfloat16, an fp32 template value splits EXACTLY into three bfloat16
terms, and every sample x term product is exact in fp32 -- so three bf16 MFMAs with fp32 accumulation
form the same products as the fp32 kernel.  numpy restatement of the device helpers bf16_rne / bf16_val."""
import numpy as np


def bf16_rne(v):
    u = np.asarray(v, np.float32).view(np.uint32).astype(np.uint64)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint32)


def bf16_val(h):
    return (h.astype(np.uint32) << 16).view(np.float32)


def split3(v):
    r = np.asarray(v, np.float32).copy()
    terms = []
    for _ in range(3):
        h = bf16_rne(r)
        t = bf16_val(h)
        terms.append(t)
        r = (r - t).astype(np.float32)          # exact: the difference has at most 16 significant bits left
    return terms, r


def test_u8_samples_are_exact_in_bf16():
    x = ((np.arange(256, dtype=np.float64) - 127.0) / 128.0).astype(np.float32)
    assert np.array_equal(bf16_val(bf16_rne(x)), x)
    assert np.array_equal(bf16_val(x.view(np.uint32) >> 16), x)          # what k_ingest does: plain truncation


def test_three_term_split_is_exact_and_products_are_exact_in_fp32():
    rng = np.random.default_rng(0)
    # template taps are |t| <= ~0.02 (pss_td / 137), down to tiny values; include zeros and both signs
    t = np.concatenate([rng.uniform(-0.03, 0.03, 400000), rng.normal(0, 1e-4, 100000), [0.0, -0.0, 1.0, -1.0, 2.0 ** -60]]).astype(np.float32)
    (t1, t2, t3), resid = split3(t)
    assert np.array_equal((t1.astype(np.float64) + t2.astype(np.float64) + t3.astype(np.float64)), t.astype(np.float64))
    assert not resid.any()
    # sample x term: 8-bit x 8-bit significands -> 16 bits, exact in fp32; the three products sum to x * t exactly in fp64
    x = ((rng.integers(0, 256, t.size) - 127.0) / 128.0).astype(np.float32)
    for ti in (t1, t2, t3):
        assert np.array_equal((x * ti).astype(np.float64), x.astype(np.float64) * ti.astype(np.float64))
    assert np.array_equal(x.astype(np.float64) * t1 + x.astype(np.float64) * t2 + x.astype(np.float64) * t3, x.astype(np.float64) * t.astype(np.float64))
