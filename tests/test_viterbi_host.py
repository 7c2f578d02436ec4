"""The GPU's tail-biting Viterbi (lte_device.h: one trellis per lane, 64 path metrics in registers, slot rotation
instead of a metric shuffle) compiled for the HOST and compared with the oracle's exhaustive decoder
(oracle/lcs_oracle.c conv_decode_tailbite <- src/lte_lib.cpp:538-551 -> itpp decode_tailbite): decoded bits and
winning start state on random LLRs, coarsely quantised LLRs (exact metric ties between paths), saturated LLRs and
encoded codewords with noise; CRC check incl. the antenna-port masks (src/searcher.cpp:1617-1636)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "host", "vit_host.cpp")
LIB = os.path.join(ROOT, "tests", "host", "libvit_host.so")


@pytest.fixture(scope="module")
def libs():
    dep = [SRC, os.path.join(ROOT, "lte-cell-scanner_amd", "csrc", "lte_device.h")]
    if not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in dep):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC",
                               "-shared", "-I" + os.path.join(ROOT, "include"), "-o", LIB, SRC])
    H = C.CDLL(LIB)
    H.vit_host_decode.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_int), C.POINTER(C.c_double)]
    H.vit_host_crc_ok.argtypes = [C.c_uint64, C.c_int]
    H.vit_host_forms_agree.argtypes = [C.POINTER(C.c_double)]
    O = C.CDLL(os.path.join(ROOT, "oracle", "liblcs_oracle.so"))
    O.orc_conv_decode_tailbite.argtypes = [C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_uint8)]
    O.orc_conv_decode_tailbite.restype = None
    O.orc_pbch_crc_ok.argtypes = [C.POINTER(C.c_uint8), C.c_int]
    return H, O


def both(libs, d):
    H, O = libs
    d = np.ascontiguousarray(d, np.float64).reshape(3, 40)
    bits, ss, met = C.c_uint64(), C.c_int(), C.c_double()
    rc = H.vit_host_decode(d.ctypes.data_as(C.POINTER(C.c_double)), C.byref(bits), C.byref(ss), C.byref(met))
    assert rc in (0, 1), "the retrace of the winning trellis (pass 2) did not reproduce pass 1's end metric"
    got = np.array([(bits.value >> t) & 1 for t in range(40)], np.uint8)
    ref = np.zeros(40, np.uint8)
    O.orc_conv_decode_tailbite(d.ctypes.data_as(C.POINTER(C.c_double)), 40, ref.ctypes.data_as(C.POINTER(C.c_uint8)))
    return got, ref, bits.value


def encode(bits):
    """Tail-biting K=7 (133,171,165)o encoder; LLR sign convention of the decoder: out bit 1 <-> negative metric term."""
    G = (0o133, 0o171, 0o165)
    state = 0
    for b in bits[-6:]:
        state = (state >> 1) | (int(b) << 5)
    out = np.zeros((3, 40))
    for t, b in enumerate(bits):
        reg = (int(b) << 6) | state
        for j in range(3):
            out[j, t] = 1.0 if bin(reg & G[j]).count("1") & 1 else -1.0
        state = reg >> 1
    return out


def test_matches_oracle_on_random_and_tied_inputs(libs):
    rng = np.random.default_rng(5)
    n_tie_cases = 0
    for k in range(120):
        if k < 40:
            d = rng.normal(0, 3, (3, 40))
        elif k < 80:                       # coarse grid: many exactly equal path metrics
            d = rng.integers(-2, 3, (3, 40)).astype(np.float64)
            n_tie_cases += 1
        elif k < 100:                      # saturated LLRs as at high SNR (trunc_log clipping), some exactly 0
            d = rng.choice([-708.396418532264, 708.396418532264, 0.0, 12.5], (3, 40))
        else:                              # a codeword in noise
            msg = rng.integers(0, 2, 40)
            d = -encode(msg) * 4.0 + rng.normal(0, 2.0, (3, 40))
        got, ref, _ = both(libs, d)
        assert np.array_equal(got, ref), f"case {k}"
        # the two forms of the step (min / compare-and-select) agree bit for bit on finite input, and the one-state-per-lane retrace
        # of EVERY start state ends on pass 1's metric
        dd = np.ascontiguousarray(d, np.float64)
        assert libs[0].vit_host_forms_agree(dd.ctypes.data_as(C.POINTER(C.c_double))) == 1, f"case {k}"
    assert n_tie_cases == 40


def test_non_finite_observations_take_the_reference_form(libs):
    """An infinite or NaN observation (a zero channel estimate upstream) switches the decoder to the compare-and-select
    form, whose `<` on NaN is the reference's: the result must still equal the oracle's exhaustive decoder."""
    rng = np.random.default_rng(9)
    for k in range(12):
        d = rng.normal(0, 3, (3, 40))
        d[rng.integers(0, 3), rng.integers(0, 40)] = [np.inf, -np.inf][k % 2]
        got, ref, _ = both(libs, d)
        assert np.array_equal(got, ref), f"case {k}"
    # a NaN observation poisons every path metric: no start state wins in either implementation (the reference then traces
    # back from an unset state -- undefined); here the decode reports "nothing" (all-zero bits, which fail the CRC)
    d = rng.normal(0, 3, (3, 40))
    d[1, 17] = np.nan
    bits, ss, met = C.c_uint64(7), C.c_int(0), C.c_double(0)
    dd = np.ascontiguousarray(d)
    assert libs[0].vit_host_decode(dd.ctypes.data_as(C.POINTER(C.c_double)), C.byref(bits), C.byref(ss), C.byref(met)) == 0
    assert ss.value == -1 and bits.value == 0


def test_decodes_clean_codewords_and_crc(libs):
    H, O = libs
    rng = np.random.default_rng(6)
    for k in range(20):
        msg = rng.integers(0, 2, 40).astype(np.uint8)
        got, ref, word = both(libs, -encode(msg) * 5.0)
        assert np.array_equal(got, msg) and np.array_equal(ref, msg)
        for n_ports in (1, 2, 4):
            assert H.vit_host_crc_ok(word, n_ports) == O.orc_pbch_crc_ok(msg.ctypes.data_as(C.POINTER(C.c_uint8)), n_ports)
    # a word with a valid CRC for exactly one port count
    payload = rng.integers(0, 2, 24).astype(np.uint8)
    for n_ports, mask in ((1, 0x0000), (2, 0xFFFF), (4, 0x5555)):
        crc = 0
        for b in payload:
            msb = ((crc >> 15) & 1) ^ int(b)
            crc = (crc << 1) & 0xFFFF
            if msb:
                crc ^= 0x1021
        crc ^= mask
        c_est = np.concatenate([payload, [(crc >> (15 - t)) & 1 for t in range(16)]]).astype(np.uint8)
        word = sum(int(b) << t for t, b in enumerate(c_est))
        for p in (1, 2, 4):
            want = int(p == n_ports)
            assert H.vit_host_crc_ok(word, p) == want == O.orc_pbch_crc_ok(c_est.ctypes.data_as(C.POINTER(C.c_uint8)), p)


def test_register_transform_building_blocks_on_the_host(libs):
    """fft16 / fft8 of lte_device.h (the in-register halves of the kernels' 128-point transform) and their composition with the
    W128 twiddles against numpy's FFT; cis_small (the polynomial exp(jx) the grid kernel uses for |x| <= 0.4) against libm over
    its whole stated range |x| <= 1."""
    L = libs[0]
    dp = C.POINTER(C.c_double)
    rng = np.random.default_rng(5)
    for n, fn in ((16, L.fft_host_16), (8, L.fft_host_8), (128, L.fft_host_128)):
        fn.argtypes = [dp, dp]
        fn.restype = None
        for _ in range(20):
            x = rng.normal(size=n) + 1j * rng.normal(size=n)
            o = np.empty(n, np.complex128)
            fn(np.ascontiguousarray(x).ctypes.data_as(dp), o.ctypes.data_as(dp))
            ref = np.fft.fft(x)
            assert np.abs(o - ref).max() < 4e-14 * np.abs(ref).max(), n
    L.cis_small_host.argtypes = [C.c_double, dp]
    L.cis_small_host.restype = None
    out = np.empty(2)
    worst = 0.0
    for x in np.concatenate([np.linspace(-1.0, 1.0, 20001), rng.uniform(-0.4, 0.4, 20000), [0.0, 1e-300, -1e-9]]):
        L.cis_small_host(float(x), out.ctypes.data_as(dp))
        worst = max(worst, abs(out[0] - np.cos(x)), abs(out[1] - np.sin(x)))
    assert worst < 2.5e-16, worst       # 1.5e-16 against extended precision + libm's own half ulp


def test_phase_walk_wrap_without_its_division_is_the_same_function(libs):
    """k_trk_prep walks the bulk phase symbol by symbol through the reference's WRAP (include/macros.h) but takes the floor of
    WRAP's quotient from a parallel prefix sum, guarded by trk_wrap_certain_interval: a k inside the interval of a floor value
    must have exactly that floor in WRAP's own arithmetic, and the walk's value must be WRAP's bit for bit.  Random phases over
    many turns (a 35 kHz LO error advances the phase by 2.5 turns per symbol), the neighbourhoods of every multiple of pi (ulp by
    ulp and across the guard band), zeros, huge values, non-finite ones."""
    L = libs[0]
    L.trk_wrap_interval_violations.argtypes = [C.POINTER(C.c_double), C.c_long, C.POINTER(C.c_long)]
    L.trk_wrap_interval_violations.restype = C.c_long
    rng = np.random.default_rng(11)
    parts = [rng.uniform(-40 * np.pi, 40 * np.pi, 3_000_000), rng.uniform(-np.pi, np.pi, 500_000), rng.normal(0, 1e-3, 100_000),
             rng.uniform(-1e5, 1e5, 200_000)]
    for mult in range(-9, 10):
        base = mult * np.pi
        near = [base]
        up, dn = base, base
        for _ in range(2000):
            up = np.nextafter(up, np.inf); dn = np.nextafter(dn, -np.inf)
            near += [up, dn]
        parts.append(np.array(near))
        parts.append(base + np.linspace(-2e-8, 2e-8, 20001))
    parts.append(np.array([0.0, -0.0, 1e300, -1e300, 1e-320, -1e-320, np.inf, -np.inf, np.nan, 2 ** 31 * 6.3, -2 ** 31 * 6.3, 1e7, -1e7]))
    x = np.ascontiguousarray(np.concatenate(parts))
    assert x.size > 4_000_000
    n_inside = C.c_long()
    assert L.trk_wrap_interval_violations(x.ctypes.data_as(C.POINTER(C.c_double)), x.size, C.byref(n_inside)) == 0
    assert n_inside.value > 0.8 * x.size          # (within 1e-9 of a turn boundary, beyond 1e6 turns, non-finite: WRAP itself)
