#!/usr/bin/env python3
"""Regenerate tests/golden/*.npz from the reference's shipped data files.

Runs only in the dev container (needs /root/reference and /opt/conda/bin/h5dump); the
resulting .npz files are committed so that tests, smoke() and bench.py never read
/root/reference at run time.  Only DATA is converted (no reference source is copied):

  capbuf_0000.npz      <- test/capbuf_0000.it        (recorded capture, exact (u8-127)/128 values,
                                                       stored as the u8 I/Q bytes) + fc
  test_peak_search.npz <- test/test_peak_search.it   (peak_search inputs + 20 golden peaks)
  test_sss_detect.npz  <- test/test_sss_detect.it    (capbuf, 24 input peaks, SSS goldens)
  test_tfg.npz         <- Matlab/test_tfg.mat        (capbuf + the input peak; the golden is
                                                       n_rb_dl==50, test/test_tfg.cpp:100)
  test_xcorr_pss.npz   <- Matlab/test_xcorr_pss.mat  (135360-sample capbuf; test_xcorr_pss.it is
                                                       missing upstream, so this is an input-only
                                                       fixture: stored re-quantised to u8, the
                                                       original differs from (u8-127)/128 by <6e-15)

MATLAB indices in the .it/.mat files are 1-based; they are stored here unchanged and the
tests subtract 1 exactly where the reference's tests do (test/test_sss_detect.cpp:56,65 ...).
"""
import importlib.util
import os
import re
import subprocess

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))

spec = importlib.util.spec_from_file_location("itfile", os.path.join(ROOT, "lte-cell-scanner_amd", "itfile.py"))
itfile = importlib.util.module_from_spec(spec)
spec.loader.exec_module(itfile)

_NUM = r"[-+]?(?:\d+\.?\d*(?:[eE][-+]?\d+)?|nan|inf)"


def h5_numbers(path, ds):
    out = subprocess.check_output(["/opt/conda/bin/h5dump", "-d", ds, "-y", "-w", "0", "-m", "%.17g", path]).decode()
    body = out[out.index("DATA {") + 6:].split("ATTRIBUTE")[0]
    return np.array(re.findall(_NUM, body), dtype=np.float64)


def h5_complex(path, ds):
    a = h5_numbers(path, ds)
    return a[0::2] + 1j * a[1::2]


def to_iq_u8(c, tol):
    """Invert the dongle conversion (u8-127)/128 (src/capbuf.cpp:174); assert it is lossless to tol."""
    re = np.round(c.real * 128 + 127)
    im = np.round(c.imag * 128 + 127)
    assert re.min() >= 0 and re.max() <= 255 and im.min() >= 0 and im.max() <= 255
    back = (re - 127) / 128 + 1j * (im - 127) / 128
    assert np.abs(back - c).max() <= tol
    iq = np.empty(2 * c.size, np.uint8)
    iq[0::2] = re.astype(np.uint8)
    iq[1::2] = im.astype(np.uint8)
    return iq


def main():
    d = itfile.read_it(f"{REF}/test/capbuf_0000.it")
    iq = to_iq_u8(d["capbuf"], 0.0)
    np.savez_compressed(f"{HERE}/capbuf_0000.npz", iq_u8=iq, fc=d["fc"])

    d = itfile.read_it(f"{REF}/test/test_peak_search.it")
    np.savez_compressed(f"{HERE}/test_peak_search.npz", **d)

    d = itfile.read_it(f"{REF}/test/test_sss_detect.it")
    np.savez_compressed(f"{HERE}/test_sss_detect.npz", **d)

    m = f"{REF}/Matlab/test_tfg.mat"
    peaks = {k: h5_numbers(m, "/peaks/" + k) for k in ["frame_start", "freq", "freq_fine", "ind", "n_id_1", "n_id_2", "pow"]}
    cp = "".join(chr(int(x)) for x in h5_numbers(m, "/peaks/cp_type"))
    assert cp == "normal"
    np.savez_compressed(f"{HERE}/test_tfg.npz", capbuf=h5_complex(m, "/capbuf"), fc=h5_numbers(m, "/fc"),
                        cp_type_is_extended=np.array([0]), expected_n_rb_dl=np.array([50]),
                        **{"peak_" + k: v for k, v in peaks.items()})

    m = f"{REF}/Matlab/test_xcorr_pss.mat"
    c = h5_complex(m, "/capbuf")
    iq = to_iq_u8(c, 1e-14)
    # parameters from Matlab/test_xcorr_pss.m:23-25
    np.savez_compressed(f"{HERE}/test_xcorr_pss.npz", iq_u8=iq, fc=np.array([739e6]), ds_comb_arm=np.array([2]),
                        f_search_set=np.array([35e3, 40e3, 45e3]))
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
