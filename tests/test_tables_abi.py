"""CPU-only checks of the product library: it loads, exports every symbol include/lcs.h
declares, refuses to create a context without a GPU, and its host tables agree with the
oracle's restatement of the reference table code."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle as O
from conftest import ROOT, load_pkg


@pytest.fixture(scope="module")
def pkg():
    return load_pkg()


def test_library_exports_every_declared_symbol(pkg):
    hdr = open(os.path.join(ROOT, "include", "lcs.h")).read()
    declared = set(re.findall(r"\b(lcs_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"lcs_ctx", "lcs_cell"}
    lib = pkg.capi.load()
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, missing
    assert declared == set(pkg.capi.EXPORTS)


def test_cell_struct_layout_and_init(pkg):
    assert C.sizeof(pkg.LcsCell) == 7 * 8 + 10 * 4 == C.sizeof(O.Cell)
    c = pkg.new_cell()
    assert np.isnan(c.fc_requested) and np.isnan(c.freq_superfine)
    assert (c.ind, c.n_id_2, c.n_id_1, c.cp_type, c.n_ports, c.n_rb_dl, c.sfn) == (-1, -1, -1, 0, -1, -1, -1)
    assert c.n_id_cell() == -1 and c.n_symb_dl() == -1


def test_no_gpu_means_loud_failure(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(pkg.SearcherError):
        pkg.Searcher(0)


def test_pss_tables_match_oracle(pkg):
    for t in range(3):
        assert np.abs(pkg.table_pss_fd(t) - O.pss_fd(t)).max() < 1e-13
        assert np.abs(pkg.table_pss_td(t) - O.pss_td(t)).max() < 1e-13


def test_sss_table_matches_oracle(pkg):
    for n1 in range(168):
        for n2 in range(3):
            for slot in (0, 10):
                assert np.array_equal(pkg.table_sss_fd(n1, n2, slot), O.sss_fd(n1, n2, slot)), (n1, n2, slot)


def test_lte_pn_matches_oracle(pkg):
    rng = np.random.default_rng(1)
    for c_init in [0, 1, 277, 503, 2**31 - 1] + [int(x) for x in rng.integers(0, 2**31, 20)]:
        assert np.array_equal(pkg.table_lte_pn(c_init, 1920), O.lte_pn(c_init, 1920))


def test_chi2cdf_inv_matches_oracle_and_scipy(pkg):
    from scipy.stats import chi2
    for k in (10, 140, 150, 300):
        p = 1 - 1e-12
        assert abs(pkg.chi2cdf_inv(p, k) - O.chi2cdf_inv(p, k)) < 1e-9 * k
        assert abs(pkg.chi2cdf_inv(p, k) - chi2.ppf(p, k)) < 1e-8 * chi2.ppf(p, k)


def test_f_search_set_matches_cli_recipe(pkg):
    # src/CellSearch.cpp:463-464: ppm 120 @ 739 MHz -> 37 hypotheses, ppm 100 -> 31, ppm 10 -> 3
    assert pkg.f_search_set_for(739e6, 120).size == 37
    assert pkg.f_search_set_for(739e6, 100).size == 31
    assert pkg.f_search_set_for(739e6, 10).size == 3
    assert pkg.f_search_set_for(715e6, 120).size == 35


def test_chain_kernels_fit_beside_one_correlation_workgroup():
    """The per-cell chain runs while the NEXT batch's correlation occupies the GPU: a chain workgroup starts where one of a
    CU's two correlation workgroups (4 waves x 228 VGPRs, 77.3 KB of LDS) retired.  On a SIMD 512 - 228 = 284 registers are
    free then, allocated in eights; a kernel that needs more has to wait for BOTH correlation workgroups of a CU to retire at
    once (round 5 measured that: k_sss_win at 256 + 27 registers waited 1 ms per batch).  Compiles the kernels that sit near
    the limit and checks the compiler's own resource report."""
    import re
    import subprocess
    csrc = os.path.join(ROOT, "lte-cell-scanner_amd", "csrc")
    limits = {"k_sss_win": None, "k_foe_win": None, "k_tfg": None}
    pairs = {"k_tfg": None, "k_tfoec_est": None, "k_chan_est": None, "k_sss_ml": None}       # two workgroups per freed slot (below)
    for f in ("sss_foe.hip", "tfg_mib.hip"):
        p = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-c", "--cuda-device-only",
                            "-Rpass-analysis=kernel-resource-usage", os.path.join(csrc, f), "-o", os.devnull], capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stderr[-2000:]
        cur = None
        for line in p.stderr.splitlines():
            m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|LDS Size \[bytes/block\]): (\S+)", line)
            if not m:
                continue
            if m.group(1) == "Function Name":
                cur = next((k for k in limits if f"{len(k)}{k}" in m.group(2)), None)      # every instantiation (one per source format)
                cur2 = next((k for k in pairs if f"{len(k)}{k}" in m.group(2)), None)
                if cur:
                    limits[cur] = (limits[cur] or []) + [{"name": m.group(2)}]
                if cur2:
                    pairs[cur2] = (pairs[cur2] or []) + [{"name": m.group(2)}]
            else:
                if cur:
                    limits[cur][-1][m.group(1).split()[0]] = int(m.group(2))
                if cur2:
                    pairs[cur2][-1][m.group(1).split()[0]] = int(m.group(2))
    for k, insts in limits.items():
        assert insts and len(insts) == 3, (k, insts)
        for v in insts:
            regs = -(-(v["VGPRs"] + v.get("AGPRs", 0)) // 8) * 8
            assert regs <= 280, f"{v['name']}: {v['VGPRs']} + {v.get('AGPRs', 0)} registers -> {regs} allocated: does not fit beside a resident correlation workgroup (284 free)"
            assert v["LDS"] <= 77 * 1024, (k, v)
    # The kernels of which TWO workgroups fit that slot (2 x 136 <= 284 registers, 2 x 43264 <= 163840 - 77312 bytes of LDS): they were
    # 155-164 registers wide until their sincos / atan2 / polynomial expansions became real calls (lte_device.h: cis_call ...), whose
    # 64-bit literals the compiler had kept in registers for the kernels' whole lifetime (profiles/r05/experiments: dense band + 2.5 %).
    for k, insts in pairs.items():
        assert insts, k
        for v in insts:
            regs = -(-(v["VGPRs"] + v.get("AGPRs", 0)) // 8) * 8
            assert regs <= 136 and v["LDS"] <= 43264 and v.get("ScratchSize", 0) == 0, (k, v)


def test_no_kernel_of_the_library_spills_or_uses_scratch():
    """Every kernel of the SHIPPED library (the code objects inside liblcs_amd.so, not a recompilation): no VGPR spills, no private
    segment -- nothing a kernel holds goes through memory (SGPRs the compiler parks in lanes of a VGPR by v_writelane are
    registers still: .sgpr_spill_count is not scratch) (round 5: k_trk_ce 22 spilled registers / 480 B, k_trk_mib 404 B, k_pbch 268 B, k_peak_search_reg 28 B --
    scratch traffic sits in HBM behind every access).  tools/code_objects.py reads the kernels' own metadata notes."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("code_objects", os.path.join(ROOT, "tools", "code_objects.py"))
    co = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(co)
    lib = os.path.join(ROOT, "lte-cell-scanner_amd", "liblcs_amd.so")
    if not os.path.exists(lib):
        pytest.skip("library not built")
    ks = co.kernels_of(lib)
    assert len(ks) >= 60, len(ks)                      # 65 kernels in round 6: the parser saw all of the bundles
    bad = {k: v for k, v in ks.items() if v["vgpr_spill_count"] or v["private_segment_fixed_size"]}
    assert not bad, bad
    for k, v in ks.items():
        assert v["vgpr_count"] + v["agpr_count"] <= 512 and v["group_segment_fixed_size"] <= 160 * 1024, (k, v)
