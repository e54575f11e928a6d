"""
The pipeline of qocx_eval_resident relies on which kernels can share a SIMD (512 registers,
DESIGN.md 4): the sweep must fit beside one two-wave K1a wave or one K3 wave, the two-wave K1a and
the skew K3 must reach their waves per SIMD, and none of them may spill. This test compiles the
device code for gfx950 with the resource remarks on (no GPU needed) and checks those budgets.
"""
import os
import re
import shutil
import subprocess

import pytest

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "qoc_amd", "csrc")


def resources(source):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    out = subprocess.run(
        [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-c",
         os.path.join(CSRC, source), "-o", os.devnull, "-Rpass-analysis=kernel-resource-usage"],
        capture_output=True, text=True, check=True).stderr
    table, name = {}, None
    for line in out.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            table[name] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and name:
            table[name][m.group(1).strip()] = int(m.group(2))
    return table


def find(table, fragment):
    hits = [v for k, v in table.items() if fragment in k]
    assert len(hits) == 1, (fragment, [k for k in table if fragment in k])
    return hits[0]


def total_registers(entry):
    return entry["VGPRs"] + entry.get("AGPRs", 0)


def test_headline_kernels_share_a_simd():
    kernels = resources("qocx_kernels.hip")
    pade2 = resources("qocx_pade2.hip")
    sweep = find(kernels, "sweep_kernelILi2ELi1ELb0ELb0E")
    sweep1 = find(kernels, "sweep_kernelILi2ELi1ELb0ELb1E")  # one operand set in LDS (round 3)
    sweep4 = find(kernels, "sweep_kernelILi2ELi4ELb0ELb0E")
    k1a = find(pade2, "pade_pq2_kernelILb1ELb0ELb1E")   # Hermitian, product build, step table (the headline's)
    k1a_norm = find(pade2, "pade_pq2_kernelILb1ELb0ELb0E")  # ... deciding order / squarings from the matrix norm
    k3 = find(kernels, "krylov_grad_skew_kernelILi2ELb0ELb0E")  # (the third flag: the copy of latency mode, sweep_umode)
    lu = find(kernels, "lu_kernelILi2E")
    for entry in (sweep, sweep1, sweep4, k1a, k1a_norm, k3, lu):
        assert entry["VGPRs Spill"] == 0 and entry["ScratchSize"] == 0
    granule = lambda r: (r + 7) // 8 * 8  # noqa: E731  (allocation granularity)
    assert granule(total_registers(k1a)) * 2 <= 512          # two K1a waves per SIMD
    assert granule(total_registers(k3)) * 3 <= 512           # three K3 waves per SIMD
    assert granule(total_registers(lu)) * 3 <= 512           # three K1b waves per SIMD
    # the sweep beside one wave of either throughput kernel
    assert granule(total_registers(sweep)) + granule(total_registers(k1a)) <= 512
    assert granule(total_registers(sweep)) + granule(total_registers(k3)) <= 512
    assert granule(total_registers(sweep1)) + granule(total_registers(k1a)) <= 512
    assert granule(total_registers(sweep1)) + granule(total_registers(k3)) <= 512
    assert granule(total_registers(sweep4)) + granule(total_registers(k3)) <= 512


def test_sixteen_tile_kernels_do_not_spill():
    """33 <= n <= 64 (qocx_pade4.hip, qocx_big.hip): a 64 x 64 complex matrix is 256 registers per
    lane of one wave, so K1a / K1b / K3 are four-wave workgroups with a quarter of the columns per
    wave. Their budgets: no scratch in K1b and K3 (the one-wave forms spilled 1000-2000 registers and
    ran 3x slower), at most a few spilled registers in K1a."""
    big = resources("qocx_big.hip")
    pade4 = resources("qocx_pade4.hip")
    lu = find(big, "lu4_kernel")
    assert lu["ScratchSize"] == 0 and lu["VGPRs Spill"] == 0
    assert (total_registers(lu) + 7) // 8 * 8 * 4 <= 512          # four K1b waves per SIMD
    for cols in ("Li16E", "Li12E"):  # 16 columns per wave, or 12 (n <= 48: the pad columns are zero)
        for frag in ("krylov4_kernelILb0ELb1E", "krylov4_kernelILb0ELb0E", "krylov4_kernelILb1ELb1E",
                     "krylov4_kernelILb1ELb0E"):
            k3 = find(big, frag + cols)
            assert k3["ScratchSize"] == 0, frag
        assert total_registers(find(big, "krylov4_kernelILb0ELb1E" + cols)) <= 256   # two per SIMD (Hermitian H)
    # two variants of each (qocx_pade4.hip): the [13/13] path inlined and the low-order path a
    # call (Lb0), or the other way round (Lb1, chosen when the host's norm bound is below theta_9)
    for frag in ("pade_pq4_kernelILi4ELb0ELi0E", "pade_pq4_explicit_kernelILi4ELb0ELi0E",
                 "pade_pq4_kernelILi3ELb0ELi0E", "pade_pq4_explicit_kernelILi3ELb0ELi0E"):
        k1a = find(pade4, frag)
        # (<= 22 spilled registers, plus the call frame of the outlined path)
        assert k1a["ScratchSize"] <= 160 and k1a["VGPRs Spill"] <= 22, (frag, k1a)
    for frag in ("pade_pq4_kernelILi4ELb1ELi0E", "pade_pq4_explicit_kernelILi4ELb1ELi0E",
                 "pade_pq4_kernelILi3ELb1ELi0E", "pade_pq4_explicit_kernelILi3ELb1ELi0E"):
        k1a = find(pade4, frag)
        # (the structured kernels and the nine-tile explicit one are capped at 256 registers - two
        # waves per SIMD: K1a 3.42 -> 1.98 ms per launch at n = 48, 4.77 -> 3.95 at n = 64 - and
        # what they spill sits in the outlined [13/13] path, which these variants do not reach)
        assert k1a["ScratchSize"] <= 1600 and k1a["VGPRs Spill"] <= 300, (frag, k1a)
    assert find(pade4, "pade_pq4_kernelILi3ELb1ELi0E")["Occupancy"] == 2
    assert find(pade4, "pade_pq4_kernelILi3ELb1ELi0E")["VGPRs Spill"] <= 32
    # round 4, Hermitian generators (two thirds of the tiles, every order inline, no call): nine tiles
    # without scratch; sixteen tiles keep their few spills inside the orders 7 / 9
    herm9 = find(pade4, "pade_pq4_kernelILi3ELb1ELi9E")
    assert herm9["Occupancy"] == 2 and herm9["ScratchSize"] == 0 and herm9["VGPRs Spill"] == 0
    herm16 = find(pade4, "pade_pq4_kernelILi4ELb1ELi9E")
    assert herm16["Occupancy"] == 2 and herm16["VGPRs Spill"] <= 80
    # ... and none in the build for norm bounds below theta_5 (orders 3 and 5 only)
    herm16_5 = find(pade4, "pade_pq4_kernelILi4ELb1ELi5E")
    assert herm16_5["Occupancy"] == 2 and herm16_5["ScratchSize"] == 0 and herm16_5["VGPRs Spill"] == 0


def test_general_path_kernels_do_not_spill():
    """65 <= n <= 256 (qocx_general.hip): workgroups of 256 threads on matrices in HBM / L2; the factor kernel
    (MFMA GEMM and the blocked inversion as calls: a frame of saved registers, no spills) fits twice on a CU."""
    general = resources("qocx_general.hip")
    factor = find(general, "factor_kernel")
    assert factor["ScratchSize"] <= 512 and factor["VGPRs Spill"] == 0 and total_registers(factor) <= 256
    k3 = find(general, "13krylov_kernel")  # (the many-state form with its product calls is a kernel of its own)
    assert k3["ScratchSize"] == 0 and k3["VGPRs Spill"] == 0 and total_registers(k3) <= 128
    assert find(general, "krylov_many_kernel")["VGPRs Spill"] <= 16
    # (every caller of the product functions is held to two waves per SIMD - the functions' registers are allocated
    # once, for the loosest caller, and the factor kernel needs its two workgroups per CU; a few spills at most)
    sweep = find(general, "sweep_kernel")
    assert sweep["ScratchSize"] <= 512 and sweep["VGPRs Spill"] <= 32 and total_registers(sweep) <= 256
    for frag in ("magnus_kernelILb0E", "magnus_kernelILb1E"):
        entry = find(general, frag)
        assert entry["VGPRs Spill"] == 0 and total_registers(entry) <= 256


def test_release_library_has_no_diagnostic_switches():
    """the knobs that return garbage (timing experiments), the stamped kernel
    builds and the QOCX_* environment switches exist in libqocx_diag.so only (-DQOCX_DIAG,
    qoc_amd/csrc/qocx_diag.h); the product library rejects the knobs and does not even contain the
    variable names."""
    from qoc_amd import engine
    lib = engine.load_library()
    assert lib.qocx_build_is_diag() == 0
    for name in (b"dbg_skip", b"sweep3_dbg", b"sweep3_stamps", b"lindblad_stamps", b"k1a_dbg",
                 b"k1a_stamps", b"peak_mode"):
        assert lib.qocx_knob_kind(name) == -2, name
    for name in (b"pade_order", b"fuse_lu", b"lu_mfma", b"bidir", b"unit_adjoint", b"sweep_impl"):
        assert lib.qocx_knob_kind(name) == 1, name
    assert lib.qocx_knob_kind(b"no_such_knob") == 0
    blob = open(engine.LIBRARY_PATH, "rb").read()
    for env in (b"QOCX_PQ1", b"QOCX_SWEEP_W", b"QOCX_SEG_WEIGHTS", b"QOCX_SWEEP_IMPL",
                b"QOCX_SWEEP_LOADER", b"QOCX_LINDBLAD_SINGLE_WAVE", b"QOCX_TRACE_HOST"):
        assert env not in blob, env
    # the stamped kernel instantiations are not in the product library either
    assert b"pade_pq2_kernelILb1ELb1E" not in blob


def test_mfma_factorisation_kernels_keep_their_budgets():
    """Round 4 (qocx_lu4m.hip): the nine-tile one-wave factorisation must reach two waves per SIMD
    (at one it was measured 12 % slower) with at most a handful of spilled registers, and the
    two-wave form of n > 48 must not touch scratch (a lane-indexed register array in its four-wave
    predecessor once became a scratch array: a trip to memory in every block of four pivots)."""
    table = resources("qocx_lu4m.hip")
    lu9 = find(table, "lu9_kernelILi2E")
    assert lu9["Occupancy"] == 2 and lu9["VGPRs Spill"] <= 32, lu9
    lu2w = find(table, "lu2w_kernel")  # 49 <= n <= 64: two waves, two tile columns each
    assert lu2w["ScratchSize"] == 0 and lu2w["VGPRs Spill"] == 0 and lu2w["Occupancy"] >= 2, lu2w


def test_tile_per_wave_lindblad_kernels_stay_out_of_scratch():
    """Round 4 (qocx_lindblad4t.hip, 17 <= n <= 32): the stage loops must stay rolled and scratch free.
    Unrolled (to keep the twelve k_j tiles in registers) the kernel was 258 KB of code against 64 KB of
    instruction cache and spilled 700-900 registers; the k_j tiles of a wave go through the seed's HBM
    scratch instead. The combine kernel must fit several workgroups per CU (LDS, registers)."""
    table = resources("qocx_lindblad4t.hip")
    for herm in ("Lb0E", "Lb1E"):
        kern = find(table, "lindblad4t_kernelI" + herm)
        assert kern["ScratchSize"] == 0 and kern["VGPRs Spill"] == 0, kern
        comb = find(table, "lindblad4t_combine_kernelI" + herm)
        assert comb["ScratchSize"] == 0 and comb["VGPRs Spill"] == 0, comb
        assert comb["LDS Size"] <= 40 * 1024 and comb["Occupancy"] >= 2, comb
