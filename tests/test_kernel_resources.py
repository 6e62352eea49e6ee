"""The register budget of the shipped kernels, read from the BUILT library (scripts/kernel_resources.py: the gfx950 code
objects inside dpilqr_amd/libdpilqr_hip.so and their AMDGPU metadata) -- a build whose hot-path kernels spill fails here.

Round 2 shipped fused workgroup sweeps with 29..48 spilled vector registers and up to 1012 spilled scalar registers each
(the register LU fall-back and loop-invariant lane terms of the fused stage set the allocation of the whole kernel) after
DESIGN.md had said the per-size occupancy caps "follow the spill-free register counts".  The table is committed under
profiles/ per round; these are the bounds it has to keep."""
import re
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "scripts"))


@pytest.fixture(scope="module")
def table():
    import kernel_resources
    if not (kernel_resources.LLVM / "llvm-readelf").exists():
        pytest.skip("no llvm-readelf")
    rows = kernel_resources.resources()
    assert len(rows) > 200          # every translation unit's code object was found
    return {r["demangled"]: r for r in rows}


def test_bench_path_kernels_do_not_spill(table):
    """cfg2's solve loop: the fused wavefront sweep (all three layouts), the line search, the rollout."""
    hot = [k for k in table if re.match(r"k_riccati_mfma<20, 10, (4|8|12), 4, 2, true>$", k)]
    hot += ["k_linesearch_wave<0, 5>", "k_rollout_wave<0, 5>"]
    assert len(hot) == 5
    for k in hot:
        r = table[k]
        assert r["vgpr_spill_count"] == 0 and r["private_segment_fixed_size"] == 0, (k, r)
        assert r["sgpr_spill_count"] <= 16, (k, r)
    # three wavefronts per SIMD need <= 168 registers, two <= 256
    assert table["k_riccati_mfma<20, 10, 12, 4, 2, true>"]["vgpr_count"] <= 168
    assert table["k_riccati_mfma<20, 10, 8, 4, 2, true>"]["vgpr_count"] <= 256


def test_workgroup_sweeps_stay_within_their_spill_bounds(table):
    """cfg3 / cfg4's sweeps (the solve loop's default is the FUSED form): at most a handful of loop-invariant values in
    scratch (reloaded a few times per step), no spilled register inside the phases; the scratch the metadata reports is the
    stack of the out-of-line register LU (lu_fallback_wg: singular Q_uu only since round 4).  Round 4's first build of the row
    swaps inside the blocked elimination put the accumulator tiles into scratch (336 B per lane at n_u = 24: a loop the
    `#pragma unroll` no longer unrolled) and ran 3.8 times slower without any other symptom: the scratch bound below is that
    build's tripwire."""
    wg = {k: r for k, r in table.items() if k.startswith("k_riccati_wg<")}
    assert len(wg) >= 40
    for k, r in wg.items():
        fused = k.endswith("true>")
        assert r["vgpr_spill_count"] <= 16, (k, r)                              # round 2: up to 48
        assert r["sgpr_spill_count"] <= 160, (k, r)                             # round 2: up to 1012 (round 4's swap bookkeeping, two forms of the panel: <= 130, into lanes of a vector register)
        assert r["private_segment_fixed_size"] <= 256, (k, r)                   # the fall-back's stack (<= 228 B); arrays in scratch: 336+
    big = table["k_riccati_wg<60, 30, 4, 2, true>"]
    assert big["vgpr_spill_count"] == 0 and big["vgpr_count"] <= 256            # cfg3's 15-unicycle clusters: two per CU


def test_line_search_of_every_baseline_config_does_not_spill(table):
    """The line search / rollout kernels BASELINE's configurations launch: cfg2 (five DoubleIntDynamics4D), cfg3 (clusters of 1..15
    UnicycleDynamics4D), cfg4 (1..10 QuadcopterDynamics6D); cfg5's twenty-agent problem takes k_forward<..., KDIRECT> (tu_bigfwd).
    Through round 5 <3,14>, <3,15>, <4,6>, <4,7>, <4,8> spilled 1..29 registers, a few of them reloaded in every step of the
    horizon loop.  What they held: the 105 pair costs of a fifteen-agent cluster loaded at once before the first add of their sum
    (now in chunks of eight, same order of adds); per-lane 64-bit pointers for the trajectory loads and candidate stores (now a
    scalar base + a 32-bit lane offset formed per step); the pair table's derived offsets and the candidate index kept alive across
    the loop for one use behind it (now made again there)."""
    names = ["k_linesearch_wave<0, 5>", "k_rollout_wave<0, 5>"]
    names += [f"k_linesearch_wave<3, {k}>" for k in range(1, 16)] + [f"k_rollout_wave<3, {k}>" for k in range(1, 16)]
    names += [f"k_linesearch_wave<4, {k}>" for k in range(1, 11)] + [f"k_rollout_wave<4, {k}>" for k in range(1, 11)]
    for k in names:
        r = table[k]
        assert r["vgpr_spill_count"] == 0 and r["private_segment_fixed_size"] == 0, (k, r)
    # (the other families' six-agent kernels: DoubleIntDynamics6D, HumanDynamicsLin6D lost their spills with the same changes;
    # HumanDynamics6D's keeps 22, Quadcopter12D's up to 54 at one wavefront per SIMD -- neither is on a BASELINE configuration's path)
    for k in ("k_linesearch_wave<1, 6>", "k_linesearch_wave<6, 6>"):
        assert table[k]["vgpr_spill_count"] == 0, (k, table[k])


def test_line_search_team_kernels_do_not_spill(table):
    """The two-wavefront line search (k_linesearch_team): the per-agent constants live in the cost wavefront's registers, the
    rollout's state in the other's -- no spills at any size (the one-wavefront quadcopter kernels spill from five agents on)."""
    lt = {k: r for k, r in table.items() if k.startswith("k_linesearch_team<")}
    assert len(lt) == 30
    for k, r in lt.items():
        # (36 B of scratch at one unicycle: the trigonometric argument reduction's table lookup, as in the one-wavefront kernel)
        assert r["vgpr_spill_count"] == 0 and r["private_segment_fixed_size"] <= 36 and r["vgpr_count"] <= 192, (k, r)


def test_in_sweep_production_kernels_do_not_spill(table):
    """The record-free wavefront sweeps of the six-state family, CarDynamics3D and unhinted four-state clusters (k_riccati_mfma_inprod): no spilled vector
    register and no scratch at either occupancy (they hold an agent's whole Jacobian, 54 doubles, in registers for a moment)."""
    ip = {k: r for k, r in table.items() if k.startswith("k_riccati_mfma_inprod<")}
    assert len(ip) == 30          # six-state 4 sizes, CarDynamics3D 6, four-state 5; one and two wavefronts per SIMD
    for k, r in ip.items():
        assert r["vgpr_spill_count"] == 0 and r["private_segment_fixed_size"] == 0, (k, r)


def test_large_cluster_forward_pass_does_not_spill(table):
    """config 5's line search (k_forward<double, 12, 4, KDIRECT, PIPE>: K[t] dx of all candidates on the matrix pipe, round 6):
    512 registers per lane at one wavefront per SIMD, none spilled (the form that walked K[t]'s columns per lane spilled 12)."""
    r = table["k_forward<double, 12, 4, true, true>"]
    assert r["vgpr_spill_count"] == 0 and r["private_segment_fixed_size"] == 0, r
    for k in ("k_forward<float, 12, 4, true, true>", "k_forward<double, 6, 3, true, true>", "k_forward<double, 4, 2, true, true>"):
        assert table[k]["vgpr_spill_count"] == 0, (k, table[k])


def test_large_cluster_sweep_does_not_spill(table):
    """k_riccati_big (n_x > 60, fp32 arm) runs sixteen wavefronts per workgroup, i.e. 128 registers per lane.  Through round 4
    it spilled 32..66 vector registers (and died with HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION in builds that spilled 82 and
    95: scalar registers spilled into the lanes of a vector register that was spilled in turn is what fitted).  Round 5 removed
    the reasons -- the substitution on the matrix pipe instead of sixteen-row register blocks per thread, the staging loops'
    lane terms formed per step, machine LICM off for the unit (it hoisted two dozen fp64 polynomial constants of sin / cos / tan
    out of the horizon loop) -- and every instantiation, with either form of the Jacobians (-DDPILQR_JAC_SINCOS), now needs at
    most 122 registers and NO scratch for spills.  Held to zero here, where no GPU is needed."""
    big = {k: r for k, r in table.items() if k.startswith("k_riccati_big<")}
    assert len(big) == 8
    for k, r in big.items():
        assert r["vgpr_spill_count"] == 0, (k, r)
        assert r["vgpr_count"] <= 128, (k, r)
        # the stack object of the four-state float instantiation's trigonometric argument reduction (36 B); no spill slots
        assert r["private_segment_fixed_size"] <= 36, (k, r)
