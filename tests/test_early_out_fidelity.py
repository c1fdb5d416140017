"""CPU tier: how far the ordered-phase early-out schedule (the GPU's, restated in the oracle) is from the
reference's serial order — the published curve of DESIGN.md §3.2, pinned on a 320x240 frame so the whole
file runs in seconds.  The GPU tier asserts that the HIP path is bit-exact against the restated schedule
(tests/test_parity_gpu.py: ordered_phases_*), so these bounds carry over to the product."""
from tests.early_out_fidelity import fidelity


def test_fidelity_curve_against_the_serial_reference_order():
    ss, rows = fidelity("320x240", [16, 32, 64])  # ~10 s
    by = {r["growth"]: r for r in rows}
    # one generation per phase: as close as the chain structure gets (upstream "mixed" order, 75 chains of 1024 generations: 0.9996)
    assert by[16]["touched_jaccard"] >= 0.995, by[16]
    # doubling phases (the exact mode's seed; 0.941 here, 0.962 at 640x480)
    assert by[32]["touched_jaccard"] >= 0.93, by[32]
    assert 1.0 <= by[32]["updates_ratio"] < 1.2, by[32]
    # coarser schedules drift further from the serial order and do more work: the default sits at the knee
    assert by[64]["touched_jaccard"] <= by[32]["touched_jaccard"] + 0.01
    assert by[64]["updates_ratio"] >= by[32]["updates_ratio"] - 0.02
    for r in rows:
        assert r["block_jaccard"] >= 0.99, r


def test_c4_geometry_update_count_within_ten_percent_of_serial():
    """2 cm voxels, 10 m rays: a frame makes far more voxel visits than the approximate set has slots."""
    ss, rows = fidelity("c4geom_small", [16, 32], with_maps=False)
    for r in rows:
        assert abs(r["updates_ratio"] - 1.0) < 0.10, r
