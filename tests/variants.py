"""Configuration variants exercised by both parity tiers (oracle vs real reference on CPU,
HIP vs oracle on GPU): every TsdfIntegratorBase::Config / SemanticConfig knob the hot path reads."""
VARIANTS = {
    "no_carving": dict(voxel_carving_enabled=0),
    "const_weight": dict(use_const_weight=1),
    "no_clear": dict(allow_clear=0),
    "no_dropoff": dict(use_weight_dropoff=0),
    "sparsity": dict(use_sparsity_compensation_factor=1, sparsity_compensation_factor=2.5),
    "short_rays": dict(min_ray_length_m=0.5, max_ray_length_m=3.0),
    "coarse_voxels": dict(voxel_size=0.1, truncation_distance=0.3),
    "subsample_1": dict(start_voxel_subsampling_factor=1.0),
    "subsample_4": dict(start_voxel_subsampling_factor=4.0),
    "clear_every_3": dict(clear_checks_every_n_frames=3),
    "p_0.9_no_dynamic": dict(semantic_measurement_probability=0.9, dynamic_labels=[]),
    "many_dynamic": dict(dynamic_labels=[3, 19, 20]),
    "max_weight_2": dict(max_weight=2.0),
    "anti_grazing": dict(enable_anti_grazing=1),
}


ORDER_MODE_NAMES = {0: "mixed", 1: "sorted", 2: "mixed_1024_groups"}   # oracle/ref_py.Reference(order_mode=...)


def random_combo(seed: int) -> dict:
    """A seeded random COMBINATION of the knobs above (interactions between options: carving x
    clearing x drop-off x weights x ray limits x set bookkeeping x colour mode)."""
    import numpy as np
    rng = np.random.default_rng(1000 + seed)
    names = sorted(VARIANTS)
    kw = {}
    for n in names:
        if n in ("coarse_voxels",):   # keeps the 5 cm geometry the synthetic frames are made for
            continue
        if rng.random() < 0.4:
            kw.update(VARIANTS[n])
    if "start_voxel_subsampling_factor" not in kw and rng.random() < 0.3:
        kw["start_voxel_subsampling_factor"] = float(rng.choice([1.5, 3.0]))
    kw["color_mode"] = int(rng.integers(0, 3))
    kw["integration_order_mode"] = int(rng.integers(0, 3))   # mixed (upstream form) | sorted | mixed, 1024 groups
    return kw
