"""The stochastic golden renders of oracle/make_goldens.py and the mirt_config each one stands for."""

# name -> (scene, depth-of-field override or None, config overrides)
RENDER_CASES = {
    "t5_default": ("Test5", None, dict(max_ray_depth=5)),
    "t5_glossy": ("Test5", None, dict(max_ray_depth=8, glossy_samples=8, shadow_samples=16)),
    "t7_dof_soft": ("Test7", (0.95, 16.12), dict(max_ray_depth=8, glossy_samples=4)),
    "t6_soft9": ("Test6", None, dict(max_ray_depth=4, shadow_samples=9)),
    "t2_texture": ("Test2", None, dict(max_ray_depth=4)),
    "t3_bump": ("Test3", None, dict(max_ray_depth=4)),
    "t4_glossy3": ("Test4", None, dict(max_ray_depth=6, glossy_samples=3)),
    "t1_cfg0": ("Test1", None, dict(max_ray_depth=3)),
    "t5_nomb_seed7": ("Test5", None, dict(max_ray_depth=5, enable_motion_blur=0, seed=7)),
}
