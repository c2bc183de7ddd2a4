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
    # the 64-spp depth-of-field configuration's scene and flags (Test7 + a glass sphere)
    "t7g_cfg5": ("Test7_glass", (0.95, 16.12), dict(max_ray_depth=8, glossy_samples=4, shadow_samples=16)),
    # normal maps on a sphere, a cube (with a colour texture and a bump map that the normal map overrides) and a plane
    "nm_default": ("NormalMap", None, dict(max_ray_depth=4)),
    "nm_notex": ("NormalMap", None, dict(max_ray_depth=3, enable_textures=0)),
}

# Frames of Raytracer::render itself (one thread, one random stream through the frame; real_*.npz):
# name -> (scene, depth-of-field override or None, config overrides).  tone_mapping_mode: 1 reinhard, 2 exposure.
REAL_CASES = {
    "real_t5_none": ("Test5", None, dict(max_ray_depth=5)),
    "real_t5_reinhard": ("Test5", None, dict(max_ray_depth=5, tone_mapping_mode=1)),
    "real_t5_exposure": ("Test5", None, dict(max_ray_depth=5, tone_mapping_mode=2, exposure=1.7)),
    "real_t4_exposure_nogamma": ("Test4", None, dict(max_ray_depth=4, tone_mapping_mode=2, exposure=2.5,
                                                     enable_gamma_correction=0)),
    "real_t7_dof": ("Test7", (0.95, 16.12), dict(max_ray_depth=4, shadow_samples=4)),
    "real_nm": ("NormalMap", None, dict(max_ray_depth=3)),
}
