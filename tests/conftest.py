import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SCENES = os.path.join(ROOT, "scenes")
sys.path.insert(0, SCENES)
import make_scenes as _make_scenes  # noqa: E402

_make_scenes.ensure()   # scenes/*.xml are generated files (scenes/make_scenes.py), not tracked
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """gpu-marked tests need a HIP device: without one (a build host, a plain `pytest`) they are skipped instead of failing.
    torch.cuda.device_count() does not initialise the HIP runtime on this image."""
    try:
        import torch
        have_gpu = torch.cuda.device_count() > 0
    except Exception:      # noqa: BLE001
        have_gpu = False
    if have_gpu:
        return
    skip = pytest.mark.skip(reason="no HIP device visible (GPU tests run with -m gpu on an MI355X box)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure)."""
    from oracle import orc as _orc
    _orc.lib()
    return _orc


@pytest.fixture(scope="session")
def mi():
    import mitsuba3dopplertof_amd as _mi
    if not os.path.exists(_mi.lib_path()):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "mitsuba3dopplertof_amd", "csrc"), "-j4"])
    _mi._lib()
    return _mi


# The parity configurations: (name, scene file, -D parameters, spp).  Sizes are chosen so that the oracle
# finishes in seconds; they cover every waveform, every time-sampling strategy, correlated/uncorrelated paths,
# odd (non power-of-two) spp, a crop-free non-square film and the instanced Domino scene.
CONFIGS = [
    ("c1_boxes_antithetic", "cornell_boxes.xml", dict(resx=32, resy=32), 16),
    # BASELINE configs[0] exactly: configs_example/scene.xml's room (cornell_boxes), sinusoidal HOMODYNE (hetero_frequency = 0), uniform time sampling
    ("c1_boxes_uniform_homodyne", "cornell_boxes.xml", dict(resx=32, resy=32, time_sampling_method="uniform", hetero_frequency=0.0,
                                                            wave_function_type="sinusoidal"), 16),
    # BASELINE configs[3] exactly: Domino (1 025 objects), rectangular low-pass, antithetic 0.5
    ("c4_domino_rectangular", "domino.xml", dict(resx=96, resy=64, wave_function_type="rectangular", time_sampling_method="antithetic",
                                                 antithetic_shift=0.5), 4),
    ("c2_wall_stratified", "cornell_wall.xml", dict(resx=32, resy=32), 16),
    ("c3_wall_mirror", "cornell_wall.xml", dict(resx=32, resy=24, time_sampling_method="antithetic_mirror", antithetic_shift=0.0), 8),
    ("boxes_uniform_rect", "cornell_boxes.xml", dict(resx=24, resy=32, time_sampling_method="uniform", wave_function_type="rectangular"), 8),
    ("boxes_tri_uncorrelated", "cornell_boxes.xml", dict(resx=32, resy=32, wave_function_type="triangular", path_correlation_depth=0), 8),
    ("boxes_trap_depth6_spp6", "cornell_boxes.xml", dict(resx=16, resy=16, wave_function_type="trapezoidal", max_depth=6, path_correlation_depth=2, time_sampling_method="stratified"), 6),
    ("boxes_tcn4", "cornell_boxes.xml", dict(resx=16, resy=16, time_correlate_number=4, time_sampling_method="antithetic"), 8),
    # ETimeSampling's last two values (sampler.h:27-34; correlated.cpp:147-152): `periodic` -- the tcn samples of a group 1 / tcn of the exposure apart --
    # and `regular` -- one shared time per group
    ("boxes_periodic_tcn4", "cornell_boxes.xml", dict(resx=16, resy=16, time_correlate_number=4, time_sampling_method="periodic"), 8),
    ("wall_regular", "cornell_wall.xml", dict(resx=24, resy=16, time_sampling_method="regular", wave_function_type="triangular"), 8),
    ("domino_small", "domino_small.xml", dict(resx=48, resy=48), 4),
    ("area_light_doppler", "cornell_area.xml", dict(resx=32, resy=32), 16),
    ("area_light_depth6_rr", "cornell_area.xml", dict(resx=24, resy=24, max_depth=6, time_sampling_method="stratified", path_correlation_depth=2), 8),
    # SURVEY 8(f)#3 material / shape families (their own test files hold the analytic checks; these pin the streams)
    ("specular_mirror_glass", "cornell_specular.xml", dict(resx=24, resy=24, max_depth=8), 8),
    ("plastic_boxes", "cornell_plastic.xml", dict(resx=24, resy=24), 8),
    ("rough_conductor_boxes", "cornell_rough.xml", dict(resx=24, resy=24, max_depth=5), 8),
    ("rough_plastic_boxes", "cornell_roughplastic.xml", dict(resx=24, resy=24, max_depth=5), 8),
    ("frosted_glass", "cornell_frosted.xml", dict(resx=24, resy=24, max_depth=6), 8),
    # the Beckmann distribution (the plugins' default `distribution`; restated exp / log / erf / erfinv, see oracle header)
    ("rough_conductor_beckmann", "cornell_rough.xml", dict(resx=24, resy=24, max_depth=5, distribution="beckmann"), 8),
    ("rough_plastic_beckmann", "cornell_roughplastic.xml", dict(resx=24, resy=24, max_depth=5, distribution="beckmann"), 8),
    ("frosted_glass_beckmann", "cornell_frosted.xml", dict(resx=24, resy=24, max_depth=6, distribution="beckmann"), 8),
    # checkerboard / bitmap textures on the diffuse reflectances (rectangles, cube texcoords, a plastic's diffuse_reflectance)
    ("textured", "cornell_textured.xml", dict(resx=32, resy=32, max_depth=4), 8),
    # textures on the other slots: specular_reflectance (conductor, roughconductor, plastic), specular_transmittance and alpha_u / alpha_v (roughdielectric), alpha (roughconductor)
    ("textured_specular", "cornell_textured_specular.xml", dict(resx=32, resy=32, max_depth=5), 8),
    # `mask` BSDFs: constant, checkerboard and bitmap opacities over diffuse / plastic, one- and two-sided; null interactions, point + area light
    ("masked", "cornell_masked.xml", dict(resx=32, resy=32, max_depth=6), 8),
    # valid_ray (dopplertofpath.cpp:101-102,252-253,279-282) in an OPEN scene: veils (`mask`), a `thindielectric` pane and an opaque card in front of the void --
    # a path of null interactions that leaves the scene returns 0, also what it gathered at the veils; the last iteration (which otherwise only looks for
    # emitter hits) still validates; max_depth 1 and 2 make it the first / second
    ("open_veils", "open_veils.xml", dict(resx=32, resy=32, max_depth=5), 8),
    ("open_veils_depth2", "open_veils.xml", dict(resx=32, resy=32, max_depth=2), 8),
    ("open_veils_depth1", "open_veils.xml", dict(resx=24, resy=24, max_depth=1), 8),
    # ... with a constant environment: visible (every ray valid from the start), and lighting the cards but hidden (hide_emitters: it must not show through the cut-outs)
    ("open_veils_env", "open_veils_env.xml", dict(resx=32, resy=32, max_depth=4), 8),
    ("open_veils_env_hidden", "open_veils_env.xml", dict(resx=32, resy=32, max_depth=4, hide_emitters="true"), 8),
    # `normalmap` BSDFs: bitmap and checkerboard normal maps around diffuse / roughconductor / plastic, inside twosided and mask; light-leak rejection
    ("normalmap", "cornell_normalmap.xml", dict(resx=32, resy=32, max_depth=5), 8),
    # `blendbsdf`: constant / checkerboard / bitmap weights, reflecting and transmitting partners, inside twosided and mask, a normal-mapped partner
    ("blend", "cornell_blend.xml", dict(resx=32, resy=32, max_depth=6), 8),
    # area emitters with a textured radiance (bitmap: importance-sampled through DiscreteDistribution2D, bilinear + nearest; checkerboard: uniform), MIS both ways
    ("textured_light", "cornell_textured_light.xml", dict(resx=32, resy=32, max_depth=4), 16),
    # `constant` environment emitter: rays that leave the scene, environment sampling with MIS, valid_ray
    # sample_visible = false: all microfacet normals are sampled (roughconductor / roughplastic weights and densities, roughdielectric with
    # Walter et al.'s roughness scaling)
    ("rough_conductor_all_normals", "cornell_rough.xml", dict(resx=32, resy=32, sample_visible="false", max_depth=5), 8),
    ("rough_plastic_all_normals", "cornell_roughplastic.xml", dict(resx=32, resy=32, sample_visible="false", distribution="beckmann"), 8),
    ("frosted_glass_all_normals", "cornell_frosted.xml", dict(resx=32, resy=32, sample_visible="false", max_depth=6), 8),
    # the bitmap texture from a baseline JPEG file (4:2:0): the product's own decoder against PIL's in the oracle loader
    ("textured_jpeg", "cornell_textured.xml", dict(resx=32, resy=32, texfile="tex_rgb.jpg"), 8),
    ("environment", "cornell_env.xml", dict(resx=32, resy=32, max_depth=4), 8),
    # `envmap` emitter (RGBE file, rotated): latitude-longitude lookup on a miss, hierarchical importance sampling with MIS
    ("envmap", "cornell_envmap.xml", dict(resx=32, resy=32, max_depth=4), 8),
    # `directional` emitters (one by `direction`, one by `to_world`): delta directions sampled from outside the bounding sphere
    ("directional", "cornell_sun.xml", dict(resx=32, resy=32, max_depth=4), 8),
    # `thinlens` sensor: one more correlated 2-D draw per lane (the aperture sample), rays that start on the lens
    ("thinlens", "cornell_thinlens.xml", dict(resx=32, resy=32, max_depth=4, path_correlation_depth=2), 8),
    ("cylinders", "cornell_cylinders.xml", dict(resx=24, resy=24, max_depth=4), 8),
    ("spot_light", "cornell_spot.xml", dict(resx=24, resy=24), 8),
    ("disks", "cornell_disk.xml", dict(resx=24, resy=24, max_depth=5), 8),
    ("spheres", "cornell_spheres.xml", dict(resx=24, resy=24), 8),
    ("sphere_light", "cornell_sphere_light.xml", dict(resx=24, resy=24, max_depth=5), 8),
]


@pytest.fixture(scope="session")
def configs():
    return CONFIGS
