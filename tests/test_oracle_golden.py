"""The CPU oracle against every golden vector the survey recorded from the
reference's own fix_ca_region (60 small cases; §8(d): the
full-size configs, marked slow) and against its own structural properties."""
import os

import numpy as np
import pytest

import pyoracle
from fixca import synth
from helpers import md5, real_crops, small_input, survey_vectors

VEC = survey_vectors()


def test_noise_generator_known_answer():
    assert list(synth.noise_bytes(1, 8)) == [193, 92, 2, 137, 236, 45, 10, 145]


@pytest.mark.parametrize("name", sorted(VEC["inputs"]))
def test_small_inputs_md5(name):
    assert md5(small_input(VEC["inputs"][name])) == VEC["inputs"][name]["md5"]


@pytest.mark.parametrize("case", VEC["small"], ids=lambda c: "%s-%s-i%d" % (c["input"].replace(" ", ""), c["pset"], c["interpolation"]))
def test_appendix_c(case):
    img = small_input(VEC["inputs"][case["input"]], case["width"], case["height"])
    p = pyoracle.make_params(interpolation=case["interpolation"], **VEC["psets"][case["pset"]])
    assert md5(pyoracle.region(img, p)) == case["md5"]


@pytest.mark.slow
@pytest.mark.parametrize("case", [c for c in VEC["full"] if c["width"] <= 6000], ids=lambda c: c["name"])
def test_full_size_configs(case):
    img = synth.noise_image(case["seed"], case["width"], case["height"], case["channels"])
    assert md5(img) == case["input_md5"]
    p = pyoracle.make_params(interpolation=case["interpolation"], **case["params"])
    out = pyoracle.region(img, p, threads=4)      # row bands are independent: same bytes as 1 thread
    assert md5(out) == case["md5"]


@pytest.mark.parametrize("crop", real_crops(), ids=lambda c: c["name"])
def test_real_crops(crop):
    assert md5(crop["image"]) == crop["input_md5"]
    for interp in (0, 1, 2):
        p = pyoracle.make_params(interpolation=interp, **crop["params"])
        assert md5(pyoracle.region(crop["image"], p)) == crop["oracle_md5"][str(interp)]


@pytest.mark.parametrize("interp", [0, 1, 2])
def test_identity_when_no_shift(interp):
    """amounts 0 and shifts 0: scale is exactly 1 and every mode returns the input."""
    img = synth.noise_image(7, 64, 48, 4)
    out = pyoracle.region(img, pyoracle.make_params(lens_x=32, lens_y=24, interpolation=interp))
    assert np.array_equal(out, img)


@pytest.mark.parametrize("interp", [0, 1, 2])
def test_green_alpha_untouched_and_row_range(interp):
    img = synth.noise_image(9, 70, 50, 4)
    p = pyoracle.make_params(blue=3.0, red=-2.0, lens_x=10, lens_y=40, interpolation=interp, x_blue=1.5, y_red=-2.25)
    full = pyoracle.region(img, p)
    assert np.array_equal(full[..., 1], img[..., 1]) and np.array_equal(full[..., 3], img[..., 3])
    part = pyoracle.region(img, p, y1=11, y2=37, dst=np.full_like(img, 0xAB))
    assert np.array_equal(part[11:37], full[11:37])
    assert np.all(part[:11] == 0xAB) and np.all(part[37:] == 0xAB)      # rows outside [y1,y2) untouched


def test_mt_equals_scalar():
    img = synth.noise_image(11, 129, 97, 3)
    p = pyoracle.make_params(blue=-3.3, red=4.7, lens_x=-1, lens_y=-1, interpolation=2, x_blue=2.5, y_red=-3)
    assert np.array_equal(pyoracle.region(img, p), pyoracle.region(img, p, threads=3))


def test_refuses_undefined_domain():
    img = synth.noise_image(3, 16, 16, 4)
    with pytest.raises(ValueError):        # max_dim + amount <= 0: the reference divides by <= 0
        pyoracle.region(img, pyoracle.make_params(blue=-30.0, lens_x=8, lens_y=8))


def test_reciprocal_division_is_exact(tmp_path):
    """k_wcol replaces get_pixel's division by 2^n - 1 (fix-ca.c:720-726) with multiply + two FMAs; that
    is the correctly rounded quotient for every u8, u16 and u32 sample value (exhaustive, ~10 s)."""
    import subprocess
    exe = tmp_path / "divcheck"
    src = os.path.join(os.path.dirname(__file__), "c", "divcheck.c")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-o", str(exe), src, "-lm"])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.split() == ["u8", "0", "u16", "0", "u32", "0"], out.stdout


def test_weight_form_error_bound(tmp_path):
    """k_wcol's 16-bit CUBIC weight form (fixca_wide.hip): over 12 million windows (random, extreme steps, nearly
    flat, ramps; random and coarse dyadic fractions) the weight form stays within 1e-8 samples of the reference's
    operation order (the kernel rechecks within 1e-6 of a rounding boundary), and no unflagged sample rounds
    differently"""
    import subprocess
    exe = tmp_path / "weightbound"
    src = os.path.join(os.path.dirname(__file__), "c", "weightbound.c")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-o", str(exe), src, "-lm"])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    worst, unflagged, flagged = out.stdout.split()
    assert out.returncode == 0 and float(worst) < 1e-8 and int(unflagged) == 0 and int(flagged) > 0, out.stdout


def test_weight_form_error_bound_linear(tmp_path):
    """the 16-bit LINEAR weight form (k_lin<u16, .., FAST>, fixca_lin.hip): over 24 million 2 x 2 windows the form
    fma(fy, h1 - h0, h0), h = fma(fx, s1 - s0, s0) stays within 1e-9 samples of the reference's operation order
    (analytic bound 2.2e-10, recheck band 1e-6), and no unflagged sample rounds differently"""
    import subprocess
    exe = tmp_path / "weightbound_linear"
    src = os.path.join(os.path.dirname(__file__), "c", "weightbound_linear.c")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-o", str(exe), src, "-lm"])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    worst, unflagged, flagged = out.stdout.split()
    assert out.returncode == 0 and float(worst) < 1e-9 and int(unflagged) == 0 and int(flagged) > 0, out.stdout
