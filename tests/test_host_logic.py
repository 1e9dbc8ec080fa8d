"""CPU tests of the host-side logic: the planner's strip widths against a numpy
model of the row-scan kernel, and the model against the oracle (<= 1 LSB)."""
import re

import numpy as np
import pytest

import fixca
import kernel_model
import pyoracle
from fixca import synth
from helpers import diff_stats

PSETS = [
    dict(blue=1.5, red=-1.5, lens_x=0.5, lens_y=0.5),                 # lens as a fraction of the size
    dict(blue=6.0, red=-2.4, lens_x=0, lens_y=0),
    dict(lens_x=0.5, lens_y=0.5, x_blue=1.0, y_blue=-7.5, y_red=4.0),
    dict(blue=-3.3, red=4.7, lens_x=-1, lens_y=-1, x_blue=2.5, x_red=-1.25, y_blue=.75, y_red=-3),
    dict(blue=-20, red=25, lens_x=0.33, lens_y=0.5, x_blue=-29, x_red=30, y_blue=30, y_red=-30),
]


def _params(kw, w, h, interp):
    kw = dict(kw)
    for k, n in (("lens_x", w), ("lens_y", h)):
        if 0 < kw[k] < 1:
            kw[k] = int(kw[k] * n)
    return kw


@pytest.mark.parametrize("size", [(257, 193, 4), (300, 70, 3), (64, 40, 4), (1000, 50, 4)])
@pytest.mark.parametrize("pset", range(len(PSETS)))
@pytest.mark.parametrize("interp", [1, 2])
def test_model_within_one_lsb(size, pset, interp, monkeypatch):
    monkeypatch.setenv("FIXCA_NO_SCAN", "1")       # the model describes the register-window kernel
    monkeypatch.setenv("FIXCA_NO_ROLL", "1")
    w, h, c = size
    kw = _params(PSETS[pset], w, h, interp)
    p = fixca.make_params(interpolation=interp, **kw)
    k, desc = fixca.plan(w, h, c, 1, p)
    assert k == 3, desc
    oc = int(re.search(r"oc=(\d+)", desc).group(1))
    img = synth.noise_image(5, w, h, c)
    ref = pyoracle.region(img, pyoracle.make_params(interpolation=interp, **kw))
    out = kernel_model.run(img, p, oc)
    mx, n = diff_stats(out, ref)
    assert mx <= 1
    assert n <= 0.02 * ref.size                   # ties at .5 are the only disagreements


def test_model_smooth_image_mostly_exact():
    img = synth.smooth_image(257, 193, 4)
    kw = dict(blue=2.0, red=-2.0, lens_x=28, lens_y=61)
    p = fixca.make_params(interpolation=2, **kw)
    ref = pyoracle.region(img, pyoracle.make_params(interpolation=2, **kw))
    out = kernel_model.run(img, p, 252)
    mx, n = diff_stats(out, ref)
    assert mx <= 1 and n <= 0.001 * ref.size
