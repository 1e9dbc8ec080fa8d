"""Shared test helpers: golden tables, image construction, comparison."""
import hashlib
import json
import os

import numpy as np

from fixca import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def md5(a) -> str:
    return hashlib.md5(np.ascontiguousarray(a).tobytes()).hexdigest()


def survey_vectors():
    with open(os.path.join(GOLDEN, "survey_vectors.json")) as f:
        return json.load(f)


def real_crops():
    with open(os.path.join(GOLDEN, "real_crops.json")) as f:
        meta = json.load(f)["crops"]
    for m in meta:
        m["image"] = np.load(os.path.join(GOLDEN, m["name"] + ".npy"))
    return meta


def small_input(desc, width=257, height=193):
    if desc["kind"] == "noise":
        return synth.noise_image(desc["seed"], width, height, desc["channels"])
    return synth.smooth_image(width, height, desc["channels"])


def diff_stats(a, b):
    """max |a-b| and the number of differing samples (uint8 images)."""
    d = np.abs(a.astype(np.int16) - b.astype(np.int16))
    return int(d.max()) if d.size else 0, int(np.count_nonzero(d))
