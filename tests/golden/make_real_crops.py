"""Cuts three lossless crops out of the reference's sample photographs
(/root/reference/img-fix-ca/*.jpg, decoded with PIL .convert('RGB')) and stores
them as raw uint8 .npy next to the md5 of the CPU oracle's output for every
interpolation mode.  The crops are DATA (decoded pixels of the reference's own
sample images); the expected md5s are produced by oracle/ after it has been
pinned on the full-size images against SURVEY.md Appendix C (this script
re-checks those five full-size md5s first and refuses to write otherwise).

Run here (needs /root/reference and PIL):  python tests/golden/make_real_crops.py
"""
import hashlib, json, os, sys
import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle  # noqa: E402

REF = "/root/reference/img-fix-ca"
md5 = lambda a: hashlib.md5(np.ascontiguousarray(a).tobytes()).hexdigest()

def main():
    vec = json.load(open(os.path.join(HERE, "survey_vectors.json")))
    decoded = {}
    for c in vec["real"]:
        if c["file"] not in decoded:
            decoded[c["file"]] = np.ascontiguousarray(np.asarray(Image.open(os.path.join(REF, c["file"])).convert("RGB")))
        im = decoded[c["file"]]
        assert md5(im) == c["decoded_md5"], "decoder drift on " + c["file"]
        out = pyoracle.region(im, pyoracle.make_params(interpolation=c["interpolation"], **c["params"]))
        assert md5(out) == c["md5"], "oracle not pinned on " + c["name"]
    crops = [
        # name, file, x, y, w, h, params for the crop treated as its own image
        ("branches_crop", "full-branches.jpg", 600, 1100, 256, 192,
         dict(blue=6.0, red=-2.4, lens_x=58, lens_y=180)),
        ("wat_crop", "full-Wat_Pathum_Wanaram.jpg", 1800, 1200, 255, 191,
         dict(blue=1.5, red=-1.5, lens_x=127, lens_y=95, x_blue=0.5, y_red=-1.25)),
        # 256-aligned (W % 4 == 0, W >= 256): a reference photograph that reaches the main RGB8 kernel (k_roll_rgb)
        ("wat_crop512", "full-Wat_Pathum_Wanaram.jpg", 1688, 1104, 512, 384,
         dict(blue=1.5, red=-1.5, lens_x=256, lens_y=192)),
    ]
    meta = []
    for name, f, x, y, w, h, params in crops:
        crop = np.ascontiguousarray(decoded[f][y:y + h, x:x + w])
        np.save(os.path.join(HERE, name + ".npy"), crop)
        exp = {}
        for interp in (0, 1, 2):
            out = pyoracle.region(crop, pyoracle.make_params(interpolation=interp, **params))
            exp[str(interp)] = md5(out)
        meta.append(dict(name=name, source=f, rect=[x, y, w, h], params=params,
                         input_md5=md5(crop), oracle_md5=exp))
    json.dump(dict(note="crops of the reference's sample JPEGs (PIL decode); oracle_md5 = output of oracle/ "
                        "(pinned on the full images against SURVEY Appendix C by this script)", crops=meta),
              open(os.path.join(HERE, "real_crops.json"), "w"), indent=1)
    print("ok", [m["name"] for m in meta])

if __name__ == "__main__":
    main()
