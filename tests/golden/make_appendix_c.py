"""Transcribes the md5 vectors SURVEY.md Appendix C and §8(d) record (outputs of
the reference's own fix_ca_region, recorded by the survey) into
tests/golden/survey_vectors.json.  Data only: parameters + expected md5.
Run from the repo root:  python tests/golden/make_appendix_c.py
"""
import json, os, re

HERE = os.path.dirname(os.path.abspath(__file__))
SURVEY = os.path.join(HERE, "..", "..", "SURVEY.md")

PSETS = {
    "P1": dict(blue=1.5, red=-1.5, lens_x=128, lens_y=96),
    "P2": dict(blue=2.0, red=-2.0, lens_x=28, lens_y=61),
    "P3": dict(blue=6.0, red=-2.4, lens_x=0, lens_y=0),
    "P4": dict(blue=0.0, red=0.0, lens_x=128, lens_y=96, x_blue=1.0, x_red=0.0, y_blue=-7.5, y_red=4.0),
    "P5": dict(blue=-3.3, red=4.7, lens_x=-1, lens_y=-1, x_blue=2.5, x_red=-1.25, y_blue=0.75, y_red=-3.0),
}
INPUTS = {
    "RGBA noise": dict(kind="noise", seed=0xF1CA, channels=4, md5="7a837b6a995669faa8d5d6159afc628c"),
    "RGBA smooth": dict(kind="smooth", channels=4, md5="2d187c1628f712112f057f8e4a1ee686"),
    "RGB noise": dict(kind="noise", seed=0xF1CA, channels=3, md5="9e7a573703b3fc6dd6399db20d869078"),
    "RGB smooth": dict(kind="smooth", channels=3, md5="d13d470269e1dcd38c9d872148053d3d"),
}

def main():
    text = open(SURVEY).read()
    app_c = text[text.index("## Appendix C"):text.index("## Appendix D")]
    cases = []
    for line in app_c.splitlines():
        m = re.match(r"\|\s*(RGBA? (?:noise|smooth))\s*\|\s*(P\d)\s*\|\s*([0-9a-f]{32})\s*\|\s*([0-9a-f]{32})\s*\|\s*([0-9a-f]{32})\s*\|", line)
        if not m:
            continue
        name, pset = m.group(1), m.group(2)
        for interp, md5 in enumerate(m.groups()[2:]):
            cases.append(dict(input=name, pset=pset, width=257, height=193,
                              interpolation=interp, md5=md5))
    assert len(cases) == 60, len(cases)
    full = [
        # SURVEY §8(d) configs 2-4 and the centre / None variants of config 3
        dict(name="config2_linear_centre", kind="noise", seed=0xF1CA0001, width=6000, height=4000, channels=4,
             input_md5="b86c5ff5c21dbe6fd57a8ac9b26bf7de",
             params=dict(blue=2.0, red=-2.0, lens_x=3000, lens_y=2000), interpolation=1,
             md5="32374e105a0feb5f66f89343ea8fb4be"),
        dict(name="config3_cubic_offcentre", kind="noise", seed=0xF1CA0001, width=6000, height=4000, channels=4,
             input_md5="b86c5ff5c21dbe6fd57a8ac9b26bf7de",
             params=dict(blue=2.0, red=-2.0, lens_x=658, lens_y=1280), interpolation=2,
             md5="5716c8780413ca4826c97d134e3c01df"),
        dict(name="config3_cubic_centre", kind="noise", seed=0xF1CA0001, width=6000, height=4000, channels=4,
             input_md5="b86c5ff5c21dbe6fd57a8ac9b26bf7de",
             params=dict(blue=2.0, red=-2.0, lens_x=3000, lens_y=2000), interpolation=2,
             md5="f20e8d3ab699e635ee5a33c759f76e4e"),
        dict(name="config3_none_centre", kind="noise", seed=0xF1CA0001, width=6000, height=4000, channels=4,
             input_md5="b86c5ff5c21dbe6fd57a8ac9b26bf7de",
             params=dict(blue=1.5, red=-1.5, lens_x=3000, lens_y=2000), interpolation=0,
             md5="71ee5284dbc7c767c77aa706b9f39895"),
        dict(name="config4_directional_cubic", kind="noise", seed=0xF1CA0002, width=8192, height=8192, channels=4,
             input_md5="3e26455940d42cf726b04bb305ea1d65",
             params=dict(blue=0.0, red=0.0, lens_x=4096, lens_y=4096, x_blue=1.0, x_red=0.0, y_blue=-7.5, y_red=4.0),
             interpolation=2, md5="8d87177982f44b32c827443198f17ea6"),
    ]
    real = [
        # Appendix C "real-image pins": pixels = PIL .convert('RGB') decode of the
        # reference's sample JPEGs (decoder is external: checked only where
        # /root/reference and the same PIL are present; crops are committed separately)
        dict(name="branches_linear_lens00", file="full-branches.jpg", decoded_md5="b781b1fc5965cfa57bf39adf7f194b5f",
             params=dict(blue=6.0, red=-2.4, lens_x=0, lens_y=0), interpolation=1, md5="d03b3619bf3c4881438883ddd22f2a9c"),
        dict(name="branches_linear_lens658", file="full-branches.jpg", decoded_md5="b781b1fc5965cfa57bf39adf7f194b5f",
             params=dict(blue=6.0, red=-2.4, lens_x=658, lens_y=1280), interpolation=1, md5="c6a42ca1ecd30e0fe393fa3b37de4980"),
        dict(name="branches_cubic_lens658", file="full-branches.jpg", decoded_md5="b781b1fc5965cfa57bf39adf7f194b5f",
             params=dict(blue=6.0, red=-2.4, lens_x=658, lens_y=1280), interpolation=2, md5="b61f0c2a50aa038c0d090a3d6136edcc"),
        dict(name="wat_none_centre", file="full-Wat_Pathum_Wanaram.jpg", decoded_md5="dcf339e923cc0f1929340c67ad1b0de2",
             params=dict(blue=1.5, red=-1.5, lens_x=1944, lens_y=1296), interpolation=0, md5="71c5fb4a77e18f001980aa90ae99ad6f"),
        dict(name="wat_linear_readme", file="full-Wat_Pathum_Wanaram.jpg", decoded_md5="dcf339e923cc0f1929340c67ad1b0de2",
             params=dict(blue=0.0, red=-1.5, lens_x=1944, lens_y=1296), interpolation=1, md5="9c0fd18064d1c0e097bcbbb078c48c80"),
    ]
    out = dict(provenance="SURVEY.md Appendix C and section 8(d): md5 of the full output buffer of the "
                          "reference's fix_ca_region (fix-ca.c:998-1348, bpc=1, show_progress=1), recorded by the survey.",
               psets=PSETS, inputs=INPUTS, small=cases, full=full, real=real)
    with open(os.path.join(HERE, "survey_vectors.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", len(cases), "small +", len(full), "full +", len(real), "real cases")

if __name__ == "__main__":
    main()
