"""CPU checks on the generated gfx950 code of the row-rolling kernel (hipcc cross-compiles
without a GPU).  Two properties of fixca_roll.hip rest on what the compiler emits:

  * the DPP byte->float conversions are hand-written asm, which the compiler's hazard
    recognizer does not see: a VALU write of a register within two instructions before a DPP
    read of it needs wait states the asm does not carry.  Their sources must come from LDS loads.
  * the 16-byte row stores must not carry the row offset in the SGPR soffset operand (hipcc emits
    no wait state for that form and on MI355X the store then picks up the next row's bytes).
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gimp-fix-ca_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"


_ASM_CACHE = {}


def _asm(tmp_path_factory, source, skip_kernels=None, only_kernels=None):
    """instruction lines of the translation unit (compiled once per source and test session); skip_kernels / only_kernels:
    a regex on the mangled names of kernels to leave out / to keep"""
    if not os.path.exists(HIPCC) and not shutil.which("hipcc"):
        pytest.skip("hipcc not available")
    if source not in _ASM_CACHE:
        out = tmp_path_factory.mktemp("isa") / (source + ".s")
        subprocess.check_call([HIPCC if os.path.exists(HIPCC) else "hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17",
                               "-ffp-contract=off", "-S", "--cuda-device-only", "-o", str(out),
                               os.path.join(CSRC, source)], stderr=subprocess.DEVNULL)
        _ASM_CACHE[source] = out.read_text().split("\n")
    lines, skipping = [], False
    for l in _ASM_CACHE[source]:
        m = re.match(r"(_Z\w+):", l)
        if m:
            skipping = bool(skip_kernels and re.search(skip_kernels, m.group(1))) or bool(only_kernels and not re.search(only_kernels, m.group(1)))
        if not skipping and l.startswith("\t") and not l.strip().startswith((";", ".")):
            lines.append(l)
    return lines


# k_roll<TAPS, P2, 1> is the instantiation for widths that are not multiples of 4 (a lane of 1-3 pixels, the lane's own
# pixels read through a byte shift): the properties below are those of the common kernel, k_roll<TAPS, P2, 0>
ODD_WIDTH_KERNELS = r"k_roll(_rgb)?ILi\dELb[01]ELi1EEE"


@pytest.fixture(scope="module")
def roll_asm(tmp_path_factory):
    return _asm(tmp_path_factory, "fixca_roll.hip", ODD_WIDTH_KERNELS)


@pytest.fixture(scope="module")
def roll_rgb_asm(tmp_path_factory):
    return _asm(tmp_path_factory, "fixca_roll_rgb.hip", ODD_WIDTH_KERNELS)


def _dst_regs(line):
    t = line.strip().replace(",", " ").split()
    if len(t) < 2:
        return set()
    m = re.match(r"v\[(\d+):(\d+)\]$", t[1])
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", t[1])
    return {int(m.group(1))} if m else set()


def _check_dpp(asm, at_least):
    ndpp = 0
    for i, l in enumerate(asm):
        if "_dpp" not in l:
            continue
        ndpp += 1
        src = int(re.match(r"v(\d+)$", l.strip().replace(",", " ").split()[2]).group(1))
        for back in (1, 2):
            p = asm[i - back].strip()
            assert not (p.startswith("v_") and src in _dst_regs(p)), "DPP hazard: %s -> %s" % (p, l.strip())
    assert ndpp >= at_least      # the conversions are there at all


def test_dpp_sources_are_not_fresh_valu_results(roll_asm):
    _check_dpp(roll_asm, 16)


def test_rgb8_tap_windows_are_picked_by_conversion_not_by_shifts(roll_rgb_asm):
    """RGB8 (fixca_roll_rgb.hip): the byte alignment of a lane's 12-byte tap window is a constant of each of four
    instantiations, so the bytes are picked by v_cvt_f32_ubyteN (own pixels) and v_cvt_f32_ubyteN_dpp (the next lane's):
    no v_alignbyte and no v_mov_dpp in the kernel, and all four byte selectors occur in DPP form.  The hand-written DPP
    conversions read LDS-load results only (the hazard the RGBA test checks)."""
    _check_dpp(roll_rgb_asm, 64)
    text = "\n".join(roll_rgb_asm)
    assert "v_alignbyte_b32" not in text and "v_mov_b32_dpp" not in text
    for k in range(4):
        assert "v_cvt_f32_ubyte%d_dpp" % k in text
    stores = [l.strip() for l in roll_rgb_asm if l.strip().startswith("buffer_store_dwordx3")]
    assert stores
    for st in stores:                                   # same store rule as RGBA: the scalar offset is the literal 0
        ops = [o.strip() for o in st.split(None, 1)[1].split(",")]
        assert ops[3].split()[0] == "0", st


@pytest.mark.parametrize("source", ["fixca_roll.hip", "fixca_roll_rgb.hip"])
def test_odd_width_instantiations_keep_the_hazard_rules(tmp_path_factory, source):
    """k_roll<TAPS, P2, 1> (widths that are not multiples of 4; edge pixels): the hand-written DPP conversions still read LDS-load
    results only, and every 12 / 16-byte row store still has the literal 0 as its scalar offset"""
    asm = _asm(tmp_path_factory, source, only_kernels=ODD_WIDTH_KERNELS)
    _check_dpp(asm, 16)
    stores = [l.strip() for l in asm if l.strip().startswith(("buffer_store_dwordx3", "buffer_store_dwordx4"))]
    assert stores
    for st in stores:
        ops = [o.strip() for o in st.split(None, 1)[1].split(",")]
        assert ops[3].split()[0] == "0", st


def test_row_stores_keep_soffset_constant(roll_asm):
    stores = [l.strip() for l in roll_asm if l.strip().startswith("buffer_store_dwordx4")]
    assert stores
    for s in stores:
        # buffer_store_dwordx4 vdata, voffset, srsrc, soffset offen ...: soffset must be the literal 0
        ops = [o.strip() for o in s.split(None, 1)[1].split(",")]
        assert ops[3].split()[0] == "0", s


def test_ring_loads_are_lds_dma_and_scalar_tables(roll_asm):
    text = "\n".join(roll_asm)
    assert re.search(r"buffer_load_dwordx4 .* lds", text)          # LDS-DMA ring fill
    assert "s_load_dwordx" in text                                 # block descriptors / weights via SMEM
    assert "v_readlane_b32" not in text                            # 3x the cost of a VALU op: tables replaced them


@pytest.mark.parametrize("source", ["fixca_roll.hip", "fixca_lin.hip", "fixca_cub.hip"])
def test_ring_loads_are_not_wrapped_in_waterfall_loops(tmp_path_factory, source):
    """An LDS-DMA instruction takes its buffer descriptor from scalar registers.  When the compiler holds the descriptor in
    vector registers (round 3: the first source row of a tile, wave-uniform but computed with fp64 vector instructions) it
    wraps every such load in a loop of 4 v_readfirstlane, 2 compares, saveexec and a branch: 1.5-3 points of the HBM peak
    on the LINEAR kernels.  k_lin and k_cub force the value into a scalar register; this keeps it that way.
    (k_move is left as the compiler builds it: see the comment there.)"""
    asm = _asm(tmp_path_factory, source)
    dma = [i for i, l in enumerate(asm) if re.search(r"buffer_load_dword\S*\s.*\blds\b", l)]
    assert len(dma) >= 8, "no LDS-DMA found in " + source
    for i in dma:
        before = " ".join(asm[max(0, i - 3):i])
        assert "saveexec" not in before, (source, asm[max(0, i - 8):i + 1])
