"""The compile-time lane tables of the prediction's covariance phase (ukf_kernel16.hpp: CovTab) checked on the CPU: the
tables are constexpr, so a host-only program prints them (tests/cpp/covtab_dump.hip, built with hipcc, no GPU call) and the
structure is verified here -- every entry of the new covariance's lower triangle is stored by exactly one lane, the affine
block is updated in place entry for entry, nothing else is written, and the row for filters that do not commit stores
nothing at all."""
import json
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def tables(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    exe = str(tmp_path_factory.mktemp("covtab") / "covtab_dump")
    subprocess.check_call([HIPCC, "-O1", "--offload-arch=gfx950", "-std=c++17",
                           "-I" + os.path.join(ROOT, "slam-pose_estimation_amd", "csrc"),
                           os.path.join(ROOT, "tests", "cpp", "covtab_dump.hip"), "-o", exe],
                          stderr=subprocess.DEVNULL)
    return json.loads(subprocess.run([exe], capture_output=True, text=True, check=True).stdout)


def tri(r, c):
    return r * (r + 1) // 2 + c


@pytest.mark.parametrize("name", ["pose_f64", "pose_f32", "orient_f32"])
def test_every_entry_has_exactly_one_owner(tables, name):
    t = tables[name]
    SZ, D, NL, TR, TC, AEL, PKS, DUM = (t[k] for k in ("SZ", "D", "NL", "TR", "TC", "AEL", "PKS", "DUM"))
    sink = DUM * SZ
    wr, rd = t["wr"], t["rd"]
    ntile = TR * TC
    # tile stores: the nonlinear block (rows < NL) and the cross block (rows >= NL, columns < NL), each entry once
    tile = [w for lane in wr[:16] for w in lane[:ntile] if w != sink]
    want = {(PKS + tri(r, c)) * SZ for r in range(D) for c in range(min(r, NL - 1) + 1)}
    assert len(tile) == len(set(tile)) and set(tile) == want
    # affine block (rows and columns >= NL): updated in place, read offset == write offset where a lane owns an entry
    aff_w = [(lane[ntile + k], lane[ntile + AEL + k]) for lane in wr[:16] for k in range(AEL)]
    owned = [(w, r) for w, r in aff_w if w != sink]
    assert all(w == r for w, r in owned)
    want_aff = {(PKS + tri(r, c)) * SZ for r in range(NL, D) for c in range(NL, r + 1)}
    assert len(owned) == len(want_aff) and {w for w, _ in owned} == want_aff
    assert not (want & want_aff) and len(want) + len(want_aff) == D * (D + 1) // 2      # together: the whole packed triangle
    # reads of lanes without an affine entry stay inside the staged covariance
    assert all(PKS * SZ <= r < (PKS + D * (D + 1) // 2) * SZ for _, r in aff_w)
    # the row of a filter that does not commit: every store goes to the sink
    assert all(w == sink for w in wr[16][:ntile + AEL])
    # plain-noise offsets of the affine entries = (row, column) of the entry that lane owns
    for lane in range(16):
        for k in range(AEL):
            w = wr[lane][ntile + k]
            if w == sink:
                continue
            e = w // SZ - PKS
            r = max(i for i in range(D) if tri(i, 0) <= e)
            c = e - tri(r, 0)
            assert rd[lane][5 + k] == (r * D + c) * SZ
    # weight of the neighbour's half sum: 1 for the two lanes that share a tile of the nonlinear block, 0 for a cross lane
    one_hi = 0x3FF00000 if SZ == 8 else None
    for lane in range(16):
        stores = [w for w in wr[lane][:ntile] if w != sink]
        ws = rd[lane][3] if SZ == 8 else rd[lane][2]
        assert ws in (0, one_hi if SZ == 8 else 0x3F800000)
        if ws == 0 and stores:      # a cross lane owns its whole tile: all rows >= NL
            assert all((w // SZ - PKS) >= tri(NL, 0) for w in stores)


def test_xcd_numbering_is_a_bijection_with_contiguous_eighths(tables):
    """group_of_block: workgroup b runs on XCD b % 8; every XCD must get ONE contiguous run of groups, all groups exactly once"""
    for nb_s, groups in tables["group_of_block"].items():
        nb = int(nb_s)
        assert sorted(groups) == list(range(nb))
        for x in range(8):
            mine = [groups[b] for b in range(x, nb, 8)]
            assert mine == list(range(mine[0], mine[0] + len(mine))) if mine else True
        starts = [groups[x] for x in range(min(8, nb))]
        assert starts == sorted(starts)
