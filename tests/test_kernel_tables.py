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


@pytest.mark.parametrize("name", ["orient_f64_o", "orient_f32_o"])
def test_orientation_lane_tables(tables, name):
    """OCovTab (round 4): the OrientationState kernels' 16-bit lane tables.  Every entry of the new covariance's lower triangle has
    exactly one owner, a tile that hangs over the 13 x 13 matrix stores nothing outside it, operand pointers and noise offsets are
    those of the lane's tile, the scale classes of the affine entries follow MT<OrientM>::aff_scale, offsets fit 16 bits."""
    t = tables[name]
    SZ, D, NL, TR, TC, AEL, PKS, DUM, NSH_SINK, TNL, LAF, ST, TRIP = (t[k] for k in ("SZ", "D", "NL", "TR", "TC", "AEL", "PKS", "DUM", "NSH_SINK", "TNL", "LAF", "ST", "TRIP"))
    wr, rd = t["wr"], t["rd"]
    ntile = TR * TC
    tile_sink, aff_sink = NSH_SINK * SZ, DUM * SZ
    assert max(max(r) for r in rd) < 65536 and max(max(w) for w in wr) < 65536
    tile = [w for lane in wr[:16] for w in lane[:ntile] if w != tile_sink]
    want = {(PKS + tri(r, c)) * SZ for r in range(D) for c in range(min(r, NL - 1) + 1)}
    assert len(tile) == len(set(tile)) and set(tile) == want
    aff = [(lane[ntile + k], lane[ntile + AEL + k]) for lane in wr[:16] for k in range(AEL)]
    owned = [(w, r) for w, r in aff if w != aff_sink]
    want_aff = {(PKS + tri(r, c)) * SZ for r in range(NL, D) for c in range(NL, r + 1)}
    assert all(w == r for w, r in owned) and len(owned) == len(want_aff) and {w for w, _ in owned} == want_aff
    assert all(PKS * SZ <= r < (PKS + D * (D + 1) // 2) * SZ for _, r in aff)
    assert all(w == tile_sink for w in wr[16][:ntile]) and all(w == aff_sink for w in wr[16][ntile:ntile + AEL])
    cls = lambda c: 0 if c < 9 else (1 if c < 12 else 2)      # noqa: E731  (gyro-bias, acc-bias, gravity)
    for lane in range(16):
        flags = rd[lane][2]
        stores = [w for w in wr[lane][:ntile] if w != tile_sink]
        rows = sorted({max(i for i in range(D) if tri(i, 0) <= w // SZ - PKS) for w in stores})
        if stores:
            r0 = rows[0]
            c0 = min((w // SZ - PKS) - tri(max(i for i in range(D) if tri(i, 0) <= w // SZ - PKS), 0) for w in stores)
            nonlin = r0 < NL
            assert (flags & 1) == (1 if nonlin else 0)
            # operand pointers: the U rows (first half) of the table / the affine factor rows, the column operand at C0
            assert rd[lane][0] == ((TNL + r0) if nonlin else (LAF + r0 - NL)) * SZ
            assert rd[lane][1] == (TNL + (0 if nonlin else TRIP * ST) + c0) * SZ
            assert rd[lane][3] == (r0 * D + c0) * SZ      # noise: tile origin (immediate offsets reach the other eight entries)
        for k in range(AEL):
            w = wr[lane][ntile + k]
            if w == aff_sink:
                continue
            e = w // SZ - PKS
            r = max(i for i in range(D) if tri(i, 0) <= e)
            c = e - tri(r, 0)
            assert rd[lane][4 + k] == (r * D + c) * SZ
            assert (flags >> (2 + 4 * k)) & 3 == cls(r) and (flags >> (4 + 4 * k)) & 3 == cls(c)


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
