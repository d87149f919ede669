"""How unit_conv2d_wgrad_group deals the units of a grouped weight-gradient grid to the XCDs (csrc/conv_wgrad.hip), checked without a GPU
through unit_conv2d_wgrad_group_layout: every tile of every (layer [, filter tap], split) exactly once, the slots of an XCD gap-free, the units
that contract over the same rows on ONE XCD (profiles/r06_exp_wgrad_gangs.txt), no rounds lost to the gangs."""
import collections
import ctypes
import os

import pytest

from unit_amd import _lib, ops

BF16 = ops.BF16


def layers_of(which):
    L = []
    if which == "res5":       # one Res5 head on 1024 RoIs: block 0 (stride 2 from 14x14) + 2 identity blocks
        n = 1024
        L += [(n, 14, 14, 1024, 512, 1, 2, 0), (n, 7, 7, 512, 512, 3, 1, 1), (n, 7, 7, 512, 2048, 1, 1, 0), (n, 14, 14, 1024, 2048, 1, 2, 0)]
        for _ in range(2):
            L += [(n, 7, 7, 2048, 512, 1, 1, 0), (n, 7, 7, 512, 512, 3, 1, 1), (n, 7, 7, 512, 2048, 1, 1, 0)]
    elif which == "res4":     # a six-block gradient bucket of res4 on four 600x1000 images
        for _ in range(6):
            L += [(4, 38, 63, 1024, 256, 1, 1, 0), (4, 38, 63, 256, 256, 3, 1, 1), (4, 38, 63, 256, 1024, 1, 1, 0)]
    elif which == "long":     # more layers than one launch holds (WG_GROUP_MAX_PROBLEMS = 20)
        for _ in range(9):
            L += [(4, 38, 63, 1024, 256, 1, 1, 0), (4, 38, 63, 256, 256, 3, 1, 1), (4, 38, 63, 256, 1024, 1, 1, 0)]
    else:                     # res3: 128-wide tiles, and one 256 -> 512 shortcut that joins them
        L += [(4, 150, 250, 256, 128, 1, 2, 0), (4, 75, 125, 128, 128, 3, 1, 1), (4, 75, 125, 128, 512, 1, 1, 0), (4, 150, 250, 256, 512, 1, 2, 0)]
        for _ in range(3):
            L += [(4, 75, 125, 512, 128, 1, 1, 0), (4, 75, 125, 128, 128, 3, 1, 1), (4, 75, 125, 128, 512, 1, 1, 0)]
    return L


def planned(which, hint=0):
    L = layers_of(which)
    pr = (ops.WgradProblem * len(L))()
    for q, (n, h, w, c, k, r, stride, pad) in zip(pr, L):
        oh, ow = ops.conv_out_size(h, w, r, r, stride, pad)
        q.x, q.dy, q.partial = 0x10000, 0x20000, 0x30000          # aligned, never read by the layout call
        q.N, q.H, q.W, q.C, q.K, q.R, q.S, q.stride, q.pad, q.OH, q.OW, q.ldy = n, h, w, c, k, r, r, stride, pad, oh, ow, k
    assert _lib.lib().unit_conv2d_wgrad_group_plan(pr, len(L), hint) == 0
    return pr


def layout(pr):
    rows = ctypes.c_int(0)
    buf = (ctypes.c_int * (9 * 4096))()
    assert _lib.lib().unit_conv2d_wgrad_group_layout(pr, len(pr), buf, 4096, ctypes.byref(rows)) == 0
    U = collections.namedtuple("U", "launch kind xcd start tiles prob tap split tile0")
    return [U(*buf[9 * i:9 * i + 9]) for i in range(rows.value)]


def in_map_3x3(q):
    return q.kind == 2 and q.R == 3 and q.stride == 1 and q.pad == 1 and q.OH * q.OW <= 512


def tap_rows(q, tap):
    kr, ks = divmod(tap, 3)
    return q.N * (q.OH - (kr != 1)) * (q.OW - (ks != 1))


@pytest.fixture
def gang_mode():
    old = os.environ.get("UNIT_WGRAD_GANG")
    def set_mode(m):
        if m is None:
            os.environ.pop("UNIT_WGRAD_GANG", None)
        else:
            os.environ["UNIT_WGRAD_GANG"] = str(m)
    yield set_mode
    set_mode(old)


@pytest.mark.parametrize("which,hint", [("res5", 0), ("res5", 4), ("res4", 0), ("res4", 3), ("res3", 0), ("long", 0)])
@pytest.mark.parametrize("mode", [None, 0, 1, 2])
def test_every_tile_once_and_slots_gap_free(which, hint, mode, gang_mode):
    gang_mode(mode)
    pr = planned(which, hint)
    units = layout(pr)
    assert units
    seen = collections.Counter()
    for u in units:
        q = pr[u.prob]
        assert u.kind == q.kind and 0 <= u.xcd < 8 and u.tiles > 0 and u.split < q.splits
        assert u.tap == 0 or in_map_3x3(q)
        for t in range(u.tile0, u.tile0 + u.tiles):
            seen[(u.prob, u.tap, u.split, t)] += 1
    want = 0
    for i, q in enumerate(pr):
        T = 256 if q.kind == 2 else 128
        tiles = (q.R * q.S * q.C // T) * (q.K // T)
        if in_map_3x3(q):
            keys = [(i, tap, s, t) for tap in range(9) for s in range(q.splits) for t in range(tiles // 9)]
        else:
            keys = [(i, 0, s, t) for s in range(q.splits) for t in range(tiles)]
        assert all(seen[k] == 1 for k in keys), (which, i)
        want += len(keys)
    assert sum(seen.values()) == want
    # an XCD's units of one launch tile its slot range without gaps or overlaps
    by = collections.defaultdict(list)
    for u in units:
        by[(u.launch, u.xcd)].append(u)
    for us in by.values():
        us.sort(key=lambda u: u.start)
        assert us[0].start == 0
        assert all(a.start + a.tiles == b.start for a, b in zip(us, us[1:]))
    if which == "long":
        assert len({u.launch for u in units if u.kind == 2}) >= 2          # 27 layers: two grids of 256-tiles


def test_same_pace_taps_of_a_split_share_an_xcd(gang_mode):
    gang_mode(None)                     # the default
    pr = planned("res5")
    units = layout(pr)
    where = collections.defaultdict(set)
    for u in units:
        if in_map_3x3(pr[u.prob]):
            kr, ks = divmod(u.tap, 3)
            where[(u.prob, u.split, (kr != 1) + (ks != 1))].add(u.xcd)
    assert len(where) == 3 * 3 * 3       # three 3x3 layers, three splits, three pace classes
    assert all(len(x) == 1 for x in where.values()), where
    # the tiles of a pointwise (layer, split) already did
    for i, q in enumerate(pr):
        if not in_map_3x3(q):
            for s in range(q.splits):
                assert len({u.xcd for u in units if u.prob == i and u.split == s}) == 1
    gang_mode(1)                        # all nine taps of a split together
    units = layout(planned("res5"))
    where = collections.defaultdict(set)
    for u in units:
        if in_map_3x3(pr[u.prob]):
            where[(u.prob, u.split)].add(u.xcd)
    assert all(len(x) == 1 for x in where.values())
    gang_mode(0)                        # round 5: tap units dealt one by one land on several XCDs
    units = layout(planned("res5"))
    where = collections.defaultdict(set)
    for u in units:
        if in_map_3x3(pr[u.prob]):
            where[(u.prob, u.split)].add(u.xcd)
    assert max(len(x) for x in where.values()) >= 4


def simulated_makespan(pr, units, fixed=6):
    """the 256-tile grid on 8 x 32 CUs: an XCD's CUs take its workgroup slots in order; a tile lasts its 64-pixel steps + a fixed part"""
    worst = 0
    for x in range(8):
        cu = [0] * 32
        for u in sorted((u for u in units if u.xcd == x and u.launch == 0 and u.kind == 2), key=lambda u: u.start):
            q = pr[u.prob]
            rows = tap_rows(q, u.tap) if in_map_3x3(q) else q.N * q.OH * q.OW
            d = -(-(-(-rows // q.splits)) // 64) + fixed
            for _ in range(u.tiles):
                m = cu.index(min(cu))
                cu[m] += d
        worst = max(worst, max(cu))
    return worst


@pytest.mark.parametrize("hint", [0, 4])
def test_gangs_cost_no_rounds(hint, gang_mode):
    """what ends the grid is the slowest XCD's list-scheduling makespan: dealt by summed weight alone (UNIT_WGRAD_DEAL=0) the 16-tile gangs lose
    ~10 % to round quantisation; the makespan rule puts them where the single tap units were"""
    old = os.environ.get("UNIT_WGRAD_DEAL")
    try:
        span = {}
        for deal in (0, 1):
            os.environ["UNIT_WGRAD_DEAL"] = str(deal)
            for mode in (0, 1, 2):
                gang_mode(mode)
                pr = planned("res5", hint)
                span[(deal, mode)] = simulated_makespan(pr, layout(pr))
        assert span[(0, 2)] > 1.03 * span[(0, 0)]                  # the problem
        assert span[(1, 2)] <= span[(0, 0)] and span[(1, 0)] <= span[(0, 0)]
        assert span[(1, 1)] <= span[(0, 1)]                        # 36-tile gangs on 32 CUs: no worse, not cured
    finally:
        if old is None:
            os.environ.pop("UNIT_WGRAD_DEAL", None)
        else:
            os.environ["UNIT_WGRAD_DEAL"] = old


def test_cached_tables_equal_fresh_ones(gang_mode):
    gang_mode(2)
    a = layout(planned("res5"))
    for which in ("res4", "long", "res3", "res5"):                # other problem lists in between; the fourth call is a cache hit
        b = layout(planned(which))
    assert a == b
    gang_mode(0)                                                    # the mode is part of the key
    assert layout(planned("res5")) != a
