"""CPU: the tile plan's blob layout (deltaconv_amd/csrc/tile_plan.h) as seen from Python (geometry/graph.py: TilePlan.section)
agrees with the library's own size / tile-count functions -- the two sides compute section offsets independently."""
import pytest
import torch

from deltaconv_amd._lib import lib


def _layout(n, nc, mx, k, P):
    # equal-sized clouds: exactly their tiles; ragged: the host-side bound (occupied ids are a dense prefix either way)
    T = nc * ((mx + P - 1) // P) if nc * mx == n else (n + P - 1) // P + nc
    PK = P * k
    r4 = lambda w: (w + 3) & ~3
    o_nu = r4(T * P)
    o_uniq = r4(o_nu + T)
    o_loc = r4(o_uniq + T * PK)
    o_self = r4(o_loc + (T * PK + 1) // 2)
    return T, r4(o_self + (T * P + 1) // 2) + 64


@pytest.mark.parametrize("n,nc,mx,k,P", [(32768, 32, 1024, 20, 64), (32768, 8, 4096, 30, 32), (1535, 3, 700, 10, 64),
                                         (3000, 3, 1000, 20, 64), (65, 1, 65, 64, 32), (0, 0, 0, 20, 64)])
def test_layout_matches_library(n, nc, mx, k, P):
    T, words = _layout(n, nc, mx, k, P)
    assert int(lib.raw("dc_tile_plan_tiles")(n, nc, mx, P)) == T
    assert int(lib.raw("dc_tile_plan_words")(T, k, P)) == words
    if nc * mx == n and n:
        assert T == nc * -(-mx // P)                   # no spare tile ids at all


def test_section_views_cover_the_blob():
    from deltaconv_amd.geometry.graph import TilePlan

    class G:                                       # the fields TilePlan reads from a Graph
        n, k, num_clouds, max_cloud = 1535, 10, 3, 700
    P = 64
    words = int(lib.raw("dc_tile_plan_words")(int(lib.raw("dc_tile_plan_tiles")(G.n, G.num_clouds, G.max_cloud, P)), G.k, P))
    blob = torch.arange(words, dtype=torch.int32)
    plan = TilePlan(G, blob, P)
    T = plan.tiles
    assert plan.section("pts").shape == (T, P) and plan.section("uniq").shape == (T, P * G.k)
    assert plan.section("loc").shape == (T, P * G.k) and plan.section("self").shape == (T, P)
    # sections are disjoint, ordered, 16-byte aligned
    starts = [int(plan.section(s).reshape(-1)[0]) for s in ("pts", "nu", "uniq")]
    assert starts == sorted(starts) and all(s % 4 == 0 for s in starts)
    assert plan.args == (G.n, T, G.k, P)


def test_max_cloud_and_rowblock_limits_exported():
    assert int(lib.raw("dc_tile_plan_max_cloud")()) == 4096
    assert int(lib.raw("dc_rowblock_max_rows")()) == 64
