"""CPU restatement of how prefill_attn_kernel numbers its workgroups (csrc/attn_prefill.hip): the XCD-aware grid (on by
default for short launches) and the in-register tile list. The device code is the authority; this spells the two maps
out so that their defining properties — every (q-head, tile) exactly once, a kv group's heads side by side on one XCD,
longest tiles first — are checked without a GPU."""
import itertools
import random

import pytest


def xcd_grid(tiles_bound, hq, hkv):
    """launcher: blocks = ceil(tiles * hkv / 8) * 8 * G;  kernel: block b -> (head, tile_rank) (may exceed the bound)."""
    g_sz = hq // hkv
    groups = tiles_bound * hkv
    blocks = (groups + 7) // 8 * 8 * g_sz
    out = []
    for b in range(blocks):
        xcd, slot = b & 7, b >> 3
        gi, g = divmod(slot, g_sz)
        j = gi * 8 + xcd
        tile_rank, kvh = divmod(j, hkv)
        out.append((b, kvh * g_sz + g, tile_rank))
    return out


@pytest.mark.parametrize("hq,hkv", [(16, 8), (32, 8), (64, 8), (8, 1), (8, 2), (4, 2), (8, 8), (2, 1), (4, 1), (2, 2), (1, 1), (40, 8)])
@pytest.mark.parametrize("tiles", [1, 2, 7, 8, 9, 33, 130])
def test_xcd_numbering_covers_every_head_and_tile_once(hq, hkv, tiles):
    m = xcd_grid(tiles, hq, hkv)
    live = [(h, t) for _, h, t in m if t < tiles]
    assert sorted(live) == sorted(itertools.product(range(hq), range(tiles)))
    # the G heads of one (tile, kv-head) group: consecutive slots of ONE XCD (hardware hands block b to XCD b % 8)
    g_sz = hq // hkv
    by_group = {}
    for b, h, t in m:
        if t < tiles:
            by_group.setdefault((t, h // g_sz), []).append(b)
    for blocks in by_group.values():
        assert len({b & 7 for b in blocks}) == 1
        slots = sorted(b >> 3 for b in blocks)
        assert slots == list(range(slots[0], slots[0] + g_sz))
    # dispatch order = longest first: tile_rank never decreases by more than one group row as b grows along an XCD
    for x in range(8):
        ranks = [t for b, _, t in m if b & 7 == x]
        assert ranks == sorted(ranks)


def tile_list(lens, qblk=128):
    """kernel, <= 64 sequences: lane i holds ceil(len_i / qblk); inclusive scan; tile_rank r -> tile = total - 1 - r;
    seq = #lanes whose inclusive prefix <= tile; q-block = tile - exclusive prefix of seq."""
    vals = [(n + qblk - 1) // qblk for n in lens] + [0] * (64 - len(lens))
    incl = list(itertools.accumulate(vals))
    total = incl[63]
    out = []
    for r in range(total):
        tile = total - 1 - r
        seq = sum(1 for s in incl if s <= tile)
        out.append((seq, tile - (incl[seq] - vals[seq])))
    return out


def test_in_register_tile_list_is_the_reversed_list_of_all_q_blocks():
    rng = random.Random(0)
    for _ in range(200):
        lens = [rng.choice([0, 1, 127, 128, 129, rng.randint(1, 5000)]) for _ in range(rng.randint(1, 64))]
        if sum(lens) == 0:
            continue
        want = [(i, b) for i, n in enumerate(lens) for b in range((n + 127) // 128)]
        assert tile_list(lens) == want[::-1]
