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


# ---------------------------------------------------------------------------------------------------------------
# The PERSISTENT form (round 4): workgroup bx of a grid of nb workgroups walks blocks bx, 2 nb - 1 - bx, 2 nb + bx, ...
# (a snake over the longest-first numbering) until the first block whose tile rank is past the end of the list.
def snake_walk(bx, nb, total_ranks, hq, hkv):
    """kernel find_next(): the (head, tile_rank) items workgroup bx processes, in order."""
    g_sz = hq // hkv
    items, rnd = [], 0
    while True:
        b = (rnd + 1) * nb - 1 - bx if rnd & 1 else rnd * nb + bx
        xcd, slot = b & 7, b >> 3
        gi, g = divmod(slot, g_sz)
        j = gi * 8 + xcd
        rank, kvh = divmod(j, hkv)
        if rank >= total_ranks:
            return items
        items.append((kvh * g_sz + g, rank, b))
        rnd += 1


@pytest.mark.parametrize("hq,hkv", [(16, 8), (32, 8), (8, 1), (64, 8), (8, 8), (2, 1)])
@pytest.mark.parametrize("tiles", [1, 5, 64, 128, 145, 1000])
def test_persistent_snake_walk_visits_every_item_exactly_once(hq, hkv, tiles):
    """Launcher: blocks = ceil(tiles x Hkv / 8) x 8 x G, grid = min(blocks, 2 x CUs = 512). Every (q-head, tile rank) with
    rank < tiles is processed by exactly one workgroup; a workgroup stops at its FIRST invalid block — valid only because
    ranks never decrease along a walk; the heads of a kv group keep meeting on one physical XCD (block ids that differ
    by 8 map to physical workgroups that differ by 8, forwards and backwards)."""
    g_sz = hq // hkv
    blocks = (tiles * hkv + 7) // 8 * 8 * g_sz
    nb = min(blocks, 512)
    seen = {}
    for bx in range(nb):
        walk = snake_walk(bx, nb, tiles, hq, hkv)
        ranks = [r for _, r, _ in walk]
        assert ranks == sorted(ranks)                       # longest first within a walk too
        for head, rank, b in walk:
            assert (head, rank) not in seen
            seen[(head, rank)] = bx
    assert sorted(seen) == sorted(itertools.product(range(hq), range(tiles)))
    if nb % 8 == 0:
        for (head, rank), bx in seen.items():
            mate = seen[(head // g_sz * g_sz, rank)]        # first head of the same (tile, kv-head) group
            assert mate & 7 == bx & 7                       # same physical XCD


def test_persistent_snake_walk_balances_uniform_batches():
    """Equal-length sequences and an even number of full rounds (what the launcher requires before it takes the persistent
    form by itself): the tiles a workgroup processes differ by at most one q-block's worth between any two workgroups.
    Work of an item = its q-block index + 1 (causal: key tiles grow with the q-block); ranks are longest-first."""
    seqs, qblocks, hq, hkv = 16, 8, 16, 8                   # 16 x 1024 tokens, Qwen3-0.6B heads: 2048 items, 512 workgroups
    tiles = seqs * qblocks
    order = [(s, b) for s in range(seqs) for b in range(qblocks)][::-1]           # tile rank -> (sequence, q-block)
    work = []
    for bx in range(512):
        work.append(sum(order[rank][1] + 1 for _, rank, _ in snake_walk(bx, 512, tiles, hq, hkv)))
    assert max(work) - min(work) <= qblocks
