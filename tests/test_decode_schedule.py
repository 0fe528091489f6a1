"""CPU restatement of the stream-K schedule of the decode attention kernels (csrc/attn_decode.hip): how the plan kernel
and the attention kernel cut a decode step's (sequence, kv-head, 32-token tile) list into equal per-wave shares, walk a
share segment by segment, and number the split partials. The device code is the authority; this checks, on random
ragged batches (with graph-padding rows of length 0), the invariants the split workspace and the merge kernel rely on:
every tile is owned by exactly one wave, a segment's partials get distinct slots 0 .. count-1 with count as the kernel
writes it into `meta`, and no slot index reaches the `slots` bound the workspace was sized with. With a shared-prefix
group (nvl_decode_plan's `shared_prefix`): the member rows' leading tiles belong to the shared-prefix pass and nobody else,
the tile of a member's newest token never does, and the pass's own partial slot (the last one) is never a stream-K slot."""
import random

K_TILE, K_WAVES, K_MIN_TILES = 32, 4, 4          # kTile, kWaves, kMinTilesPerWave


def stream_slots(max_context):                   # inline int stream_slots(int64_t max_context)
    return max_context // (K_TILE * K_MIN_TILES) + 3          # (the last slot belongs to the shared-prefix pass)


def grid_waves(batch, hkv, max_context, cus=256):     # mfma8_grid(...) * kWaves with a plan (2 workgroups per CU)
    grid = cus * 2
    max_tiles = batch * hkv * ((max_context + K_TILE - 1) // K_TILE)
    max_wg = (max_tiles + K_WAVES * K_MIN_TILES - 1) // (K_WAVES * K_MIN_TILES)
    return max(1, min(grid, max_wg)) * K_WAVES


def shared_tiles(ctx, shared_blocks, member, tiles_per_block=8):
    """decode_plan_kernel's clamp: the pass takes `shared_blocks` blocks of every member, but never the tile that holds a
    member's newest token: sh <= min over live members of floor((len - 1) / 32); 0 without a live member."""
    lows = [(n - 1) // K_TILE for n, m in zip(ctx, member) if n > 0 and m]
    return min(shared_blocks * tiles_per_block, min(lows)) if (lows and shared_blocks > 0) else 0


def own_tiles(ctx, sh=0, member=None):
    """Tiles of every row's OWN share (chunk_prefix with `skip`): all of them, minus the shared ones for a member."""
    return [max((n + K_TILE - 1) // K_TILE - (sh if (member and member[i]) else 0), 0) if n > 0 else 0
            for i, n in enumerate(ctx)]


def plan(ctx, hkv, nwaves, sh=0, member=None):
    """decode_plan_kernel: header (total, per) + the first segment (b, h, t0, nb) of every wave's share."""
    nbs = own_tiles(ctx, sh, member)
    pre = [0]
    for n in nbs:
        pre.append(pre[-1] + n)
    total = pre[-1] * hkv
    per = max((total + nwaves - 1) // nwaves, K_MIN_TILES)
    ents = []
    for w in range(nwaves):
        g = w * per
        if g >= total:
            ents.append(None)
            continue
        lo = max(b for b in range(len(ctx)) if pre[b] * hkv <= g)          # largest b with hkv * pre[b] <= g
        nb = pre[lo + 1] - pre[lo]
        r = g - pre[lo] * hkv
        ents.append((lo, r // nb, r % nb, nb))
    return total, per, ents, pre


def walk(ctx, hkv, total, per, ents, sh=0, member=None):
    """decode_mfma8_kernel's share loop: yields (wave, b, h, first tile, tiles, partial slot k, partial count written);
    tile indices are relative to the row's OWN share (tile t of the share is tile t + sh of a member's sequence)."""
    nbs = own_tiles(ctx, sh, member)
    for wid, e in enumerate(ents):
        if e is None:
            continue
        b, h, t0, nb = e
        g, g1 = wid * per, min(total, (wid + 1) * per)
        while g < g1:
            run = min(nb - t0, g1 - g)
            seg0 = g - t0
            first = seg0 // per
            yield wid, b, h, t0, run, wid - first, (seg0 + nb - 1) // per - first + 1
            g += run
            if g >= g1:
                break
            if h + 1 < hkv:                                                     # advance(): next kv head, or the next
                h += 1                                                          # sequence that has tiles
            else:
                b += 1
                while nbs[b] == 0:
                    b += 1
                h = 0
            nb, t0 = nbs[b], 0


def test_every_tile_has_one_owner_and_partials_fit_their_slots():
    rng = random.Random(1)
    for trial in range(80):
        batch = rng.choice([1, 2, 7, 64, 131, 256])
        hkv = rng.choice([1, 2, 8])
        max_context = rng.choice([256, 4096, 16384])
        ctx = [0 if rng.random() < 0.1 else rng.randint(1, max_context) for _ in range(batch)]
        if trial % 7 == 0:
            ctx = [max_context] * batch                                          # every sequence at the bound
        if sum(ctx) == 0:
            continue
        nwaves = grid_waves(batch, hkv, max_context, cus=rng.choice([8, 64, 256]))
        total, per, ents, pre = plan(ctx, hkv, nwaves)
        slots = stream_slots(max_context)
        owner, parts = {}, {}
        for wid, b, h, t0, run, k, count in walk(ctx, hkv, total, per, ents):
            assert 0 <= k < slots - 1, (k, slots, ctx[b], per)
            for t in range(t0, t0 + run):
                assert (b, h, t) not in owner
                owner[(b, h, t)] = wid
            parts.setdefault((b, h), []).append((k, count))
        want = {(b, h, t) for b, n in enumerate(ctx) for h in range(hkv) for t in range((n + K_TILE - 1) // K_TILE if n else 0)}
        assert set(owner) == want
        for (b, h), kc in parts.items():
            ks, counts = [k for k, _ in kc], {c for _, c in kc}
            assert len(counts) == 1 and sorted(ks) == list(range(counts.pop())), (b, h, kc)


def test_shared_prefix_group_splits_the_tiles_between_the_pass_and_the_stream_k_shares():
    rng = random.Random(5)
    seen_clamp = seen_full = 0
    for trial in range(80):
        batch = rng.choice([2, 7, 64, 131, 256])
        hkv = rng.choice([1, 2, 8])
        max_context = rng.choice([2048, 4096, 16384])
        blocks = rng.choice([1, 2, 5])
        ctx = [0 if rng.random() < 0.1 else rng.randint(1, max_context) for _ in range(batch)]
        member = [rng.random() < 0.8 for _ in range(batch)]          # (padding rows may carry a stale flag)
        if trial % 3:                                                # mostly: members really are longer than the shared blocks
            ctx = [max(n, blocks * 256 + rng.randint(1, 40)) if (n and m) else n for n, m in zip(ctx, member)]
        if not any(n and m for n, m in zip(ctx, member)):
            continue
        sh = shared_tiles(ctx, blocks, member)
        seen_clamp += sh < blocks * 8
        seen_full += sh == blocks * 8
        nwaves = grid_waves(batch, hkv, max_context, cus=rng.choice([8, 64, 256]))
        total, per, ents, pre = plan(ctx, hkv, nwaves, sh, member)
        slots = stream_slots(max_context)
        owner = {}
        for wid, b, h, t0, run, k, count in walk(ctx, hkv, total, per, ents, sh, member):
            assert 0 <= k < slots - 1                                 # slot slots - 1 is the pass's
            off = sh if member[b] else 0
            for t in range(t0 + off, t0 + off + run):
                assert (b, h, t) not in owner
                owner[(b, h, t)] = wid
        passed = {(b, h, t) for b, n in enumerate(ctx) if n > 0 and member[b] for h in range(hkv) for t in range(sh)}
        want = {(b, h, t) for b, n in enumerate(ctx) for h in range(hkv) for t in range((n + K_TILE - 1) // K_TILE if n else 0)}
        assert not (set(owner) & passed) and set(owner) | passed == want
        for b, n in enumerate(ctx):                                   # the newest token's tile is always the row's own
            if n > 0:
                assert all((b, h, (n - 1) // K_TILE) in owner for h in range(hkv))
    assert seen_clamp and seen_full
