"""CPU restatement of how linear_wide_kernel (csrc/gemm_wide.hip) lays an x tile out in LDS, for both k steps (128 columns
per step; 64 in the one-row-group form at 193-256 rows, round 4): the LOADER wave's LDS-DMA instructions write a
lane-linear image (64 lanes x 16 B = 1 KiB per instruction) and apply the XOR swizzle on the SOURCE side; the CONSUMER
waves read MFMA B fragments with ds_read_b128 at `frag_off`. The device code is the authority; this spells both maps out
and checks, without a GPU, that (1) every fragment read returns exactly the 16-byte chunk of x it is supposed to, and
(2) the 16 lanes of a fragment read that are served together (fixed k-chunk, rows 0-15 of a row tile) fall on 16
different 16-byte slots of the 256-byte bank row: no LDS bank conflicts."""
import itertools

import pytest


def loader_image(bk, mt, nl):
    """LDS byte offset (within a stage) -> (row, 16-byte chunk of that row's k step) as the loader wave(s) fill it."""
    row_b = bk * 2
    piece_rows = 1024 // row_b
    pieces = mt * 16 // piece_rows
    image = {}
    for lw in range(nl):                                   # loader wave lw stages pieces lw, lw + nl, ...
        for i in range(pieces // nl):
            piece = i * nl + lw
            for lane in range(64):
                prow, slot = (lane >> 4, lane & 15) if bk == 128 else (lane >> 3, lane & 7)
                row = piece_rows * piece + prow
                chunk = slot ^ (row & 15) if bk == 128 else slot ^ ((row >> 1) & 7)
                off = piece * 1024 + lane * 16             # global_load_lds: M0 + 16 * lane
                assert off not in image
                image[off] = (row, chunk)
    return image


def frag_off(bk, l15, lq, kb):
    row_b = bk * 2
    c = kb * 4 + lq
    return l15 * row_b + ((c ^ l15) << 4) if bk == 128 else l15 * row_b + ((c ^ ((l15 >> 1) & 7)) << 4)


@pytest.mark.parametrize("bk,mt,nl", [(128, 9, 1), (128, 9, 2), (128, 12, 1), (128, 16, 2), (64, 16, 1), (64, 16, 2), (64, 13, 1)])
def test_consumer_fragments_read_what_the_loader_staged(bk, mt, nl):
    if (mt * 16 // (1024 // (bk * 2))) % nl:
        pytest.skip("pieces do not divide among the loader waves (not instantiated)")
    image = loader_image(bk, mt, nl)
    row_b = bk * 2
    assert len(image) == mt * 16 * row_b // 16             # the image is a bijection onto the stage
    for t, l15, lq, kb in itertools.product(range(mt), range(16), range(4), range(bk // 32)):
        off = t * 16 * row_b + frag_off(bk, l15, lq, kb)
        assert image[off] == (t * 16 + l15, kb * 4 + lq), (t, l15, lq, kb)


@pytest.mark.parametrize("bk", [128, 64])
def test_fragment_reads_are_bank_conflict_free(bk):
    for lq, kb in itertools.product(range(4), range(bk // 32)):
        slots = {(frag_off(bk, l15, lq, kb) >> 4) & 15 for l15 in range(16)}       # 16-byte slot within the 256-byte bank row
        assert len(slots) == 16, (bk, lq, kb, sorted(slots))


def test_a_loader_instruction_reads_whole_contiguous_row_segments():
    """What makes the loader's global reads cheap (45-60 B/clk/CU against 15 for fragment-shaped loads): the 64 lanes of
    one LDS-DMA instruction cover whole rows of the k step — 4 rows x 256 B, or 8 rows x 128 B — permuted WITHIN a row."""
    for bk in (128, 64):
        image = loader_image(bk, 16, 1)
        for piece in range(len(image) * 16 // 1024):
            lanes = [image[piece * 1024 + lane * 16] for lane in range(64)]
            rows = sorted({r for r, _ in lanes})
            assert len(rows) == 1024 // (bk * 2) and rows == list(range(rows[0], rows[0] + len(rows)))
            for r in rows:
                assert sorted(c for rr, c in lanes if rr == r) == list(range(bk // 8))
