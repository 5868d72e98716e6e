"""CPU checks of what csrc/linear_dw.hip and csrc/linear_train.hip rest on (oracle/bf16_planes.py): the three-plane bf16 split is exact
over fp32's range, and the re-dealt reduction order makes the transposing LDS reads conflict-free where the natural order cannot be."""
import numpy as np

from oracle import bf16_planes as B


def test_three_truncation_planes_sum_to_the_value_exactly():
    rng = np.random.default_rng(5)
    x = (rng.standard_normal(200000) * np.power(10.0, rng.uniform(-30, 30, 200000))).astype(np.float32)
    x = np.concatenate([x, np.float32([0.0, -0.0, 1.0, -1.0, 3.4e38, -3.4e38, 1.17549435e-38, 1.0 + 2.0 ** -23, 1.0 - 2.0 ** -24])])
    hi, mid, lo = B.split3(x)
    for pl in (hi, mid, lo):
        assert not np.any(pl.view(np.uint32) & np.uint32(0xFFFF))                  # each plane is a bf16 value
        assert np.all((pl == 0) | (np.sign(pl) == np.sign(x)))                      # truncation: every plane carries the value's sign
    assert np.array_equal((hi.astype(np.float64) + mid.astype(np.float64) + lo.astype(np.float64)).astype(np.float32), x)
    assert np.array_equal(hi.astype(np.float64) + mid.astype(np.float64) + lo.astype(np.float64), x.astype(np.float64))
    assert np.all(np.abs(mid) <= np.abs(hi) * 2.0 ** -7) and np.all(np.abs(lo) <= np.abs(hi) * 2.0 ** -15)


def _conflicts(banks):
    seen = {}
    worst = 1
    for b in (x for pair in banks for x in pair):
        seen[b] = seen.get(b, 0) + 1
        worst = max(worst, seen[b])
    return worst


def test_redealt_reduction_order_is_conflict_free_and_the_natural_one_is_not():
    dealt = lambda g: [4 * g + q for q in range(4)]                                # this half: rows 0..7 (the other half: 8..15; 16.. for the second read)
    natural = lambda g: [8 * g + q for q in range(4)]                              # group g = indices 8g..8g+7: rows 0..3 and 8..11
    for block in range(0, 192, 32):                                                # the six 16-column blocks of a 96-column tile
        assert _conflicts(B.tr_read_banks(224, dealt, block)) == 1
    assert _conflicts(B.tr_read_banks(224, natural)) > 1
    # ... at every pitch whose 32-byte runs are bank-aligned the natural order puts rows r and r + 8 on the same banks
    assert all(_conflicts(B.tr_read_banks(p, natural)) > 1 for p in range(32, 1024, 32))
    # and the dealt order is free exactly at pitches of 32 bytes x odd
    free = [p for p in range(32, 1024, 32) if _conflicts(B.tr_read_banks(p, dealt)) == 1]
    assert free == [p for p in range(32, 1024, 32) if (p // 32) % 2 == 1]
