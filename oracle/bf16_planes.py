"""TEST INFRASTRUCTURE (oracle/): numpy restatement of two facts the round-5 training kernels rest on, checked on the CPU.

1. `split3` (csrc/linear_dw.hip:55-69, csrc/linear_train.hip:47-61): an fp32 value as three bf16 planes by truncation,
   hi + mid + lo == value exactly.
2. The dealing of a 32-long reduction chunk to the four 16-lane groups of `ds_read_b64_tr_b16` (csrc/linear_dw.hip:10-16, tr_frag):
   group g reads rows 4g..4g+3 and 16+4g..16+4g+3 of a [row][column] image of 16-bit elements with a 224-byte row pitch.  The LDS bank
   of byte address a is (a / 4) % 64 for this instruction and conflicts count per 32-lane half (MI355X guide, LDS section).

Only tests import this module."""
import numpy as np


def split3(x):
    """x: float32 array -> (hi, mid, lo) float32 arrays whose low 16 bits are zero (= bf16 values), truncation splits."""
    x = np.asarray(x, dtype=np.float32)
    mask = np.uint32(0xFFFF0000)
    hi = (x.view(np.uint32) & mask).view(np.float32)
    r1 = (x - hi).astype(np.float32)
    mid = (r1.view(np.uint32) & mask).view(np.float32)
    lo = (r1 - mid).astype(np.float32)
    return hi, mid, lo


def tr_read_banks(pitch_bytes, rows_of_group, col_block_bytes=0):
    """Banks touched by each lane of one 32-lane half (two 16-lane groups) of a ds_read_b64_tr_b16: lane (q, p) of group g supplies the
    address of row rows_of_group(g)[q], columns 4p..4p+3 (8 bytes).  Returns a list of 32 (bank, bank + 1) pairs."""
    out = []
    for g in range(2):
        rows = rows_of_group(g)
        for li in range(16):
            q, p = li >> 2, li & 3
            a = rows[q] * pitch_bytes + col_block_bytes + 8 * p
            out.append(((a // 4) % 64, (a // 4 + 1) % 64))
    return out
