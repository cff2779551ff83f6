#!/usr/bin/env python3
"""Bank-conflict count of the SH tile of preprocess_fwd / preprocess_bwd (csrc/mrgs_preprocess.hip), per layout.

A wave moves its 64 SH rows between global memory and a per-wave LDS tile in 16-byte pieces (lane l of round k holds elements
4 (64 k + l) + j of the run, one ds_write_b32 / ds_read_b32 per j) and then every lane works on its own row.  ds_*_b32 instructions are
served in two groups of 32 lanes over 32 banks (MI355X_MICROARCH.md, LDS); an instruction group costs one extra cycle per extra distinct
address on its busiest bank.  Printed: extra cycles summed over the transposing instructions of one direction, and over the lanes' own
accesses (one row per lane), for
  * the row-major tile with row stride 49 (rounds 1-4; what the unsplit [P,16,3] tensor still uses),
  * the same tile holding the SPLIT tensors' rows (DC 3 floats + rest 45 floats): the layout round 4 profiled at 3.4 M conflict cycles,
  * the verbatim copy of the split tensors' runs (round 5: 16-byte ds accesses, nothing transposed; the lanes' rows have stride 3 / 45).
"""


def transposing(addr, L, split):
    tot = n = 0
    n4 = 64 * L // 4
    for k in range(12):
        for j in range(4):
            for grp in range(2):
                banks = {}
                for lane in range(grp * 32, grp * 32 + 32):
                    q = k * 64 + lane
                    if q >= n4:
                        continue
                    e = 4 * q + j
                    r = e // L
                    a = addr(r, e - L * r + (3 if split else 0))
                    banks.setdefault(a % 32, set()).add(a)
                if banks:
                    tot += max(len(v) for v in banks.values()) - 1
                    n += 1
    if split:                                   # the DC rows: element t of the 192-float run, one per lane
        for t0 in range(0, 192, 64):
            for grp in range(2):
                banks = {}
                for lane in range(grp * 32, grp * 32 + 32):
                    t = t0 + lane
                    r = t // 3
                    a = addr(r, t - 3 * r)
                    banks.setdefault(a % 32, set()).add(a)
                tot += max(len(v) for v in banks.values()) - 1
                n += 1
    return tot, n


def own_rows(addr, ncoef=48):
    tot = 0
    for c in range(ncoef):
        for grp in range(2):
            banks = {}
            for r in range(grp * 32, grp * 32 + 32):
                a = addr(r, c)
                banks.setdefault(a % 32, set()).add(a)
            tot += max(len(v) for v in banks.values()) - 1
    return tot


if __name__ == "__main__":
    row49 = lambda r, c: r * 49 + c
    print("row-major stride 49, unsplit rows of 48: transposing extra cycles %d over %d instruction groups; own rows %d" %
          (*transposing(row49, 48, False), own_rows(row49)))
    print("row-major stride 49, split rows (3 + 45): transposing extra cycles %d over %d instruction groups; own rows %d" %
          (*transposing(row49, 45, True), own_rows(row49)))
    lin = lambda r, c: r * 3 + c if c < 3 else 192 + r * 45 + (c - 3)
    print("verbatim copy of the split runs (stride 3 / 45): nothing transposed (linear 16-byte ds accesses); own rows %d" % own_rows(lin))
