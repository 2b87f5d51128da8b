#!/usr/bin/env python3
"""LDS bank-conflict model for gfx950 fragment reads/writes.

Lane groups and bank widths follow the per-instruction servicing rules of the
CDNA4 LDS (64 dword banks; ds_read_b64/b128 bank = (addr/4) % 64,
ds_read_b32 and all ds_write bank = (addr/4) % 32).  For a candidate tile
pitch the script prints the worst-case serialisation factor per lane group, so
pads in eigen_amd/csrc/gemm_kernels.hip can be checked without a GPU.
"""
import sys

G_B128 = [
    list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
    list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
    list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)),
    list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64)),
]
G_HALF = [list(range(0, 32)), list(range(32, 64))]
G_W64 = [list(range(16 * g, 16 * g + 16)) for g in range(4)]
G_W128 = [list(range(8 * g, 8 * g + 8)) for g in range(8)]


def cycles(addr_of_lane, width, groups, nbanks):
    """max over groups of the worst bank multiplicity (distinct addresses)."""
    worst = 0
    for g in groups:
        banks = {}
        for l in g:
            a = addr_of_lane(l)
            for d in range(width // 4):
                b = ((a // 4) + d) % nbanks
                banks.setdefault(b, set()).add((a // 4 + d))
        worst = max(worst, max(len(v) for v in banks.values()))
    return worst


def read_cfg(es):
    if es == 4:
        return 4, G_HALF, 32
    if es == 8:
        return 8, G_HALF, 64
    return 16, G_B128, 64


def write_cfg(nbytes):
    if nbytes == 4:
        return 4, G_HALF, 32
    if nbytes == 8:
        return 8, G_W64, 32
    return 16, G_W128, 32


def frag_read(es, tm, pitch, kcontig):
    w, groups, nb = read_cfg(es)
    def addr(l):
        xi, kk = l % tm, l // tm
        idx = xi * pitch + kk if kcontig else kk * pitch + xi
        return idx * es
    return cycles(addr, w, groups, nb)


def main():
    cases = [("f32", 4, 32, 128, 32), ("f64", 8, 16, 128, 16),
             ("c32", 8, 32, 128, 16), ("c64", 16, 16, 128, 8)]
    for name, es, tm, bx, bk in cases:
        print(f"== {name}: elem {es} B, MFMA tile {tm}, BX {bx}, BK {bk}")
        okx = [p for p in range(0, 33) if frag_read(es, tm, bx + p, False) == 1
               and ((bx + p) * es) % 16 == 0]
        okk = [p for p in range(0, 17) if frag_read(es, tm, bk + p, True) == 1]
        okk16 = [p for p in okk if ((bk + p) * es) % 16 == 0]
        print("  X-contig conflict-free pads (16B rows):", okx[:8])
        print("  K-contig conflict-free pads:", okk[:8], " with 16B-aligned rows:", okk16[:8])


if __name__ == "__main__":
    main()
