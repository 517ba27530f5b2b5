#!/usr/bin/env python3
"""ds_read_b128 bank-conflict model of the conv kernels' halo image (MI355X_MICROARCH.md, LDS table: 64 banks x 4 B, four
16-lane groups per wave-instruction).  Prints LDS cycles per fragment read (4 = conflict-free) for candidate pixel strides.
Round 2 used it to replace the 48 B / 80 B strides (2-way conflicts on every read) by 32 B / 96 B."""
groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
groups += [[l + 32 for l in g] for g in groups]


def cycles(addr):
    tot = 0
    for g in groups:
        banks = {}
        for l in g:
            for b in range(4):
                banks.setdefault(((addr[l] // 4) + b) % 64, set()).add(addr[l])
        tot += max(len(v) for v in banks.values())
    return tot


def fragment_read(P, UPB):
    """quads read units (tap, c) = unit 4s+q of a K-step: UPB 2 -> (t,c0),(t,c1),(t',c0),(t',c1); UPB 4 -> (t,c0..c3)"""
    worst = 0
    for base in range(64):
        for dx in ((0, 0), (0, 1), (1, 0)):
            addr = [0] * 64
            for lane in range(64):
                q, l = lane >> 4, lane & 15
                tap, c = (q >> 1, q & 1) if UPB == 2 else (0, q)
                addr[lane] = (base + l + (dx[tap] if UPB == 2 else 0)) * P + c * 16
            worst = max(worst, cycles(addr))
    return worst


if __name__ == "__main__":
    for UPB, Ps in ((2, (32, 48, 64)), (4, (64, 80, 96, 112, 144))):
        for P in Ps:
            print("units/block %d, pixel stride %3d B: %d LDS cycles per ds_read_b128 (4 = conflict-free)" % (UPB, P, fragment_read(P, UPB)))
