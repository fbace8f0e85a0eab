#!/usr/bin/env python3
"""LDS images of the leaf blocks of k_tile2g (dct_pruning_amd/csrc/tile2g.hip, G2Layout<M, PB>): check the committed
layouts and search new ones.

A pass holds PB blocks of M x M floats; element (row r, column c) of block g lives at g*BS + r*RS + c. While the
columns are transformed (axis A) lane g*LWA + j reads / writes (i, j) of block g for i = 0..M-1, one i per instruction;
while the rows are transformed (axis B) lane g*LWB + j accesses (j, i). Lanes outside repeat the address of the nearest
active lane of their 32-lane group (G2LaneMap), i.e. a broadcast. The LDS serves 32 lanes per cycle from 32 banks of
4 bytes: an instruction is conflict-free when, inside each 32-lane group, different addresses fall into different banks.

usage: tools/g2_layout_search.py            # verify the layouts in tile2g.hip
       tools/g2_layout_search.py M PB       # search (RS, BS, LWA, LWB) for a new shape, smallest BS first
"""
import re
import sys

BANKS = 32


def lane_map(M, PB, LW):
    act = [(l // LW < PB and l % LW < M) for l in range(64)]
    src = list(range(64))
    for l in range(64):
        if act[l]:
            continue
        lo = (l // 32) * 32
        best = None
        for d in range(1, 32):
            if l - d >= lo and act[l - d]:
                best = l - d
                break
            if l + d < lo + 32 and act[l + d]:
                best = l + d
                break
        if best is None:
            best = max(t for t in range(64) if act[t])
        src[l] = best
    return [(s // LW, s % LW) for s in src]


def conflict_free(M, PB, RS, BS, LW, axis):
    lanes = lane_map(M, PB, LW)
    for i in range(M):
        for half in (0, 32):
            banks = {}
            for l in range(half, half + 32):
                g, j = lanes[l]
                addr = g * BS + (i * RS + j if axis == "A" else j * RS + i)
                b = addr % BANKS
                if banks.setdefault(b, addr) != addr:
                    return False
    return True


def check(M, PB, RS, BS, LWA, LWB):
    fits = RS >= M and BS >= (M - 1) * RS + M and PB * max(LWA, LWB) <= 64 + (max(LWA, LWB) - M)
    return fits and conflict_free(M, PB, RS, BS, LWA, "A") and conflict_free(M, PB, RS, BS, LWB, "B")


def committed():
    src = open(__file__.replace("tools/g2_layout_search.py", "dct_pruning_amd/csrc/tile2g.hip")).read()
    pat = r"struct G2Layout<(\d+), (\d+)> \{\s*static constexpr int RS = (\d+), BS = (\d+), LWA = (\d+), LWB = (\d+);"
    return [tuple(int(v) for v in m) for m in re.findall(pat, src)]


def search(M, PB):
    out = []
    for RS in range(M, M + 12):
        lo = (M - 1) * RS + M
        for BS in range(lo, lo + 96):
            for LWA in range(M, 64 // PB + 1 if PB > 1 else 65):
                if (PB - 1) * LWA + M > 64 or not conflict_free(M, PB, RS, BS, LWA, "A"):
                    continue
                for LWB in range(M, 64 // PB + 1 if PB > 1 else 65):
                    if (PB - 1) * LWB + M > 64:
                        continue
                    if conflict_free(M, PB, RS, BS, LWB, "B"):
                        out.append((BS, RS, LWA, LWB))
                        break
                break
    return sorted(out)


if __name__ == "__main__":
    if len(sys.argv) == 3:
        M, PB = int(sys.argv[1]), int(sys.argv[2])
        for BS, RS, LWA, LWB in search(M, PB)[:10]:
            print("G2Layout<%d, %d>: RS = %d, BS = %d, LWA = %d, LWB = %d" % (M, PB, RS, BS, LWA, LWB))
    else:
        bad = 0
        for M, PB, RS, BS, LWA, LWB in committed():
            ok = check(M, PB, RS, BS, LWA, LWB)
            bad += not ok
            print("G2Layout<%d, %d> RS %d BS %d LWA %d LWB %d: %s" % (M, PB, RS, BS, LWA, LWB, "conflict-free" if ok else "CONFLICTS"))
        sys.exit(1 if bad else 0)
