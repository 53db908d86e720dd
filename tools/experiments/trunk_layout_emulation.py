"""Emulation of the class-tiled cell order of k_tower8_c128 (4 positions per workgroup) with explicit swizzle keys
(cchess_zero_amd/csrc/cz_conv_kernel.h): the row map is a bijection, its inverse is the kernel's decode, which (tile, tap) pairs
are off the board, and how many lanes of each 16-lane ds_read_b128 group share a slot (bank conflicts) per tile and tap.  A "group"
here is 16 consecutive GEMM rows of a tile: the kernel relabels its lanes (m31) so that the hardware's lane groups {0-3, 12-15,
20-27} / {4-11, 16-19, 28-31} own exactly those.  usage: python tools/experiments/trunk_layout_emulation.py
(tests/test_abi_cpu.py imports it and holds the kernel's constants to it)."""


def row_of(p, y, x):
    if y == 0:
        return 8 + 8 * p + x if x < 8 else 2 * p + (x - 8)
    if x == 0:
        return 40 + 8 * p + (y - 1)
    if x == 9:
        return 168 + 8 * p + (y - 1)
    if y == 8:
        return 72 + 8 * p + (x - 1)
    j = 56 * p + 8 * (y - 1) + (x - 1)
    return 104 + j if j < 64 else 136 + j


def key_of(p, y, x):
    return 8 * ((y + p) & 1) + ((x + y) & 7)


def cell_of_row(k):
    """The inverse as the kernel computes it (per lane, once)."""
    if k < 8:
        return (k // 2, 0, 8 + k % 2)
    if k < 40:
        return ((k - 8) // 8, 0, (k - 8) % 8)
    if k < 72:
        q = k - 40
        return (q // 8, q % 8 + 1, 0)
    if k < 104:
        q = k - 72
        return (q // 8, 8, q % 8 + 1)
    if k < 168:
        j = k - 104
    elif k < 200:
        q = k - 168
        return (q // 8, q % 8 + 1, 9)
    else:
        j = k - 136
    p, r = divmod(j, 56)
    return (p, r // 8 + 1, r % 8 + 1)


def m31(l):
    """GEMM row (within its tile) of lane l31."""
    return l if l < 4 else l + 12 if l < 12 else l - 8 if l < 16 else l + 8 if l < 20 else l - 12 if l < 28 else l


HW_GROUPS = ([0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31])


def analyse():
    cells = {}
    for p in range(4):
        for y in range(9):
            for x in range(10):
                k = row_of(p, y, x)
                assert k not in cells, (p, y, x, k)
                cells[k] = (p, y, x)
    assert sorted(cells) == list(range(360))
    for k in range(360):
        assert cell_of_row(k) == cells[k], k
    assert sorted(m31(l) for l in range(32)) == list(range(32))
    conflicts, skippable = {}, []
    for tile in range(12):
        for tap in range(9):
            dy, dx = tap // 3 - 1, tap % 3 - 1
            allinv = True
            for grp, lanes in enumerate(HW_GROUPS):
                keys = []
                for l in lanes:
                    k = 32 * tile + m31(l) - 24
                    if k < 0:
                        continue    # a padding lane: its key is its row number, chosen freely
                    p, y, x = cells[k]
                    yy, xx = y + dy, x + dx
                    if 0 <= yy < 9 and 0 <= xx < 10:
                        allinv = False
                    keys.append(key_of(p, yy, xx))
                c = len(keys) - len(set(keys))
                if c:
                    conflicts[(tile, tap, grp)] = c
            if allinv:
                skippable.append((tile, tap))
    return conflicts, skippable


def skiptab_of(skippable):
    """The kernel's per-cell-group table: two bits per tap, bit 0 = the group's first tile (wr), bit 1 = its second (wr + 4)."""
    tabs = [0, 0, 0, 0]
    for tile, tap in skippable:
        wr, slot = tile % 4, tile // 4
        assert slot < 2, (tile, tap)
        tabs[wr] |= (1 << slot) << (2 * tap)
    return tabs


if __name__ == "__main__":
    conflicts, skippable = analyse()
    by_tile = {}
    for (tile, tap, grp), c in conflicts.items():
        by_tile[tile] = by_tile.get(tile, 0) + c
    print("lanes sharing a slot with another lane of their ds_read_b128 group, by tile:", by_tile, "(total %d)" % sum(conflicts.values()))
    print("off-board (tile, tap) pairs:", skippable, "=", len(skippable), "of 108")
    print("skip tables of the four cell groups:", [hex(t) for t in skiptab_of(skippable)])
