"""Row maps of conv3_halo_kernel's part tiles (conv_halo_bf16.hip, HALF = 1 / 2): every valid voxel of the tile is covered exactly
once by the M tiles a workgroup runs, and the ds_read_b128 of an A fragment is conflict-free (or, for the w strip, 2-way) under the
instruction's 16-lane service groups (MI355X_MICROARCH.md)."""
GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
SP, HWp, HHp = 40, 12, 10


def rowmap(half, edge_h, wm, i, l):
    mt = wm * 4 + i
    par = ((l >> 2) ^ (l >> 3) ^ (l >> 4)) & 1
    aw, k = ((l >> 3) & 1) * 4 + (l & 3), (l >> 3) * 4 + (l & 3)
    dd, hh, ww = mt >> 1, l >> 2, (mt & 1) * 4 + (l & 3)
    if half == 1 and edge_h:
        hh, ww = (mt & 1) * 4 + par + 2 * (l >> 4), aw
    if half == 2:
        dd, ww = 2 * wm + i, l & 3
        if edge_h:
            hh, ww = par + 2 * (l >> 4), aw
        if i == 2:
            if edge_h:
                dd, hh, ww = 2 * wm + (l >> 4), 4 + par, aw
            else:
                dd, hh, ww = 2 * wm + par, k >> 1, 4 + (k & 1)
    return dd, hh, ww


def main():
    for half, tiles, vh, vw in ((0, (0, 1, 2, 3), 8, 8), (1, (0, 2), 8, 4), (1, (0, 2), 4, 8), (2, (0, 1, 2), 8, 6), (2, (0, 1, 2), 6, 8)):
        edge_h = vh < 8
        seen = {}
        worst = 0
        for wm in range(2):
            for i in tiles:
                vox = [rowmap(half, edge_h, wm, i, l) for l in range(32)]
                for v in vox:
                    seen[v] = seen.get(v, 0) + 1
                for g in GROUPS:
                    for hi in range(2):
                        for tap_off in range(3):
                            slots = [((((vox[l][0] * HHp + vox[l][1]) * HWp + vox[l][2] + tap_off) * SP + 8 * hi) * 2 // 16) % 16 for l in g]
                            worst = max(worst, max(slots.count(x) for x in slots))
        want = {(d, h, w) for d in range(4) for h in range(vh) for w in range(vw)}
        assert set(seen) == want and all(c == 1 for c in seen.values()), (half, edge_h)
        print('HALF=%d %s edge: %d voxels covered once by %d M tiles, worst ds_read_b128 multiplicity %d'
              % (half, 'h' if edge_h else 'w', len(seen), 2 * len(tiles), worst))


if __name__ == '__main__':
    main()
