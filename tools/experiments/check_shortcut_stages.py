"""CPU check of the index arithmetic of tools/experiments/conv_t32_shortcut_stages.patch (written while GPU access was closed):
where the LDS-DMA pieces of a 1x1 chunk's centre patch land, and what every lane's fragment read of the staged loop finds there
-- pixel (row_base + 2 j + lr, lcx), channel group 2 ks + kh -- for the four tile shapes (TH 8 / 16, 4 / 8 waves), plus the bank
check (every 16-lane group of a ds_read_b128 touches 16 distinct 16-byte bank quads).    python tools/experiments/check_shortcut_stages.py"""


def check(TH, NW):
    NT, BM, TM = NW * 64, TH * 16, TH // 8
    R1 = BM * 4 // NT
    lds = {}
    for r in range(R1):                      # issue1: piece = r * NT + tid -> stage offset r * NT * 16 + wave * 1024 + lane * 16
        for tid in range(NT):
            piece = r * NT + tid
            pp, pj = piece >> 2, piece & 3
            off = r * NT * 16 + (tid >> 6) * 1024 + (tid & 63) * 16
            assert off not in lds
            lds[off] = (pp >> 4, pp & 15, pj ^ (((pp & 15) >> 2) & 3))           # (tile row, tile column, source channel group)
    assert len(lds) == BM * 4 and max(lds) == BM * 64 - 16
    for w in range(NW):
        row_base = (w & 3) * (TH // 4)
        for l in range(64):
            q, kh = l & 31, l >> 5
            lr, lcx = q >> 4, q & 15
            pf, pk = ((row_base + lr) * 16 + lcx) * 64, (lcx >> 2) & 3
            for j in range(TM):
                for ks in range(2):
                    a = pf + j * 2048 + (((2 * ks + kh) ^ pk) << 4)
                    assert lds[a] == (row_base + 2 * j + lr, lcx, 2 * ks + kh), (TH, NW, w, l, j, ks)
        for ks in range(2):
            for j in range(TM):
                for g in range(4):
                    quads = set()
                    for l in range(16 * g, 16 * g + 16):
                        q, kh = l & 31, l >> 5
                        lr, lcx = q >> 4, q & 15
                        a = ((row_base + lr) * 16 + lcx) * 64 + j * 2048 + (((2 * ks + kh) ^ ((lcx >> 2) & 3)) << 4)
                        quads.add((a >> 4) & 15)
                    assert len(quads) == 16
    print(f"TH={TH} waves={NW}: {R1} DMA rounds per chunk, placement / reads / banks consistent")


if __name__ == "__main__":
    for th in (8, 16):
        for nw in (4, 8):
            check(th, nw)
