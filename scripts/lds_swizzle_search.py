"""Search for the LDS slot swizzle of the shared-activation-row convolution tile (igemm.hip, SWZ).

An activation slab row is 128 bytes = eight 16-byte chunks; chunk q of row R is stored in slot q ^ h(R).  A wave reads
fragment rows R = lr + kw (lr = lane & 15, kw = 0, 1, 2: the filter tap) with ds_read_b128, which the LDS serves in
four groups of 16 lanes; a group is conflict-free when its 16 lanes touch 16 different 16-byte positions of the
256-byte bank window, i.e. different (R & 1, slot) pairs.  Backtracking over h: Z16 -> Z8 finds the first table that is
conflict-free for all three shifts; it depends on R >> 1 only and is what igemm.hip packs into SWZ."""

GROUPS = [[(0, r) for r in (0, 1, 2, 3, 12, 13, 14, 15)] + [(1, r) for r in range(4, 12)],
          [(0, r) for r in range(4, 12)] + [(1, r) for r in (0, 1, 2, 3, 12, 13, 14, 15)],
          [(2, r) for r in (0, 1, 2, 3, 12, 13, 14, 15)] + [(3, r) for r in range(4, 12)],
          [(2, r) for r in range(4, 12)] + [(3, r) for r in (0, 1, 2, 3, 12, 13, 14, 15)]]   # (lane >> 4, lane & 15)


def conflict_free(h, shifts=(0, 1, 2)):
    for group in GROUPS:
        for kw in shifts:
            seen = set()
            for lq, lr in group:
                row = (lr + kw) & 15
                if h[row] is None:
                    continue
                key = (row & 1, lq ^ h[row])
                if key in seen:
                    return False
                seen.add(key)
    return True


def search():
    h = [None] * 16

    def rec(i):
        if i == 16:
            return True
        for v in range(8):
            h[i] = v
            if conflict_free(h) and rec(i + 1):
                return True
        h[i] = None
        return False

    return h if rec(0) else None


if __name__ == "__main__":
    h = search()
    print("h(R)       =", h)
    print("g(R >> 1)  =", h[0::2], "packed:", hex(sum(v << (4 * i) for i, v in enumerate(h[0::2]))))
    print("plain (R >> 1) & 7 swizzle conflict-free per shift:", [conflict_free([(r >> 1) & 7 for r in range(16)], (k,)) for k in (0, 1, 2)])
