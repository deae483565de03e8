"""Bank-conflict check of the LDS swizzle keys of csrc/conv3x3.hip against the ds_read_b128 model of
MI355X_MICROARCH.md (LDS): a wave64 access is served in 4 lane groups, one LDS cycle each when no two DISTINCT addresses of a
group share a bank (64 banks x 4 B; identical addresses broadcast).  4 cycles per instruction = conflict free.

Rows are 64 bytes (4 chunks of 16 B); lane l of a fragment read wants chunk (l >> 4) of row rows[l & 15], stored in slot
(chunk ^ key).  Access shapes:
  plain      16 consecutive rows from any start (weight tiles; halo rows of the stride-1 convs)
  ups, o=0   rows b + (i >> 1)       (nearest-x2 upsampling, tap kw = 1)
  ups, o=1   rows b + ((i + 1) >> 1) (taps kw = 0, 2: nine rows)
"""
G = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]
G += [[l + 32 for l in g] for g in G]


def cycles(addr):
    tot = 0
    for g in G:
        banks = {}
        for l in g:
            for d in range(4):
                banks.setdefault(((addr[l] >> 2) + d) & 63, set()).add(addr[l])
        tot += max(len(v) for v in banks.values())
    return tot


def key_plain(x):
    return (x >> 1) & 3


def key_ups(x):
    j = x >> 2
    return ((j & 1) << 1) | ((j >> 1) & 1)


def addresses(rows, key):
    return [rows[l & 15] * 64 + (((l >> 4) ^ key(rows[l & 15])) << 4) for l in range(64)]


def worst(key):
    plain = max(cycles(addresses([s + i for i in range(16)], key)) for s in range(64))
    ups = [max(cycles(addresses([b + ((o + i) >> 1) for i in range(16)], key)) for b in range(64)) for o in (0, 1)]
    return plain, ups[0], ups[1]


if __name__ == "__main__":
    for name, key in (("(x >> 1) & 3", key_plain), ("bitrev2(x >> 2)", key_ups)):
        p, u0, u1 = worst(key)
        print(f"key {name:16s}: plain {p} cycles, ups aligned {u0}, ups odd {u1}   (4 = conflict free)")
