"""Bank / correctness model of the V^T tile layout of the one-pass attention kernels (csrc/llama_ops.hip: flash_prefill_kernel,
xattn_kernel), against the LDS rules of MI355X_MICROARCH.md (LDS).

The P.V MFMA of those kernels enumerates the keys of a 32-key step as (4*lg + r, 16 + 4*lg + r), so a lane needs two groups of four
consecutive keys, 16 apart.  Read as two 8-byte loads the compiler merges them into ds_read2_b64 (two accesses of 4 x 16 CONTIGUOUS
lanes, banks mod 32, half the rate of ds_read_b128) and 16 consecutive rows of a 144-byte pitch are then 2-way conflicted: the 32-38 %
bank-conflict cycles of rounds 2-4.  Layout checked here: inside each 32-key block position lg*8 + half*4 + r holds key
half*16 + lg*4 + r, so a fragment is ONE ds_read_b128; rows of 64 keys are padded by 32 bytes, rows of 32 keys (64 bytes) use the XOR
key of conv3x3's 64-byte rows instead (no pad: the 512-channel single-head instance keeps two workgroups per CU)."""
G128 = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]
G128 += [[l + 32 for l in g] for g in G128]


def read_b128_cycles(addr):      # 4 lane groups, 64 banks x 4 B
    tot = 0
    for g in G128:
        banks = {}
        for l in g:
            for d in range(4):
                banks.setdefault(((addr[l] >> 2) + d) & 63, set()).add(addr[l])
        tot += max(len(v) for v in banks.values())
    return tot


def write_b64_cycles(addr):      # 4 groups of 16 contiguous lanes, banks mod 32
    tot = 0
    for g0 in range(0, 64, 16):
        banks = {}
        for l in range(g0, g0 + 16):
            for d in range(2):
                banks.setdefault(((addr[l] >> 2) + d) & 31, set()).add(addr[l])
        tot += max(len(v) for v in banks.values())
    return tot


def pitch(KT):
    return KT if KT == 32 else KT + 16           # elements (bf16)


def quad_off(KT, row, quad):                     # element offset of 16-byte quad `quad` of row `row`
    q = quad ^ ((row >> 1) & 3) if KT == 32 else quad
    return row * pitch(KT) + (q << 3)


def write_offsets(KT, row, cc):
    """element offsets the two 8-byte halves (keys 8cc .. 8cc+3, 8cc+4 .. 8cc+7) of source chunk cc of a row go to"""
    qa = (cc >> 2) * 4 + 2 * (cc & 1)
    e = ((cc >> 1) & 1) * 4
    return quad_off(KT, row, qa) + e, quad_off(KT, row, qa + 1) + e


def check(KT, HD):
    rows = HD
    # ---- function: write a tile whose element (row, key) = row * 1000 + key, read the fragments back
    lds = {}
    for row in range(rows):
        for cc in range(KT // 8):
            a, b = write_offsets(KT, row, cc)
            for j in range(4):
                lds[a + j] = row * 1000 + 8 * cc + j
                lds[b + j] = row * 1000 + 8 * cc + 4 + j
    assert len(lds) == rows * KT
    for d in range(HD // 16):
        for pr in range(KT // 32):
            for lane in range(64):
                lr, lg = lane & 15, lane >> 4
                o = quad_off(KT, d * 16 + lr, pr * 4 + lg)
                got = [lds[o + j] for j in range(8)]
                want = [(d * 16 + lr) * 1000 + pr * 32 + (j >> 2) * 16 + lg * 4 + (j & 3) for j in range(8)]
                assert got == want, (KT, d, pr, lane, got, want)
    # ---- banks: fragment reads (every d, pr) and staging writes (every group of 64 consecutive chunks)
    rd = max(read_b128_cycles([2 * quad_off(KT, d * 16 + (l & 15), pr * 4 + (l >> 4)) for l in range(64)])
             for d in range(HD // 16) for pr in range(KT // 32))
    vch = KT // 8
    wr = 0
    for c0 in range(0, rows * vch, 64):
        for half in (0, 1):
            addr = []
            for l in range(64):
                row, cc = (c0 + l) // vch, (c0 + l) % vch
                a, b = write_offsets(KT, row, cc)
                first = (a, b) if not (row & 1) or KT == 32 else (b, a)   # padded rows: odd rows store their halves in the other order
                addr.append(2 * first[half])
            wr = max(wr, write_b64_cycles(addr))
    return rd, wr


if __name__ == "__main__":
    for KT, HD in ((64, 64), (64, 128), (64, 192), (64, 32), (32, 512), (32, 768)):
        rd, wr = check(KT, HD)
        print(f"KT={KT} HD={HD}: pitch {2 * pitch(KT)} B, fragment read {rd} cycles (4 = conflict free), staging ds_write_b64 {wr} cycles (4 = conflict free)")
