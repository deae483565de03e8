"""Combine the per-(kernel, grid) means of two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; tools/pmc_summary.py) into the
HBM bytes per launch of the kernel classes bench.py reports rooflines for.  Corrections as /opt/skills/guides/MI355X_MICROARCH.md
prescribes for gfx950: counter unit KiB; FETCH_SIZE x 2 for 16 B/lane coalesced streams (128-byte requests tallied at 64 bytes).
Usage: python tools/pmc_traffic.py <fetch.json> <write.json> <out.json> "<command the passes profiled>" """
import json
import sys

CLASSES = {
    "decode_attn": ["decode_attn"],
    "decode_gemm": ["dgemm_kernel", "dg3_kernel"],
    "conv3x3": ["conv3x3_kernelIDF16b"],
    "igemm": ["igemm_kernelIDF16b", "gemm256l_kernel"],
}


def per_class(rows, counter, scale):
    out = {}
    for cls, pats in CLASSES.items():
        n = tot = 0
        for r in rows:
            if r["counter"] == counter and any(p in r["kernel"] for p in pats):
                n += r["launches"]
                tot += r["total"] * 1024.0 * scale
        if n:
            out[cls] = (n, tot / n)
    return out


def main():
    fetch, write, out, cmd = sys.argv[1:5]
    rd = per_class(json.load(open(fetch)), "FETCH_SIZE", 2.0)
    wr = per_class(json.load(open(write)), "WRITE_SIZE", 1.0)
    res = {"source": f"rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes, of `{cmd}`; per-(kernel, grid) means in the two input files",
           "corrections": "counter unit KiB (x1024). FETCH_SIZE x2: every read of these kernels is a 16 B/lane coalesced stream (global_load_dwordx4 / "
                          "global_load_lds), which gfx950 tallies at 64 B per 128 B request (MI355X_MICROARCH.md, HBM section). WRITE_SIZE x1.",
           "per_launch_bytes": {}, "detail": {}}
    for cls in CLASSES:
        if cls in rd or cls in wr:
            r, w = rd.get(cls, (0, 0.0)), wr.get(cls, (0, 0.0))
            res["per_launch_bytes"][cls] = r[1] + w[1]
            res["detail"][cls] = {"kernel_name_filter": CLASSES[cls], "launches": max(r[0], w[0]), "read_bytes_per_launch": r[1],
                                  "write_bytes_per_launch": w[1]}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res["per_launch_bytes"], indent=1))


if __name__ == "__main__":
    main()
