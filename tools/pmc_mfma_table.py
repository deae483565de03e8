"""MFMA-busy / LDS-active / bank-conflict table per (kernel, grid) from the summary of a
``rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT`` pass (tools/pmc_summary.py).
GRBM_GUI_ACTIVE is summed over the 8 XCDs, MFMA busy over the 1,024 SIMDs, the LDS counters over the 256 CUs:
  mfma_busy% = MFMA_BUSY / (GUI / 8 * 1024),  lds_active% = LDS_IDX_ACTIVE / (GUI / 8 * 256),  conflict% = BANK_CONFLICT / LDS_IDX_ACTIVE
Usage: python tools/pmc_mfma_table.py <pmc_mfma.json>"""
import json
import sys
from collections import defaultdict


def main():
    rows = json.load(open(sys.argv[1]))
    d = defaultdict(dict)
    for r in rows:
        d[(r["kernel"], r["blocks"])][r["counter"]] = (r["mean"], r["launches"])
    print(f"{'kernel':60s} {'blocks':>7s} {'launches':>8s} {'Mcycles':>8s} {'mfma_busy%':>10s} {'lds_active%':>11s} {'conflict%':>9s}")
    out = []
    for (k, b), c in d.items():
        if "GRBM_GUI_ACTIVE" not in c:
            continue
        gui = c["GRBM_GUI_ACTIVE"][0] / 8.0
        n = c["GRBM_GUI_ACTIVE"][1]
        mf = c.get("SQ_VALU_MFMA_BUSY_CYCLES", (0, 0))[0]
        la = c.get("SQ_LDS_IDX_ACTIVE", (0, 0))[0]
        bc = c.get("SQ_LDS_BANK_CONFLICT", (0, 0))[0]
        out.append((gui * n, f"{k[:60]:60s} {b:7d} {n:8d} {gui / 1e6:8.2f} {100 * mf / (gui * 1024):10.1f} {100 * la / (gui * 256):11.1f} "
                             f"{(100 * bc / la if la else 0):9.1f}"))
    for _, line in sorted(out, reverse=True)[:45]:
        print(line)


if __name__ == "__main__":
    main()
