#!/usr/bin/env python3
"""Per-kernel table from rocprofv3 --pmc passes over SQ / GRBM counters (separate runs of the same bench command).

    python tools/summarize_sq.py <pass1_counter_collection.csv> [<pass2...>] > table.md

Per kernel name (template instances kept apart), launch-weighted means of:
  us          dispatch duration (End - Start timestamps of the counter pass: a profiled pass, slower than a bare run)
  clk GHz     GRBM_GUI_ACTIVE / 8 XCDs / duration: the clock the chip held during the kernel
  MFMA busy   SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x kernel cycles): share of matrix-core cycles in use
  MFMA/peak   the bf16 rate that share gives at the held clock, over the 2.5 PFLOP/s nominal peak (2.4 GHz)
  issue       SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES: share of wave-cycles in which the wave issues something
  wait        SQ_WAIT_ANY / SQ_WAVE_CYCLES (parked on s_waitcnt or a barrier), stall = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES
  LDS busy    SQ_LDS_IDX_ACTIVE / (256 CUs x kernel cycles); conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  LDS stall   SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES
rocprofv3 reports each counter summed over the device (all XCDs, SEs, CUs).
"""
import collections
import csv
import re
import sys

CUS, SIMDS, XCDS = 256, 4, 8


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)


def main():
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))   # kernel -> counter -> [n, sum]
    dur = collections.defaultdict(lambda: [0, 0.0])
    for path in sys.argv[1:]:
        seen = set()
        for r in csv.DictReader(open(path)):
            k = short(r["Kernel_Name"])
            a = acc[k][r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
            key = (path, r["Dispatch_Id"])
            if key not in seen:
                seen.add(key)
                d = dur[k]
                d[0] += 1
                d[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
    rows = []
    for k, c in acc.items():
        m = {name: v[1] / v[0] for name, v in c.items() if v[0]}
        us = dur[k][1] / dur[k][0]
        n = max(v[0] for v in c.values())
        cyc = m.get("GRBM_GUI_ACTIVE", 0.0) / XCDS
        rows.append((us * n, k, n, us, cyc, m))
    rows.sort(reverse=True)
    print("| kernel | launches | us | clk GHz | MFMA busy | MFMA/peak | issue | wait | stall | LDS busy | conflict | LDS stall |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|")

    def pct(x):
        return "-" if x is None else f"{100 * x:.0f} %"

    for _, k, n, us, cyc, m in rows[:40]:
        wc = m.get("SQ_WAVE_CYCLES")
        clk = cyc / (us * 1e3) if cyc and us else None
        busy = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (SIMDS * CUS * cyc) if cyc and "SQ_VALU_MFMA_BUSY_CYCLES" in m else None
        frac = busy * clk / 2.4 if busy is not None and clk else None
        issue = m["SQ_ACTIVE_INST_ANY"] / wc if wc and "SQ_ACTIVE_INST_ANY" in m else None
        wait = m["SQ_WAIT_ANY"] / wc if wc and "SQ_WAIT_ANY" in m else None
        stall = m["SQ_WAIT_INST_ANY"] / wc if wc and "SQ_WAIT_INST_ANY" in m else None
        lds = m["SQ_LDS_IDX_ACTIVE"] / (CUS * cyc) if cyc and "SQ_LDS_IDX_ACTIVE" in m else None
        conf = m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"] if m.get("SQ_LDS_IDX_ACTIVE") and "SQ_LDS_BANK_CONFLICT" in m else None
        ldsst = m["SQ_WAIT_INST_LDS"] / wc if wc and "SQ_WAIT_INST_LDS" in m else None
        print(f"| `{k}` | {n} | {us:.1f} | {clk:.2f} | {pct(busy)} | {pct(frac)} | {pct(issue)} | {pct(wait)} | {pct(stall)} | "
              f"{pct(lds)} | {pct(conf)} | {pct(ldsst)} |" if clk else f"| `{k}` | {n} | {us:.1f} | - | - | - | {pct(issue)} | {pct(wait)} | {pct(stall)} | - | {pct(conf)} | {pct(ldsst)} |")


if __name__ == "__main__":
    main()
