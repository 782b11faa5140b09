#!/usr/bin/env python3
"""Per-kernel table from rocprofv3 --pmc passes over SQ / GRBM counters (separate runs of the same bench command).

    python tools/summarize_sq.py <pass1_counter_collection.csv> [<pass2...>] > table.md

Per kernel name (template instances kept apart), launch-weighted means of:
  us          dispatch duration (End - Start timestamps of the counter pass: a profiled pass, slower than a bare run)
  clk GHz     SQ_BUSY_CU_CYCLES / 256 CUs / duration: the clock the chip held, for kernels that keep every CU busy
              from start to end (the persistent convolution kernels; it reads low for kernels that do not).
              GRBM_GUI_ACTIVE / 8 / duration is printed beside it: it counts activity around the dispatch as well
              and reads 10-20 % high on dispatches of under 0.1 ms.
  MFMA busy   SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES): share of the busy CUs' matrix-core cycles in
              use (the counter adds 16 per v_mfma_f32_16x16x32_bf16, 32 per 32x32x16: checked against SQ_INSTS_MFMA)
  MFMA/peak   that share at the held clock over the nominal 2.4 GHz: what fraction of the 2.5 PFLOP/s figure the
              kernel's issued MFMAs amount to (padded channels included)
  issue / wait / stall   SQ_ACTIVE_INST_ANY, SQ_WAIT_ANY (parked on s_waitcnt or a barrier), SQ_WAIT_INST_ANY (an
              instruction waiting to issue) over SQ_WAVE_CYCLES
  LDS busy    SQ_LDS_IDX_ACTIVE / SQ_BUSY_CU_CYCLES (4 per ds_read_b128: checked against SQ_INSTS_LDS);
              conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE; LDS stall = SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES
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
    print("| kernel | launches | us | clk GHz (CU-busy / GRBM) | MFMA busy | MFMA/peak | issue | wait | stall | LDS busy | conflict | LDS stall |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|")

    def pct(x):
        return "-" if x is None else f"{100 * x:.0f} %"

    for _, k, n, us, cyc, m in rows[:40]:
        wc = m.get("SQ_WAVE_CYCLES")
        cu = m.get("SQ_BUSY_CU_CYCLES")
        clk = cu / CUS / (us * 1e3) if cu and us else None
        clk_grbm = cyc / (us * 1e3) if cyc and us else None
        busy = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (SIMDS * cu) if cu and "SQ_VALU_MFMA_BUSY_CYCLES" in m else None
        frac = busy * clk / 2.4 if busy is not None and clk else None
        issue = m["SQ_ACTIVE_INST_ANY"] / wc if wc and "SQ_ACTIVE_INST_ANY" in m else None
        wait = m["SQ_WAIT_ANY"] / wc if wc and "SQ_WAIT_ANY" in m else None
        stall = m["SQ_WAIT_INST_ANY"] / wc if wc and "SQ_WAIT_INST_ANY" in m else None
        lds = m["SQ_LDS_IDX_ACTIVE"] / cu if cu and "SQ_LDS_IDX_ACTIVE" in m else None
        conf = m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"] if m.get("SQ_LDS_IDX_ACTIVE") and "SQ_LDS_BANK_CONFLICT" in m else None
        ldsst = m["SQ_WAIT_INST_LDS"] / wc if wc and "SQ_WAIT_INST_LDS" in m else None
        clks = "-" if clk is None else f"{clk:.2f} / {clk_grbm:.2f}" if clk_grbm else f"{clk:.2f}"
        print(f"| `{k}` | {n} | {us:.1f} | {clks} | {pct(busy)} | {pct(frac)} | {pct(issue)} | {pct(wait)} | {pct(stall)} | "
              f"{pct(lds)} | {pct(conf)} | {pct(ldsst)} |")


if __name__ == "__main__":
    main()
