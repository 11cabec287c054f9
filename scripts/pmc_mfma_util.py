"""Per-kernel MFMA utilisation from one rocprofv3 PMC pass:
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d <dir> -- python3 bench.py ...
    python scripts/pmc_mfma_util.py <dir>/**/*_counter_collection.csv

utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles x 1024 SIMDs): the fraction of SIMD-cycles of the kernel's lifetime in
which the matrix pipe was busy (MI355X: 256 CUs x 4 SIMDs; MI355X_MICROARCH.md: MFMA_BUSY counts shader cycles, 32 per
32x32x16 bf16 MFMA -- checked: 75 GFLOP per launch = 2.29 M MFMAs = 73 Mcycles, the counter reads 63-75 M).  rocprofv3 reports
GRBM_GUI_ACTIVE summed over the 8 XCDs (a 90 us launch reads 1.4 Mcycles = 8 x 177 k), so kernel cycles = GUI_ACTIVE / 8.
Counters are summed over the launches of a kernel.  `eff GHz` = GUI_ACTIVE / 8 over the launches' wall time (timestamps of the same
pass): the clock the chip actually ran the kernel at (MI355X_MICROARCH.md, DVFS give-back: it clocks to its power budget -- dense
MFMA bodies on random operands sit near 1.9-2.0 GHz of the nominal 2.4), i.e. what the 2.5 PF datasheet peak is worth there."""
import csv
import re
import sys
from collections import defaultdict

csv.field_size_limit(1 << 30)


def short(name: str) -> str:
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_:]+(<[^(>]*>)?)", name)
    return (m.group(1) if m else name)[:48]


acc: dict[str, dict[str, float]] = defaultdict(lambda: defaultdict(float))
n: dict[str, int] = defaultdict(int)
for path in sys.argv[1:]:
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            k = short(r["Kernel_Name"])
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                n[k] += 1
                if r.get("Start_Timestamp") and r.get("End_Timestamp"):
                    acc[k]["wall_ns"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
print(f"{'kernel':48s} {'launches':>8s} {'GUI_ACTIVE Mcyc':>16s} {'MFMA_BUSY Mcyc':>15s} {'MFMA util':>10s} {'eff GHz':>8s}")
tot_b = tot_a = 0.0
for k, c in sorted(acc.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
    a, b = c.get("GRBM_GUI_ACTIVE", 0.0), c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    if a <= 0:
        continue
    tot_a, tot_b = tot_a + a, tot_b + b
    ghz = f"{a / 8 / c['wall_ns']:8.2f}" if c.get("wall_ns", 0) > 0 else f"{'':8s}"
    print(f"{k:48s} {n[k]:8d} {a / 1e6:16.2f} {b / 1e6:15.2f} {b / (a / 8 * 1024):10.3f} {ghz}")
print(f"{'all kernels (sum of lifetimes, two streams overlap)':48s} {'':8s} {tot_a / 1e6:16.2f} {tot_b / 1e6:15.2f} {tot_b / (tot_a / 8 * 1024):10.3f}")
