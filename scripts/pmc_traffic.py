"""Per-kernel HBM traffic from rocprofv3 --pmc passes (FETCH_SIZE in one pass, WRITE_SIZE in another: they do not fit in
one -- MI355X_MICROARCH.md "rocprofv3 PMC slots").

    python scripts/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> [--json out.json]

Units / corrections (MI355X_MICROARCH.md, HBM section): rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB; on gfx950
FETCH_SIZE counts 128-byte requests at 64 bytes, i.e. reports HALF the bytes of wide coalesced streaming reads -> doubled
here.  WRITE_SIZE is uncalibrated in the guide: it is calibrated in the same run on `adamw_k`, whose byte counts are known
exactly (reads 16 B/param: p, g, m, v; writes 12 B/param: p, m, v); the calibration factors are printed.
"""

import csv
import json
import re
import sys
from collections import defaultdict

csv.field_size_limit(1 << 30)


def short(name: str) -> str:
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_:]+(<[^(>]*>)?)", name)
    return (m.group(1) if m else name)[:60]


def load(path: str) -> dict[str, dict[str, list[float]]]:
    out: dict[str, dict[str, list[float]]] = defaultdict(lambda: defaultdict(list))
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            out[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return out


def main() -> None:
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    jpath = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
    fetch, write = load(args[0]), load(args[1])
    n_params = 39_927_296  # DiT-S/2 arena (parameters + alignment padding), what adamw_k sweeps
    rows = {}
    for k in sorted(set(fetch) | set(write)):
        fv, wv = fetch.get(k, {}).get("FETCH_SIZE", []), write.get(k, {}).get("WRITE_SIZE", [])
        if not fv and not wv:
            continue
        rows[k] = {"launches": max(len(fv), len(wv)),
                   "fetch_MB": 2.0 * 1024 * (sum(fv) / len(fv)) / 1e6 if fv else None,  # KiB -> bytes, x2 gfx950 correction
                   "write_MB": 1024 * (sum(wv) / len(wv)) / 1e6 if wv else None}
    cal = rows.get("adamw_k")
    if cal:
        print(f"# calibration on adamw_k: fetch {cal['fetch_MB']:.1f} MB (expect {16 * n_params / 1e6:.1f}), "
              f"write {cal['write_MB']:.1f} MB (expect {12 * n_params / 1e6:.1f})")
    print(f"{'kernel':60s} {'launches':>8s} {'fetch MB/launch':>16s} {'write MB/launch':>16s}")
    for k, r in sorted(rows.items(), key=lambda kv: -((kv[1]['fetch_MB'] or 0) + (kv[1]['write_MB'] or 0)) * kv[1]['launches']):
        f = f"{r['fetch_MB']:.2f}" if r["fetch_MB"] is not None else "-"
        w = f"{r['write_MB']:.2f}" if r["write_MB"] is not None else "-"
        print(f"{k:60s} {r['launches']:8d} {f:>16s} {w:>16s}")
    if jpath:
        json.dump(rows, open(jpath, "w"), indent=1)


if __name__ == "__main__":
    main()
