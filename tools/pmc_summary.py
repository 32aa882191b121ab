"""Aggregate rocprofv3 --pmc CSV output (counter_collection.csv) per kernel name.
    python tools/pmc_summary.py gpurun_out/prof
Prints, per kernel: dispatches and the SUM of every collected counter, plus derived HBM bytes per dispatch
(FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE under-counts wide coalesced reads by 2x:
MI355X_MICROARCH.md s.HBM -> both the raw and the x2-corrected read figure are printed)."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


XCDS = 8        # MI355X: 8 XCDs, one GRBM each


def short(name):
    name = re.sub(r"^void ", "", name)
    name = name.replace("(anonymous namespace)::", "")      # (else the argument-list strip below eats the whole name)
    name = re.sub(r"\(.*$", "", name)
    return name[:70]


def main(root):
    agg = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(lambda: defaultdict(int))
    for f in glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = short(row.get("Kernel_Name", "?"))
                c = row.get("Counter_Name")
                v = float(row.get("Counter_Value", 0) or 0)
                agg[k][c] += v
                calls[k][c] += 1
    counters = sorted({c for k in agg for c in agg[k]})
    print("counters:", " ".join(counters))
    rows = []
    for k in agg:
        n = max(calls[k].values())
        fetch = agg[k].get("FETCH_SIZE", 0.0) * 1024
        write = agg[k].get("WRITE_SIZE", 0.0) * 1024
        rows.append((fetch * 2 + write, k, n, fetch, write))
    print("%-70s %6s %14s %14s %14s | per dispatch: read(x2) MB  write MB" % ("kernel", "calls", "FETCH bytes", "FETCHx2 bytes", "WRITE bytes"))
    for tot, k, n, fetch, write in sorted(rows, reverse=True)[:40]:
        nf = max(calls[k].get("FETCH_SIZE", 0), 1)
        nw = max(calls[k].get("WRITE_SIZE", 0), 1)
        print("%-70s %6d %14.4g %14.4g %14.4g | %9.2f %9.2f" % (k, n, fetch, 2 * fetch, write, 2 * fetch / nf / 1e6, write / nw / 1e6))
    print()
    for k in sorted(agg, key=lambda k: -agg[k].get("SQ_BUSY_CYCLES", 0))[:25]:
        a = agg[k]
        if "SQ_VALU_MFMA_BUSY_CYCLES" not in a:
            continue
        # SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD summed; GRBM_GUI_ACTIVE = wall cycles of the dispatch
        # GRBM_GUI_ACTIVE arrives SUMMED over the 8 XCDs (one GRBM per XCD, each counting the dispatch's wall cycles): the wall clock
        # of the dispatch is gui / 8.  Round 1 divided by the raw sum and printed 5.5 % for a GEMM whose matrix pipes were busy 44 %
        # of the time.  MfmaUtil = MFMA-busy cycles / (wall cycles x 1024 SIMDs).
        gui = a.get("GRBM_GUI_ACTIVE", 0.0) / XCDS
        mfma = a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        util = mfma / (gui * 1024) if gui else 0.0          # 256 CUs x 4 SIMDs
        print("%-70s MFMA busy %.3g cyc, wall (GUI active / 8 XCDs) %.3g cyc -> MfmaUtil %.1f%%  (waves: wait_any %.0f%%, wait_inst %.0f%%, active %.0f%% of wave cycles)" % (
            k, mfma, gui, 100 * util, 100 * a.get("SQ_WAIT_ANY", 0) / max(a.get("SQ_WAVE_CYCLES", 1), 1),
            100 * a.get("SQ_WAIT_INST_ANY", 0) / max(a.get("SQ_WAVE_CYCLES", 1), 1), 100 * a.get("SQ_ACTIVE_INST_ANY", 0) / max(a.get("SQ_WAVE_CYCLES", 1), 1)))
    for k in sorted(agg, key=lambda k: -agg[k].get("TCC_HIT_sum", 0))[:15]:
        a = agg[k]
        if "TCC_HIT_sum" in a:
            print("%-70s L2 hit rate %.1f%%" % (k, 100 * a["TCC_HIT_sum"] / max(a["TCC_HIT_sum"] + a.get("TCC_MISS_sum", 0), 1)))


    # machine-readable total for bench.py's roofline.traffic: HBM-side bytes of the GEMM family per profiled step
    # steps that ran under the counters = launches of the optimizer kernel (one per step; bench.py's warm-up, timed and
    # host-enqueue-measurement steps all count), unless PMC_STEPS says otherwise
    adam = [k for k in agg if "adamw_seg_kernel" in k]
    steps = int(os.environ.get("PMC_STEPS", "0")) or (max(calls[adam[0]].values()) if adam else 3)
    gemm = [k for k in agg if "gemm" in k]
    rd = sum(agg[k].get("FETCH_SIZE", 0.0) for k in gemm) * 1024 * 2
    wr = sum(agg[k].get("WRITE_SIZE", 0.0) for k in gemm) * 1024
    launches = sum(calls[k].get("FETCH_SIZE", 0) for k in gemm)
    import json
    print("JSON " + json.dumps(dict(gemm_read_bytes_per_step=rd / steps, gemm_write_bytes_per_step=wr / steps, gemm_launches_per_step=launches / steps,
                                    steps=steps, measured_at_head=os.environ.get("DH_HEAD") or None, note="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), KiB -> bytes, FETCH_SIZE x2 (gfx950 correction, MI355X_MICROARCH.md s.HBM); L2-miss traffic incl. Infinity-Cache hits")))


if __name__ == "__main__":
    main(sys.argv[1])
