#!/usr/bin/env python3
"""Condense the rocprofv3 outputs of tools/profile.sh into the files kept under profiles/.

    python tools/summarize_profile.py gpurun_out/<name> profiles/<prefix>

writes <prefix>_kernel_stats.csv (the --kernel-trace --stats table as is) and <prefix>_pmc_summary.json (mean counter
value per launch of the cycle kernel, one entry per counter of the separate --pmc passes).
"""
import csv
import glob
import json
import os
import shutil
import sys


def main():
    src, prefix = sys.argv[1], sys.argv[2]
    ks = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))
    if ks:
        shutil.copy(ks[0], prefix + "_kernel_stats.csv")
    out = {}
    kinfo = None
    for f in sorted(glob.glob(os.path.join(src, "pmc_*", "*", "*_counter_collection.csv"))):
        acc = {}
        for row in csv.DictReader(open(f)):
            if "dwbc_cycle_kernel" not in row["Kernel_Name"]:
                continue
            if kinfo is None:
                kinfo = dict(kernel=row["Kernel_Name"][:110], grid=row["Grid_Size"], wg=row["Workgroup_Size"], lds=row["LDS_Block_Size"],
                             scratch=row["Scratch_Size"], vgpr=row["VGPR_Count"], agpr=row["Accum_VGPR_Count"], sgpr=row["SGPR_Count"])
            a = acc.setdefault(row["Counter_Name"], [0.0, 0])
            a[0] += float(row["Counter_Value"])
            a[1] += 1
        for k, (tot, n) in acc.items():
            out[k] = dict(mean_per_launch=tot / n, launches=n)
    out["_kernel"] = kinfo
    out["_note"] = ("rocprofv3 --pmc passes (one counter group per run, no tracing) of: python3 bench.py --steps 20 --warmup 3 "
                    "--no-cpu-baseline.  FETCH_SIZE / WRITE_SIZE in KiB.  SQ_* cycle counters count per wave in units of 4 clocks.")
    json.dump(out, open(prefix + "_pmc_summary.json", "w"), indent=1)
    print("wrote", prefix + "_kernel_stats.csv", prefix + "_pmc_summary.json")


if __name__ == "__main__":
    main()
