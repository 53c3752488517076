#!/usr/bin/env python3
"""gpurun_out/prof_TAG (tools/profile_round.sh) -> profiles/TAG_* and profiles/traffic.json.

    python tools/summarise_profiles.py TAG

Copies the per-kernel duration tables and the per-launch counter averages, and writes, per workload, the HBM bytes per
launch of the contraction kernel (gfx950 correction: 2 x FETCH_SIZE + WRITE_SIZE, in KiB; MI355X_MICROARCH.md "HBM")
together with the md5 of the library build they were measured on -- bench.py reports `traffic` only for that build."""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_launch(path, counter, kernel_substr):
    for r in csv.DictReader(open(path)):
        if r["counter"] == counter and kernel_substr in r["kernel"]:
            return float(r["avg_per_launch"]), r["kernel"]
    return None, None


def main():
    tag = sys.argv[1]
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    dst = os.path.join(ROOT, "profiles")
    md5 = open(os.path.join(src, "lib_md5.txt")).read().strip()
    tj_path = os.path.join(dst, "traffic.json")
    try:
        tj = json.load(open(tj_path))
        if "hbm_bytes_per_launch" in tj:            # round-1 layout (one kernel, no build id)
            tj = {}
    except Exception:
        tj = {}
    for f in sorted(glob.glob(os.path.join(src, "*.csv"))):
        shutil.copy(f, os.path.join(dst, "%s_%s" % (tag, os.path.basename(f))))
    for f in sorted(glob.glob(os.path.join(src, "*_pmc_FETCH_SIZE.csv"))):
        w = os.path.basename(f)[:-len("_pmc_FETCH_SIZE.csv")]
        fetch, kname = per_launch(f, "FETCH_SIZE", "gemm")
        write, _ = per_launch(f.replace("FETCH_SIZE", "WRITE_SIZE"), "WRITE_SIZE", "gemm")
        if fetch is None or write is None:
            continue
        entry = {"kernel": kname, "FETCH_SIZE_KiB_per_launch": fetch, "WRITE_SIZE_KiB_per_launch": write,
                 "correction": "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (MI355X_MICROARCH.md: FETCH_SIZE reports 1/2 of wide coalesced reads on gfx950)",
                 "hbm_bytes_per_launch": (2.0 * fetch + write) * 1024.0, "lib_md5": md5,
                 "source": "profiles/%s_%s_pmc_FETCH_SIZE.csv, profiles/%s_%s_pmc_WRITE_SIZE.csv (separate rocprofv3 --pmc passes, averages over both contraction launches of an iteration)" % (tag, w, tag, w)}
        sf, sk = per_launch(f, "FETCH_SIZE", "sweep")
        sw, _ = per_launch(f.replace("FETCH_SIZE", "WRITE_SIZE"), "WRITE_SIZE", "sweep")
        if sf is not None and sw is not None:
            entry["sweep"] = {"kernel": sk, "hbm_bytes_per_launch": (2.0 * sf + sw) * 1024.0}
        tj[w] = entry
    json.dump(tj, open(tj_path, "w"), indent=1)
    print(json.dumps(tj, indent=1))


if __name__ == "__main__":
    main()
