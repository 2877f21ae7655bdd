#!/usr/bin/env python3
"""HBM traffic of the one-rank kernel from the two rocprofv3 --pmc passes of tools/profiles.sh, written as the
profiles/<tag>_n1_allreduce_1GiB_pmc.md table and the profiles/traffic.json entry bench.py quotes.  Every figure in
the table is computed here from the csv files (kernel time from the --kernel-trace --stats csv of the same script run,
HIP-event times from the bench lines of the same runs), so table and csv cannot drift apart.

    python tools/pmc_summary.py gpurun_out/r03 r03
"""
import csv
import glob
import json
import os
import sys

KERNEL = "raLocalKernel"


def per_launch(dirname, counter):
    files = glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True)
    vals = [float(r["Counter_Value"]) for f in files for r in csv.DictReader(open(f))
            if KERNEL in r["Kernel_Name"] and r["Counter_Name"] == counter]
    return sum(vals) / len(vals), len(vals)


def main():
    src, tag = sys.argv[1], sys.argv[2]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fetch_kib, nf = per_launch(os.path.join(src, "pmc_FETCH_SIZE"), "FETCH_SIZE")
    write_kib, nw = per_launch(os.path.join(src, "pmc_WRITE_SIZE"), "WRITE_SIZE")
    fetch_b = fetch_kib * 1024 * 2            # gfx950 tallies 128-byte read requests as 64 bytes
    write_b = write_kib * 1024
    total = int(round(fetch_b + write_b))
    alg = 2 * (1 << 30)
    stats = glob.glob(os.path.join(src, "prof_n1", "**", "*kernel_stats.csv"), recursive=True)[0]
    row = [r for r in csv.DictReader(open(stats)) if KERNEL in r["Name"]][0]
    under = json.loads([l for l in open(os.path.join(src, f"{tag}_bench_n1_under_rocprof.json")) if l.startswith("{")][-1])
    alone = json.loads([l for l in open(os.path.join(src, f"{tag}_bench_n1.json")) if l.startswith("{")][-1])
    md = f"""# {tag} — HBM traffic of the dominant kernel at N=1 (rocprofv3 --pmc, separate passes)

Written by `tools/pmc_summary.py` from the csv files of one `tools/profiles.sh {tag}` run (each counter in its own run, from
/tmp with TMPDIR=/tmp):

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d <out> -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d <out> -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1

Kernel: `{row['Name'].split('(')[0].replace('void ', '')}` (the one-rank ncclAllReduce path = device copy of the 1 GiB bucket), {nf} / {nw} launches in the passes.

| counter | raw per launch (KiB) | correction | bytes per launch |
|---|---|---|---|
| FETCH_SIZE | {fetch_kib:.1f} | x2 (gfx950 tallies 128-B read requests as 64 B) | {int(round(fetch_b))} |
| WRITE_SIZE | {write_kib:.1f} | none | {int(round(write_b))} |

Traffic per launch = {total} B; algorithmic bytes per launch = 2 x 1073741824 = {alg} B (ratio {total / alg:.4f}).

Kernel time, all from the same script run: `{tag}_n1_allreduce_1GiB_kernel_stats.csv` (rocprofv3 --kernel-trace --stats of
`python3 bench.py --no-cpu-baseline --steps 20 --warmup 5`): {row['Calls']} calls, average {float(row['AverageNs']) / 1e3:.1f} us; the HIP-event
average bench.py printed in that run is {under['roofline']['kernel_ms']:.4f} ms (`{tag}_bench_n1_under_rocprof.json`), {alone['roofline']['kernel_ms']:.4f} ms without the
profiler (`{tag}_bench_n1.json`).
"""
    open(os.path.join(root, "profiles", f"{tag}_n1_allreduce_1GiB_pmc.md"), "w").write(md)
    tfile = os.path.join(root, "profiles", "traffic.json")
    t = json.load(open(tfile)) if os.path.exists(tfile) else {}
    t["allreduce_fp32_1073741824_n1"] = total
    json.dump(t, open(tfile, "w"), indent=1)
    print(md)


if __name__ == "__main__":
    main()
