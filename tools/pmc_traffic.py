"""HBM bytes per launch of one kernel from two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE).

    python tools/pmc_traffic.py --fetch DIR --write DIR --kernel 'k_partials_dna_fused<4, 4>' \
        --grid 800768 200192 --algorithmic 462000000 --out profiles/traffic_c2.json --trim profiles/r1_c2_pmc

Method (profiles/README.md, MI355X_MICROARCH.md "HBM / rocprofv3"): the two counters do not share a
pass on gfx950; values are KiB; FETCH_SIZE counts 64 B per 128-B read request, i.e. reports half
of a streamed read (calibrated with tools/pmc_calib.hip) and is doubled here; WRITE_SIZE is exact.
--grid keeps only dispatches with these Grid_Size values (the launches bench.py's roofline leg
times); --trim writes the rows used as small CSVs for the record."""
import argparse
import csv
import glob
import json
import os


def rows(d, counter):
    out = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                if r.get("Counter_Name") == counter:
                    out.append(r)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fetch", required=True)
    ap.add_argument("--write", required=True)
    ap.add_argument("--kernel", required=True)
    ap.add_argument("--grid", type=int, nargs="*", default=[])
    ap.add_argument("--algorithmic", type=float, required=True, help="algorithmic bytes per launch (bench.py's figure)")
    ap.add_argument("--out", required=True)
    ap.add_argument("--trim")
    a = ap.parse_args()
    res = {}
    for name, d in (("FETCH_SIZE", a.fetch), ("WRITE_SIZE", a.write)):
        sel = [r for r in rows(d, name) if a.kernel in r["Kernel_Name"] and (not a.grid or int(r["Grid_Size"]) in a.grid)]
        if not sel:
            raise SystemExit(f"no {name} rows for {a.kernel}")
        res[name] = sel
        if a.trim:
            keep = ["Dispatch_Id", "Grid_Size", "Kernel_Name", "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp"]
            with open(f"{a.trim}_{name}.csv", "w", newline="") as fh:
                w = csv.DictWriter(fh, fieldnames=keep, extrasaction="ignore")
                w.writeheader()
                for r in rows(d, name):
                    if "k_partials" in r["Kernel_Name"] or "k_edge" in r["Kernel_Name"]:
                        w.writerow(r)
    nf, nw = len(res["FETCH_SIZE"]), len(res["WRITE_SIZE"])
    fetch = sum(float(r["Counter_Value"]) for r in res["FETCH_SIZE"])
    write = sum(float(r["Counter_Value"]) for r in res["WRITE_SIZE"])
    per_launch = (2.0 * fetch / nf + write / nw) * 1024.0
    import subprocess
    try:
        commit = subprocess.check_output(['git', '-C', os.path.dirname(os.path.abspath(__file__)), 'rev-parse', '--short', 'HEAD'], text=True, stderr=subprocess.DEVNULL).strip()
    except Exception:
        commit = os.environ.get('PLL_COMMIT', 'unknown (no git on the GPU box: set PLL_COMMIT)')
    grids = sorted({int(r['Grid_Size']) for r in res['WRITE_SIZE']})
    out = dict(hbm_bytes_per_launch=int(round(per_launch)), commit=commit, grid_sizes_seen=grids, algorithmic_bytes_per_launch=int(a.algorithmic),
               ratio=round(per_launch / a.algorithmic, 4), kernel=a.kernel, grid_sizes=a.grid,
               method="rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (with --kernel-trace only) around "
                      "`python3 bench.py --steps 3 --warmup 1 --no-cpu`; mean over the dispatches of the kernel with the grid sizes "
                      "bench.py's roofline leg launches; counters are KiB; FETCH_SIZE doubled (gfx950 counts 64 B per 128 B read "
                      "request, calibrated with tools/pmc_calib.hip); WRITE_SIZE exact",
               fetch_KiB_mean=round(fetch / nf, 2), write_KiB_mean=round(write / nw, 2), dispatches=[nf, nw])
    with open(a.out, "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
