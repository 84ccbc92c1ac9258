#!/usr/bin/env python3
"""profiles/pmc_traffic.json from two rocprofv3 PMC passes of bench.py (separate runs: --pmc FETCH_SIZE, --pmc WRITE_SIZE).
Corrections per /opt/skills/guides/MI355X_MICROARCH.md and profiles/r01_c_pmc_calibration.txt: unit KB, FETCH_SIZE x2 on gfx950,
WRITE_SIZE exact.  Records the commit and the kernel-source hash the passes ran on; bench.py quotes the figure only for
byte-identical kernel source.   Usage: pmc_traffic_json.py <fetch-dir> <write-dir> <E> <round> [head]"""
import csv, glob, json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def is_dominant(name):
    """The fused weight-gradient + Adam launches: wgrad_adam_rows_kernel<POL> (round 2 default) or conv_wgrad_kernel<.., ADAM = true, ..>."""
    if "wgrad_adam_rows_kernel" in name or "wgrad_adam_fwd_kernel" in name:
        return True
    return "conv_wgrad_kernel" in name and "true" in name.split("conv_wgrad_kernel")[1][:40]


FUSED_SEEN = set()          # which form of the dominant kernel the passes actually ran (decides whose source hash the record carries)


def mean_for(path, counter):
    n, tot = 0, 0.0
    for f in glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] == counter and is_dominant(row["Kernel_Name"]):
                    n += 1
                    tot += float(row["Counter_Value"])
                    FUSED_SEEN.add("wgrad_adam_fwd_kernel" in row["Kernel_Name"])
    return n, (tot / n if n else 0.0)


def main():
    fdir, wdir, E, rnd = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    head = sys.argv[5] if len(sys.argv) > 5 else "unknown"
    import bench
    typed = fdir.startswith("mean:")
    if typed:            # pmc_traffic_json.py mean:<fetch KB>:<calls> mean:<write KB>:<calls> E round head  (from a committed summary)
        # numbers typed on the command line carry NO provenance: the record gets no kernel-source hash, so bench.py never quotes
        # it as `roofline.traffic` (it is a note for humans only)
        fetch, nf = float(fdir.split(":")[1]), int(fdir.split(":")[2])
        write, nw = float(wdir.split(":")[1]), int(wdir.split(":")[2])
    else:
        nf, fetch = mean_for(fdir, "FETCH_SIZE")
        nw, write = mean_for(wdir, "WRITE_SIZE")
    if not typed and len(FUSED_SEEN) != 1:
        raise SystemExit("the passes ran %s forms of the dominant kernel: one record cannot describe both" % ("no" if not FUSED_SEEN else "both"))
    fused = (not typed) and FUSED_SEEN == {True}
    alg = 6 * 4 * E * (512 * 2304 + 512 * 4608 + 512 * 256) / 3 / 1e6       # mean of the three shapes, MB per launch
    out = {"_comment": "HBM traffic per launch of the fused weight-gradient + Adam kernel from PMC counters (separate --pmc FETCH_SIZE / "
                       "--pmc WRITE_SIZE runs of bench.py; FETCH_SIZE x2 on gfx950, WRITE_SIZE x1, unit KB; calibration: "
                       "profiles/r01_c_pmc_calibration.txt).  bench.py copies `traffic` from here only when its episodes-per-step AND the "
                       "kernel-source hash match.",
           "round": rnd, "head": head, "kernel_source_sha16": None if typed else bench.kernel_source_sha(fused), "episodes_per_step": E,
           "kernel": ("wgrad_adam_fwd_kernel (weight gradient + Adam + the next inner step's forward of the layer from the tiles just updated; its "
                      "traffic includes the next step's activation rows each workgroup re-reads through L2)" if fused else
                      "wgrad_adam_rows_kernel (fused weight gradient + Adam; conv_wgrad_kernel<64,64,ADAM> in round 1)"), "launches": nf, "fetch_kb_mean_raw": round(fetch, 1), "write_kb_mean_raw": round(write, 1),
           "traffic_mb_per_launch": round((2.0 * fetch + write) * 1024 / 1e6, 1), "algorithmic_mb_per_launch": round(alg, 1)}
    with open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w") as f:
        json.dump(out, f, indent=2)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
