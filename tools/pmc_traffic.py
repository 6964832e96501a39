"""tools/pmc_traffic.py -- turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py into
profiles/r01_pmc_traffic.json: HBM bytes per launch per hot kernel.

Corrections per /opt/skills/guides/MI355X_MICROARCH.md (section HBM): the counters are in KiB-like units
(x1024); on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide coalesced read stream, so the
read side is doubled ("read_bytes_corrected"); WRITE_SIZE is uncalibrated and taken as is.

    python tools/pmc_traffic.py gpurun_out/pmc_fetch/*_counter_collection.csv gpurun_out/pmc_write/*_counter_collection.csv
"""
import collections, csv, json, sys


def per_kernel(path, counter):
    tot, n = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"]
        name = ("geo_mlp_tangent" if ("nm_geo_mlp_kernel<true" in k or "nm_geo_mlp_h_kernel<true" in k)
                else "geo_mlp" if ("nm_geo_mlp_kernel<false" in k or "nm_geo_mlp_h_kernel<false" in k)
                else "color_mlp" if "nm_col_mlp" in k else "knn_distance" if "nm_distance_kernel" in k else None)
        if name:
            tot[name] += float(r["Counter_Value"])
            n[name] += 1
    return {k: (tot[k] / n[k], n[k]) for k in tot}


def main(fetch_csv, write_csv):
    f, w = per_kernel(fetch_csv, "FETCH_SIZE"), per_kernel(write_csv, "WRITE_SIZE")
    out = {}
    for k in f:
        rd = f[k][0] * 1024.0
        wr = w.get(k, (0.0, 0))[0] * 1024.0
        out[k] = {"launches": f[k][1], "FETCH_SIZE_avg": f[k][0], "WRITE_SIZE_avg": w.get(k, (0.0, 0))[0],
                  "read_bytes_raw": rd, "read_bytes_corrected": 2.0 * rd, "write_bytes": wr,
                  "hbm_bytes_per_launch": 2.0 * rd + wr}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
