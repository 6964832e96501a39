"""tools/pmc_report.py -- turn the rocprofv3 --pmc passes of tools/profile_round.sh into the JSON summaries kept under
profiles/ (bench.py reads profiles/<tag>_pmc_traffic.json, _pmc_mfma.json, _pmc_knn.json, _pmc_traffic_stress5.json).

    python tools/pmc_report.py traffic  FETCH.csv WRITE.csv [CALIB.json]  > profiles/r02_pmc_traffic.json
    python tools/pmc_report.py mfma     MFMA.csv                          > profiles/r02_pmc_mfma.json
    python tools/pmc_report.py knn      SQ.csv                            > profiles/r02_pmc_knn.json
    python tools/pmc_report.py calib    FETCH.csv WRITE.csv               > profiles/r02_pmc_calibration.json

Corrections per /opt/skills/guides/MI355X_MICROARCH.md (section HBM): FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE reports half of the bytes of a wide coalesced read stream (doubled here: "read_bytes_corrected"); WRITE_SIZE
is calibrated against stores of known size in the calib pass (factor applied when a calibration file is given).
"""
import collections, csv, json, sys

N_SIMD, N_XCD = 1024, 8


def classify(k):
    if "nm_geo_mlp" in k:
        return "geo_mlp_tangent" if ("kernel<true" in k) else "geo_mlp"
    if "nm_col_mlp" in k:
        return "color_mlp"
    if "nm_distance_kernel" in k:
        return "knn_distance"
    if "nm_probe_bounds" in k:
        return "knn_probe_bounds"
    if "nm_rays_upsample" in k or "nm_rays_finalize" in k or "nm_rays_composite" in k or "nm_rays_order" in k:
        return "per_ray_kernels"
    if "calib_write" in k or "FillFunctor" in k or "fill" in k.lower():
        return "calib_fill"
    if "rocclr_copyBuffer" in k:
        return "calib_copy"
    return None


def per_kernel(path, counters):
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(collections.Counter)
    for r in csv.DictReader(open(path)):
        name = classify(r["Kernel_Name"])
        if name and r["Counter_Name"] in counters:
            tot[name][r["Counter_Name"]] += float(r["Counter_Value"])
            n[name][r["Counter_Name"]] += 1
    return tot, n


def traffic(fetch_csv, write_csv, calib=None):
    f, nf = per_kernel(fetch_csv, {"FETCH_SIZE"})
    w, nw = per_kernel(write_csv, {"WRITE_SIZE"})
    wf = json.load(open(calib))["write_size_factor"] if calib else 1.0
    out = {}
    for k in sorted(set(f) | set(w)):
        launches = max(nf[k]["FETCH_SIZE"], nw[k]["WRITE_SIZE"], 1)
        rd = f[k]["FETCH_SIZE"] * 1024.0 / launches
        wr = w[k]["WRITE_SIZE"] * 1024.0 / launches
        out[k] = {"launches": launches, "read_bytes_raw_per_launch": rd, "read_bytes_corrected_per_launch": 2.0 * rd,
                  "write_bytes_raw_per_launch": wr, "write_bytes_calibrated_per_launch": wr * wf,
                  "hbm_bytes_per_launch": 2.0 * rd + wr * wf}
    out["_notes"] = {"write_size_factor": wf, "read_factor": 2.0,
                     "units": "bytes per launch; FETCH_SIZE x 1024 x 2 (gfx950 wide-read correction), WRITE_SIZE x 1024 x calibration factor"}
    print(json.dumps(out, indent=1))


def mfma(path):
    v, n = per_kernel(path, {"SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_MFMA", "GRBM_GUI_ACTIVE", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES"})
    out = {}
    for k in v:
        if "mlp" not in k:
            continue
        active = v[k]["GRBM_GUI_ACTIVE"] / N_XCD
        out[k] = {"launches": n[k]["GRBM_GUI_ACTIVE"], "mfma_instructions": v[k]["SQ_INSTS_MFMA"],
                  "mfma_busy_cycles_all_simds": v[k]["SQ_VALU_MFMA_BUSY_CYCLES"], "active_cycles_per_xcd": active,
                  "mfma_pipe_utilisation": v[k]["SQ_VALU_MFMA_BUSY_CYCLES"] / (N_SIMD * active) if active else None}
    print(json.dumps(out, indent=1))


def knn(path):
    names = {"SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU",
             "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "GRBM_GUI_ACTIVE", "SQ_BUSY_CYCLES", "SQ_WAVES"}
    v, n = per_kernel(path, names)
    out = {}
    for k in v:
        if not k.startswith("knn"):
            continue
        c = v[k]
        wc = c["SQ_WAVE_CYCLES"] or 1.0
        out[k] = {"launches": max(n[k].values()), "counters": dict(c),
                  "wave_life_issuing": c["SQ_ACTIVE_INST_ANY"] / wc, "wave_life_parked_on_waitcnt": c["SQ_WAIT_ANY"] / wc,
                  "wave_life_issue_stalled": c["SQ_WAIT_INST_ANY"] / wc,
                  "valu_instructions_per_wave": c["SQ_INSTS_VALU"] / c["SQ_WAVES"] if c.get("SQ_WAVES") else None,
                  "salu_instructions_per_wave": c["SQ_INSTS_SALU"] / c["SQ_WAVES"] if c.get("SQ_WAVES") else None,
                  "smem_instructions_per_wave": c["SQ_INSTS_SMEM"] / c["SQ_WAVES"] if c.get("SQ_WAVES") else None}
    print(json.dumps(out, indent=1))


def calib(fetch_csv, write_csv):
    """tools/pmc_calib.py writes 2 GiB with a plain 16-byte-per-lane store kernel (torch fill) and copies 2 GiB."""
    f, nf = per_kernel(fetch_csv, {"FETCH_SIZE"})
    w, nw = per_kernel(write_csv, {"WRITE_SIZE"})
    GiB2 = 2.0 * 2 ** 30
    fill_w = w["calib_fill"]["WRITE_SIZE"] * 1024.0 / max(nw["calib_fill"]["WRITE_SIZE"], 1)
    copy_w = w["calib_copy"]["WRITE_SIZE"] * 1024.0 / max(nw["calib_copy"]["WRITE_SIZE"], 1) if "calib_copy" in w else None
    copy_r = f["calib_copy"]["FETCH_SIZE"] * 1024.0 / max(nf["calib_copy"]["FETCH_SIZE"], 1) if "calib_copy" in f else None
    print(json.dumps({"known_bytes": GiB2, "fill_WRITE_SIZE_bytes": fill_w, "write_size_factor": GiB2 / fill_w if fill_w else None,
                      "copy_WRITE_SIZE_bytes": copy_w, "copy_FETCH_SIZE_bytes": copy_r,
                      "copy_fetch_factor": GiB2 / copy_r if copy_r else None}, indent=1))


if __name__ == "__main__":
    {"traffic": traffic, "mfma": mfma, "knn": knn, "calib": calib}[sys.argv[1]](*sys.argv[2:])
