"""tools/pmc_mfma.py -- matrix-pipe utilisation of the MLP kernels from one rocprofv3 --pmc pass
(SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE ...) of bench.py.

SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over all SIMDs (= 32 x the number of 32x32x16 MFMAs,
MI355X_MICROARCH.md); GRBM_GUI_ACTIVE is summed over the 8 XCDs.  utilisation = busy / (1024 SIMDs x
active cycles per XCD).

    python tools/pmc_mfma.py gpurun_out/pmc_mfma/p_counter_collection.csv > profiles/r01_pmc_mfma.json
"""
import collections, csv, json, sys

N_SIMD, N_XCD = 1024, 8


def main(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        name = ("geo_mlp_tangent" if "nm_geo_mlp_h_kernel<true" in k else "geo_mlp" if "nm_geo_mlp_h_kernel<false" in k
                else "color_mlp" if "nm_col_mlp_h" in k else None)
        if name:
            agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                n[name] += 1
    out = {}
    for k, v in agg.items():
        active = v["GRBM_GUI_ACTIVE"] / N_XCD
        out[k] = {"launches": n[k], "mfma_instructions": v["SQ_INSTS_MFMA"], "mfma_busy_cycles_all_simds": v["SQ_VALU_MFMA_BUSY_CYCLES"],
                  "active_cycles_per_xcd": active, "mfma_pipe_utilisation": v["SQ_VALU_MFMA_BUSY_CYCLES"] / (N_SIMD * active)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
