"""gpurun_out/pmc_r06_*.txt (tools/pmc_r06.sh) -> profiles/r06_pmc_digest.json

  land_surface   HBM bytes per pixel and model step of the land-surface kernels of the resident step, three forms: the
                 separate launches of rounds 1-5 (k_canopy + k_scale_rows + k_soil_fused + k_soil_stragglers), the one pass of
                 round 6 (k_soil_fused<.., CANOPY> + k_soil_stragglers), and the one pass with the optional maps left out;
                 k_pixel_aggregates full / lean beside them
  workloads      bench.py's schema (pmc_traffic_r03): the cone launches of the model step with structures
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")


def parse(path):
    k = {}
    for line in open(path):
        m = re.match(r"(\S.*?)\s+launches=(\d+)\s+read_GB=(\S+)\s+write_GB=(\S+)\s+total_GB=(\S+)", line)
        if m:
            k[m.group(1).strip()] = dict(launches=int(m.group(2)), read_GB=float(m.group(3)), write_GB=float(m.group(4)))
    return k


def main():
    size, steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000, 4
    pix_steps = size * size * steps
    land = {}
    for mode in ("separate", "fused", "lean"):
        k = parse(os.path.join(OUT, "pmc_r06_land_%s.txt" % mode))
        names = [n for n in k if n.startswith(("k_canopy", "k_scale_rows", "k_soil_fused", "k_soil_stragglers"))]
        tot = sum(k[n]["read_GB"] + k[n]["write_GB"] for n in names) * 1e9
        agg = k["k_pixel_aggregates"]
        land[mode] = dict(kernels={n: k[n] for n in names + ["k_pixel_aggregates"]},
                          land_surface_bytes_per_pixel_step=round(tot / pix_steps, 1),
                          pixel_aggregates_bytes_per_pixel_step=round((agg["read_GB"] + agg["write_GB"]) * 1e9 / pix_steps, 1),
                          source="profiles/pmc_r06_land_%s.txt" % mode)
    land["algorithmic_bytes_per_pixel_step"] = dict(separate=3 * (200 + 504) + 56 + 25, fused=3 * 576 + 24,
                                                    lean=3 * (576 - 56) + 24, pixel_aggregates=824)
    k = parse(os.path.join(OUT, "pmc_r06_structures.txt"))
    cone = [n for n in k if n.startswith("k_fused_cones<") and ", true, false, 64>" in n][0]      # STRUCT = true: the fused call
    split = [n for n in k if n.startswith("k_fused_cones_split")]
    model_steps, cells, nsub = 4, 3000 * 3000, 24
    tot = (k[cone]["read_GB"] + k[cone]["write_GB"] + sum(k[n]["read_GB"] + k[n]["write_GB"] for n in split)) * 1e9
    launches = k[cone]["launches"] + sum(k[n]["launches"] for n in split)
    w = dict(source="profiles/pmc_r06_structures.txt", model_steps=model_steps,
             hbm_bytes_per_cell_substep=round(tot / model_steps / (cells * nsub), 1),
             kernels={"k_fused_cones (with structures) + k_fused_cones_split": dict(
                 launches=launches, hbm_read_bytes_per_launch=(k[cone]["read_GB"] + sum(k[n]["read_GB"] for n in split)) * 1e9 / launches,
                 hbm_write_bytes_per_launch=(k[cone]["write_GB"] + sum(k[n]["write_GB"] for n in split)) * 1e9 / launches)})
    d = dict(tag="r06", fetch_correction=2.0, land_surface=land, workloads={"fused_structures_deep_3000": w})
    json.dump(d, open(os.path.join(ROOT, "profiles", "r06_pmc_digest.json"), "w"), indent=1)
    print(json.dumps({m: (land[m]["land_surface_bytes_per_pixel_step"], land[m]["pixel_aggregates_bytes_per_pixel_step"])
                      for m in ("separate", "fused", "lean")}), w["hbm_bytes_per_cell_substep"])


if __name__ == "__main__":
    main()
