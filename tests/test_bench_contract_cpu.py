"""The shape of bench.py's JSON line, checked on the committed result of the last full run on an MI355X
(profiles/r03_bench_final.json): the driver's contract (metric / value / unit / n_gpus / steps / warmup / ms_per_step /
higher_is_better / scaling / vs_baseline / dtype / data / config.workload) plus the two objects of this tier, `roofline`
and `cpu_baseline`, and the internal consistency of the numbers."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_follows_the_contract():
    text = open(os.path.join(ROOT, "profiles", "r03_bench_final.json")).read().strip().splitlines()
    assert len(text) == 1                                # ONE JSON line on stdout
    d = json.loads(text[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["dtype"] == "f64" and d["data"] == "synthetic"
    assert d["vs_baseline"] is None                      # BASELINE.md holds no published number for this metric
    assert "workload" in d["config"] and "model" not in d["config"]
    cells = d["config"]["cells"]
    assert abs(d["value"] - cells / d["ms_per_step"] / 1e3) < 1e-3 * d["value"]      # Mcell-steps/s = cells per ms / 1e3
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-6
    # achieved = algorithmic bytes per launch / mean launch duration of the dominant kernel
    assert abs(r["achieved"] - r["alg_bytes_per_launch"] / (r["mean_launch_us"] * 1e-6) / 1e9) < 1e-3 * r["achieved"]
    assert r["alg_bytes_per_cell_step"] == 48.0
    if r["traffic"] is not None:                         # PMC bytes per launch: at least the algorithmic bytes, not 2x them
        assert 1.0 <= r["traffic"] / r["alg_bytes_per_launch"] < 2.0
        assert os.path.exists(os.path.join(ROOT, r["traffic_source"]))
    # the whole step cannot be faster than its dominant kernel's launches
    assert r["launches_per_step"] * r["mean_launch_us"] * 1e-3 <= d["ms_per_step"] * 1.02
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["unit"] == d["unit"]
    assert c.get("cpu_model") and c.get("host_cpus", 0) >= c["cores"]      # core count and CPU model stated
    # every routing leg of the line carries its counter traffic with the file it came from
    for fam in ("deep", "river"):
        rl = d["other_workloads"][fam]["roofline"]
        assert rl["traffic"] and os.path.exists(os.path.join(ROOT, rl["traffic_source"])), fam
    fr = d["other_workloads"]["model_step_24_substeps_split"]["fused"]["roofline"]
    assert fr["hbm_bytes_per_cell_substep"] < 200 and os.path.exists(os.path.join(ROOT, fr["traffic_source"]))


def test_committed_force_dist_line_is_one_json_line_with_both_partitions():
    """the N > 1 code path run with one rank on an MI355X box (profiles/r03_force_dist_1rank.json): stdout is the ONE JSON
    line (RCCL's banner goes to stderr), it carries the row-block model step and the catchment partition, and the two
    partitions' discharge sums agree"""
    text = open(os.path.join(ROOT, "profiles", "r03_force_dist_1rank.json")).read().strip().splitlines()
    assert len(text) == 1
    d = json.loads(text[0])
    assert d["n_gpus"] == 1 and d["finite"] and d["scaling"] == "strong"
    assert d["model_step_24_substeps_split_row_blocks"]["finite"]
    assert d["catchment_partition"]["finite"]
    assert d["row_block_vs_catchment_partition_sumQ_rel_diff"] < 1e-12


def test_committed_compact_line_round5():
    """Since round 5 stdout carries a COMPACT line (the driver keeps only the tail of stdout, and the 8 KB line of round 4
    lost its first legs there) and everything else goes to the sidecar bench_detail.json: the committed pair of the last
    full run (profiles/r05_bench_line.json + profiles/r05_bench_detail.json)."""
    text = open(os.path.join(ROOT, "profiles", "r05_bench_line.json")).read().strip().splitlines()
    assert len(text) == 1 and len(text[0]) <= 4096
    d = json.loads(text[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "legs", "detail"):
        assert k in d, k
    assert d["dtype"] == "f64" and d["vs_baseline"] is None and "workload" in d["config"]
    assert abs(d["value"] - d["config"]["cells"] / d["ms_per_step"] / 1e3) < 1e-3 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-6
    assert abs(r["achieved"] - r["alg_bytes_per_launch"] / (r["mean_launch_us"] * 1e-6) / 1e9) < 1e-3 * r["achieved"]
    assert r["launches_per_step"] * r["mean_launch_us"] * 1e-3 <= d["ms_per_step"] * 1.02
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["one_core"] and c["all_cores_value"] and c["all_cores"] >= c["cores"]
    legs = d["legs"]
    for k in ("route_deep", "route_river", "soil_wet", "soil_single_substep", "model_step_deep_5000", "hot_path_deep_5000",
              "hot_path_river_5000"):
        assert k in legs and legs[k]["ms"] > 0, k
    assert list(legs)[:2] == ["route_deep", "route_river"]                # the latency-bound legs first
    assert legs["soil_wet"]["frac"] >= 0.31 and legs["soil_wet"]["traffic_ratio"] <= 1.5      # round-4 review, item 1
    assert legs["soil_single_substep"]["frac"] >= 0.55
    st = legs["hot_path_deep_5000"]["stages"]
    assert set(st) >= {"canopy", "soil_columns", "pixel_aggregates", "overland", "channel_wavefront"}
    full = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_detail.json")))
    assert full["value"] == d["value"] and "other_workloads" in full and "sample" in full["cpu_baseline"]


def test_committed_force_dist_line_round5_ran_on_the_real_rccl():
    text = open(os.path.join(ROOT, "profiles", "r05_force_dist_1rank.json")).read().strip().splitlines()
    assert len(text) == 1
    d = json.loads(text[0])
    assert d["rccl_library"].startswith("/opt/rocm") and d["finite"]
    assert d["model_step_24_substeps_split_row_blocks"]["finite"] and d["catchment_partition"]["finite"]
    assert d["row_block_vs_catchment_partition_sumQ_rel_diff"] < 1e-12


def test_compact_line_is_a_function_of_the_sidecar():
    """bench.compact_line(detail) reproduces the committed stdout line from the committed sidecar: nothing in the line that
    is not in the detail file, nothing hand-edited.  Round 6 changed the short form of `cpu_baseline` (physical cores, team
    rates, soil / model-step / LF_ETRS89 CPU figures): the round-6 pair is compared whole, the round-5 pair without it."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    for tag, whole in (("r05", False), ("r06", True)):
        path = os.path.join(ROOT, "profiles", tag + "_bench_detail.json")
        if not os.path.exists(path):
            assert tag == "r06"
            continue
        full = json.load(open(path))
        line = json.loads(open(os.path.join(ROOT, "profiles", tag + "_bench_line.json")).read().strip())
        again = json.loads(json.dumps(bench.compact_line(full, line["detail"])))
        if not whole:
            again.pop("cpu_baseline"); line.pop("cpu_baseline")
            for k in set(again) - set(line):                          # top-level keys added since (stage_bound)
                again.pop(k)
            for k in set(again["legs"]) - set(line["legs"]):          # legs added since
                again["legs"].pop(k)
            for leg in again["legs"]:                                  # keys added to old legs since
                if isinstance(again["legs"][leg], dict):
                    again["legs"][leg] = {k: x for k, x in again["legs"][leg].items() if k in line["legs"][leg]}
            again["legs_keys"] = line["legs_keys"]
        assert again == line, tag


def test_committed_compact_line_round6():
    """Round 6: the CPU baseline is a bound, first-touch team sweep with soil / model-step / LF_ETRS89 figures beside it,
    every leg names what bounds it, the land surface is one stage, the real catchment has a leg."""
    text = open(os.path.join(ROOT, "profiles", "r06_bench_line.json")).read().strip().splitlines()
    assert len(text) == 1 and len(text[0]) <= 4096
    d = json.loads(text[0])
    assert d["dtype"] == "f64" and d["vs_baseline"] is None and d["n_gpus"] == 1 and "workload" in d["config"]
    assert abs(d["value"] - d["config"]["cells"] / d["ms_per_step"] / 1e3) < 1e-3 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-6 and r["traffic"] >= r["alg_bytes_per_launch"]
    assert r["launches_per_step"] * r["mean_launch_us"] * 1e-3 <= d["ms_per_step"] * 1.02
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == d["unit"] and c["physical_cores"] >= c["cores"] >= 1
    teams = {int(k): x for k, x in c["team_rates"].items()}
    assert 1 in teams and c["physical_cores"] in teams and max(teams.values()) >= c["value"] * 0.5
    assert c["soil"]["unit"] == "Mcolumn-steps/s" and c["model_step"]["unit"] == "Mpixel-steps/s" and c["etrs89"]["ms_per_model_step"] > 0
    legs = d["legs"]
    for k, x in legs.items():
        if k not in ("pixel_order_call", "errors") and not k.startswith("hot_path"):
            assert x.get("bound") in ("hbm", "level-latency", "launch-latency", "valu"), k
    assert legs["route_deep"]["bound"] == "level-latency" and legs["route_deep"]["us_per_level"] < 0.5
    assert legs["etrs89_chain"]["dis_dev"] < 1e-6 and legs["etrs89_chain"]["cpu_ms"] > legs["etrs89_chain"]["ms"]
    st = legs["hot_path_deep_5000"]["stages"]
    assert "land_surface" in st and "canopy" not in st and st["land_surface"][0] < 14.8     # canopy 2.9 + soil 11.9 in round 5
    # (round 5: 23.6-24.6 ms; round 6: 22.4-23.7 depending on the box the line was taken on -- the boxes differ by ~5 %)
    assert legs["hot_path_deep_5000"]["ms"] < 24.6 and legs["hot_path_deep_5000"]["ms_unreported_maps_left_out"] < 20.6
