"""Rows of DESIGN.md section 6's round-4 table from profiles/r4_bench_W.json, profiles/r4_kernel_stats_W.csv and profiles/traffic.json."""
import csv, json
ALL = ["mono_1M_normal_5x64_S1", "mono_10M_studentt_posenc_5x64_S8", "laue_5M_normal_5x64_S1", "dw_50M_normal_5x64_S1", "mono_10M_cli_default_20x10_S1",
       "mono_10M_studentt_posenc_20x10_S8", "mono_10M_studentt_posenc_4x64_img1_S8", "mono_2M_studentt_3x128_S4"]
tr = json.load(open("profiles/traffic.json"))
for w in ALL:
    d = json.loads(open(f"profiles/r4_bench_{w}.json").read().strip().splitlines()[-1]); r = d["roofline"]
    rows = list(csv.DictReader(open(f"profiles/r4_kernel_stats_{w}.csv")))
    dom = [x for x in rows if any(k in x["Name"] for k in ("elbo_mlp_kernel", "elbo_lane_kernel", "elbo_narrow", "wide_"))]
    per_step = sum(float(x["TotalDurationNs"]) for x in dom) / 13 / 1e6
    t = tr.get(w, {})
    print(f"| `{w}` | {d['value']:.3g} | {d['ms_per_step']:.3f} | {r['kernel_ms']:.3f} ({per_step:.3f}) | {r['frac']:.3f} ({r['frac_on_step_time']:.3f}) | "
          f"{t.get('hbm_bytes_per_launch', 0) / 1e9:.2f} GB of HBM traffic per step | {r.get('kernel', '')[:70]}")
