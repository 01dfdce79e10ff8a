"""End-to-end wall time of `python -m careless_amd mono` on a large synthetic unmerged MTZ, by stage (cProfile, cumulative):
formatting (host), upload + training (GPU), output step (host + GPU).  Usage: e2e_profile.py N_ROWS ITERATIONS [extra CLI args]"""
import cProfile, os, pstats, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
n, its = int(sys.argv[1]), sys.argv[2]
tmp = os.environ.get("E2E_TMP", "/tmp/e2e"); os.makedirs(tmp, exist_ok=True)
mode = os.environ.get("E2E_MODE", "mono")            # poly: a synthetic Laue file (scripts/gen_big_laue_mtz.py), metadata keys BATCH,X,Y,Wavelength
mtz = os.path.join(tmp, f"big_{mode}_{n}.mtz")
if not os.path.exists(mtz):
    subprocess.check_call([sys.executable, os.path.join(os.path.dirname(__file__), "gen_big_mtz.py" if mode == "mono" else "gen_big_laue_mtz.py"), str(n), mtz])
from careless_amd.parser import parser
from careless_amd import careless
args = parser.parse_args([mode, "--iterations", its, "--disable-progress-bar"] + sys.argv[3:] +
                         ["BATCH,XDET,YDET" if mode == "mono" else "BATCH,X,Y,Wavelength", mtz, os.path.join(tmp, "out")])
pr = cProfile.Profile(); t = time.time(); pr.enable()
careless.run_careless(args)
pr.disable(); print("TOTAL wall s", round(time.time() - t, 2))
st = pstats.Stats(pr); st.sort_stats("cumtime")
want = ("_format", "format_files", "train_model", "output_step", "get_results", "_prediction_tables", "get_predictions", "write_table_mtz",
        "build_model", "read_mtz", "to_asu", "describe", "_ngroup", "results_tables", "_build_obs", "__init__")
for (f, l, name), (cc, nc, tt, ct, callers) in sorted(st.stats.items(), key=lambda kv: -kv[1][3]):
    if ct > 0.3 and ("careless_amd" in f or "numpy" in f):
        print(f"{ct:8.2f}s cum {tt:8.2f}s self  {os.path.basename(f)}:{l} {name}")
