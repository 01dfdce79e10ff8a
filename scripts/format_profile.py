"""cProfile of the formatting step alone (host code) on the synthetic unmerged MTZ of scripts/gen_big_mtz.py: format_profile.py N_ROWS"""
import cProfile, os, pstats, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
n = int(sys.argv[1])
tmp = os.environ.get("E2E_TMP", "/tmp/e2e"); os.makedirs(tmp, exist_ok=True)
mtz = os.path.join(tmp, f"big_mono_{n}.mtz")
if not os.path.exists(mtz):
    subprocess.check_call([sys.executable, os.path.join(os.path.dirname(__file__), "gen_big_mtz.py"), str(n), mtz])
from careless_amd._lib import get_lib; get_lib()
from careless_amd.parser import parser
from careless_amd import careless
args = parser.parse_args(["mono", "BATCH,XDET,YDET", mtz, os.path.join(tmp, "out")])
careless._format(args)
pr = cProfile.Profile(); t = time.time(); pr.enable()
careless._format(args)
pr.disable(); print("format s", round(time.time() - t, 3))
pstats.Stats(pr).sort_stats("tottime").print_stats(16)
