"""Dev script: bench.py's `cli` side leg on its own (no bench process holding a GPU context beside it)."""
import json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import importlib.util
spec = importlib.util.spec_from_file_location("bench_mod", ROOT / "bench.py")
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
from deepemia_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
sd = synth.random_d2_state_dict(101, 2, seed=0)
print(json.dumps(bench.cli_leg(sd, 101, "cuda:0", n, 700.0), indent=1))
