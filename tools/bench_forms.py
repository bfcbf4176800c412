"""bench.py's per-form launch times only (no CPU baseline, no verification): A/B of library variants.
usage: [MGX_LIB_PATH=...] python tools/bench_forms.py [bench.py arguments]"""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", "--no-verify", "--no-secondary"] + sys.argv[1:],
                     capture_output=True, cwd=root)
d = json.loads(out.stdout.decode().strip().splitlines()[-1])
forms = {"kCheb": d["roofline"], "kPlain": d["roofline_matvec"], **d["roofline_forms"]}
print("step %.3f ms  matvec %.3f  V-cycle %.3f | " % (d["ms_per_step"], d["matvec_ms"], d["vcycle_ms"]) +
      "  ".join("%s %.1f us (%.3f)" % (k, 1e3 * v["avg_launch_ms"], v["frac"]) for k, v in forms.items() if v))
