import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(sys.argv[1] if len(sys.argv) > 1 else "", round(d["value"]), "sub/s", round(d["ms_per_step"], 3), "ms/step  frac", round(d["roofline"]["frac"], 4),
      "avg_launch_ms", round(d["roofline"]["avg_launch_ms"], 3), {k: round(v, 3) for k, v in d["kernel_ms_per_step"].items()})
