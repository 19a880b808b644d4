"""stdin: bench.py's output; prints `<name> <kind> us/step X frac Y` from the last JSON line (A/B helper)."""
import json, sys
name, kind = (sys.argv + ["", ""])[1:3]
lines = [l for l in sys.stdin.read().splitlines() if l.startswith("{")]
if not lines:
    print("%-24s %-6s no result" % (name, kind))
else:
    d = json.loads(lines[-1])
    r = d.get("roofline") or {}
    print("%-24s %-6s us/step %.2f  frac %.3f" % (name, kind, d["ms_per_step"] * 1e3, r.get("frac") or 0.0))
