#!/bin/bash
# One GPU runs, one after the other, every rank's strip of an N-rank job (bench.py --emulate-strip r/N: redundant halo
# rows, camera at rest => no communication is needed, so the per-rank time IS what that rank would take), serial and with
# two frames in flight.  Strong-scaling expectation = full frame / slowest strip.  -> gpurun_out/<tag>/emulated_strips.json
TAG=${1:-emu}; OUT=gpurun_out/$TAG; mkdir -p $OUT
python - $OUT "${COMMIT:-unknown}" <<'PY'
import json, subprocess, sys
out, commit = sys.argv[1], sys.argv[2]
def run(extra):
    r = subprocess.run([sys.executable, "bench.py", "--workload", "4k", "--steps", "200", "--warmup", "20", "--no-cpu-baseline", "--no-secondary"] + extra,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-1500:]
    return json.loads(r.stdout.strip().splitlines()[-1])
res = {"_source": commit, "_note": "3840x2160, 4 segments, N = 5, redundant halo; ms per frame of ONE rank's strip measured alone on one MI355X "
       "(bench.py --emulate-strip r/N --steps 200); speed-up = full frame / slowest strip of the job", "full_frame": {}, "strips": {}}
for fl in (1, 2):
    d = run(["--frames-in-flight", str(fl)])
    res["full_frame"][f"in_flight_{fl}"] = d["ms_per_step"]
    print("full", fl, d["ms_per_step"], flush=True)
for n in (2, 4, 8):
    for fl in (1, 2):
        ms = []
        for r in range(n):
            d = run(["--emulate-strip", f"{r}/{n}", "--frames-in-flight", str(fl)])
            ms.append(d["ms_per_step"])
            print(n, fl, r, d["ms_per_step"], flush=True)
        res["strips"][f"{n}_ranks_in_flight_{fl}"] = {"ms_per_strip": ms, "slowest": max(ms),
                                                       "speedup_vs_full_frame_same_mode": round(res["full_frame"][f"in_flight_{fl}"] / max(ms), 3),
                                                       "speedup_vs_serial_full_frame": round(res["full_frame"]["in_flight_1"] / max(ms), 3)}
json.dump(res, open(out + "/emulated_strips.json", "w"), indent=1)
print(json.dumps({k: (v["slowest"], v["speedup_vs_full_frame_same_mode"]) for k, v in res["strips"].items()}))
PY
