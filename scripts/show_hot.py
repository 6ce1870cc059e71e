import sys, json
for ln in sys.stdin:
    if not ln.startswith("{"): continue
    d = json.loads(ln)
    print("cfg %d %-28s loop %.4f ms  ev-med %.4f  min %.4f | sync'd %.4f | two-ctx %.4f | %.3e f/s  handed %s" % (d["cfg"], d.get("knob", d["lib"][-24:]), d["ms_per_call_loop"], d["ms_per_call_events_median"], d["ms_per_call_events_min"], d["ms_single_synchronised_call"], d["two_streams_ms_per_call"], d["frames"] / d["ms_per_call_events_median"] * 1e3, d["handed"]))
