"""One line per JSON-line result file of tools/bench_samples.py / tools/svc_rate.py / bench.py (usage: summ.py files...)."""
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().split("\n")[-1])
        ss = d.get("search_service") or d.get("config", {}).get("search_service") or {}
        keep = {k: (round(v) if isinstance(v, float) and v > 10 else (round(v, 3) if isinstance(v, float) else v)) for k, v in ss.items()
                if k in ("busy_fraction", "cycles_per_frame", "help_cycles_per_frame", "launches", "remote_help", "mode", "timeline_ms")}
        print("%-44s %8.0f  ok=%s  %s" % (f.split("/")[-1], d.get("value", d.get("frames_per_s")), d.get("records_equal_oracle", d.get("slots_equal_plain_run", d.get("slots_identical"))), keep))
    except Exception as e:  # noqa: BLE001
        print("%-44s ERR %s | %s" % (f.split("/")[-1], e, open(f).read()[-300:].replace("\n", " ")))
