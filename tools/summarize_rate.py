"""Condense the JSON lines of tools/svc_rate.py in a log (stdin) to one line each."""
import json, sys
for ln in sys.stdin:
    ln = ln.rstrip()
    if ln.startswith('{'):
        try:
            d = json.loads(ln)
        except Exception:
            print('   ', ln[:300]); continue
        if 'frames_per_s' not in d:
            print('   ', ln[:300]); continue
        ss = d.get('search_service') or {}
        sm = d.get('stage_ms') or {}
        print('   %.0f f/s  ms/pass %.3f  eq %s slow %s | cyc/frame %s help/frame %s busy %s launches %s | stage_ms %s' % (
            d['frames_per_s'], d['ms_per_pass'], d['slots_equal_plain_run'], d.get('slow_submits'), ss.get('cycles_per_frame') and round(ss['cycles_per_frame']),
            ss.get('help_cycles_per_frame') and round(ss['help_cycles_per_frame']), ss.get('busy_fraction') and round(ss['busy_fraction'], 2), ss.get('launches'),
            {k: round(v, 3) for k, v in sm.items()}))
        if d.get('scan_profile_cycles_per_frame'):
            print('      scan:', {k: round(v) for k, v in d['scan_profile_cycles_per_frame'].items()})
    elif not ln.startswith(('_ZN', 'k_map', '[gpurun] merged')):
        print(ln[:300])
