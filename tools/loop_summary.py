import sys,json
for ln in sys.stdin:
    try: d=json.loads(ln)
    except Exception: continue
    print({k:d.get(k) for k in ('ms_per_scan_median','hz','ms_per_scan_mean','end_to_end_hz','pipeline_hz_steady_state','prefetch_thread','sweeps_in_pinned_host_memory','producer_ms_median','mapping_thread_waits_for_producer_ms_median','mapper_stopwatches_ms_median')})
