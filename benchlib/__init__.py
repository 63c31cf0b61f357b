"""bench.py's workloads, one module each (bench.py itself keeps the driver contract: arguments, rank start-up, the timed region, the JSON line).
Nothing here is product code; like bench.py it drives the C ABI through aom-av1-psy_amd/capi.py and uses oracle/ only as the checker."""
