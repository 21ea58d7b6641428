"""Scratch: which environment variables rocprofv3 hands to the profiled program."""
import os
for k in sorted(os.environ):
    if "ROCP" in k.upper() or "PMC" in k.upper() or "COUNTER" in k.upper():
        print(k, "=", os.environ[k][:100])
