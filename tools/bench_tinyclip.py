"""BASELINE config 5 on one device (bench.py's tinyclip_config5 leg alone): TinyCLIP distillation step, pairs/s."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import bench

print(json.dumps(bench.tinyclip_config5_leg(batch=int(sys.argv[1]) if len(sys.argv) > 1 else 256)))
