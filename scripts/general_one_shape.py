"""One shape of scripts/general_shapes_rate.py (for a rocprofv3 pass): python scripts/general_one_shape.py 512,512 512,512 256 300"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from general_shapes_rate import rate  # noqa: E402

hp = tuple(int(x) for x in sys.argv[1].split(","))
hq = tuple(int(x) for x in sys.argv[2].split(","))
print(json.dumps(rate(hp, hq, 42, 7, int(sys.argv[3]), int(sys.argv[4]), os.environ.get("SAC_GENERAL") == "1")))
