import json
import sys

d = json.loads(sys.stdin.read())
print("   ms/step %.4f   host enqueue ms/step %.4f" % (d["ms_per_step"], d.get("host_enqueue_ms_per_step", 0.0)))
