# which leg pays: two ReaxFF legs in one process, then (second process) a leg whose args come from the OPLS defaults
import sys, json, copy, argparse
sys.argv = ["bench.py", "--force-field", "reax", "--steps", "4", "--warmup", "2", "--no-cpu-baseline"]
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, torch, torch.distributed as dist
ap_main = bench.main
# re-create the args as main() does: reuse main's parser by monkeypatching run_leg to capture args
cap = {}
orig = bench.run_leg
def grab(args, *a, **k):
    cap['args'] = args; cap['rest'] = a
    return orig(args, *a, **k)
bench.run_leg = grab
import io, contextlib
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
d1 = json.loads([l for l in buf.getvalue().splitlines() if l.startswith('{')][-1])
print("first leg (native reax args):", round(d1['value'], 1), flush=True)
a2 = copy.copy(cap['args'])
r2 = orig(a2, *cap['rest'])
print("second leg, same args, same process:", round(r2['value'], 1), flush=True)
