# bench.py with N idle hardware queues (CU-masked streams) alive in the process: what `--sequence auto` does when the device's
# hardware-queue slots are short.  usage: python profiles/tools/bench_with_extra_queues.py N [bench.py flags]
import sys, ctypes as C, runpy, json, io, contextlib
sys.path[:0] = ['/root/repo', '/root/repo/ship-track-estimators_amd']
import torch
from track_estimators._hip import binding
n = int(sys.argv[1])
lib = binding.load()
dev = torch.device('cuda:0')
keep = []
for _ in range(n):
    h = C.c_void_p()
    binding.check(lib.ste_stream_create_cu_range(0, 256, C.byref(h)), "create")
    s = torch.cuda.ExternalStream(h.value, device=dev)
    with torch.cuda.stream(s):
        torch.zeros(16, device=dev).add_(1)
    keep.append((h, s))
torch.cuda.synchronize()
sys.argv = ['bench.py'] + sys.argv[2:]
buf = io.StringIO()
try:
    with contextlib.redirect_stdout(buf):
        runpy.run_path('/root/repo/bench.py', run_name='__main__')
except SystemExit:
    pass
j = json.loads([l for l in buf.getvalue().splitlines() if l.startswith('{')][-1])
a = j['config']['sequence_auto']
print(f"extra queues {n}: {j['ms_per_step']:.3f} ms per step, chosen {a['chosen']} on {a['streams']}: scheduled {a['scheduled_launches_ms']} per-step {a['per_step_launches_ms']:.2f} ms", flush=True)
