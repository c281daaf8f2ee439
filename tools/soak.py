"""Soak: N bench steps in one process (default 3000 at 192 chunks, ~1 min), then the allocator's high-water marks and a second, equally long
leg - throughput and memory must not drift between the legs.  python tools/soak.py [steps]"""
import io, json, os, sys, contextlib
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
import bench
steps = sys.argv[1] if len(sys.argv) > 1 else '3000'
for leg in (1, 2):
    sys.argv = ['bench.py', '--steps', steps, '--warmup', '10', '--no-cpu-baseline', '--no-kernel-timing']
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.main()
    line = json.loads(buf.getvalue().strip().splitlines()[-1])
    print(f"leg {leg}: {steps} steps, {line['value']} clips/s, {line['ms_per_step']} ms/step; max allocated {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB, "
          f"reserved {torch.cuda.memory_reserved() / 2**30:.2f} GiB")
