"""Can two ranks share the single GPU of a box under RCCL? (the trace of collectives beside backward GEMMs needs it)
python tools/rccl_two_ranks_one_gpu.py"""
import os, sys, subprocess, socket
if 'RANK' not in os.environ:
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK='0', WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)], env=env))
    rc = [p.wait(timeout=240) for p in procs]
    print('exit codes', rc)
    sys.exit(max(rc))
import torch, torch.distributed as dist
torch.cuda.set_device(0)
try:
    dist.init_process_group('nccl', rank=int(os.environ['RANK']), world_size=2, device_id=torch.device('cuda:0'))
    x = torch.full((1 << 20,), float(int(os.environ['RANK']) + 1), device='cuda:0')
    dist.all_reduce(x)
    torch.cuda.synchronize()
    print('rank', os.environ['RANK'], 'all_reduce ->', x[0].item(), flush=True)
    dist.destroy_process_group()
except Exception as e:
    print('rank', os.environ['RANK'], 'FAILED:', type(e).__name__, str(e)[:400], flush=True)
    sys.exit(3)
