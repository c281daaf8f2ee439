import os, sys, torch
sys.path.insert(0, '/root/repo')
os.chdir('/root/repo')
import bench
sys.argv = ['bench.py', '--steps', '400', '--warmup', '10', '--no-cpu-baseline', '--no-kernel-timing']
bench.main()
print('max allocated GB', torch.cuda.max_memory_allocated() / 2**30, 'reserved GB', torch.cuda.memory_reserved() / 2**30)
