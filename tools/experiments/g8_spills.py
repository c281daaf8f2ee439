"""Where do the register spills of a gemm8 instantiation sit? Disassembles gemm8.hip (device only) and counts scratch loads / stores inside
the K loop (the span of the v_mfma instructions) and outside it (prologue + epilogue), per instantiation matching argv[1] (default: the LayerNorm
epilogue modes). A reload inside the K loop comes with a vmcnt(0) that drains the LDS-DMA prefetch - the thing to avoid.
python tools/experiments/g8_spills.py ['ILi4E|ILi5E']"""
import os, re, subprocess, sys, tempfile
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pat = re.compile(sys.argv[1] if len(sys.argv) > 1 else r'gemm8_kernelILi[45]E')
out = os.path.join(tempfile.gettempdir(), 'g8_spills.s')
subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '--cuda-device-only', '-S', os.path.join(root, 'pseldnets_amd/csrc/gemm8.hip'), '-o', out])
lines = open(out).read().split('\n')
for i0, l in enumerate(lines):
    m = re.match(r'(_ZN\S*gemm8_kernel\S*):', l)
    if not m or not pat.search(m.group(1)): continue
    i1 = next(i for i in range(i0, len(lines)) if lines[i].startswith('.Lfunc_end'))
    body = lines[i0:i1]
    mf = [i for i, x in enumerate(body) if 'v_mfma' in x]
    sc = [i for i, x in enumerate(body) if 'scratch_' in x]
    lo = max(i for i, x in enumerate(body) if x.startswith('.LBB') and i < mf[0])
    hi = min(i for i, x in enumerate(body) if 's_cbranch' in x and i > mf[-1])
    print(f"{m.group(1)[20:60]}: {len(body)} lines, K loop {lo}-{hi}; scratch ops: {sum(1 for i in sc if lo <= i <= hi)} in the K loop, "
          f"{sum(1 for i in sc if i < lo)} before, {sum(1 for i in sc if i > hi)} after (epilogue)")
