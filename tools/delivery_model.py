"""The delivered-bytes law of the tile loops against the isolated tables of the evidence run: for every forward / input-
gradient product of stages 2-3 (tools/gemm8_check.py shapes, COLD=1) and every weight gradient (tools/wgrad8_check.py shapes) the bytes a
launch DELIVERS to its workgroups - every tile stages its A rows and its B rows for the whole K, whatever cache they come from - plus the
epilogue's operand loads and the stores, divided by the measured time.  python tools/delivery_model.py profiles/r06_gemm8_shapes_cold.log
profiles/r06_wgrad8_shapes.log"""
import re, sys, math


def cdiv(a, b): return (a + b - 1) // b


rows = []
for line in open(sys.argv[1]):
    m = re.match(r"(s\d) (\S+ \S+)\s+M=\s*(\d+) K=\s*(\d+) N=\s*(\d+) (\w+)\s*: r256c256\s+([\d.]+) \| r256c192\s+([\d.]+)", line)
    if not m: continue
    st, name, M, K, N, mode, t256, t192 = m.group(1), m.group(2), int(m.group(3)), int(m.group(4)), int(m.group(5)), m.group(6), float(m.group(7)), float(m.group(8))
    for bn, t in ((256, t256), (192, t192)):
        if bn == 192 and N % 192: continue
        tiles = cdiv(M, 256) * cdiv(N, bn)
        staged = tiles * (256 + bn) * K * 2
        epi = M * N * 2 if mode in ('resid', 'mulaux') else 0
        stores = M * N * 2 * (2 if mode == 'gelu' else 1)
        alg = 2 * (M * K + N * K) + epi + stores
        rows.append((f"{st} {name}", f"{M}x{N}x{K} {mode}", f"256x{bn}", t, staged, epi + stores, alg))
for line in open(sys.argv[2]):
    m = re.match(r"(s\d) (\S+)\s+wgrad dW\[\s*(\d+),\s*(\d+)\] over\s+(\d+) tokens: bn256\s+([\d.]+) us.*?\| bn192\s+([\d.]+) us", line)
    if not m: continue
    st, name, N, K, T, t256, t192 = m.group(1), m.group(2), int(m.group(3)), int(m.group(4)), int(m.group(5)), float(m.group(6)), float(m.group(7))
    for bn, t in ((256, t256), (192, t192)):
        tiles = cdiv(N, 256) * cdiv(K, bn)
        staged = tiles * (256 + bn) * 2 * T
        slabs = 256 * 256 * bn * 4                      # one resident round of 256 workgroups, one fp32 tile each (written, then read by the reduction)
        alg = 2 * T * (N + K) + 4 * N * K
        rows.append((f"{st} {name} wgrad", f"dW[{N},{K}] / {T} tok", f"256x{bn}", t, staged, 2 * slabs, alg))
print(f"{'product':22s} {'shape':28s} {'tile':8s} {'us':>7s} {'staged MB':>10s} {'+ epilogue/stores':>18s} {'delivered TB/s':>15s} {'algorithmic TB/s':>17s}")
rates = []
for name, shape, tile, t, staged, other, alg in rows:
    r = (staged + other) / t / 1e6
    rates.append((r, staged / (staged + other)))
    print(f"{name:22s} {shape:28s} {tile:8s} {t:7.1f} {staged / 1e6:10.0f} {other / 1e6:18.0f} {r:15.2f} {alg / t / 1e6:17.2f}")
big = sorted(r for r, f in rates)
print(f"\n{len(rows)} launches: delivered rate min / median / max = {big[0]:.2f} / {big[len(big) // 2]:.2f} / {big[-1]:.2f} TB/s "
      f"(algorithmic - what the HBM roofline is priced on - is the last column: the same launches at 1.7-3.7 TB/s)")
