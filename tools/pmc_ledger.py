"""Byte ledger of the training step: rocprofv3's in-order tables of `PSELD_STAGE_MARKERS=1 PSELD_WGRAD_STREAM=0 PSELD_FEATURE_PREFETCH=0
python3 bench.py ...` cut at the stage markers (as tools/pmc_stages.py) and, inside every stage, summed PER KERNEL SYMBOL: launches, kernel
time, bytes read (FETCH_SIZE x 2: MI355X_MICROARCH.md's gfx950 correction) and written (WRITE_SIZE) per step, the rate, and - for the
layer-wise stages of HTS-AT (1-3) - the traffic expressed in ROW TENSORS of the stage ([tokens, C] bf16: 75.5 / 37.7 / 18.9 MB at 192
chunks) per block next to the algorithmic count of the layers that symbol runs ("Byte ledger"): the difference is
what a kernel re-reads or spills beyond what its layers must move.
python tools/pmc_ledger.py fetch_dir write_dir steps out.json [chunks=192]"""
import csv, glob, json, re, sys
sys.path.insert(0, __file__.rsplit('/', 2)[0])
from pseldnets_amd.ops import STAGES

fdir, wdir, steps, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
chunks = int(sys.argv[5]) if len(sys.argv) > 5 else 192

# algorithmic row tensors per block by kernel symbol (substring), layer-wise stages: what the layers a symbol runs must read + write once
# (row tensor = [tokens, C] bf16; a hidden / qkv tensor counts 4 / 3). Forward: LN 2 each; qkv 1 + 3; attention 3 + 1; proj 1 + 1 + 1;
# fc1 1 + 4 + 4 (GELU pair); fc2 4 + 1 + 1. Backward: dgrad fc2 1 + 4 + 4; dgrad fc1 4 + 1; LN backward 4 each; dgrad proj 1 + 1;
# attention backward 3 + 1 + 1 + 3; dgrad qkv 3 + 1; weight gradients: fc2 1 + 4, fc1 4 + 1, proj 1 + 1, qkv 3 + 1 (their fp32 slabs
# and the reductions are NOT algorithmic: they are this build's split-K price and show up as the excess of the wgrad rows)
ALGO_RT = [('gemm8_kernel<0, false', 13, 'qkv fwd 4 + qkv dgrad 4 + fc1 dgrad 5'), ('gemm8_kernel<1, true', 9, 'proj fwd 3 + fc2 fwd 6'),
           ('gemm8_kernel<3, false', 9, 'fc1 fwd + GELU pair 9'), ('gemm8_kernel<2, true', 9, 'fc2 dgrad x gelu\' 9'),
           ('gemm8_kernel<0, true', 2, 'proj dgrad 2'), ('gemm8w_kernel', 16, 'weight gradients: fc2 5 + fc1 5 + proj 2 + qkv 4 (slabs extra)'),
           ('reduce_slabs', 0, 'slab reductions (split-K price: not algorithmic)'), ('ln_fwd_kernel', 4, 'norm1 + norm2 forward 2 + 2'),
           ('ln_bwd_kernel', 8, 'norm2 + norm1 backward 4 + 4'), ('attn_fwd24p_kernel', 4, 'window attention forward 3 + 1'),
           ('attn_bwd24_kernel', 8, 'window attention backward 8')]
DEPTH = {'stage1': 2, 'stage2': 6, 'stage3': 2}
ROW_MB = {'stage1': chunks * 1024 * 192 * 2 / 1e6, 'stage2': chunks * 256 * 384 * 2 / 1e6, 'stage3': chunks * 64 * 768 * 2 / 1e6}


def rows(d, pat):
    r = []
    for path in glob.glob(d + '/' + pat):
        with open(path, newline='') as f:
            r += list(csv.DictReader(f))
    return r


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    name = re.sub(r'\(.*$', '', name)                      # drop the argument list
    m = re.match(r'_ZN12_GLOBAL__N_1(\d+)([a-zA-Z0-9_]+)', name)
    if m:
        n = int(m.group(1)); name = (m.group(2)[:n] + ' [' + m.group(2)[n:n + 40] + ']')
    return name.strip()


def family(name):
    """Tile-shape variants of one kernel are one ledger row: gemm8_kernel<MODE, SCALED, *>, gemm8w_kernel<*>, ln_fwd / ln_bwd."""
    m = re.match(r'gemm8_kernel<(\d), (true|false),', name)
    if m:
        return f'gemm8_kernel<{m.group(1)}, {m.group(2)}, *>'
    if name.startswith('gemm8w_kernel<'):
        return 'gemm8w_kernel<*>'
    for k in ('ln_fwd_kernel', 'ln_bwd_kernel', 'reduce_slabs'):
        if name.startswith(k):
            return k + ('*' if k == 'reduce_slabs' else '')
    return name


def staged(rs):
    rs.sort(key=lambda r: int(r['Dispatch_Id']))
    cur = 'other'
    for r in rs:
        m = re.search(r'stage_marker_kernel<(\d+)>', r['Kernel_Name'])
        if m:
            t = int(m.group(1)); cur = STAGES[t] if t < len(STAGES) else 'other'
            continue
        yield cur, r


def counter(d, cname):
    acc = {}
    rs = [r for r in rows(d, '*counter_collection.csv') if r['Counter_Name'] == cname or 'stage_marker_kernel' in r['Kernel_Name']]
    for st, r in staged(rs):
        k = (st, family(short(r['Kernel_Name'])) if st in DEPTH else short(r['Kernel_Name']))
        acc[k] = acc.get(k, 0.0) + float(r['Counter_Value'])
    return acc


fetch, write = counter(fdir, 'FETCH_SIZE'), counter(wdir, 'WRITE_SIZE')
t, n = {}, {}
for st, r in staged(rows(wdir, '*kernel_trace.csv')):
    k = (st, family(short(r['Kernel_Name'])) if st in DEPTH else short(r['Kernel_Name']))
    t[k] = t.get(k, 0.0) + (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-6
    n[k] = n.get(k, 0) + 1

res = {"unit": "per training step; read = FETCH_SIZE x 2 (gfx950 correction), written = WRITE_SIZE; KiB counters", "chunks": chunks, "stages": {}}
for st in list(STAGES) + ['other']:
    ks = sorted([k for k in t if k[0] == st], key=lambda k: -t[k])
    if not ks:
        continue
    tab, tot_r, tot_w, tot_t = [], 0.0, 0.0, 0.0
    for k in ks:
        rd, wr = fetch.get(k, 0.0) * 2 * 1024 / 1e9 / steps, write.get(k, 0.0) * 1024 / 1e9 / steps
        ms = t[k] / steps
        row = {"kernel": k[1], "launches": round(n[k] / steps, 1), "ms": round(ms, 4), "read_gb": round(rd, 3), "written_gb": round(wr, 3),
               "tb_per_s": round((rd + wr) / ms, 2) if ms else None}
        if st in DEPTH:
            row["row_tensors_per_block"] = round((rd + wr) * 1e3 / ROW_MB[st] / DEPTH[st], 1)
            for pat, rt, what in ALGO_RT:
                if pat in k[1]:
                    row["algorithmic_row_tensors_per_block"], row["layers"] = rt, what
                    break
        tab.append(row)
        tot_r += rd; tot_w += wr; tot_t += ms
    entry = {"ms": round(tot_t, 3), "read_gb": round(tot_r, 3), "written_gb": round(tot_w, 3), "tb_per_s": round((tot_r + tot_w) / tot_t, 2), "kernels": tab}
    if st in DEPTH:
        entry["row_tensor_mb"] = round(ROW_MB[st], 1)
        entry["row_tensors_per_block"] = round((tot_r + tot_w) * 1e3 / ROW_MB[st] / DEPTH[st], 1)
        entry["algorithmic_row_tensors_per_block"] = sum(r.get("algorithmic_row_tensors_per_block", 0) for r in tab)
    res["stages"][st] = entry
json.dump(res, open(out, 'w'), indent=1)
for st, e in res["stages"].items():
    extra = f", {e['row_tensors_per_block']} row tensors per block (algorithmic {e['algorithmic_row_tensors_per_block']})" if 'row_tensors_per_block' in e else ''
    print(f"== {st}: {e['ms']} ms, read {e['read_gb']} GB, written {e['written_gb']} GB, {e['tb_per_s']} TB/s{extra}")
    for r in e["kernels"][:14]:
        rt = f" | {r['row_tensors_per_block']:6.1f} RT/block (algorithmic {r.get('algorithmic_row_tensors_per_block', '-')})" if 'row_tensors_per_block' in r else ''
        print(f"   {r['kernel'][:70]:70s} x{r['launches']:5.1f} {r['ms']:7.3f} ms  r {r['read_gb']:6.3f} w {r['written_gb']:6.3f} GB {r['tb_per_s']} TB/s{rt}")
