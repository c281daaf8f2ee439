"""Reduce two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; counter_collection.csv each) of `bench.py --steps 1 --warmup 1
--no-cpu-baseline --no-kernel-timing` to HBM bytes per launch of the forward / input-gradient GEMM family
(gemm_dma_kernel<..> and gemm_kernel<.., TA = false, ..>), with the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md.
python tools/pmc_reduce.py fetch.csv write.csv out.json"""
import csv, json, re, sys


def family(name):
    if 'gemm_dma_kernel' in name:
        return True
    m = re.search(r'gemm_kernelI\w+?Li\d+ELi\d+ELb([01])E', name)          # mangled (bf16 names do not demangle): TA flag
    if m:
        return m.group(1) == '0'
    m = re.search(r'gemm_kernel<([^>]*)>', name)
    if not m:
        return False
    args = [a.strip() for a in m.group(1).split(',')]
    return len(args) >= 4 and 'true' not in args[-3:-1] and args[-2] in ('false', '0') if len(args) == 5 else \
        (len(args) >= 5 and args[4] in ('false', '0'))               # TA = false: forward / input gradient


def total(path, counter):
    s, n = 0.0, 0
    with open(path, newline='') as f:
        for row in csv.DictReader(f):
            if row['Counter_Name'] == counter and family(row['Kernel_Name']):
                s += float(row['Counter_Value']); n += 1
    return s, n


fetch, nf = total(sys.argv[1], 'FETCH_SIZE')
write, nw = total(sys.argv[2], 'WRITE_SIZE')
assert nf == nw and nf > 0, (nf, nw)
rd = fetch * 1024 * 2 / nf
wr = write * 1024 / nw
out = {"kernel": "pseld_gemm fwd/dgrad family (gemm_dma_kernel + gemm_kernel<TA=false>)", "launches_counted": nf,
       "fetch_size_kb_sum": fetch, "write_size_kb_sum": write, "read_bytes_per_launch_corrected": rd, "write_bytes_per_launch": wr,
       "traffic_bytes_per_launch": rd + wr,
       "correction": "FETCH_SIZE x2 (gfx950: 128-B requests tallied at 64 B for 16-B/lane streaming reads, MI355X_MICROARCH.md HBM section); WRITE_SIZE exact",
       "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing (two separate passes)"}
json.dump(out, open(sys.argv[3], 'w'), indent=1)
print(json.dumps(out))
