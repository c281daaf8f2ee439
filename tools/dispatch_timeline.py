"""Dispatch-by-dispatch account of the timed step (why do isolated kernel gains vanish in the two-stream step?).

Inputs: two `rocprofv3 --kernel-trace` CSVs of the SAME bench command on the SAME box,
  A = everything on one stream (PSELD_WGRAD_STREAM=0 PSELD_FEATURE_PREFETCH=0): a kernel's time with the chip to itself,
  B = the step as timed (weight gradients + next batch's features on their own streams).
The host issues its launches in the same order in both, so dispatch i of a step in A is dispatch i of a step in B.

For every dispatch of the median steady-state step of B: queue, start offset, duration in B, duration in A, which other-queue
kernels ran beside it (overlap in us), and the idle gap on its own queue in front of it.  Summary: the main queue's busy time and
gaps, how much the main chain is stretched by what runs beside it, how much side-stream work is actually hidden.

python tools/dispatch_timeline.py A_kernel_trace.csv B_kernel_trace.csv out.json [out.txt]"""
import csv, json, re, sys, statistics


def load(path):
    rows = list(csv.DictReader(open(path, newline='')))
    out = []
    rows.sort(key=lambda r: int(r['Dispatch_Id']))            # host launch order: a step is then the same run of dispatches in every step
    for r in rows:
        out.append(dict(name=r['Kernel_Name'], q=r['Queue_Id'], s=int(r['Start_Timestamp']), e=int(r['End_Timestamp']),
                        grid=int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])), vgpr=int(r['VGPR_Count']) + int(r['Accum_VGPR_Count']),
                        lds=int(r['LDS_Block_Size'])))
    return out


def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    n = re.sub(r'\(.*$', '', n)
    return n[:64]


def steps(rows):
    """Cut at the optimiser's last kernel (adamw): a step = the dispatches after one adamw up to and including the next."""
    ends = [i for i, r in enumerate(rows) if 'adamw' in r['name']]
    return [(a + 1, b + 1) for a, b in zip(ends, ends[1:])]


def main():
    A, B = load(sys.argv[1]), load(sys.argv[2])
    sa, sb = steps(A), steps(B)
    # steady state: skip the first 3 steps (warm-up), take the step with the median wall time
    def wall(rows, ab):
        return rows[ab[1] - 1]['e'] - rows[ab[0] - 1]['e']
    sa, sb = sa[3:], sb[3:]
    # steady-state steps only: the most common dispatch count (the last step of a run has no next batch to prefetch features for)
    import collections
    for cuts in (sa, sb):
        n = collections.Counter(b - a for a, b in cuts).most_common(1)[0][0]
        cuts[:] = [c for c in cuts if c[1] - c[0] == n]
    ia = sorted(range(len(sa)), key=lambda i: wall(A, sa[i]))[len(sa) // 2]
    ib = sorted(range(len(sb)), key=lambda i: wall(B, sb[i]))[len(sb) // 2]
    a = A[sa[ia][0]:sa[ia][1]]
    b = B[sb[ib][0]:sb[ib][1]]
    assert len(a) == len(b), (len(a), len(b))
    # the k-th launch of a symbol within a step of A is the k-th launch of that symbol within a step of B (the prefetched feature
    # launch sits elsewhere in the order, everything else is issued in the same order)
    def occ(rows):
        seen, out = {}, []
        for r in rows:
            k = short(r['name']); seen[k] = seen.get(k, 0) + 1; out.append((k, seen[k]))
        return out
    def per_occ(R, cuts):
        acc = {}
        for s0, s1 in cuts:
            for r, key in zip(R[s0:s1], occ(R[s0:s1])):
                acc.setdefault(key, []).append(r['e'] - r['s'])
        return {k: statistics.median(v) for k, v in acc.items()}
    medA, medB = per_occ(A, sa), per_occ(B, sb)
    keysB = occ(b)
    assert sorted(occ(a)) == sorted(keysB)
    durA = [medA[k] for k in keysB]
    durB_med = [medB[k] for k in keysB]
    t0 = B[sb[ib][0] - 1]['e']  # end of the previous step's adamw
    wallB = wall(B, sb[ib])
    wallA = wall(A, sa[ia])
    qs = sorted({r['q'] for r in b}, key=lambda q: -sum(1 for r in b if r['q'] == q))
    main_q = qs[0]
    order = sorted(range(len(b)), key=lambda i: b[i]['s'])
    last_end = {}
    table = []
    for i in order:
        r = b[i]
        gap = (r['s'] - last_end[r['q']]) if r['q'] in last_end else (r['s'] - t0 if r['q'] == main_q else 0)
        last_end[r['q']] = r['e']
        beside = {}
        for j, o in enumerate(b):
            if o['q'] == r['q']:
                continue
            lo, hi = max(r['s'], o['s']), min(r['e'], o['e'])
            if hi > lo:
                beside[short(o['name'])] = beside.get(short(o['name']), 0) + (hi - lo)
        table.append(dict(i=i, kernel=short(r['name']), queue=r['q'], start_us=(r['s'] - t0) / 1e3, us_in_step=(r['e'] - r['s']) / 1e3,
                          us_alone=durA[i] / 1e3, us_in_step_median=durB_med[i] / 1e3, gap_us=gap / 1e3, wgs=r['grid'], regs=r['vgpr'], lds=r['lds'],
                          beside_us={k: round(v / 1e3, 1) for k, v in beside.items()}))
    mainrows = [t for t in table if t['queue'] == main_q]
    side = [t for t in table if t['queue'] != main_q]
    sum_main_B = sum(t['us_in_step'] for t in mainrows)
    sum_main_A = sum(t['us_alone'] for t in mainrows)
    gaps_main = sum(max(0, t['gap_us']) for t in mainrows)
    big_gaps = sorted([t for t in mainrows if t['gap_us'] > 5], key=lambda t: -t['gap_us'])
    ov = [t for t in mainrows if t['beside_us']]
    no = [t for t in mainrows if not t['beside_us']]
    # union of busy time of all queues
    ev = sorted([(r['s'], 1) for r in b] + [(r['e'], -1) for r in b])
    busy, depth, prev, two = 0, 0, None, 0
    for t, d in ev:
        if depth > 0:
            busy += t - prev
        if depth > 1:
            two += t - prev
        depth += d
        prev = t
    summ = dict(
        wall_one_stream_ms=wallA / 1e6, wall_as_timed_ms=wallB / 1e6, dispatches=len(b), main_queue=main_q,
        main_queue_kernels=len(mainrows), side_queue_kernels=len(side),
        main_kernels_sum_alone_ms=sum_main_A / 1e3, main_kernels_sum_in_step_ms=sum_main_B / 1e3,
        main_chain_stretch_ms=(sum_main_B - sum_main_A) / 1e3, main_queue_gaps_ms=gaps_main / 1e3,
        side_kernels_sum_alone_ms=sum(t['us_alone'] for t in side) / 1e3, side_kernels_sum_in_step_ms=sum(t['us_in_step'] for t in side) / 1e3,
        any_queue_busy_ms=busy / 1e6, two_or_more_kernels_running_ms=two / 1e6, gpu_idle_ms=(wallB - busy) / 1e6,
        main_overlapped=dict(n=len(ov), alone_ms=sum(t['us_alone'] for t in ov) / 1e3, in_step_ms=sum(t['us_in_step'] for t in ov) / 1e3),
        main_not_overlapped=dict(n=len(no), alone_ms=sum(t['us_alone'] for t in no) / 1e3, in_step_ms=sum(t['us_in_step'] for t in no) / 1e3),
        largest_main_gaps=[dict(before=t['kernel'], i=t['i'], gap_us=round(t['gap_us'], 1)) for t in big_gaps[:12]])
    # by symbol
    sym = {}
    for t in table:
        k = (t['kernel'], 'main' if t['queue'] == main_q else 'side')
        d = sym.setdefault(k, dict(n=0, alone=0., instep=0., beside=0., gaps=0.))
        d['n'] += 1; d['alone'] += t['us_alone']; d['instep'] += t['us_in_step']; d['beside'] += min(t['us_in_step'], sum(t['beside_us'].values())); d['gaps'] += max(0, t['gap_us'])
    bysym = [dict(kernel=k[0], stream=k[1], launches=d['n'], alone_ms=round(d['alone'] / 1e3, 3), in_step_ms=round(d['instep'] / 1e3, 3),
                  ratio=round(d['instep'] / d['alone'], 3), ms_with_company=round(d['beside'] / 1e3, 3), gap_in_front_ms=round(d['gaps'] / 1e3, 3))
             for k, d in sorted(sym.items(), key=lambda kv: -kv[1]['instep'])]
    json.dump(dict(summary=summ, by_symbol=bysym, dispatches=table), open(sys.argv[3], 'w'), indent=1)
    lines = ["== summary =="] + [f"{k}: {v if not isinstance(v, float) else round(v, 3)}" for k, v in summ.items() if k != 'largest_main_gaps']
    lines += ["largest gaps on the main queue: " + ", ".join(f"{g['gap_us']} us before #{g['i']} {g['before'][:30]}" for g in summ['largest_main_gaps'])]
    lines += ["", "== by symbol (in-step ms, descending) ==", f"{'kernel':66s} {'strm':4s} {'n':>3s} {'alone':>7s} {'instep':>7s} {'ratio':>6s} {'w/comp':>7s} {'gaps':>6s}"]
    for s in bysym:
        lines.append(f"{s['kernel']:66s} {s['stream']:4s} {s['launches']:3d} {s['alone_ms']:7.3f} {s['in_step_ms']:7.3f} {s['ratio']:6.3f} {s['ms_with_company']:7.3f} {s['gap_in_front_ms']:6.3f}")
    lines += ["", "== every dispatch of the median step, in start order ==",
              f"{'#':>3s} {'q':>1s} {'start':>8s} {'instep':>7s} {'alone':>7s} {'gap':>6s} {'wgs':>5s} {'regs':>4s} kernel | beside"]
    for t in table:
        lines.append(f"{t['i']:3d} {t['queue']:>1s} {t['start_us']:8.1f} {t['us_in_step']:7.1f} {t['us_alone']:7.1f} {t['gap_us']:6.1f} {t['wgs']:5d} {t['regs']:4d} {t['kernel'][:50]:50s} | "
                     + ", ".join(f"{k[:28]} {v}" for k, v in t['beside_us'].items()))
    txt = "\n".join(lines)
    if len(sys.argv) > 4:
        open(sys.argv[4], 'w').write(txt + "\n")
    print("\n".join(lines[:lines.index("== every dispatch of the median step, in start order ==")]))


main()
