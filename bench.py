#!/usr/bin/env python
"""Headline benchmark: train clips/sec (60 s x 4-ch FOA @ 24 kHz) of HTS-AT mACCDOA, bf16, on N MI355X.

One step = one pass of the hot path over one batch of synthetic input already resident in HBM:
  32 clips/GPU = 192 ten-second chunks -> fused STFT/log-mel/IV -> scalar BN -> HTS-AT (Swin) -> mACCDOA head ->
  ADPIT loss -> hand-written backward -> (bucketed RCCL all-reduce) -> clip(1.0) -> AdamW.
Contract: python bench.py --gpus N --steps K --warmup W. N > 1: either under torch.distributed.run (one rank per GPU, the
driver's launch line) or plain `python bench.py --gpus N`, which then starts the N ranks itself (child processes, spawned
before this process has touched a GPU; never an exec). Rank 0 prints ONE JSON line (see DESIGN.md "Measurement" for the
roofline / cpu_baseline definitions). `--chunks 32` times the reference-native chunk batch (configs/experiment/
synth_maccdoa.yaml:8) instead of 32 whole clips.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

CLIP_SECONDS, CHUNKS_PER_CLIP, FS, CLASSES = 60, 6, 24000, 170
GFLOP_PER_CHUNK_TRAIN = 37.61       # 37.026 (net fwd+bwd) + 0.583 (features)
GFLOP_PER_CHUNK_TRAIN_CRNN = 285.4   # 3 x 94.94 (SURVEY §6 probe, CNN14-Conformer forward) + features
GFLOP_PER_CHUNK_TRAIN_EINV2 = 72.28  # 3 x 23.90 (SURVEY §6 probe, EINV2-HTSAT forward) + features
GFLOP_PER_CHUNK_TRAIN_PASST = 207.9  # 3 x (patch 1.65 + 7 blocks x (12 E^2 N + 4 N^2 E) = 67.8) + features; N=602, E=768
PEAK_BF16_TFLOPS = 2516.6           # dense MFMA bf16 peak, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0
PEAK_FLOPS, ESIZE = PEAK_BF16_TFLOPS, 2.0


class AttrDict(dict):
    __getattr__ = dict.__getitem__


def make_cfg():
    return AttrDict(data=AttrDict(n_mels=64, sample_rate=FS, hoplen=240, nfft=1024, window='hann',
                                  audio_feature='logmelIV'), adapt=AttrDict())


def synthetic_batch(clips, device, seed, chunks=None):
    """wave = 0.1*N(0,1) f32[clips, 4, 1 440 000] chunked as segment_index(chunklen=hoplen=10 s)
    would (6 full chunks per clip); adpit_label: track A0 only, act ~ Bernoulli(0.02), unit DOA. chunks: the
    reference-native chunk batch instead (f32[chunks, 4, 240 000], configs/experiment/synth_maccdoa.yaml:8)."""
    g = torch.Generator(device=device).manual_seed(seed)
    if chunks is not None:
        wave = 0.1 * torch.randn(chunks, 4, 10 * FS, generator=g, device=device)
    else:
        chunks = clips * CHUNKS_PER_CLIP
        wave = 0.1 * torch.randn(clips, 4, CHUNKS_PER_CLIP, 10 * FS, generator=g, device=device)
        wave = wave.permute(0, 2, 1, 3).reshape(chunks, 4, 10 * FS).contiguous()
    act = (torch.rand(chunks, 100, CLASSES, generator=g, device=device) < 0.02).float()
    doa = torch.randn(chunks, 100, 3, CLASSES, generator=g, device=device)
    doa = doa / doa.norm(dim=2, keepdim=True).clamp_min(1e-6)
    label = torch.zeros(chunks, 100, 6, 4, CLASSES, device=device)
    label[:, :, 0, 0] = act
    label[:, :, 0, 1:] = doa * act.unsqueeze(2)
    return wave, {'adpit_label': label}


from pseldnets_amd import ops as ops_mod   # (its stage tag names the part of the step a launch belongs to)


class KernelTimer:
    """HIP-event timing of every C-ABI launch on the stream it is enqueued on (torch's current stream)."""

    def __init__(self, lib):
        self.lib, self.records, self.on = lib, [], False
        self._orig = {}

    def install(self, names):
        for n in names:
            if n.endswith('_workspace') or n.endswith('_supported') or n in ('pseld_last_error', 'pseld_gemm_set_debug_buffer', 'pseld_attn_set_debug_buffer', 'pseld_mlp_set_debug_buffer', 'pseld_mlp_supported', 'pseld_passt_grid_t', 'pseld_gemm_last_kernel', 'pseld_adamw_bias_corrections', 'pseld_stage_marker', 'pseld_gemm8_set_debug_buffer', 'pseld_set_knob', 'pseld_unset_knob',
                                                                          'pseld_gemm8_force_tile', 'pseld_gemm_wgrad_timing', 'pseld_gemm_wgrad_timing_count', 'pseld_gemm_wgrad_timing_read',
                                                                          'pseld_gemm_wgrad_timing_symbol'):
                continue                                   # host-only queries: nothing is launched
            fn = getattr(self.lib, n)
            self._orig[n] = fn

            def wrapped(*a, _fn=fn, _n=n):
                if not self.on:
                    return _fn(*a)
                s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
                widx = self.lib.pseld_gemm_wgrad_timing_count() if _n == 'pseld_gemm_wgrad' else -1
                s.record()
                rc = _fn(*a)
                e.record()
                sym = self.lib.pseld_gemm_last_kernel().decode() if _n in ('pseld_gemm', 'pseld_gemm_wgrad') else None
                if widx >= 0 and self.lib.pseld_gemm_wgrad_timing_count() != widx + 1:
                    widx = -1                                  # (the library did not stamp this call: timing off)
                self.records.append((_n, a, s, e, sym, ops_mod._stage['name'], widx))
                return rc
            setattr(self.lib, n, wrapped)

    def detail(self):
        rows = {}
        for n, a, s, e, _sym, _st, _w in self.records:
            if n == 'pseld_gemm':
                key = ('gemm', a[1], a[2], a[6], a[7], a[8], a[19], a[20])      # ta, tb, M, N, K, epi, pro
            elif n == 'pseld_gemm_wgrad':
                key = ('wgrad', 1, 1, a[5], a[6], a[7], 0, a[11])
            else:
                continue
            d = rows.setdefault(key, [0.0, 0])
            d[0] += s.elapsed_time(e); d[1] += 1
        for key, (t, c) in sorted(rows.items(), key=lambda kv: -kv[1][0]):
            kind, ta, tb, M, N, K, epi, pro = key
            fl = 2.0 * M * N * K
            print(f"{kind:5s} ta{ta} tb{tb} M={M:7d} N={N:5d} K={K:5d} epi={epi:2d} pro={pro}: {c // 2:3d} launches/step, "
                  f"{t / c * 1e3:7.1f} us each, {fl / (t / c * 1e-3) / 1e12:6.0f} TF/s, {t / 2:7.3f} ms/step", file=sys.stderr)

    def by_symbol(self):
        """{kernel symbol: [ms, launches, flops, algorithmic bytes, bytes incl. fused operands]} over EVERY GEMM launch of the step: the forward /
        input-gradient launches of pseld_gemm (HIP events around the call = the kernel) and the weight-gradient kernels of
        pseld_gemm_wgrad, whose call also launches a slab reduction: there the library's own event pair around the GEMM kernel alone
        (pseld_gemm_wgrad_timing) is read."""
        torch.cuda.synchronize()
        out = {}
        for n, a, s, e, sym, _st, widx in self.records:
            if not sym:
                continue
            if n == 'pseld_gemm':
                M, N, K, epi = a[6], a[7], a[8], a[19]
                extra = M * N * ((1 if epi & 2 else 0) + (1 if epi & (4 | 32) else 0) + (1 if epi & 16 else 0))   # resid / aux rows read, second GELU output written
                ms, fl = s.elapsed_time(e), 2.0 * M * N * K
                nb = ESIZE * (M * K + N * K + M * N)                      # operands + result once (SURVEY 8d / DESIGN 4 definition)
                nb2 = ESIZE * (M * K + N * K + M * N + extra)              # + the fused epilogue operands the launch must also move
            elif n == 'pseld_gemm_wgrad' and widx >= 0:
                Mtok, N, K = a[5], a[6], a[7]
                ms, fl = self.lib.pseld_gemm_wgrad_timing_read(widx), 2.0 * Mtok * N * K
                if ms < 0:
                    continue
                nb = ESIZE * (Mtok * N + Mtok * K) + 4.0 * N * K           # both operands + the fp32 gradient once
                nb2 = nb
            else:
                continue
            d = out.setdefault(sym, [0.0, 0, 0.0, 0.0, 0.0])
            d[0] += ms; d[1] += 1; d[2] += fl; d[3] += nb; d[4] += nb2
        return out

    def by_stage(self):
        """{part of the step: [ms, launches]} (ops.stage names: features, front, stage0..3, head+loss, optimizer)."""
        torch.cuda.synchronize()
        out = {}
        for n, a, s, e, _sym, st, _w in self.records:
            d = out.setdefault(st, [0.0, 0])
            d[0] += s.elapsed_time(e); d[1] += 1
        return out

    def summary(self):
        torch.cuda.synchronize()
        agg = {}
        for n, a, s, e, _sym, _st, _w in self.records:
            key = n
            flops = 0.0
            nbytes = 0.0
            if n == 'pseld_gemm':
                key = 'gemm_kernel(fwd/dgrad)'
                M, N, K = a[6], a[7], a[8]
                flops = 2.0 * M * N * K
                nbytes = ESIZE * (M * K + N * K + M * N)            # operands + result once (fused extras not counted)
            elif n == 'pseld_gemm_wgrad':
                key = 'gemm_kernel(wgrad)+reduce'
                M, N, K = a[5], a[6], a[7]
                flops = 2.0 * M * N * K
                nbytes = ESIZE * (M * N + M * K) + 4.0 * N * K
            t = s.elapsed_time(e)
            # roofline time of this launch: whichever of the two ceilings binds its shape
            roof_ms = max(flops / (PEAK_FLOPS * 1e12), nbytes / (PEAK_HBM_GBS * 1e9)) * 1e3
            d = agg.setdefault(key, [0.0, 0, 0.0, 0.0, 0.0, 0.0])
            d[0] += t; d[1] += 1; d[2] += flops; d[3] += nbytes; d[4] += roof_ms
            d[5] += t if flops / (PEAK_FLOPS * 1e12) >= nbytes / (PEAK_HBM_GBS * 1e9) else 0.0
        return agg


def pmc_traffic(args, sym=None):
    """HBM bytes per launch of the dominant kernel from the PMC counters. Counters cannot be read from inside this process: the figure is the
    one measured with rocprofv3 on this same command (separate --pmc passes, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950),
    reduced per launch of one kernel symbol by tools/pmc_kernel.py and committed under profiles/r06_pmc/ (r05_pmc/ before) - one file per symbol that has led
    the step (two instantiations of the eight-phase kernel are within 2 % of each other and trade places between runs)."""
    if args.backbone != 'htsat' or args.dtype != 'bf16' or args.clips != 32 or args.chunks:
        return None
    cands = []
    for rnd in ('r06', 'r05'):          # the newest round's passes first
        if sym:
            cands.append(os.path.join(ROOT, 'profiles', rnd + '_pmc', 'pmc_' + ''.join(c if c.isalnum() else '_' for c in sym) + '.json'))
        cands.append(os.path.join(ROOT, 'profiles', rnd + '_dominant_kernel_pmc.json'))
    for path in cands:
        if os.path.exists(path):
            with open(path) as f:
                d = json.load(f)
            if sym is None or d.get('kernel') == sym:
                d['_path'] = os.path.relpath(path, ROOT)
                return d
    return None


def cpu_baseline(chunks=4, steps=3, threads=None):
    """The CPU oracle (a port: the reference's Python cannot travel) timed on this host: features + HTS-AT mACCDOA
    fwd + ADPIT + backward + clip + AdamW on `chunks` 10 s chunks, fp32; `threads` = torch intra-op threads (None: all)."""
    from oracle import feature as of, htsat as oh, losses as ol, optim as oo
    torch.manual_seed(0)
    all_threads = torch.get_num_threads()
    if threads is not None:
        torch.set_num_threads(threads)
    cfg = dict(embed_dim=96, depths=(2, 2, 6, 2), num_heads=(4, 8, 16, 32), drop_path_rate=0.0)
    sd = oh.formula_state('multi_accdoa', CLASSES, 7, cfg)
    names = [k for k, v in sd.items() if v.is_floating_point() and 'running' not in k]
    m = [torch.zeros_like(sd[n]) for n in names]
    v = [torch.zeros_like(sd[n]) for n in names]
    wave = 0.1 * torch.randn(chunks, 4, 10 * FS)
    label = torch.zeros(chunks, 100, 6, 4, CLASSES)
    label[:, :, 0, 0] = (torch.rand(chunks, 100, CLASSES) < 0.02).float()
    times = []
    for it in range(steps + 1):
        t0 = time.perf_counter()
        p = {k: (t.detach().requires_grad_(True) if k in names else t) for k, t in sd.items()}
        feat = of.logmel_iv(wave)
        out = oh.accdoa_htsat_forward(feat, p, cfg, training=True, key='multi_accdoa')
        loss = ol.adpit(out, {'adpit_label': label})['loss_all']
        loss.backward()
        with torch.no_grad():
            plist = [sd[n] for n in names]
            oo.adamw_step(plist, [p[n].grad for n in names], m, v, it + 1, 1e-4)
        times.append(time.perf_counter() - t0)
    t = sorted(times[1:])[len(times[1:]) // 2]
    used = torch.get_num_threads()
    torch.set_num_threads(all_threads)
    return {"value": round(chunks / CHUNKS_PER_CLIP / t, 4), "unit": "clips/s", "cores": used,
            "kind": "port", "sample": f"{steps} timed train steps of {chunks} ten-second chunks (fp32 oracle, "
            f"features+fwd+bwd+clip+AdamW) after 1 warm-up, median step {t:.2f} s"}


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as CHILD processes of this one (which has not
    touched a GPU: torch is imported, no HIP call has been made), one per GPU, rendezvous on 127.0.0.1. Rank 0's stdout (the
    JSON line) is relayed to ours; any failing rank fails the run and the others are stopped by their exact PIDs."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), PSELD_BENCH_CHILD='1')
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr))
    # rank 0's pipe is drained by a thread while EVERY child is polled: a rank that dies at start-up (bad device index, out of
    # memory) would otherwise leave rank 0 in the rendezvous until the store / RCCL timeout, many minutes later
    import threading
    box = []
    th = threading.Thread(target=lambda: box.append(procs[0].stdout.read()), daemon=True)
    th.start()
    codes = [None] * n
    while any(c is None for c in codes):
        for r, pr in enumerate(procs):
            if codes[r] is None:
                codes[r] = pr.poll()
        if any(c not in (None, 0) for c in codes):
            for r, pr in enumerate(procs):                 # first failure: stop the others by their exact PIDs
                if codes[r] is None:
                    pr.terminate()
            for r, pr in enumerate(procs):
                if codes[r] is None:
                    try:
                        codes[r] = pr.wait(timeout=30)
                    except subprocess.TimeoutExpired:
                        pr.kill()
                        codes[r] = pr.wait()
            break
        time.sleep(0.2)
    th.join(timeout=30)
    out0 = box[0] if box else b''
    # only the JSON line travels up: RCCL writes a version banner to the ranks' C-level stdout at exit
    text = out0.decode() if out0 else ''
    sys.stdout.write(''.join(ln + '\n' for ln in text.splitlines() if ln.startswith('{')))
    sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad:
        raise SystemExit(f"bench.py: ranks failed (rank, exit code): {bad}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--clips', type=int, default=32, help='60 s clips per GPU per step')
    ap.add_argument('--chunks', type=int, default=0,
                    help='N > 0: time the reference-native chunk batch (N ten-second chunks per GPU per step, '
                         'configs/experiment/synth_maccdoa.yaml:8 batch_size 32) instead of --clips whole clips; clips/s = (N/6)*world/step')
    ap.add_argument('--sync-bn', default='auto', choices=['auto', 'on', 'off'],
                    help="all-reduce every BatchNorm's train-mode statistics over the ranks (configs/trainer/gpu.yaml:9 sync_batchnorm: True). "
                         "auto = on when N > 1 (the reference's DDP recipe)")
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f32'])
    ap.add_argument('--backbone', default='htsat', choices=['htsat', 'passt', 'htsat_einv2', 'crnn', 'passt_einv2', 'crnn_einv2'],
                    help='htsat = the headline workload (BASELINE.json configs[1]); htsat_einv2 = configs[2] (dual-branch, tPIT); '
                         'passt = the PaSST backbone, same data')
    ap.add_argument('--augment', default='none', choices=['none', 'augmix'],
                    help="augmix = configs/augment/augmix.yaml (every shipped synth_* experiment trains with it): the batch is "
                         "tripled, rotated / mixed as waveforms and masked / shifted as features on the device; `value` still counts "
                         "the ORIGINAL clips. Not the headline configuration.")
    ap.add_argument('--decoder', default='conformer', choices=['conformer', 'gru', 'none'],
                    help="--backbone crnn only: configs/model/crnn.yaml's conformer (1 block), configs/model/default.yaml's gru (2 layers) or Identity")
    ap.add_argument('--adapt', default='none', choices=['none', 'adapter'],
                    help="adapter = configs/adapt/adapter.yaml fine-tuning (HTS-AT only): adapters + biases + head train. Not the headline.")
    ap.add_argument('--graph', default='auto', choices=['auto', 'on', 'off'],
                    help='replay the training step as one hipGraph (trainer.py use_graph). auto = off: measured, the replay is no faster than the eager launches at 32 or at 192 chunks (the step is GPU-bound at both)')
    ap.add_argument('--grad-dtype', default='f32', choices=['f32', 'bf16'],
                    help='payload of the gradient all-reduce (N > 1): f32 = the arena in place (138 MB), bf16 = cast buckets (69 MB), '
                         'summed on the wire in bf16 and accumulated back into the fp32 arena')
    ap.add_argument('--comm', default='torch', choices=['torch', 'rccl', 'rccl_direct'],
                    help="gradient all-reduce (N > 1): torch = torch.distributed.all_reduce on the RCCL process group (default); rccl = this build's "
                         "own RCCL layer (include/pseld_comm.h), ncclAllReduce; rccl_direct = the same layer, grouped point-to-point reduce-scatter + "
                         "all-gather to all peers at once (S / W bytes per xGMI link per phase instead of a ring's 2 (W-1)/W S over one link)")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-timing', action='store_true')
    ap.add_argument('--gemm-detail', action='store_true', help='per-shape GEMM launch table on stderr')
    args = ap.parse_args()

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        return spawn_ranks(args.gpus)                 # nothing in this process has touched a GPU yet
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; the two must agree "
                         "(a one-rank run must not be reported as an N-GPU number)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # PSELD_BENCH_BACKEND=gloo: control-flow check of the multi-rank path on a box with fewer GPUs than ranks (ranks
    # then share devices; RCCL refuses that). The measured configuration is always nccl (= RCCL), one rank per GPU.
    backend = os.environ.get('PSELD_BENCH_BACKEND', 'nccl')
    if backend != 'nccl':
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    group = None
    if world == 1 and os.environ.get('PSELD_BENCH_FORCE_GROUP'):
        # stream-placement check on one GPU: a world-size-1 RCCL group makes the trainer issue its bucketed all-reduces
        # (identity collectives on RCCL's own stream) so that a kernel trace shows them beside the backward kernels
        import torch.distributed as dist
        dist.init_process_group('nccl', init_method='tcp://127.0.0.1:29877', rank=0, world_size=1, device_id=device)
        group = dist.group.WORLD
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=device)
        else:
            dist.init_process_group(backend)
        group = dist.group.WORLD

    from pseldnets_amd import _lib
    from pseldnets_amd.models import multi_accdoa
    from pseldnets_amd.trainer import FusedTrainer
    from pseldnets_amd.utils.config import get_afextractor

    cfg = make_cfg()
    if args.adapt == 'adapter':
        from pseldnets_amd.train import ADAPT_GROUPS
        cfg['adapt'] = ADAPT_GROUPS['adapter']
    torch.manual_seed(2024)
    if args.backbone == 'htsat':
        net = multi_accdoa.HTSAT(cfg, CLASSES, 7, pretrained_path=None)      # configs/model/htsat.yaml geometry
    elif args.backbone == 'htsat_einv2':
        from pseldnets_amd.models import einv2
        net = einv2.HTSAT(cfg, CLASSES, 7, pretrained_path=None)             # einv2.py:189-327: SED + DOA encoders
    elif args.backbone == 'passt_einv2':
        from pseldnets_amd.models import einv2
        cfg['model'] = AttrDict(decoder=None, num_decoder_layers=2, ps_gap=2)       # configs/model/passt.yaml:4-6
        net = einv2.PASST(cfg, CLASSES, 7, pretrained_path=None)                    # einv2.py:446-575
    elif args.backbone == 'crnn_einv2':
        from pseldnets_amd.models import einv2
        cfg['model'] = AttrDict(decoder=None if args.decoder == 'none' else args.decoder,
                                num_decoder_layers=2 if args.decoder == 'gru' else 1)
        net = einv2.CRNN(cfg, CLASSES, 7, encoder='CNN12', pretrained_path=None,
                         num_features=[64, 128, 256, 512, 1024, 2048])              # einv2.py:17-174, configs/model/crnn.yaml kwargs
    elif args.backbone == 'crnn':
        cfg['model'] = AttrDict(decoder=None if args.decoder == 'none' else args.decoder,
                                num_decoder_layers=2 if args.decoder == 'gru' else 1)   # configs/model/{crnn,default}.yaml:5-6
        net = multi_accdoa.CRNN(cfg, CLASSES, 7, encoder='CNN12', pretrained_path=None,
                                num_features=[64, 128, 256, 512, 1024, 2048])   # configs/model/crnn.yaml kwargs
    else:
        net = multi_accdoa.PASST(cfg, CLASSES, 7, pretrained_path=None)      # configs/model/passt.yaml geometry
    net.compute_dtype = torch.bfloat16 if args.dtype == 'bf16' else torch.float32
    net.to(device)
    if world > 1:
        import torch.distributed as dist
        for p in net.parameters():            # identical initial weights on every rank
            dist.broadcast(p.data, 0)
    einv2_mode = args.backbone.endswith('_einv2')
    sync_bn = group is not None and args.sync_bn != 'off'      # every BatchNorm (scalar front, conv stack, Conformer) synchronised
    wave, target = synthetic_batch(args.clips, device, 2024 + rank, chunks=args.chunks or None)
    n_chunks = wave.shape[0]
    use_graph = group is None and args.adapt == 'none' and args.warmup >= 2 and \
        args.graph == 'on'
    trainer = FusedTrainer(net, get_afextractor(cfg).to(device), 'tpit' if einv2_mode else 'adpit', lr=1e-4, max_norm=1.0,
                           process_group=group, sync_bn=sync_bn, use_graph=use_graph, graph_warmup=min(3, args.warmup - 1),
                           comm=args.comm if backend == 'nccl' else 'torch')
    trainer.grad_dtype = args.grad_dtype
    clips_per_step = n_chunks / CHUNKS_PER_CLIP
    if einv2_mode:     # track-wise labels: track 0 carries the ADPIT A0 events, tracks 1-2 silent
        lab = target['adpit_label']
        act = lab[:, :, 0, 0]                                        # [chunks, 100, C]
        sed = torch.zeros(act.shape[0], 100, 3, CLASSES, device=device)
        first = (act.cumsum(-1) == 1) & (act > 0)                    # one class per track: the first active one
        sed[:, :, 0] = first.float()
        doa = torch.zeros(act.shape[0], 100, 3, 3, device=device)
        doa[:, :, 0] = (lab[:, :, 0, 1:] * first.unsqueeze(2)).sum(-1)
        target = {'sed_label': sed, 'doa_label': doa}

    # steady-state pipeline: the features of step i + 1 (here the same synthetic waveform) are extracted on a second stream while
    # step i is in its backward; every timed step still contains exactly one feature extraction (PSELD_FEATURE_PREFETCH=0: in line)
    prefetch = os.environ.get('PSELD_FEATURE_PREFETCH', '1') == '1'
    step = lambda: trainer.training_step(wave, target, next_x=wave if prefetch else None)
    if args.augment == 'augmix':
        if einv2_mode:
            raise SystemExit("--augment augmix is wired for the ADPIT workloads")
        from pseldnets_amd.models.model_module import SELDModelModule
        from pseldnets_amd.train import SyntheticDataset, compose
        acfg = compose(['experiment=synth_maccdoa', 'augment=augmix'])
        aug = SELDModelModule(acfg, SyntheticDataset(acfg))
        aug.af_extractor = trainer.af
        target['ov'] = ['1'] * wave.shape[0]

        def step():
            feats, tgt = aug.augment_step(wave, target)
            return trainer.training_step(feats, tgt, is_features=True)

    lib = _lib.lib()
    global PEAK_FLOPS, ESIZE
    PEAK_FLOPS = PEAK_BF16_TFLOPS if args.dtype == 'bf16' else 157.3
    ESIZE = 2.0 if args.dtype == 'bf16' else 4.0
    timer = None
    if rank == 0 and not args.no_kernel_timing:
        timer = KernelTimer(lib)
        timer.install(_lib.declared_symbols())

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        loss = step()
    barrier()
    if group is not None:
        trainer.enable_comm_diag()        # HIP events around every collective wait of the timed steps (the N > 1 line explains itself)
    # HIP events at the step boundaries on the launch stream: per-step durations (median) beside the whole-region clock
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        loss = step()
        marks[i + 1].record()
    barrier()
    elapsed = time.perf_counter() - t0
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    median_ms = step_ms[len(step_ms) // 2]
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    ms_per_step = 1e3 * elapsed / args.steps
    clips_per_s = clips_per_step * world / (elapsed / args.steps)
    loss_val = float(loss['loss_all'].item())

    crnn_names = {'conformer': 'CNN12 x2 + 6 Conformer blocks', 'gru': 'CNN12 x2 + 6 two-layer BiGRUs', 'none': 'CNN12 x2, Identity decoders'}
    name = {'htsat': 'HTS-AT', 'passt': 'PaSST', 'htsat_einv2': 'HTS-AT EINV2', 'passt_einv2': 'PaSST EINV2',
            'crnn_einv2': f'CRNN EINV2 ({crnn_names[args.decoder]})', 'crnn': {'conformer': 'CNN14-Conformer (CRNN: CNN12 + 1 Conformer block)', 'gru': 'CNN14-GRU (CRNN: CNN12 + 2-layer BiGRU)', 'none': 'CNN14 (CRNN: CNN12, Identity decoder)'}[args.decoder]}[args.backbone]
    gflop_chunk = {'htsat': GFLOP_PER_CHUNK_TRAIN, 'passt': GFLOP_PER_CHUNK_TRAIN_PASST, 'htsat_einv2': GFLOP_PER_CHUNK_TRAIN_EINV2,
                   'passt_einv2': 2 * GFLOP_PER_CHUNK_TRAIN_PASST - 0.583,       # two encoders, one feature extraction
                   'crnn_einv2': 2 * GFLOP_PER_CHUNK_TRAIN_CRNN - 0.583,         # approximate: two conv stacks (decoders vary)
                   'crnn': GFLOP_PER_CHUNK_TRAIN_CRNN}[args.backbone]
    out = {
        "metric": f"train clips/sec (60 s 4-ch FOA) {name}" + ("" if einv2_mode else " mACCDOA") + (" + AugMix" if args.augment == 'augmix' else "") + (" (adapter fine-tuning)" if args.adapt == 'adapter' else ""), "value": round(clips_per_s, 2), "unit": "clips/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"{name}{' dual-branch (tPIT)' if einv2_mode else ' mACCDOA'} {args.dtype}, "
                               + (f"reference-native chunk batch: {n_chunks} ten-second chunks (= {clips_per_step:.2f} clips of 60 s) per GPU per step, "
                                  if args.chunks else f"{args.clips} clips x 60 s FOA @ 24 kHz per GPU = {n_chunks} ten-second chunks/step, ")
                               + f"170 classes, {'tPIT' if einv2_mode else 'ADPIT'}, clip 1.0, AdamW, "
                               f"{'dropout 0.1' if args.backbone.startswith('crnn') else 'drop_path 0.0' if args.backbone.startswith('passt') else 'drop_path 0.1'}, BN train mode, {'AugMix augmentations (x3 chunks through the network)' if args.augment == 'augmix' else 'no augmentation'}",
                   "global_clips": round(clips_per_step * world, 3), "global_chunks": n_chunks * world, "parallelism": f"dp{world}",
                   "sync_batchnorm": bool(sync_bn), "comm": args.comm},
        "loss": round(loss_val, 6),
        # per-step HIP-event durations on rank 0's launch stream (median over >= 100 steps when --steps >= 100)
        "ms_per_step_median": round(median_ms, 3), "ms_per_step_min": round(step_ms[0], 3), "ms_per_step_p90": round(step_ms[int(0.9 * (len(step_ms) - 1))], 3),
        "value_at_median_step": round(clips_per_step * world / (median_ms * 1e-3), 2),
        "rccl_ranks": (torch.distributed.get_world_size() if world > 1 else 1), "comm_backend": backend if world > 1 else None,
        "comm_allreduce": trainer.comm_kind if group is not None else None,
        "hip_graph": bool(use_graph and trainer._graph is not None),      # the timed steps were replays of one captured hipGraph
        "wgrad_side_stream": ops_mod._wgrad_stream['on'] and args.backbone.startswith('htsat'),
        # every timed step issues ONE feature extraction - that of the next step's batch, on a second stream (the pipeline a loader with
        # one batch of look-ahead gives); the step itself consumes the extraction issued by its predecessor
        "feature_prefetch": bool(prefetch and args.augment == 'none'),
    }
    if args.augment == 'augmix':
        gflop_chunk = 3 * gflop_chunk                     # every original chunk goes through the network three times
    if group is not None:
        # what the compute stream paid for the collectives: per bucket (issued back to front while earlier layers are still in backward)
        # the time from issue to the end of its wait, and the part of it the stream actually stalled; rank 0's view
        out["comm"] = trainer.comm_report()
        trainer.enable_comm_diag(False)
    step_tflops = clips_per_s * CHUNKS_PER_CLIP * gflop_chunk / 1e3
    out["roofline_step"] = {"bound": "mfma", "achieved": round(step_tflops / world, 2), "peak": PEAK_BF16_TFLOPS,
                            "unit": "TFLOP/s", "frac": round(step_tflops / world / PEAK_BF16_TFLOPS, 4)}
    if not args.no_kernel_timing:
        # separate instrumented steps (HIP events around every launch on the launch stream, recorded on rank 0). EVERY
        # rank runs them: a training step contains the gradient all-reduce, so rank 0 cannot step alone.
        if timer is not None:
            timer.on = True
            lib.pseld_gemm_wgrad_timing(1)           # the weight-gradient GEMM kernels alone (their call also launches the slab reduction)
        trainer.use_graph = False                             # the instrumented steps launch kernel by kernel
        # ... and on ONE stream: in the timed region the weight gradients run on a second stream beside the main chain
        # (htsat.py:_wgrad), where a kernel's duration includes the time it shares the CUs with another kernel - the per-kernel
        # roofline below is that of the kernel running alone (rocprofv3 of `PSELD_WGRAD_STREAM=0 PSELD_FEATURE_PREFETCH=0 python3 bench.py` agrees with it)
        ops_mod.set_wgrad_stream(False)
        prefetch = False
        trainer._prefetched = None
        for _ in range(2):
            step()
        barrier()
    if rank == 0 and timer is not None:
        agg = timer.summary()
        if args.gemm_detail:
            timer.detail()
        timer.on = False
        lib.pseld_gemm_wgrad_timing(0)
        total = sum(v[0] for v in agg.values())
        top = sorted(agg.items(), key=lambda kv: -kv[1][0])
        gem = [v for k, v in agg.items() if k.startswith('gemm_kernel(fwd')]
        syms = timer.by_symbol()
        if syms:
            # the single dominant kernel of the step: the symbol with the largest launch-time total (HIP events on the launch stream)
            sym, (tms, n, fl, nb, nb_moved) = max(syms.items(), key=lambda kv: kv[1][0])
            ach_tf = fl / (tms * 1e-3) / 1e12
            gbs = nb / (tms * 1e-3) / 1e9
            # which ceiling binds this symbol's launches: their aggregate arithmetic intensity against the ridge (peak flops / peak bytes)
            hbm = (fl / nb) < (PEAK_FLOPS * 1e12) / (PEAK_HBM_GBS * 1e9)
            pmc = pmc_traffic(args, sym) or {}
            same = pmc.get('kernel') == sym
            out["roofline"] = {"bound": "hbm" if hbm else "mfma", "kernel": sym,
                               "measured": "kernel alone: the instrumented steps keep the weight gradients on the main stream (PSELD_WGRAD_STREAM=0)",
                               "achieved": round(gbs if hbm else ach_tf, 2), "peak": PEAK_HBM_GBS if hbm else PEAK_FLOPS,
                               "unit": "GB/s" if hbm else "TFLOP/s",
                               "frac": round((gbs / PEAK_HBM_GBS) if hbm else (ach_tf / PEAK_FLOPS), 4),
                               "launches_per_step": n // 2, "avg_launch_ms": round(tms / n, 4),
                               "algorithmic_bytes_per_launch": round(nb / n, 1), "flops_per_launch": round(fl / n, 1),
                               "bytes_incl_fused_epilogue_operands_per_launch": round(nb_moved / n, 1),
                               "achieved_tflops": round(ach_tf, 2), "achieved_gbs": round(gbs, 1),
                               "achieved_gbs_incl_fused_operands": round(nb_moved / (tms * 1e-3) / 1e9, 1),
                               # counters cannot be read from inside this process: rocprofv3 --pmc passes of this same command,
                               # reduced per launch of this kernel symbol by tools/pmc_kernel.py and committed (static figures)
                               "traffic": round(pmc['traffic_bytes_per_launch'], 1) if same else None,
                               "mfma_busy": pmc.get('mfma_busy') if same else None,
                               "rocprof_avg_launch_ms": pmc.get('rocprof_avg_launch_ms') if same else None,
                               "pmc_source": (pmc.get('_path', '') + " (static: rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE | "
                                              "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE, separate passes over this command)") if same else None,
                               # the ranking the dominant symbol was picked from: every GEMM kernel symbol of the step, weight gradients included
                               "ranked_symbols": [{"kernel": k, "ms_per_step": round(v[0] / 2, 3), "launches_per_step": v[1] // 2,
                                                   "tflops": round(v[2] / (v[0] * 1e-3) / 1e12, 1), "algorithmic_gbs": round(v[3] / (v[0] * 1e-3) / 1e9, 1)}
                                                  for k, v in sorted(syms.items(), key=lambda kv: -kv[1][0])[:6]]}
        if gem:
            tms, n, fl, nb, roof_ms, mfma_bound_ms = gem[0]
            out["roofline_family"] = {"kernel": "every pseld_gemm forward + input-gradient launch (all tile variants)",
                                      "launches_per_step": n // 2, "avg_launch_ms": round(tms / n, 4),
                                      "achieved_tflops": round(fl / (tms * 1e-3) / 1e12, 2), "achieved_gbs": round(nb / (tms * 1e-3) / 1e9, 1),
                                      "frac_shape_aware": round(roof_ms / tms, 4),
                                      "time_share_mfma_bound_shapes": round(mfma_bound_ms / tms, 4)}
        # where the step's time is, part by part (HIP events of the instrumented one-stream steps; HBM GB per part: profiles/r04_stage_table.json,
        # rocprofv3 --pmc passes of this command cut at stage markers by tools/pmc_stages.py)
        st = timer.by_stage()
        st_pmc = {}
        sp = next((q for q in (os.path.join(ROOT, 'profiles', r + '_stage_table.json') for r in ('r06', 'r05', 'r04')) if os.path.exists(q)),
                  os.path.join(ROOT, 'profiles', 'r04_stage_table.json'))
        if args.backbone == 'htsat' and args.dtype == 'bf16' and args.clips == 32 and not args.chunks and os.path.exists(sp):
            with open(sp) as f:
                st_pmc = json.load(f).get('stages', {})
        out["stage_table"] = {k: {"ms_per_step": round(v[0] / 2, 3), "launches": v[1] // 2,
                                  "hbm_gb_per_step": st_pmc.get(k, {}).get("hbm_gb_per_step")} for k, v in sorted(st.items(), key=lambda kv: -kv[1][0])}
        out["kernel_time_share"] = {k: {"ms_per_step": round(v[0] / 2, 3), "launches": v[1] // 2,
                                        "share": round(v[0] / total, 4)} for k, v in top[:(40 if args.gemm_detail else 10)]}
        for k, v in agg.items():
            if k.startswith('gemm_kernel(wgrad') and k in out["kernel_time_share"]:
                out["kernel_time_share"][k]["frac_shape_aware"] = round(v[4] / v[0], 4)
    if rank == 0:
        if not args.no_cpu_baseline and world == 1:
            try:
                # the headline CPU figure is BASELINE.md section 3's comparability line: 8 threads (the survey's probe of the REFERENCE on
                # 8 threads: 0.45 clips/s). Every host core is reported beside it: on these shared 128-thread hosts the oversubscribed run
                # is both slower and erratic (0.06 - 0.28 clips/s between boxes), so it is not the number to compare against
                out["cpu_baseline"] = cpu_baseline(threads=8)
                out["cpu_baseline"]["at_all_host_threads"] = cpu_baseline()
            except Exception as e:  # the checker must never take the measurement down
                out["cpu_baseline"] = {"error": repr(e)}
        # RCCL's version banner sits in the C library's stdout buffer until exit and would land BEHIND the JSON line: flush it
        # out first and leave stdout closed to the C side afterwards, so that the JSON line is the last line of rank 0's output
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        print(json.dumps(out), flush=True)
        if world > 1 or os.environ.get('PSELD_BENCH_FORCE_GROUP') == '1':
            try:
                devnull = os.open(os.devnull, os.O_WRONLY)
                os.dup2(devnull, 1)                    # whatever the libraries print at teardown goes nowhere
            except OSError:
                pass
    if world > 1 or group is not None:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
