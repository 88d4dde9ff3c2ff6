#!/usr/bin/env python3
"""Benchmark of the AMT-SAGA hot path on MI355X: audio windows/sec through the
detect -> subtract loop (BASELINE.json metric), one process per GPU.

    python bench.py --gpus 1 --steps K --warmup W [--workload c3|c2|c5]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one batch of synthetic windows that are
already resident in HBM: STFT(2048) -> mag/phase/max -> `iters` x [features ->
RDCNN heads -> note decision -> guess lookup -> subtract] -> event gather.
Weak scaling: every rank processes the same number of windows.

Prints ONE JSON line on rank 0 with the driver contract plus
  roofline      dominant kernel(s) of the step, timed with HIP events recorded around every conv layer in the one-stream
                measurement pass: conv mode 3 (default) -- the FFT-domain form of the timing heads' 4 x 16 layers (fc_gemm_kernel +
                fc_row_kernel per layer): algorithmic HBM bytes of the frequency tensors / time; modes 0-2 -- the
                largest direct convolution class on the matrix pipe: algorithmic TFLOP/s
  roofline_mfma (mode 3) the largest DIRECT split-fp16 convolution class, as `roofline` reports it in mode 2
  roofline_stft the north-star HBM kernel pair (STFT->mag(/phase)/max, subtract): algorithmic GB/s over the
                bytes the step CONSUMES (no phase plane when no iteration reads it), HIP events on the launch stream
  cpu_baseline  the numpy oracle ("port") timed on this host on a bounded sample: one process x one thread,
                and a multiprocessing Pool (one window per task, one BLAS thread each: the reference's
                worker model, training.py:623-630)
  value_f32_mfma          the same step with the strict-f32 convolutions (--conv-mode 0), same run
  value_one_stream_profiled     the measurement pass behind the roofline objects: the same steps with both timing networks
                          on ONE stream (a kernel's duration is then that of a kernel that owns the chip) and HIP events
                          around every conv layer / STFT / subtract launch, outside the timed region (the product's default
                          is one stream too)
  value_two_timing_streams  the same step with timing_end on a second HIP stream under timing_start (TranscriptionLoop.
                          timing_streams = 2, opt-in; parity-tested at the metric size since round 4's fix: DESIGN 10.1)
  value_h2d_inclusive     the same step with the batch's audio copied host -> HBM inside the timed region
                          (SURVEY 8d defines the metric including that copy; never `value`)
  value_h2d_overlapped    the same with that copy on a second stream, overlapped with the previous batch's compute
                          (TranscriptionLoop.run_stream), steady state
  prepare_ms              the untimed per-batch set-up (STFT + song-level CQT normalisers)
  value_prepare_inclusive the same step with that set-up inside the timed region (a fresh batch every step)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'amt-saga_amd')]

import numpy as np          # noqa: E402
import torch                # noqa: E402

WORKLOADS = {
    # BASELINE.json configs[1]
    'c2': dict(B=256, heads=('pitch',), iters=1, subtract=False, notes=(1, 1), groups=(0,), seed=2,
               name='C2: 256 single-note piano windows, STFT(2048) + RDCNN pitch head'),
    # BASELINE.json configs[2] -- the largest single-GPU config and the first with the full loop
    'c3': dict(B=1024, heads=('timing', 'pitch', 'velocity'), iters=1, subtract=True, notes=(3, 3),
               groups=(0,), seed=3,
               name='C3: 1024 polyphonic piano windows, STFT(2048) + pitch+velocity+timing(start,end) '
                    'heads + 1 subtraction iteration'),
    # per-GPU shard of BASELINE.json configs[3] (4096 windows / 4 GPUs)
    'c4': dict(B=1024, heads=('pitch', 'instrument'), iters=3, subtract=True, notes=(1, 3),
               groups=(0, 1, 2), seed=4,
               name='C4 shard: 1024 mixed-instrument windows per GPU, instrument+pitch heads, 3 iterations'),
    # per-GPU shard of BASELINE.json configs[4] (16384 windows / 8 GPUs)
    'c5': dict(B=2048, heads=('timing', 'pitch', 'instrument', 'velocity'), iters=5, subtract=True,
               notes=(2, 4), groups=(0, 1, 2), seed=5,
               name='C5 shard: 2048 mixed-instrument windows per GPU, all heads, 5 iterations'),
}
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TF = 157.3       # MI355X_MICROARCH.md: dense FP32 matrix peak
MFMA_BF16_PEAK_TF = 2500.0     # MI355X_MICROARCH.md: dense BF16 matrix peak


def build(workload, B):
    from amt_saga.hyperparams import Hyperparams
    from amt_saga.loop import TranscriptionLoop
    p = Hyperparams(N=2048)
    loop = TranscriptionLoop(p, heads=workload['heads'], iters=workload['iters'],
                             subtract=workload['subtract'], groups=workload['groups'])
    loop.setup_device()
    return p, loop


def _cpu_worker_setup(wl_name, n_fft):
    """Build the oracle loop in a process that never touches the GPU (host-side objects only: the
    product loop is constructed for its seeded weights, never set up on a device)."""
    from oracle.loop import LoopOracle
    from oracle import synth as osynth
    from amt_saga import synth
    from amt_saga.hyperparams import Hyperparams
    from amt_saga.loop import TranscriptionLoop
    wl = WORKLOADS[wl_name]
    p = Hyperparams(N=n_fft)
    loop = TranscriptionLoop(p, heads=wl['heads'], iters=wl['iters'], subtract=wl['subtract'], groups=wl['groups'])
    bank = osynth.guess_bank_waves(wl['groups'], p.pitch_low, p.pitch_high, sr=p.sr) if wl['subtract'] else None
    remap = np.zeros(3, np.int32)
    for i, g in enumerate(wl['groups']):
        remap[g] = i
    return LoopOracle(p, wl['heads'], {k: n.weights for k, n in loop.nets.items()}, iters=wl['iters'],
                      subtract=wl['subtract'], prog_group=remap[synth.prog_group_table(p.instrument_classes)],
                      bank_waves=bank)


_ORC = None
_DATA = None


def _cpu_pool_init_shared():
    from threadpoolctl import threadpool_limits
    threadpool_limits(limits=1)


def _cpu_pool_task(i):
    refs = {k[4:]: _DATA[k][i] for k in _DATA if k.startswith('ref_')}
    t0 = time.perf_counter()
    _ORC.run_window(_DATA['wave'][i], refs, i)
    return time.perf_counter() - t0


def cpu_baseline_main(wl_name, n_fft, path, budget_s):
    """Runs in a child process started by bench.py (no GPU initialised here, so forking a Pool is safe).
    Leg 1: one process, every numeric library pinned to ONE thread.  Leg 2: multiprocessing.Pool over the
    host cores this job may use, one window per task, one BLAS thread per worker."""
    import multiprocessing as mp
    from threadpoolctl import threadpool_limits
    data = np.load(path)
    n = data['wave'].shape[0]
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    out = {}
    with threadpool_limits(limits=1):
        orc = _cpu_worker_setup(wl_name, n_fft)
        refs0 = {k[4:]: data[k][0] for k in data.files if k.startswith('ref_')}
        orc.run_window(data['wave'][0], refs0, 0)                     # first call: page faults, BLAS start-up
        done, t_total = 0, 0.0
        for i in range(n):
            refs = {k[4:]: data[k][i] for k in data.files if k.startswith('ref_')}
            t0 = time.perf_counter()
            orc.run_window(data['wave'][i], refs, i)
            t_total += time.perf_counter() - t0
            done += 1
            if t_total > budget_s / 2:
                break
    out['single'] = dict(value=done / t_total, unit='windows/s', cores=1, kind='port',
                         sample='%d window(s) of the same workload, numpy oracle (oracle/loop.py), 1 process x 1 thread, '
                                '%.1f s of CPU work' % (done, t_total))
    # the GPU box grants this job a share of its cores (16 per GPU); more workers than that only queue.  Leg 2 uses
    # that share, leg 3 every core the affinity mask shows (Pool(os.cpu_count())), each bounded in time.
    # The oracle object is built ONCE here and inherited by the forked workers (copy-on-write: no per-worker copy of
    # the weights and the guess bank).
    global _ORC, _DATA
    _ORC, _DATA = orc, {k: np.asarray(data[k]) for k in data.files}      # arrays, not the lazy NpzFile (its file handle must not be shared by forks)
    per_win = t_total / done
    ctx = mp.get_context('fork')

    def pool_leg(workers, budget):
        tasks = int(min(4 * n, max(workers, workers * max(1, int(budget / max(per_win, 1e-3))))))
        with ctx.Pool(workers, initializer=_cpu_pool_init_shared) as pool:
            pool.map(_cpu_pool_task, [i % n for i in range(workers)], chunksize=1)       # warm every worker
            t0 = time.perf_counter()
            pool.map(_cpu_pool_task, [i % n for i in range(tasks)], chunksize=1)
            wall = time.perf_counter() - t0
        return dict(value=tasks / wall, unit='windows/s', cores=workers, kind='port',
                    sample='%d window task(s) of the same workload over multiprocessing.Pool(%d), one window per task, '
                           'one BLAS thread per worker (training.py:623-630 worker model), %.1f s wall; host has %d '
                           'logical cores, %d in this process\'s affinity mask' % (tasks, workers, wall, os.cpu_count() or 0, avail))
    share = max(1, min(avail, int(os.environ.get('AMT_CPU_WORKERS', '16'))))
    out['pool'] = pool_leg(share, budget_s / 2)
    if avail > share:
        out['pool_all_cores'] = pool_leg(avail, budget_s / 3)
    print('CPU_BASELINE ' + json.dumps(out), flush=True)


def cpu_baseline(p, wl_name, wave_cpu, refs_cpu, budget_s):
    """Oracle loop on the host cores: same algorithm, same weights, same inputs, in a child process."""
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, 'cpu_sample.npz')
        np.savez(path, wave=wave_cpu, **{'ref_' + k: v for k, v in refs_cpu.items()})
        env = dict(os.environ)
        env.pop('RANK', None)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), '--cpu-baseline-worker', wl_name, str(p.N), path,
                            str(budget_s)], capture_output=True, text=True, env=env, timeout=600)
    for line in r.stdout.splitlines():
        if line.startswith('CPU_BASELINE '):
            legs = json.loads(line[len('CPU_BASELINE '):])
            # headline = the better of the two Pool legs (its `cores` says which); all three legs are reported
            best = dict(max([legs[k] for k in ('pool', 'pool_all_cores') if k in legs], key=lambda l: l['value']))
            best['single_thread'] = legs['single']
            best['pool_gpu_share'] = legs['pool']
            best['pool_all_cores'] = legs.get('pool_all_cores')
            return best
    raise RuntimeError('cpu baseline worker failed: ' + r.stderr[-2000:])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--workload', default='c3', choices=sorted(WORKLOADS))
    ap.add_argument('--windows', type=int, default=0, help='windows per GPU (default: the workload size)')
    ap.add_argument('--cpu-seconds', type=float, default=24.0, help='CPU work of the baseline sample (both legs)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extras', action='store_true',
                    help='skip the f32-MFMA and H2D-inclusive legs (profiling runs)')
    ap.add_argument('--conv-mode', type=int, default=None, choices=(0, 1, 2, 3),
                    help='convolution arithmetic: 3 = FFT-domain form of the 4 x 16 layers on the large images + split-fp16 elsewhere '
                         '(default), 2 = split-fp16 everywhere, 1 = split-bf16, 0 = f32 MFMA; all f32-equivalent')
    args = ap.parse_args()

    from amt_saga import dist as adist, synth
    rank, world, local = adist.init()
    if not torch.cuda.is_available():
        raise RuntimeError('bench.py needs a GPU: the HIP path has no CPU fallback')
    dev = torch.device('cuda', torch.cuda.current_device())
    wl = WORKLOADS[args.workload]
    B = args.windows or wl['B']
    p, loop = build(wl, B)
    if args.conv_mode is not None:
        for n in loop.nets.values():
            n.set_mode(args.conv_mode)
    conv_mode = int(getattr(next(iter(loop.nets.values())), 'mode', 0)) if loop.nets else 0
    L = p.H * (p.timing_frames - 1)                         # 263,680 samples -> exactly 516 frames

    # synthetic windows, generated on the device (inputs resident in HBM before timing)
    wave, _ = synth.make_windows(B, L, seed=wl['seed'] * 1000 + rank, notes_per_window=wl['notes'],
                                 groups=wl['groups'], sr=p.sr, device=dev)
    window0 = rank * B
    # song-level constants once per batch (training.py:269-282 computes them once per song): untimed set-up,
    # reported as prepare_ms
    loop.prepare(wave)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loop.prepare(wave)
    torch.cuda.synchronize()
    prepare_ms = (time.perf_counter() - t0) * 1e3
    refs = loop.refs
    timing_nets = [loop.nets[k] for k in ('timing_start', 'timing_end') if k in loop.nets]
    prof_nets = timing_nets or list(loop.nets.values())

    def step(w=wave):
        ev, b = loop.run(w, window0=window0, refs=refs)
        return adist.gather_events(ev.reshape(-1, 7), n_total=B * world * wl['iters']), b

    def timed_steps(n, fn):
        """n steps bracketed by barrier + synchronize, max over ranks -> seconds."""
        adist.barrier()
        torch.cuda.synchronize()
        t = time.perf_counter()
        out = None
        for _ in range(n):
            out = None                  # release the previous batch first: the caching allocator then reuses
            out = fn()                  # its blocks instead of hipMalloc-ing inside the timed region
        torch.cuda.synchronize()
        adist.barrier()
        return adist.max_over_ranks(time.perf_counter() - t), out

    for _ in range(args.warmup):
        step()
    # ---- the timed region: the product as it runs -----------------
    dt, (events, last) = timed_steps(args.steps, step)
    value = B * world * args.steps / dt
    # ---- per-kernel measurement pass, OUTSIDE the timed region: the same steps on ONE stream, so that a kernel's duration
    # (the rooflines' denominator) is that of a kernel that owns the chip, with HIP events around every conv layer and
    # around the STFT / subtract launches (events on the launch stream = torch's current stream)
    streams_default = loop.timing_streams
    loop.timing_streams = 1
    step()
    for n in prof_nets:
        n.profile(True)
        n.profile_read(reset=True)
    from amt_saga.audio import AudioBatch
    ev_pairs = []
    orig_stft, orig_sub = AudioBatch.stft, AudioBatch.subtract

    def timed(fn, tag):
        def w(self, *a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            # (this pass only) keep the stream busy for ~1 ms before the first event: the STFT opens a step on an idle
            # GPU, and without this the host's enqueue latency (three allocations and a ctypes call, ~0.15 ms) sits
            # between the two events and is priced as kernel time (round 4: 1.105 ms by the events, 0.937 by rocprofv3)
            torch.cuda._sleep(2000000)
            e0.record()
            r = fn(self, *a, **k)
            e1.record()
            info = self.ph is not None if tag == 'stft' else None
            if tag == 'subtract':
                # what the launch touched: the whole windows, or (amt_subtract_span) the guess's frames of every window
                gfr = a[3] if len(a) > 3 else k.get('guess_frames')
                off = a[4] if len(a) > 4 else k.get('offset_frames')
                info = dict(span=self._fmax is not None and bool(k.get('span', False)), gfr=gfr, off=off,
                            guess_T=int(a[0].shape[1]), T=int(self.mag.shape[1]))
            ev_pairs.append((tag, e0, e1, info))
            return r
        return w
    AudioBatch.stft, AudioBatch.subtract = timed(orig_stft, 'stft'), timed(orig_sub, 'subtract')
    dt_one, _ = timed_steps(args.steps, step)
    AudioBatch.stft, AudioBatch.subtract = orig_stft, orig_sub
    loop.timing_streams = streams_default

    # ---- roofline of the dominant kernel (HIP events around every conv launch) -------------
    rows = []
    for n in prof_nets:
        rows += n.profile_read(reset=True)
        n.profile(False)
    by_class = {}
    for r in rows:
        key = (r['kh'], r['kw'], r['cin'], r['cout'], r['H'], r['W'])
        a = by_class.setdefault(key, dict(ms=0.0, flops=0.0, windows=0.0))
        a['ms'] += r['ms']
        a['flops'] += r['flops_per_window'] * r['windows']
        a['windows'] += r['windows']
    # conv mode 3: the 32 -> 32 (4 x 16) layers on the large images run in the FFT domain (three kernels per layer, timed
    # together by the per-layer events); the roofline object stays on the dominant SINGLE kernel -- the largest direct
    # split-fp16 class -- and the FFT-domain layers get their own object (roofline_fft: bound by the HBM traffic of the
    # frequency tensors)
    fft_keys = [k for k in by_class if conv_mode == 3 and k[:4] == (4, 16, 32, 32) and k[5] + 15 <= 576 and k[4] <= 20]
    pk_keys = [k for k in by_class if conv_mode == 3 and k == (4, 16, 64, 64, 10, 64)]       # packed-image form (amt_fftpk.hip)
    direct = {k: v for k, v in by_class.items() if k not in fft_keys and k not in pk_keys} or by_class
    dom_key = max(direct, key=lambda k: direct[k]['ms'])
    dom = by_class[dom_key]
    conv_ms_total = sum(a['ms'] for a in by_class.values())
    achieved_tf = dom['flops'] / (dom['ms'] * 1e-3) / 1e12
    arith = 2 if conv_mode == 3 else conv_mode
    split = arith == 2 or (arith == 1 and dom_key[2] <= 64)       # (the 128-channel layers run split-fp16 too; split-bf16 stops at 64)
    nmf = 3 if arith == 2 else 6                           # MFMAs per f32-equivalent product block
    # split modes: every algorithmic f32 MAC costs nmf f16 / bf16 MFMA MACs, so the MFMA roof for the
    # ALGORITHMIC flops of this kernel is the dense 16-bit peak / nmf
    peak_tf = round(MFMA_BF16_PEAK_TF / nmf, 1) if split else MFMA_F32_PEAK_TF
    if split and arith == 2 and dom_key[3] == 32:
        kname = ('conv_f16x3s_kernel<%d,%d,%d> (Cout %d) on %dx%d (3 x v_mfma_f32_16x16x32_f16 per f32 product '
                 'block of a tap pair)') % dom_key
    else:
        kname = (('conv_f16x3s_kernel<%d,%d,%d> (Cout %d, 32-wide N-slices) on %dx%d (3 x v_mfma_f32_16x16x32_f16 per f32 product block)'
                  if arith == 2 else
                  'conv_bf16x6_kernel<%d,%d,%d,%d> on %dx%d (6 x v_mfma_f32_32x32x16_bf16 per f32 product block)')
                 if split else 'conv_mfma_kernel<%d,%d,%d,%d> on %dx%d (v_mfma_f32_32x32x2_f32)') % dom_key
    roofline_fft = None
    if fft_keys:
        fk = max(fft_keys, key=lambda k: by_class[k]['ms'])
        fa = by_class[fk]
        Hh, Ww = fk[4], fk[5]
        # algorithmic HBM bytes of one chained layer and window: read Xf, write Yf (GEMM); read Yf, write Xf (inverse +
        # epilogue + forward); every second layer also reads a shortcut and writes its spatial output
        freq = 289 * Hh * 64 * 4
        spat = Hh * Ww * 32 * 4
        per_layer_window = 4 * freq + spat
        gbs = per_layer_window * fa['windows'] / (fa['ms'] * 1e-3) / 1e9
        fft_traffic = None
        try:
            pj = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')))
            per_w = lambda k: pj[k]['hbm_bytes_per_launch'] / pj[k]['windows_per_launch']
            row = per_w('fc_row')
            if 'fc_row_inregs' in pj:           # launches weighted: the layers with / without shortcut and spatial output
                n0, n1 = pj['fc_row']['dispatches_in_class'], pj['fc_row_inregs']['dispatches_in_class']
                row = (row * n0 + per_w('fc_row_inregs') * n1) / (n0 + n1)
            fft_traffic = int((per_w('fc_gemm') + row) * min(B, 1024))
        except Exception:
            fft_traffic = None
        roofline_fft = dict(bound='hbm', achieved=round(gbs, 1), peak=HBM_PEAK_GBS, unit='GB/s', frac=round(gbs / HBM_PEAK_GBS, 4),
                            traffic=fft_traffic, kernel='fc_gemm_kernel + fc_row_kernel<true> per layer (amt_fftconv.hip), %dx%d 32->32 (4x16)' % (Hh, Ww),
                            algorithmic_bytes_per_layer_window=per_layer_window,
                            ms_per_layer_per_1024_windows=round(fa['ms'] / (fa['windows'] / 1024.0), 3),
                            direct_form_equivalent_tflops=round(fa['flops'] / (fa['ms'] * 1e-3) / 1e12, 1),
                            share_of_conv_time=round(fa['ms'] / conv_ms_total, 3))
    roofline_pk = None
    if pk_keys:
        pa = by_class[pk_keys[0]]
        # one chained layer and window: read Xf, write Yf (GEMM); read Yf, write Xf (inverse + epilogue + forward):
        # 4 x 577 x 128 x 4 B; every second layer also reads a shortcut and writes its spatial output (10 x 64 x 64 x 4 B each)
        per_layer_window = 4 * 577 * 128 * 4 + 10 * 64 * 64 * 4
        gbs = per_layer_window * pa['windows'] / (pa['ms'] * 1e-3) / 1e9
        pk_traffic = None
        try:
            pj = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')))
            per_w = lambda k: pj[k]['hbm_bytes_per_launch'] / pj[k]['windows_per_launch']
            n0, n1 = pj['pk_row']['dispatches_in_class'], pj['pk_row_inregs']['dispatches_in_class']
            pk_traffic = int((per_w('pk_gemm') + (per_w('pk_row') * n0 + per_w('pk_row_inregs') * n1) / (n0 + n1)) * min(B, 1024))
        except Exception:
            pk_traffic = None
        roofline_pk = dict(bound='hbm', achieved=round(gbs, 1), peak=HBM_PEAK_GBS, unit='GB/s', frac=round(gbs / HBM_PEAK_GBS, 4),
                           traffic=pk_traffic, kernel='pk_gemm_kernel + pk_row_kernel<true> per layer (amt_fftpk.hip), 10x64 64->64 (4x16)',
                           algorithmic_bytes_per_layer_window=per_layer_window,
                           ms_per_layer_per_1024_windows=round(pa['ms'] / (pa['windows'] / 1024.0), 3),
                           direct_form_equivalent_tflops=round(pa['flops'] / (pa['ms'] * 1e-3) / 1e12, 1),
                           share_of_conv_time=round(pa['ms'] / conv_ms_total, 3))
    traffic = None
    tf = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    pmc = {}
    if os.path.exists(tf):
        try:
            # per-launch HBM bytes of the step's OWN launch of each kernel (selected by grid size,
            # scripts/make_pmc_traffic.py), FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950
            pmc = json.load(open(tf))
            t = pmc.get(('conv_f16x3' if arith == 2 else 'conv_bf16x6') if split else 'conv_mfma')
            if t and conv_mode == 3 and ('<%d, %d, %d,' % dom_key[:3]) not in t.get('kernel', ''):
                t = None                 # the file's direct-conv class is not the one this run reports
            if t and t.get('windows_per_launch'):
                traffic = int(t['hbm_bytes_per_launch'] * min(B, 1024) / t['windows_per_launch'])
        except Exception:
            traffic = None
    # Nothing in this object is a literal: `achieved` comes from HIP events of this run, the sustained-rate
    # denominator from the probe kernel run in this process right here (amt_probe_mfma_f16: back-to-back
    # v_mfma_f32_16x16x32_f16 on random non-zero register operands, two waves per SIMD, ~60 ms launches, best of 3),
    # and what only a separate rocprofv3 --pmc pass can give (HBM traffic, matrix-pipe busy share, held clock) is
    # copied verbatim from profiles/pmc_derived.json under `from_profiles`, with the files and the commit it was
    # collected at (scripts/collect_profiles.sh writes it; absent file -> null).
    sustained = {}
    if split and arith == 2 and not args.no_extras:
        import ctypes as C
        from amt_saga import _lib as alib
        tf_, ms_ = C.c_double(0.0), C.c_double(0.0)
        alib.check(alib.load().amt_probe_mfma_f16(2, 200000, 1, 3, C.byref(tf_), C.byref(ms_), None))
        sustained = {'sustained_f16_mfma_tflops_measured': round(tf_.value, 1),
                     'sustained_probe': 'amt_probe_mfma_f16(2 waves/SIMD, random operands), %.0f ms launch, this run' % ms_.value,
                     'executed_vs_sustained': round(achieved_tf * nmf / tf_.value, 3)}
    # profiles/pmc_derived.json: kernel name -> counters of a separate --pmc pass (scripts/pmc_conv.sh, pmc_fftconv.sh); an
    # object only carries the entry of the kernel it names
    derived = {}
    pd = os.path.join(ROOT, 'profiles', 'pmc_derived.json')
    if os.path.exists(pd):
        try:
            derived = json.load(open(pd))
        except Exception:
            derived = {}

    def derived_for(*patterns):
        out = {k: v for k, v in derived.items() if isinstance(v, dict) and any(pt in k for pt in patterns)}
        return out or None
    from_profiles = derived_for('conv_f16x3s_kernel<%d, %d, %d,' % dom_key[:3]) if (split and arith == 2) else None
    if roofline_fft:
        roofline_fft['from_profiles'] = derived_for('fc_gemm_kernel', 'fc_row_kernel<true')
    if roofline_pk:
        roofline_pk['from_profiles'] = derived_for('pk_gemm_kernel', 'pk_row_kernel<true')
    roofline = dict(bound='mfma', achieved=round(achieved_tf, 2), peak=peak_tf, unit='TFLOP/s',
                    frac=round(achieved_tf / peak_tf, 4), traffic=traffic, kernel=kname,
                    peak_note=('dense bf16/f16 MFMA peak 2500 / %d MFMAs per f32-equivalent product block' % nmf
                               if split else 'dense f32 MFMA peak'),
                    executed_mfma_tflops=round(achieved_tf * (nmf if split else 1), 1),
                    **sustained,
                    vs_f32_mfma_peak=round(achieved_tf / MFMA_F32_PEAK_TF, 3),
                    share_of_conv_time=round(dom['ms'] / conv_ms_total, 3),
                    conv_ms_per_step=round(conv_ms_total / args.steps, 2),
                    from_profiles=from_profiles, profiles_provenance=derived.get('_provenance'))
    # ---- the north star's HBM pair: only the bytes the step consumes ------------------------
    F, T, ldf = p.N // 2 + 1, p.timing_frames, (p.N // 2 + 1 + 3) & ~3
    stft_ms = sum(e0.elapsed_time(e1) for tag, e0, e1, _ in ev_pairs if tag == 'stft')
    sub_ms = sum(e0.elapsed_time(e1) for tag, e0, e1, _ in ev_pairs if tag == 'subtract')
    n_stft = sum(1 for tag, _, _, _ in ev_pairs if tag == 'stft')
    n_sub = sum(1 for tag, _, _, _ in ev_pairs if tag == 'subtract')
    with_phase = any(ph for tag, _, _, ph in ev_pairs if tag == 'stft')
    stft_bytes = B * (4 * L + 4 * F * T + (8 * F * T if with_phase else 0))     # wave in, mag (+ unit phase) out
    # subtract: residual read + written, guess read -- over the whole window (amt_subtract) or over the frames the guess
    # covers (amt_subtract_span: sum over the windows of min(offset + guess frames, T) - offset, from the launch's own tables)
    sub_total, span_launches, span_frames = 0, 0, 0
    for tag, _, _, info in ev_pairs:
        if tag != 'subtract':
            continue
        if info['span'] and isinstance(info['gfr'], torch.Tensor) and isinstance(info['off'], torch.Tensor):
            o = info['off'].clamp(min=0).long()
            fr = (torch.minimum(o + info['gfr'].long(), torch.full_like(o, info['T'])) - o).clamp(min=0)
            nfr = int(fr.sum().item())
            sub_total += 3 * 4 * F * nfr + 2 * 4 * nfr + 4 * B * info['T']
            span_launches += 1
            span_frames += nfr
        else:
            sub_total += B * (2 * 4 * F * T + 4 * F * loop.bank_frames)
    sub_bytes = sub_total / n_sub if n_sub else 0
    hbm_ms = stft_ms + sub_ms
    hbm_gbs = (n_stft * stft_bytes + n_sub * sub_bytes) / (hbm_ms * 1e-3) / 1e9 if hbm_ms > 0 else 0.0
    stft_traffic = None
    try:
        key = 'stft' if with_phase else 'stft_mag_only'
        stft_traffic = int(pmc[key]['hbm_bytes_per_launch'] * B / pmc[key]['windows_per_launch'] +
                           (pmc['subtract_span' if span_launches else 'subtract']['hbm_bytes_per_launch'] * B /
                            pmc['subtract_span' if span_launches else 'subtract']['windows_per_launch'] if n_sub else 0))
    except Exception:
        stft_traffic = None
    roofline_stft = dict(bound='hbm', achieved=round(hbm_gbs, 1), peak=HBM_PEAK_GBS, unit='GB/s',
                         frac=round(hbm_gbs / HBM_PEAK_GBS, 4), traffic=stft_traffic,
                         algorithmic_bytes=int(stft_bytes + sub_bytes),
                         kernel='stft_mag_kernel<2048,%s> + %s' % ('phase' if with_phase else 'mag only',
                                                                   'subtract_span_kernel' if span_launches else 'subtract_kernel'),
                         phase_plane_stored=bool(with_phase),
                         stft_gbs=round(n_stft * stft_bytes / (stft_ms * 1e-3) / 1e9, 1) if stft_ms else None,
                         subtract_gbs=round(n_sub * sub_bytes / (sub_ms * 1e-3) / 1e9, 1) if sub_ms else None,
                         subtract_form=('span: %.1f of %d frames per window' % (span_frames / max(span_launches, 1) / B, T)
                                        if span_launches else 'whole window'),
                         ms_per_step=round(hbm_ms / args.steps, 3))

    # ---- extra legs, same run ----------------------------------------------------------------
    extras = {}
    if not args.no_extras and 'timing' in loop.heads and loop.timing_streams == 1:
        # (0) timing_end on a second stream under timing_start (opt-in mode of the product)
        loop.timing_streams = 2
        step()
        k2s = max(1, min(args.steps, 5))
        dt2s, (ev2s, _) = timed_steps(k2s, step)
        loop.timing_streams = 1
        extras['value_two_timing_streams'] = round(B * world * k2s / dt2s, 2)
        extras['two_timing_streams_same_events'] = bool(torch.equal(ev2s, events))
    if not args.no_extras:
        # (1) strict-f32 convolutions
        if conv_mode != 0 and loop.nets:
            for n in loop.nets.values():
                n.set_mode(0)
            step()
            k0 = max(1, min(args.steps, 3))
            dt0, _ = timed_steps(k0, step)
            extras['value_f32_mfma'] = round(B * world * k0 / dt0, 2)
            for n in loop.nets.values():
                n.set_mode(conv_mode)
            step()
        # (1a) the step INCLUDING the per-batch song-level normalisers (prepare(): whole-CQT maxima of every window,
        # each window standing for its song -- training.py:271-282 computes them once per song): what a stream of
        # fresh batches costs when nothing about a batch is known in advance
        def step_fresh():
            ev, b = loop.run(wave, window0=window0, refs=None)
            return adist.gather_events(ev.reshape(-1, 7), n_total=B * world * wl['iters']), b
        step_fresh()
        kp = max(1, min(args.steps, 5))
        dtp, _ = timed_steps(kp, step_fresh)
        extras['value_prepare_inclusive'] = round(B * world * kp / dtp, 2)
        # (2) host -> HBM copy of the batch's audio inside the timed region (pinned host memory, one copy
        # per step on the launch stream, no overlap with compute)
        host = torch.empty((B, L), dtype=torch.float32).pin_memory()
        host.copy_(wave)
        stage = torch.empty_like(wave)

        def step_h2d():
            stage.copy_(host, non_blocking=True)
            return step(stage)
        step_h2d()
        k1 = max(1, min(args.steps, 5))
        dt1, _ = timed_steps(k1, step_h2d)
        extras['value_h2d_inclusive'] = round(B * world * k1 / dt1, 2)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); stage.copy_(host, non_blocking=True); e1.record(); torch.cuda.synchronize()
        extras['h2d_gbs'] = round(B * L * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)
        # (3) the same with the copy of batch i+1 overlapped with the compute of batch i
        # (TranscriptionLoop.run_stream: copy stream + two staging buffers); steady state: the first batch's
        # copy is issued before the clock starts, k1 copies and k1 computes lie inside it
        gen = loop.run_stream((host for _ in range(k1 + 2)), refs=refs, window0=rank * B)
        next(gen)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k1):
            ev_s, _b = next(gen)
            adist.gather_events(ev_s.reshape(-1, 7), n_total=B * world * wl['iters'])
            ev_s = _b = None
        torch.cuda.synchronize()
        dt2 = time.perf_counter() - t0
        gen.close()
        torch.cuda.synchronize()
        extras['value_h2d_overlapped'] = round(B * world * k1 / dt2, 2)
        del host, stage

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        n_cpu = min(B, 64)
        refs_cpu = {k: v[:n_cpu].cpu().numpy() for k, v in refs.items()}
        cpu = cpu_baseline(p, args.workload, wave[:n_cpu].cpu().numpy(), refs_cpu, args.cpu_seconds)

    if rank == 0:
        out = {
            'metric': 'audio windows/sec (44.1 kHz, 2048-pt STFT) through full detect->subtract loop',
            'value': round(value, 2), 'unit': 'windows/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 2),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': {0: 'f32', 1: 'f32 (conv: split-bf16 x3 operands, f32 accumulate)',
                      2: 'f32 (conv: split-fp16 x2 operands, f32 accumulate)',
                      3: 'f32 (conv: split-fp16 x2 operands, f32 accumulate; 4x16 layers on the large images in the FFT domain)'}[conv_mode],
            'data': 'synthetic (additive-synth windows, seeded synthetic weights with calibrated BN statistics)',
            'config': {'workload': wl['name'], 'windows_per_gpu': B, 'n_fft': p.N, 'hop': p.H,
                       'frames': T, 'iters': wl['iters'], 'heads': list(wl['heads']),
                       'parallelism': 'windows sharded x%d, event all-gather' % world},
            # the dominant kernel(s) of the step: in conv mode 3 the FFT-domain layer pair (GEMM + row transforms; bound by
            # the HBM traffic of the frequency tensors), else the largest direct convolution class (matrix pipe), which in
            # mode 3 is reported as roofline_mfma
            'roofline': roofline_fft if (roofline_fft and roofline_fft['share_of_conv_time'] > roofline['share_of_conv_time']) else roofline,
            'roofline_mfma': roofline if (roofline_fft and roofline_fft['share_of_conv_time'] > roofline['share_of_conv_time']) else None,
            'roofline_pk': roofline_pk,
            'roofline_stft': roofline_stft, 'cpu_baseline': cpu,
            # what `value` is: the bench contract's rate -- inputs resident in HBM when the clock starts, event all-gather
            # inside.  SURVEY 8d words the metric with the host -> HBM copy of the audio inside: that figure is
            # value_h2d_overlapped (copy of batch i + 1 under the compute of batch i, TranscriptionLoop.run_stream) /
            # value_h2d_inclusive (serial copy)
            'value_definition': 'inputs resident in HBM, event all-gather inside the timed region (bench contract); the '
                                'PCIe-inclusive rates of SURVEY 8d are value_h2d_overlapped / value_h2d_inclusive',
            'rccl_world': adist.world_size_seen(), 'windows_per_gpu': [B] * world,
            'timing_streams': streams_default,
            # the measurement pass behind the roofline objects: the same steps on one stream with HIP events around every
            # conv layer / STFT / subtract launch (outside the timed region)
            'value_one_stream_profiled': round(B * world * args.steps / dt_one, 2),
            'prepare_ms': round(prepare_ms, 1),
            'events_checksum': int(events.to(torch.int64).sum().item()),
            'distinct_decisions': {k: int(torch.unique(events[:, c]).numel())
                                   for c, k in ((2, 'pitch'), (4, 'velocity'), (5, 'onset_frame'), (6, 'end_frame'))},
            'arithmetic_note': {
                0: 'convolutions on the f32 matrix pipe (v_mfma_f32_32x32x2_f32)',
                1: 'f32-equivalent: every f32 operand split exactly into 3 bf16 terms, 6 MFMAs per product '
                   'block, f32 accumulate; same parity bars as --conv-mode 0 (tests/test_gpu_rdcnn.py)',
                2: 'f32-equivalent: every f32 operand = f16 h + 2^-11 f16 l (22 significand bits, per-window power-of-two '
                   'range scaling from measured maxima), 3 MFMAs per product block, f32 accumulate, ~1e-7 rms / <= 2^-21 '
                   'worst case per product; same parity bars as --conv-mode 0 (tests/test_gpu_rdcnn.py); value_f32_mfma '
                   'is the same step on the f32 MFMA',
            }[arith] + (' The 32->32 (4x16) layers on the 20x516 images run in the FFT domain (amt_fftconv.hip): 576-point f32 FFTs '
                        'of channel pairs, per-frequency-pair GEMMs in the same split-fp16 arithmetic; same parity bars.' if conv_mode == 3 else ''),
        }
        out.update(extras)
        if cpu:
            out['speedup_vs_cpu_baseline'] = round(value / cpu['value'], 1)
        print(json.dumps(out), flush=True)
    adist.shutdown()


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == '--cpu-baseline-worker':
        cpu_baseline_main(sys.argv[2], int(sys.argv[3]), sys.argv[4], float(sys.argv[5]))
    else:
        main()
