#!/usr/bin/env python3
"""Headline benchmark: DeepID2 Joint identification-verification training images/sec.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step is one minibatch (128 images per GPU, 55x55x3, 10575 identities; net =
cuda-convnet_amd/cfg/deepid2_10575.cfg, the reference's deepid2/net-define-Joint-m0-10575.cfg)
through fprop + bprop (softmax/logreg + L2-SN verification cost) + momentum/weight-decay update,
fp32, on the `_ConvNet` extension (C++ layer graph -> C ABI -> hand-written HIP kernels).
The synthetic data batch is resident in HBM before the timed region starts.  With N > 1 every
rank is one replica (weak scaling, 128 images per GPU); weight gradients are all-reduced over
RCCL inside the step.  The launcher plumbing (barrier, broadcast of the RCCL unique id, max over
ranks) is a loopback TCP group (cuda_convnet_amd/rendezvous.py); torch is not imported.

Prints ONE JSON line on rank 0 (see the README of the driver contract), including
  roofline     -- the operation that takes the most device time in a step, timed live with HIP
                  events on the net's own stream (the `_ConvNet` profiling pass right after the timed
                  region, same process, same shapes) against its algorithmic FLOPs / bytes;
  kernels      -- the same figures for every operation of the step;
  cpu_baseline -- the CPU oracle (oracle/, a multithreaded C++ restatement of the reference
                  semantics) timed on this host's cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_HBM_GBS = 8000.0          # HBM3E spec


def op_work(layers, label, N, fused_pools=()):
    """Algorithmic work of one profiled operation: (flops, bytes, class).  class 'mfma' = a product
    (conv / local / FC): roofline_entry prices it against BOTH roofs and reports the one that binds.  fused_pools: names of
    max-pool layers whose undo (and the relu gradient under it) ran inside the backward-data launch of the layer above
    (mcn_img_acts_max_undo): that launch reads the pool's input and output and writes the pool input's gradient instead of
    writing its own target, and its algorithmic bytes say so."""
    lname, op = label.rsplit('.', 1)
    idx = 0
    if '[' in op:
        op, i = op[:-1].split('[')
        idx = int(i)
    if lname == 'net' and op == 'update':  # every matrix in one launch
        nW = sum(w.size for x in layers if 'weights' in x for w in x['weights'])
        nConv = sum(w.size for x in layers if x['type'] == 'conv' for w in x['weights'])
        nB = sum(x['biases'].size for x in layers if 'weights' in x)
        return 3.0 * nW, 4.0 * (4 * nW + nConv + 5 * nB), 'hbm'
    if lname == 'net' and op == 'biasGrads':  # the row sums of every layer's gradient matrix in one call (mcn_add_sum_multi)
        n = sum(N * x['outputs'] for x in layers if 'weights' in x)
        return n, 4.0 * n, 'hbm'
    l = next((x for x in layers if x['name'] == lname), None)
    if l is None:  # an operation of the net itself that has no entry above
        return 0.0, 0.0, 'hbm'
    t = l['type']
    if t in ('conv', 'local'):
        M, F = l['modules'], l['filters']
        C, fc, fp = l['channels'][idx], l['filterChannels'][idx], l['filterPixels'][idx]
        img, out, nW = N * C * l['imgPixels'][idx], N * F * M, l['weights'][idx].size
        if op == 'bpropActs' and layers[l['inputs'][idx]]['name'] in fused_pools:
            pool_in = N * layers[l['inputs'][idx]]['numInputs'][0]
            return 2.0 * N * M * F * fc * fp, 4.0 * (out + nW + img + 2 * pool_in), 'mfma'
        if op in ('fprop', 'bpropActs', 'bpropWeights'):
            return 2.0 * N * M * F * fc * fp, 4.0 * (img + out + nW), 'mfma'
        if op == 'bpropBiases':
            return out, 4.0 * out, 'hbm'
    if t == 'fc':
        I, O = l['numInputs'][idx], l['outputs']
        if op in ('fprop', 'bpropActs', 'bpropWeights'):
            return 2.0 * N * I * O, 4.0 * (N * I + N * O + I * O), 'mfma'
        if op == 'bpropBiases':
            return N * O, 4.0 * N * O, 'hbm'
    if op == 'update':
        n = sum(w.size for w in l['weights'])
        per = 5 if t == 'conv' else 4  # useGrad: 3R + 2W, else 2R + 2W
        return 3.0 * n, 4.0 * (per * n + 5 * l['biases'].size), 'hbm'
    if t == 'pool':
        inp, out = N * l['numInputs'][0], N * l['outputs']
        if op == 'fprop':
            return inp, 4.0 * (inp + out), 'hbm'
        return inp, 4.0 * (2 * inp + 2 * out), 'hbm'  # max undo: imgs, maxActs, maxGrads -> target
    if t == 'neuron':
        n = N * l['outputs']
        return n, (8.0 if op == 'fprop' else 12.0) * n, 'hbm'
    if t == 'softmax':
        return 4.0 * N * l['outputs'], 8.0 * N * l['outputs'], 'hbm'
    if t in ('rnorm', 'cmrnorm'):
        n = N * l['outputs']
        return 2.0 * l['size'] * n, (12.0 if op == 'fprop' else 24.0) * n, 'hbm'
    if t == 'cost.logreg':
        K = layers[l['inputs'][1]]['outputs']
        return N * K, 4.0 * N * K, 'hbm'
    if t in ('cost.l2-sn', 'cost.l2-reg'):
        D = layers[l['inputs'][1]]['outputs']
        return 3.0 * N * D, (4.0 if op == 'fprop' else 12.0) * N * D, 'hbm'
    n = N * l['outputs']
    return n, 8.0 * n, 'hbm'


def roofline_entry(layers, label, count, ms, N, fused_pools=()):
    flops, nbytes, bound = op_work(layers, label, N, fused_pools)
    t = ms / count * 1e-3
    e = {'op': label, 'launches': count, 'avg_us': round(t * 1e6, 2), 'bound': bound}
    if bound == 'mfma':
        a = flops / t / 1e12
        e.update(achieved=round(a, 3), peak=PEAK_FP32_MFMA_TFLOPS, unit='TFLOP/s', frac=round(a / PEAK_FP32_MFMA_TFLOPS, 4),
                 hbm_GBs=round(nbytes / t / 1e9, 1), mfma_frac=round(a / PEAK_FP32_MFMA_TFLOPS, 4))
        # the roof that binds is the one with the longer minimum time: a product that carries a pool undo (or a first layer
        # with 48-deep filters) moves more bytes per flop than the ridge of 157.3 TFLOP/s : 8 TB/s = 20 flop/byte allows
        if nbytes / (PEAK_HBM_GBS * 1e9) > flops / (PEAK_FP32_MFMA_TFLOPS * 1e12):
            g = nbytes / t / 1e9
            e.update(bound='hbm', achieved=round(g, 1), peak=PEAK_HBM_GBS, unit='GB/s', frac=round(g / PEAK_HBM_GBS, 4),
                     mfma_TFLOPs=round(a, 3))
    else:
        a = nbytes / t / 1e9
        e.update(achieved=round(a, 1), peak=PEAK_HBM_GBS, unit='GB/s', frac=round(a / PEAK_HBM_GBS, 4))
    return e


def host_breakdown(profile, steps):
    """Where the submitting threads' time goes in an untimed extra block of `steps` steps (getProfile `host.*`, microseconds
    per step): the main thread's launch calls, the time it is blocked on the side thread or on stream ticks, its event calls,
    and the same for the side thread."""
    out = {}
    for k, (c, ms) in sorted(profile.items()):
        if k.startswith('host.'):
            out[k[5:]] = round(ms * 1e3 / max(steps, 1), 2)
            if k.endswith('launch_calls'):
                out[k[5:].replace('launch_calls', 'launches')] = round(c / max(steps, 1), 1)
    return out


def cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_baseline(preset, budget_s=9.0, budget_one_s=6.0):
    """The whole-net CPU oracle on a bounded sample of the same workload (images/sec on host cores): all cores the
    process may use, then ONE core, with the CPU model."""
    import ctypes
    import ccn_loader
    ccn_loader.load()
    from cuda_convnet_amd import CFG_DIR, layer
    from cuda_convnet_amd.convnet import PRESETS
    from cuda_convnet_amd.data import SyntheticDataProvider
    from oracle.net import OracleNet
    ldef, lpar, kind, img = PRESETS[preset]
    dp = SyntheticDataProvider(kind, 128, img=img)
    dims = {i: dp.get_data_dims(i) for i in range(len(dp.dims))}
    L = layer.parse_layers(os.path.join(CFG_DIR, ldef), os.path.join(CFG_DIR, lpar), dims, dp.get_num_classes(), seed=3)
    net = OracleNet(L, 128)
    data = dp.get_next_batch()[2]
    cores = int(os.environ.get('OMP_NUM_THREADS', 0)) or len(os.sched_getaffinity(0))
    try:
        gomp = ctypes.CDLL('libgomp.so.1')
    except OSError:
        gomp = None

    def sample(threads, budget):
        if gomp is not None:
            gomp.omp_set_num_threads(threads)
        net.run_batch(data)  # warm-up (page in the library, first-touch the buffers)
        t0 = time.time()
        net.run_batch(data)
        one = time.time() - t0
        reps = int(max(1, min(512, budget / max(one, 1e-3))))
        t0 = time.time()
        for _ in range(reps):
            net.run_batch(data)
        return reps * 128 / (time.time() - t0), reps

    v_all, r_all = sample(cores, budget_s)
    out = {'value': round(v_all, 1), 'unit': 'images/sec', 'cores': cores, 'kind': 'port', 'cpu_model': cpu_model(),
           'sample': '%d training minibatches of 128 (fprop+bprop+update), same net and synthetic data' % r_all}
    if gomp is not None:
        v_one, r_one = sample(1, budget_one_s)
        out['one_core'] = {'value': round(v_one, 1), 'unit': 'images/sec', 'cores': 1, 'sample': '%d training minibatches of 128' % r_one}
        gomp.omp_set_num_threads(cores)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--workload', default='deepid2', help='preset name (cuda_convnet_amd.convnet.PRESETS)')
    ap.add_argument('--minibatch', type=int, default=128)
    ap.add_argument('--resident-minibatches', type=int, default=16)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-profile', action='store_true')
    ap.add_argument('--force-dp', action='store_true', help='run the replica-exchange path even with one rank (rehearsal)')
    ap.add_argument('--profile-steps', type=int, default=20)
    ap.add_argument('--min-seconds', type=float, default=0.5, help='repeat the timed block of --steps steps until this much has been timed; 0: once')
    ap.add_argument('--host-data', action='store_true', help='(default since round 3; kept for old command lines)')
    ap.add_argument('--no-host-data', action='store_true',
                    help='skip the second loop in which every data batch comes from host memory (prefetchBatch: uploaded over the copy '
                         'stream while the previous batch trains), reported as upload_inclusive next to the resident value: what the '
                         "reference's own logs time (gpumodel.py:145-162, data.cu:72-98)")
    ap.add_argument('--pageable', action='store_true', help='--host-data: plain numpy arrays instead of page-locked ones')
    a = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('MCN_BENCH_DEVICE', os.environ.get('LOCAL_RANK', 0)))  # (override: rehearsing ranks on one GPU)
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit('bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d' % (a.gpus, a.gpus))
        sys.exit('--gpus %d does not match WORLD_SIZE %d' % (a.gpus, world))

    # stdout carries exactly one JSON line: whatever native libraries print (RCCL's version banner) goes to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import numpy as np

    import ccn_loader
    ccn_loader.load()
    from cuda_convnet_amd import CFG_DIR, layer
    from cuda_convnet_amd.convnet import PRESETS, import_native
    from cuda_convnet_amd.data import SyntheticDataProvider

    from cuda_convnet_amd.rendezvous import Group

    lib = import_native()
    group = Group(rank, world)  # loopback control plane; no torch in the GPU processes (see rendezvous.py)
    ldef, lpar, kind, img = PRESETS[a.workload]
    mb = a.minibatch
    cases = mb * a.resident_minibatches
    # every rank draws its own images (seed offset), identical initial weights (same parser seed)
    dp = SyntheticDataProvider(kind, cases, img=img, minibatch=mb, seed=1 + 1000 * rank)
    dims = {i: dp.get_data_dims(i) for i in range(len(dp.dims))}
    layers = layer.parse_layers(os.path.join(CFG_DIR, ldef), os.path.join(CFG_DIR, lpar), dims, dp.get_num_classes(), seed=3)
    data = dp.get_next_batch()[2]

    lib.initModel(layers, mb, local)
    if world > 1 or a.force_dp:
        uid = group.broadcast(lib.commUniqueId() if rank == 0 else None)
        lib.initDataParallel(rank, world, uid, 1 if a.force_dp else 0)
    lib.loadBatch(data)  # inputs resident in HBM from here on

    barrier = group.barrier

    lib.startSteps(a.warmup, 0, 0)
    lib.finishBatch()

    def timed_block():
        """EXACTLY a.steps steps between two barriers, the device drained on both sides; max over ranks"""
        barrier()
        t0 = time.perf_counter()
        lib.startSteps(a.steps, 0, 0)
        costs, ncases = lib.finishBatch()  # blocks until the stream has drained (device synchronised)
        barrier()
        dt = time.perf_counter() - t0
        hp = lib.getProfile()
        assert ncases == a.steps * mb
        for k, v in costs.items():
            assert all(np.isfinite(v)), ('non-finite cost', k, v)
        return group.max(dt), hp, costs, ncases

    # K steps at 0.25 ms are a 5 ms region: one clock ramp or stall moves it by percent.  The block is repeated until at least
    # --min-seconds have been timed (the count follows from the first block's max-over-ranks time, so every rank agrees) and
    # the MEDIAN block is reported, with the spread.
    blocks = [timed_block()]
    repeats = 1 if a.min_seconds <= 0 else int(min(200, max(1, -(-a.min_seconds // max(blocks[0][0], 1e-6)))))
    while len(blocks) < repeats:
        blocks.append(timed_block())
    blocks.sort(key=lambda b: b[0])
    dt, host_prof, costs, ncases = blocks[len(blocks) // 2]
    dt_min, dt_max = blocks[0][0], blocks[-1][0]
    host_enqueue_ms = host_prof.get('host.enqueue', (0, 0.0))[1]

    inclusive = None
    if world == 1 and not a.no_host_data:
        # distinct host batches, cycled; batch k+1 is uploaded while batch k trains (the reference uploads on the compute
        # thread: data.cu:72-98); the timed region starts with batch 0 still on the host
        nb = max(8, -(-a.steps // a.resident_minibatches))
        host = []
        for b in range(3):
            d = SyntheticDataProvider(kind, cases, img=img, minibatch=mb, seed=50 + b).get_next_batch()[2]
            if not a.pageable:
                pin = [lib.pinnedArray(*x.shape) for x in d]
                for p_, x in zip(pin, d):
                    p_[...] = x
                d = pin
            host.append(d)
        lib.startBatch(host[0], 0)
        lib.finishBatch()
        t0 = time.perf_counter()
        lib.startBatch(host[0], 0)
        for k in range(1, nb):
            lib.prefetchBatch(host[k % 3])
            lib.finishBatch()
            lib.startBatch([], 0)
        lib.finishBatch()
        dti = time.perf_counter() - t0
        inclusive = {'value': round(nb * cases / dti, 1), 'unit': 'images/sec', 'batches': nb, 'cases_per_batch': cases,
                     'host_memory': 'pageable' if a.pageable else 'page-locked', 'bytes_per_image': int(sum(x.shape[0] for x in host[0]) * 4)}

    prof, prof_min = {}, {}
    if not a.no_profile:
        # three passes; per operation the MEDIAN pass is what is reported and summed (a step that ran), the fastest pass is
        # kept beside it as min_us: one stalled launch (a clock ramp, another tenant of the box) in 20 would otherwise name
        # the wrong operation as the dominant one, and a sum of per-operation minima is a step that never ran
        passes = {}
        for _ in range(3):
            lib.startSteps(a.profile_steps, 0, 1)
            lib.finishBatch()
            for k, (c, ms) in lib.getProfile().items():
                passes.setdefault(k, []).append((c, ms))
        for k, v in passes.items():
            v.sort(key=lambda x: x[1] / max(x[0], 1))
            prof[k] = v[len(v) // 2]
            prof_min[k] = v[0][1] / max(v[0][0], 1) * 1e3
    rccl = None
    if world > 1 or a.force_dp:
        # what the replica group's communicator says about itself (ncclCommCount: the ranks RCCL saw, not the ranks the
        # launcher started), the buckets of one step, and the collectives' time on the exchange stream (events around every
        # call, in an untimed extra block)
        nranks, ver, is_rccl = lib.commInfo()
        lib.getCommLog()
        lib.startSteps(a.profile_steps, 0, 3)
        lib.finishBatch()
        cp = lib.getProfile()
        log = lib.getCommLog()
        per = max(len(log) // max(a.profile_steps, 1), 1)
        step0 = log[:per]
        rccl = {'nranks': nranks, 'version': ver, 'transport': 'rccl' if is_rccl else 'shared-memory rehearsal (test build)',
                'buckets_bytes': [4 * c for k, o, c in step0 if k == 'allreduce'],
                'allgather_bytes_per_rank': [4 * c for k, o, c in step0 if k == 'allgather'],
                'allreduce_us_per_step': round(cp.get('comm.allreduce', (0, 0.0))[1] * 1e3 / max(a.profile_steps, 1), 2),
                'allgather_us_per_step': round(cp.get('comm.allgather', (0, 0.0))[1] * 1e3 / max(a.profile_steps, 1), 2),
                'collectives_per_step': per}
    breakdown = None
    if world == 1 and not a.no_profile:
        lib.startSteps(a.steps, 0, 0)
        lib.finishBatch()
        breakdown = host_breakdown(lib.getProfile(), a.steps)
    lib.destroyModel()

    if rank == 0:
        out = {
            'metric': 'DeepID2 train images/sec (node)', 'value': round(a.steps * mb * world / dt, 1), 'unit': 'images/sec',
            'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': round(dt / a.steps * 1e3, 4),
            'repeats': len(blocks), 'ms_per_step_min': round(dt_min / a.steps * 1e3, 4), 'ms_per_step_max': round(dt_max / a.steps * 1e3, 4),
            'host_enqueue_ms_per_step': round(host_enqueue_ms / a.steps, 4), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': '%s (%s) fprop+bprop+update, %dx%dx%d, %d ids, minibatch %d/GPU'
                                   % (a.workload, ldef, img[0], img[1], img[2], dp.get_num_classes(), mb),
                       'global_batch': mb * world, 'parallelism': 'dp%d' % world,
                       'final_costs': {k: [x / ncases for x in v] for k, v in costs.items()}},
        }
        if prof:
            prof = {k: v for k, v in prof.items() if not k.startswith(('host.', 'phase.'))}
            # a max pool whose backward span launched nothing had its undo done by the layer above (spans without launches are dropped)
            fused_pools = {l['name'] for l in layers if l['type'] == 'pool' and l.get('pool') == 'max'
                           and '%s.fprop[0]' % l['name'] in prof and not any(k.startswith(l['name'] + '.bpropActs') for k in prof)}
            ks = [roofline_entry(layers, lab, c, ms, mb, fused_pools) for lab, (c, ms) in prof.items() if c > 0]
            for e in ks:
                e['min_us'] = round(prof_min[e['op']], 2)
            ks.sort(key=lambda e: -e['avg_us'] * e['launches'])
            step_us = sum(e['avg_us'] * e['launches'] for e in ks) / a.profile_steps
            # the dominant launch among those a roofline says anything about: the L2-SN cost (pair statistics of 160 x 128
            # features, in data-parallel mode an all-gather) is latency, SURVEY 8d files it under "latency / launch"
            big = [e for e in ks if op_work(layers, e['op'], mb, fused_pools)[1] >= 1e6]
            top = dict((big or ks)[0])
            out['roofline'] = {'bound': top['bound'], 'achieved': top['achieved'], 'peak': top['peak'], 'unit': top['unit'],
                               'frac': top['frac'], 'traffic': None, 'kernel': top['op'], 'avg_us': top['avg_us']}
            # HBM bytes per launch from the PMC counters: separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), committed
            # under profiles/ with the gfx950 correction applied (they cannot be collected inside this process)
            pmc = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'r03_pmc_traffic_by_op.json')
            if a.workload in ('deepid2', 'deepid2plus') and mb == 128 and os.path.exists(pmc):
                rec = json.load(open(pmc)).get(top['op'] if a.workload == 'deepid2' else 'deepid2plus ' + top['op'])
                # a record is only as good as the kernel it was taken from: it names the device symbol, which must still be
                # in the library this run loaded (round 1 shipped a record of a kernel variant the tree no longer launched)
                from cuda_convnet_amd import capi
                blob = open(capi.LIB_PATH, 'rb').read()
                if rec and all(sym.encode() in blob for sym in rec['kernel_symbols']):
                    out['roofline']['traffic'] = rec['traffic_bytes_per_launch']
                    out['roofline']['traffic_unit'] = 'bytes/launch (PMC, corrected); algorithmic %d' % rec['algorithmic_bytes_per_launch']
                elif rec:
                    out['roofline']['traffic_unit'] = 'stale PMC record dropped: %s not in this build' % ','.join(rec['kernel_symbols'])
            out['device_us_per_step'] = round(step_us, 1)
            # second half of BASELINE.json's metric: filterActs (conv / local forward) as a fraction of the fp32 MFMA peak
            conv_names = {l['name'] for l in layers if l['type'] in ('conv', 'local')}
            out['filter_acts_frac_of_fp32_mfma_peak'] = {e['op']: e.get('mfma_frac', e['frac']) for e in ks
                                                         if e['op'].split('.')[0] in conv_names and '.fprop[' in e['op']}
            out['kernels'] = ks
        if inclusive:
            inclusive['fraction_of_resident'] = round(inclusive['value'] / out['value'], 3)
            out['upload_inclusive'] = inclusive
        if breakdown:
            out['host_breakdown_us_per_step'] = breakdown
        if rccl:
            out['rccl'] = rccl
            if rccl['transport'] != 'rccl':  # a rehearsal of the launch line on one GPU: not a measurement of anything
                out['rehearsal_images_per_sec'] = out['value']
                out['value'] = None
        if world == 1 and not a.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(a.workload)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + '\n').encode())
    group.barrier()
    group.close()


if __name__ == '__main__':
    main()
