#!/usr/bin/env python3
"""profiles/r03_pmc_traffic_by_op.json from the FETCH_SIZE / WRITE_SIZE passes of pmc_round3.sh (gpurun_out/pmc_<tag>_{fetch,write}.csv).

Per bench op label: the kernels one call of that op launches, their HBM traffic per launch (PMC, corrected as
MI355X_MICROARCH.md's HBM / rocprofv3 section prescribes for gfx950: FETCH_SIZE counts 64 B per 128 B request of the wide reads
these kernels issue -> x2; WRITE_SIZE exact; both in KB), the algorithmic bytes of the same call, and the mangled symbols of the
kernels -- bench.py drops a record whose symbols are no longer in libmiconvnet.so (a stale record must not be quoted)."""
import csv
import json
import os
import re
import shutil
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, 'gpurun_out')
PROF = os.path.join(ROOT, 'profiles')
N = 128
F4 = 4


def conv_bytes(imgY, imgX, C, F, fs, mY, mX, conv):
    img, hid = C * imgY * imgX * N * F4, F * mY * mX * N * F4
    w = C * fs * fs * F * F4 * (1 if conv else mY * mX)
    return img, hid, w


CONV2 = conv_bytes(26, 26, 16, 32, 3, 24, 24, True)
LOCAL3 = conv_bytes(12, 12, 32, 48, 3, 10, 10, False)
LOCAL4 = conv_bytes(5, 5, 48, 64, 2, 4, 4, False)
CONV1 = conv_bytes(55, 55, 3, 16, 4, 52, 52, True)
CONV2P = conv_bytes(26, 26, 32, 128, 3, 24, 24, True)
LOCAL3P = conv_bytes(12, 12, 128, 128, 3, 10, 10, False)
FC6 = (10575 * 160 * F4, 160 * N * F4, 10575 * N * F4)  # W, X, Y
ROUND = 'r03'


def undo_bytes(g):  # imgActs continued through pool + relu: hidActs, W, the pool's input (4x the image) and output, the pool input's gradient
    return g[1] + g[2] + 4 * g[0] + g[0] + 4 * g[0]


# op label: (tag, [kernel name prefixes as rocprofv3 prints them], algorithmic bytes per call, note)
OPS = {
    'conv2.bpropActs[0]': ('iau2', ['void k_img_acts<1, 1, true, 2, 2, 2>'], undo_bytes(CONV2), 'mcn_img_acts_max_undo (imgActs + max-pool undo + relu gradient in one launch)'),
    'local3.bpropActs[0]': ('iau3', ['void k_img_acts<1, 1, true, 2, 2, 2>'], undo_bytes(LOCAL3), 'mcn_img_acts_max_undo'),
    'local4.bpropActs[0]': ('ia4', ['void k_img_acts<1, 1, true, 2, 4, 2>'], undo_bytes(LOCAL4), 'mcn_img_acts_max_undo, four waves per pixel'),
    'fc6.fprop[0]': ('fc6', ['void k_fc_nn_lds<10>'], FC6[0] + FC6[1] + FC6[2], 'forward product + bias + softmax stage 1'),
    'fc6.bpropActs[0]': ('fc6', ['void k_fc_tn_split<5, 6>', 'k_fc_tn_reduce'], FC6[0] + FC6[1] + FC6[2],
                         'K = 10575 split over 60 blocks per tile: partial slabs written and read once more'),
    'fc6.bpropWeights[0]': ('fc6', ['void k_fc_nt_lds<10>'], 2 * FC6[0] + FC6[1] + FC6[2], 'dW = mom * dW + eps/N dY X^T: dW read and written'),
    'conv1.bpropWeights[0]': ('wa1', ['void k_conv_wgrad_strip<3, 1>', 'k_reduce_chunks'], CONV1[0] + CONV1[1] + CONV1[2],
                              'kbench runs the reduced entry point; in the single-GPU step the chunk sum rides in the update launch'),
    'conv2.bpropWeights[0]': ('wa2', ['void k_weight_acts<2, 2, 2>', 'k_reduce_chunks'], CONV2[0] + CONV2[1] + CONV2[2], ''),
    'local3.bpropWeights[0]': ('wa3', ['void k_weight_acts<3, 1, 2>'], LOCAL3[0] + LOCAL3[1] + 2 * LOCAL3[2], 'inc = mom * inc + eps/N g: the increments are read and written'),
    'local4.bpropWeights[0]': ('wa4', ['void k_weight_acts<1, 1, 2>'], LOCAL4[0] + LOCAL4[1] + 2 * LOCAL4[2], ''),
    'local3.fprop[0]': ('fa', ['void k_filter_acts<3, 1, 9, true, true, 8>'], sum(LOCAL3), ''),
    'local4.fprop[0]': ('fa', ['void k_filter_acts<2, 1, 6, true, true, 8>'], sum(LOCAL4), ''),
    # DeepID2+ (tools/kbench.py '+' shapes): the round-3 kernels
    'deepid2plus local3.fprop[0]': ('fa', ['void k_filter_acts<4, 1, 4, true, true, 4>'], sum(LOCAL3P), '64 filters x 64 images per block, K split over four waves: each image half reads the module\'s weight slab'),
    'deepid2plus local3.bpropWeights[0]': ('wa3', ['void k_weight_acts_wide<3>'], LOCAL3P[0] + LOCAL3P[1] + 2 * LOCAL3P[2], 'k_weight_acts_wide; increments read and written'),
    'deepid2plus conv2.bpropWeights[0]': ('wa2p', ['void k_weight_acts_wide<3>', 'k_reduce_chunks'], sum(CONV2P), 'k_weight_acts_wide over 144 chunks of 4 modules (21 MB of partials written and read once) + k_reduce_chunks'),
    'deepid2plus local3.bpropActs[0]': ('ian', ['void k_img_acts<2, 1, true, 1, 1, 2>'], undo_bytes(LOCAL3P), 'one wave per (pixel, 32 channels, 64 images), a block = the four channel tiles of one pixel (img_acts_wide = 0, what the layer graph runs)'),
    'deepid2plus conv2.bpropActs[0]': ('ian2', ['void k_img_acts<2, 1, true, 1, 1, 2>'], undo_bytes(CONV2P), 'one wave per (pixel, 32 channels, 64 images) (img_acts_wide = 0)'),
}


def averages(path):
    acc, cnt = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(path)):
        acc[r['Kernel_Name']] += float(r['Counter_Value'])
        cnt[r['Kernel_Name']] += 1
    return {k: acc[k] / cnt[k] for k in acc}


def symbols(blob, prefixes):
    base = {re.match(r'(?:void )?(\w+)', p).group(1) for p in prefixes}
    found = []
    for b in base:
        cands = sorted(set(re.findall(rb'_Z\d+' + b.encode() + rb'\w*', blob)))
        dem = subprocess.run(['c++filt'] + [c.decode() for c in cands], capture_output=True, text=True).stdout.split('\n')
        for c, d in zip(cands, dem):
            if any(d.startswith(p) or ('void ' + d).startswith(p) for p in prefixes):
                found.append(c.decode())
    return found


def main():
    blob = open(os.path.join(ROOT, 'cuda-convnet_amd', 'libmiconvnet.so'), 'rb').read()
    out = {}
    for op, (tag, prefixes, algo, note) in OPS.items():
        f, w = averages(os.path.join(OUT, 'pmc_%s_fetch.csv' % tag)), averages(os.path.join(OUT, 'pmc_%s_write.csv' % tag))
        kernels, total = {}, 0.0
        for p in prefixes:
            names = [k for k in f if k.startswith(p)]
            assert len(names) == 1, (op, p, names)
            k = names[0]
            kernels[p] = {'FETCH_SIZE_KB': round(f[k], 1), 'WRITE_SIZE_KB': round(w[k], 1)}
            total += (2 * f[k] + w[k]) * 1024
        syms = symbols(blob, prefixes)
        assert len(syms) >= len(prefixes), (op, syms)
        out[op] = {'kernels': kernels, 'kernel_symbols': syms,
                   'correction': 'gfx950: FETCH_SIZE counts 64 B per 128 B request of wide reads -> x2; WRITE_SIZE exact; KB',
                   'traffic_bytes_per_launch': int(total), 'algorithmic_bytes_per_launch': int(algo),
                   'traffic_over_algorithmic': round(total / algo, 2),
                   'collected_with': 'pmc_round3.sh: pmc_passes.sh %s (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs over tools/kbench.py)' % tag}
        if note:
            out[op]['note'] = note
        for ps in ('fetch', 'write', 'sq1', 'sq2'):
            src = os.path.join(OUT, 'pmc_%s_%s.csv' % (tag, ps))
            if os.path.exists(src):
                shutil.copy(src, os.path.join(PROF, '%s_pmc_%s_%s.csv' % (ROUND, tag, ps)))
    json.dump(out, open(os.path.join(PROF, ROUND + '_pmc_traffic_by_op.json'), 'w'), indent=1)
    for op, r in out.items():
        print('%-24s %6.1f MB / launch  (algorithmic %6.1f MB, x%.2f)  %s' % (op, r['traffic_bytes_per_launch'] / 1e6, r['algorithmic_bytes_per_launch'] / 1e6,
                                                                               r['traffic_over_algorithmic'], ' '.join(r['kernel_symbols'])))


if __name__ == '__main__':
    sys.exit(main())
