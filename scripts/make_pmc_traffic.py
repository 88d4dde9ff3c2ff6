#!/usr/bin/env python3
"""profiles/pmc_traffic.json from the per-(kernel, grid) FETCH_SIZE / WRITE_SIZE rows of scripts/summarize_pmc.py
(separate --pmc passes of `bench.py --windows W --steps 1`, scripts/collect_profiles.sh).

    python scripts/make_pmc_traffic.py <pmc_fetch.csv> <pmc_write.csv> <windows W> [tag]

Per MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports half of the
bytes of wide coalesced reads, so it is doubled; WRITE_SIZE is taken as is.

For every kernel family the STEP'S OWN launch class is selected by grid size -- the largest grid of that kernel in the
run: the step's STFT covers all W windows (the guess bank's STFT and a small last conv layer have smaller grids), the
20 x 516 convolution launches of a chunk all share one grid -- and the figure is that class's sum / its dispatches.
(Round 2 divided a kernel's total by all its dispatches, mixing unequal launches; its file also kept stale families:
the output is rewritten from scratch every time.)"""
import csv, json, os, subprocess, sys

fetch, write, windows = sys.argv[1], sys.argv[2], int(sys.argv[3])
tag = sys.argv[4] if len(sys.argv) > 4 else ''
FAMILIES = {'conv_f16x3': 'conv_f16x3s_kernel<4, 16, 128, true',        # the largest direct class of a mode-3 step (5 x 8 images)
            'conv_mfma': 'conv_mfma_kernel<4, 16, 32, 32',
            'stft': 'stft_mag_kernel<2048, true', 'stft_mag_only': 'stft_mag_kernel<2048, false',
            'subtract': 'subtract_kernel', 'subtract_span': 'subtract_span_kernel', 'compress_bands': 'compress_bands_kernel',
            'cqt_window_max': 'cqt_blocks_kernel<false', 'cqt_window_max_mfma': 'cqt_max_mfma_kernel',
            # conv mode 3: the FFT-domain layers (fc_row_kernel<true> = inverse + epilogue [+ forward]: its launches of a
            # chain differ in what they read and write -- shortcut, spatial output -- and are averaged)
            'fc_gemm': 'fc_gemm_kernel', 'fc_row': 'fc_row_kernel<true, 2', 'fc_row_inregs': 'fc_row_kernel<true, 1',
            'fc_row_first': 'fc_row_kernel<false',
            # the packed-image form of the 10 x 64 layers (amt_fftpk.hip)
            'pk_gemm': 'pk_gemm_kernel', 'pk_row': 'pk_row_kernel<true, 2', 'pk_row_inregs': 'pk_row_kernel<true, 1',
            'pk_row_first': 'pk_row_kernel<false'}


def load(path):
    out = {}
    for r in csv.DictReader(open(path)):
        cols = list(r.keys())
        out[(r['kernel'], int(r['grid_size']))] = (int(r['dispatches']), float(r[cols[3]]))
    return out


def pick(table, pat):
    """The (kernel, grid) class with the largest grid among the kernels matching pat."""
    keys = [k for k in table if pat in k[0]]
    return max(keys, key=lambda k: k[1]) if keys else None


f, w = load(fetch), load(write)
res = {}
for fam, pat in FAMILIES.items():
    kf, kw = pick(f, pat), pick(w, pat)
    if not kf or not kw or kf[1] != kw[1]:
        continue
    (nf, sf), (nw, sw) = f[kf], w[kw]
    fk, wk = sf / nf, sw / nw
    res[fam] = {'kernel': kf[0].split('(')[0], 'grid_size': kf[1], 'dispatches_in_class': nf,
                'fetch_size_kb_per_launch': round(fk, 1), 'write_size_kb_per_launch': round(wk, 1),
                'windows_per_launch': windows,
                'hbm_bytes_per_launch': round((2 * fk + wk) * 1024, 1),
                'hbm_bytes_per_window': round((2 * fk + wk) * 1024 / windows, 1)}
try:
    commit = os.environ.get('AMT_HEAD_COMMIT') or subprocess.check_output(['git', 'rev-parse', '--short', 'HEAD'], cwd=os.path.dirname(os.path.abspath(__file__)),
                                     text=True).strip()
except Exception:
    commit = None
res['_provenance'] = {'source': [os.path.basename(fetch), os.path.basename(write)], 'profiles_dir': tag, 'commit': commit,
                      'note': 'FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of wide coalesced reads); WRITE_SIZE '
                              'as is; separate --pmc passes of bench.py --windows %d --steps 1; the step\'s own launch class '
                              '(largest grid) of each kernel' % windows}
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles', 'pmc_traffic.json')
json.dump(res, open(dst, 'w'), indent=1)
print(json.dumps(res, indent=1))
