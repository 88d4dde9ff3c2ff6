#!/usr/bin/env python3
"""profiles/pmc_traffic.json from the per-kernel FETCH_SIZE / WRITE_SIZE sums of
scripts/collect_profiles.sh (separate --pmc passes of `bench.py --windows 256 --steps 1`).

    python scripts/make_pmc_traffic.py <pmc_fetch.csv> <pmc_write.csv> [windows]

Per MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports half
of the bytes of wide coalesced reads, so it is doubled; WRITE_SIZE is taken as is.  The
dominant dispatch class of each kernel family (largest per-dispatch traffic) is reported."""
import csv, json, os, sys

fetch, write = sys.argv[1], sys.argv[2]
windows = int(sys.argv[3]) if len(sys.argv) > 3 else 256
FAMILIES = {'conv_f16x3': 'conv_f16x3s_kernel<4, 16, 32, false>',
            'conv_bf16x6': 'conv_bf16x6_kernel<4, 16, 32, 32, false>',
            'conv_mfma': 'conv_mfma_kernel<4, 16, 32, 32',
            'stft': 'stft_mag_kernel<2048, true>', 'stft_mag_only': 'stft_mag_kernel<2048, false>',
            'subtract': 'subtract_kernel'}


def load(path):
    out = {}
    for r in csv.DictReader(open(path)):
        cols = list(r.keys())
        out[r['kernel']] = (int(r['dispatches']), float(r[cols[3]]))
    return out


f, w = load(fetch), load(write)
res = {}
for fam, pat in FAMILIES.items():
    kf = [k for k in f if pat in k]
    kw = [k for k in w if pat in k]
    if not kf or not kw:
        continue
    fk, wk = f[kf[0]][1], w[kw[0]][1]
    res[fam] = {
        'fetch_size_kb_per_dispatch': round(fk, 1), 'write_size_kb_per_dispatch': round(wk, 1),
        'windows_per_dispatch': windows,
        'hbm_bytes_per_window_per_launch': round((2 * fk + wk) * 1024 / windows, 1),
        'note': 'FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of wide coalesced reads); '
                'WRITE_SIZE as is; separate --pmc passes, bench.py --windows %d --steps 1' % windows}
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles', 'pmc_traffic.json')
old = json.load(open(dst)) if os.path.exists(dst) else {}
old.update(res)
json.dump(old, open(dst, 'w'), indent=1)
print(json.dumps(res, indent=1))
