"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py into HBM bytes per launch per kernel.

  python scripts/pmc_summary.py gpurun_out/pmc_fetch/f_counter_collection.csv \
                                gpurun_out/pmc_write/w_counter_collection.csv profiles/r01b

FETCH_SIZE / WRITE_SIZE are reported in KiB-units of 1024 B? -- rocprofv3 derives them as (requests x size)/1024, so
bytes = value * 1024.  gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts 128-byte read requests as 64 B
-> doubled; WRITE_SIZE is exact for 16-byte-per-lane stores and uncalibrated for the 4/8-byte-per-lane epilogue
stores of the conv kernels (kept as reported)."""
import collections
import csv
import json
import sys


def per_kernel(path, counter):
    agg = collections.defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r['Counter_Name'] == counter:
                agg[r['Kernel_Name']].append(float(r['Counter_Value']))
    return agg


def short(name):
    """Kernel names come demangled or mangled depending on the row: reduce both to the bare function name."""
    import re
    name = name.replace('(anonymous namespace)::', '').replace('void ', '')
    m = re.match(r'_ZN12_GLOBAL__N_1(\d+)', name)
    if m:
        start = m.end()
        base = name[start:start + int(m.group(1))]
        rest = name[start + int(m.group(1)):]
        targs = re.match(r'I(DF16b|f)?((?:L[ib]\d+E)*)E', rest)
        if targs:
            dt = {'DF16b': ['bf16'], 'f': ['float'], None: []}[targs.group(1)]
            ints = re.findall(r'L[ib](\d+)E', targs.group(2))
            return f"{base}<{', '.join(dt + ints)}>"
        return base
    return name.split('(')[0]


def main():
    fetch = per_kernel(sys.argv[1], 'FETCH_SIZE')
    write = per_kernel(sys.argv[2], 'WRITE_SIZE')
    prefix = sys.argv[3]
    merged = collections.defaultdict(lambda: [[], []])
    for k, v in fetch.items():
        merged[short(k)][0] += v
    for k, v in write.items():
        merged[short(k)][1] += v
    rows = []
    for k, (f, w) in merged.items():
        n = max(len(f), len(w))
        fb = 2 * 1024 * sum(f) / max(len(f), 1)
        wb = 1024 * sum(w) / max(len(w), 1)
        rows.append((k, n, fb, wb))
    rows.sort(key=lambda r: -(r[2] + r[3]) * r[1])
    with open(prefix + '_hbm_traffic_per_kernel.csv', 'w') as f:
        f.write('kernel,launches,read_bytes_per_launch(2xFETCH_SIZE),write_bytes_per_launch(WRITE_SIZE),total_GB_all_launches\n')
        for k, n, fb, wb in rows:
            f.write(f'"{k}",{n},{fb:.0f},{wb:.0f},{(fb + wb) * n / 1e9:.3f}\n')
    fam = [r for r in rows if r[0].startswith(('conv3x3_db_kernel', 'conv3x3_k32n64_kernel', 'conv3x3_k32_kernel'))]
    # the dominant conv3x3 kernel: the 16x16x32 kernel of the 128-channel n-tiles (bf16), else conv3x3_db_kernel (fp32)
    dom = 'conv3x3_k32_kernel' if any(r[0].startswith('conv3x3_k32_kernel') for r in rows) else 'conv3x3_db_kernel'
    conv = [r for r in rows if r[0].startswith(dom)]
    n = sum(r[1] for r in conv)
    tot = sum((r[2] + r[3]) * r[1] for r in conv)
    nf = sum(r[1] for r in fam)
    totf = sum((r[2] + r[3]) * r[1] for r in fam)
    out = {'kernel': dom, 'what': 'all instantiations: the fprop + dgrad launches with GEMM-N % 128 == 0', 'launches': n,
           'hbm_bytes_per_launch': tot / max(n, 1),
           'family': {'kernel': 'conv3x3_db_kernel + conv3x3_k32n64_kernel + conv3x3_k32_kernel', 'launches': nf,
                      'hbm_bytes_per_launch': totf / max(nf, 1)},
           'method': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --steps 1 '
                     '--warmup 1`; bytes = 2 x FETCH_SIZE + WRITE_SIZE (gfx950: 128-B read requests are tallied as 64 B)'}
    with open(prefix + '_conv_traffic.json', 'w') as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))
    for r in rows[:14]:
        print(f'{r[0][:60]:60s} x{r[1]:4d}  read {r[2] / 1e6:9.1f} MB  write {r[3] / 1e6:9.1f} MB')


if __name__ == '__main__':
    main()
