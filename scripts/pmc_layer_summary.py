"""Compact summary of the counter passes scripts/pmc_layer.sh leaves in a directory:

    python scripts/pmc_layer_summary.py gpurun_out/<dir> [kernel-name substring] > profiles/<name>.json

Per kernel (launch-averaged): wave cycles, the share of them spent waiting, LDS activity and bank conflicts, matrix-pipe
busy share, VALU / LDS / SALU / VMEM instruction counts per MFMA-busy cycle, L2 hit rate and requests, duration."""
import collections
import csv
import glob
import json
import os
import re
import sys


def short(name):
    name = name.replace('(anonymous namespace)::', '').replace('void ', '')
    m = re.match(r'_ZN12_GLOBAL__N_1(\d+)', name)
    if m:
        return name[m.end():m.end() + int(m.group(1))]
    return name.split('(')[0]


def main():
    d = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else ''
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, '*_counter_collection.csv')):
        for r in csv.DictReader(open(f)):
            k = short(r['Kernel_Name'])
            if want in k:
                agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
    dur = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, 'trace_kernel_trace.csv')):
        for r in csv.DictReader(open(f)):
            k = short(r['Kernel_Name'])
            if want in k:
                dur[k].append((float(r['End_Timestamp']) - float(r['Start_Timestamp'])) / 1e3)
    out = {}
    for k, c in agg.items():
        a = {n: sum(v) / len(v) for n, v in c.items()}
        if a.get('SQ_WAVE_CYCLES', 0) < 1e6:
            continue                                     # packing / reduction helpers
        e = {'launches_sampled': len(c.get('SQ_WAVE_CYCLES', [])),
             'duration_us_unprofiled_trace': sum(dur[k]) / len(dur[k]) if dur.get(k) else None,
             'gui_active_cycles_sum_over_xcds': a.get('GRBM_GUI_ACTIVE')}
        wc = a.get('SQ_WAVE_CYCLES')
        if wc:
            for n in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_WAIT_INST_LDS', 'SQ_ACTIVE_INST_ANY'):
                if n in a:
                    e[n.lower() + '_per_wave_cycle'] = a[n] / wc
        # GRBM_GUI_ACTIVE comes summed over the 8 XCDs, SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs (pmc_mfma_util.py)
        if a.get('GRBM_GUI_ACTIVE') and 'SQ_VALU_MFMA_BUSY_CYCLES' in a:
            e['mfma_busy_fraction'] = a['SQ_VALU_MFMA_BUSY_CYCLES'] / (a['GRBM_GUI_ACTIVE'] / 8.0 * 1024)
        if 'SQ_LDS_IDX_ACTIVE' in a:
            e['lds_bank_conflict_per_lds_active'] = a.get('SQ_LDS_BANK_CONFLICT', 0.0) / max(a['SQ_LDS_IDX_ACTIVE'], 1.0)
        for n in ('SQ_INSTS_VALU', 'SQ_INSTS_LDS', 'SQ_INSTS_SALU', 'SQ_ACTIVE_INST_VMEM'):
            if n in a:
                e[n.lower()] = a[n]
        if 'TCC_HIT_sum' in a and 'TCC_MISS_sum' in a:
            e['l2_hit_rate'] = a['TCC_HIT_sum'] / max(a['TCC_HIT_sum'] + a['TCC_MISS_sum'], 1.0)
            e['l2_requests'] = a.get('TCC_REQ_sum')
            e['l2_fabric_read_requests'] = a.get('TCC_EA0_RDREQ_sum')
        out[k] = e
    json.dump(out, sys.stdout, indent=1)


if __name__ == '__main__':
    main()
