"""MFMA utilisation per kernel family from one `rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES
SQ_INSTS_VALU_MFMA_MOPS_BF16 -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline` pass (rocpd database).

  python scripts/pmc_mfma_util.py gpurun_out/pmc_mfma/m_results.db profiles/rNN_mfma_util.json

GRBM_GUI_ACTIVE comes summed over the 8 XCDs, SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs:
  util          = MFMA_BUSY / ((GUI_ACTIVE / 8) * 256 CUs * 4 SIMDs)      fraction of cycles the matrix pipes are busy
  flop_per_clk  = MOPS_BF16 * 512 / (GUI_ACTIVE / 8)                       against 1024 SIMDs * 1017 = 1.04 M at peak
"""
import collections
import json
import re
import sqlite3
import sys


def main():
    db, out = sys.argv[1:3]
    c = sqlite3.connect(db)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    fam = r'(conv3x3_db_kernel|conv3x3_k32n64_kernel|conv3x3_k32_kernel|wgrad5_bf16_kernel|wgrad4_bf16_kernel|wgrad3_bf16_kernel|wgradT_bf16_kernel|pw_db_kernel|wgrad_bf16_kernel)'
    for name, cn, v in c.execute('select kernel_name, counter_name, value from counters_collection'):
        m = re.search(fam, name)
        if m:
            agg[m.group(1)][cn].append(v)
    res = {}
    for k, d in agg.items():
        def avg(x):
            return sum(d[x]) / max(len(d[x]), 1)
        gui = avg('GRBM_GUI_ACTIVE') / 8.0
        res[k] = {'launches': len(d['GRBM_GUI_ACTIVE']), 'cycles_per_launch': gui,
                  'mfma_busy_fraction': avg('SQ_VALU_MFMA_BUSY_CYCLES') / (gui * 256 * 4),
                  'flop_per_cycle': avg('SQ_INSTS_VALU_MFMA_MOPS_BF16') * 512 / gui,
                  'fraction_of_at_clock_peak': avg('SQ_INSTS_VALU_MFMA_MOPS_BF16') * 512 / gui / (1024 * 1017.0)}
    # the conv3x3 fprop + dgrad family as bench.py's roofline object means it: every instantiation, launch-weighted
    conv = [v for k, v in res.items() if k.startswith('conv3x3_')]
    if conv:
        n = sum(v['launches'] for v in conv)
        cyc = sum(v['cycles_per_launch'] * v['launches'] for v in conv)
        res['conv3x3_fprop_dgrad'] = {
            'launches': n, 'cycles_per_launch': cyc / n,
            'mfma_busy_fraction': sum(v['mfma_busy_fraction'] * v['cycles_per_launch'] * v['launches'] for v in conv) / cyc,
            'fraction_of_at_clock_peak': sum(v['fraction_of_at_clock_peak'] * v['cycles_per_launch'] * v['launches']
                                             for v in conv) / cyc}
    with open(out, 'w') as f:
        json.dump(res, f, indent=1)
    for k, v in res.items():
        print(k, {a: round(b, 3) if isinstance(b, float) else b for a, b in v.items()})


if __name__ == '__main__':
    main()
