"""Turn what scripts/profile_round.sh left under gpurun_out/<tag>/ into the committed summaries under profiles/:

  python scripts/profile_collect.py r02

  profiles/<tag>_bench_default.json            the driver's command, as printed
  profiles/<tag>_bench_under_rocprof.json      the same command under rocprofv3 --kernel-trace --stats
  profiles/<tag>_bench_kernel_stats.csv        per-kernel totals of that trace
  profiles/<tag>_mfma_util.json                matrix-pipe busy fraction per kernel family (counter pass)
  profiles/<tag>_hbm_traffic_per_kernel.csv, <tag>_conv_traffic.json   FETCH_SIZE / WRITE_SIZE passes
  profiles/<tag>_other_configs.jsonl           bench.py --config for BASELINE.json configs[2..4]
"""
import glob
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag = sys.argv[1]
    src = os.path.join(ROOT, 'gpurun_out', tag)
    dst = os.path.join(ROOT, 'profiles', tag)
    py = sys.executable
    for name in ('bench_default.json', 'bench_under_rocprof.json', 'other_configs.jsonl'):
        shutil.copy(os.path.join(src, name), f'{dst}_{name}')
    db = glob.glob(os.path.join(src, 'trace', '**', '*_results.db'), recursive=True)[0]
    subprocess.run([py, os.path.join(ROOT, 'scripts', 'rocpd_to_csv.py'), 'stats', db, f'{dst}_bench_kernel_stats.csv'], check=True)
    db = glob.glob(os.path.join(src, 'pmc_mfma', '**', '*_results.db'), recursive=True)[0]
    subprocess.run([py, os.path.join(ROOT, 'scripts', 'pmc_mfma_util.py'), db, f'{dst}_mfma_util.json'], check=True)
    f = glob.glob(os.path.join(src, 'pmc_fetch', '**', '*counter_collection.csv'), recursive=True)[0]
    w = glob.glob(os.path.join(src, 'pmc_write', '**', '*counter_collection.csv'), recursive=True)[0]
    subprocess.run([py, os.path.join(ROOT, 'scripts', 'pmc_summary.py'), f, w, dst], check=True)
    # stamp the two files bench.py quotes counters from with the source state they were measured on
    import json
    sha = subprocess.run(['git', '-C', ROOT, 'rev-parse', '--short', 'HEAD'], capture_output=True, text=True).stdout.strip()
    dirty = subprocess.run(['git', '-C', ROOT, 'status', '--porcelain', '--', 'pytorch-unet_amd', 'bench.py'],
                           capture_output=True, text=True).stdout.strip()
    for name in ('mfma_util.json', 'conv_traffic.json'):
        path = f'{dst}_{name}'
        doc = json.load(open(path))
        doc['commit'] = sha + ('+uncommitted changes' if dirty else '')
        json.dump(doc, open(path, 'w'), indent=1)


if __name__ == '__main__':
    main()
