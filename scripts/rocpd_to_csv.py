"""rocprofv3's default output on this image is a rocpd SQLite file; turn it into the two CSV shapes the other
scripts and profiles/ use.

  python scripts/rocpd_to_csv.py stats    gpurun_out/prof/x_results.db  profiles/rNN_bench_kernel_stats.csv
  python scripts/rocpd_to_csv.py counters gpurun_out/pmc/x_results.db   gpurun_out/pmc/x_counter_collection.csv
"""
import csv
import sqlite3
import sys


def main():
    mode, db, out = sys.argv[1:4]
    c = sqlite3.connect(db)
    with open(out, 'w', newline='') as f:
        w = csv.writer(f)
        if mode == 'stats':
            w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage'])
            for r in c.execute('select name, total_calls, total_duration, average, percentage from top_kernels'):
                w.writerow(r)
        else:
            w.writerow(['Kernel_Name', 'Counter_Name', 'Counter_Value'])
            for r in c.execute('select kernel_name, counter_name, value from counters_collection'):
                w.writerow(r)


if __name__ == '__main__':
    main()
