"""Prints the top kernels of a rocprofv3 --kernel-trace --stats CSV directory."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 14]:
    print("%-100s calls %5s avg_us %10.1f total_ms %9.2f" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3,
                                                            float(r["TotalDurationNs"]) / 1e6))
