"""Per-kernel bandwidth table of the headline command from the committed profiles: average launch duration (rocprofv3
kernel trace), algorithmic bytes per launch (DESIGN.md section 3 formulas; the workload's P, R from the bench line) and
the PMC traffic per launch.  usage: python scripts/roofline_table.py [R_per_view]   -> markdown on stdout"""
import csv, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P, K, V = 800000, 4, 12
R = float(sys.argv[1]) if len(sys.argv) > 1 else 11767021.0
T = 120 * 68
pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))["bytes_per_launch"]
avg = {}
for row in csv.DictReader(open(os.path.join(ROOT, "profiles", "r02_bench_kernel_stats.csv"))):
    avg[row["Name"]] = float(row["AverageNs"]) * 1e-3

def us(sub):
    return next(v for k, v in avg.items() if sub in k)

def tr(sub):
    return next(v for k, v in pmc.items() if sub in k)

rows = [
    # name, key in csv/pmc, algorithmic bytes per launch, what they are
    ("k_preprocess<1>", "k_preprocess<", (44 + 12 * K) * P + V * P * (75 + 64),
     "inputs once; per view a 64-B Splat line, key, count, radius + the 64-B gradient record cleared"),
    ("k_radix_hist<u32> (depth, x4)", "k_radix_hist<unsigned int>", 4 * P * V, "keys read"),
    ("k_radix_scatter<8,u32> (depth, x4)", "k_radix_scatter<8", 16 * P * V, "key+id read, key+id written"),
    ("k_duplicate<u16>", "k_duplicate<", V * (20 * P + 6 * R), "16-B Splat slice + id per Gaussian; 2-B key + 4-B id per pair"),
    ("k_radix_hist<u16> (tile, x2)", "k_radix_hist<unsigned short>", 2 * R * V, "keys read"),
    ("k_radix_scatter<7,u16>", "k_radix_scatter<7", 12 * R * V, "6 B read + 6 B written per pair"),
    ("k_radix_scatter<6,u16>", "k_radix_scatter<6", 12 * R * V, "same"),
    ("k_preprocess_backward<1>", "k_preprocess_backward<", (44 + 12 * K) * P + V * P * (48 + 44 + 5) + P * (12 * 13 + 4 + 12 + 12 + 16 + 12 + 24),
     "inputs once; per view 48 B of the Splat line, the 44-B gradient record, radius + clamp flags; every gradient array once (all 13 dL_dsh rows)"),
]
print("| kernel | µs per launch (12 views) | algorithmic MB | algorithmic TB/s | of 8 TB/s | measured MB (2·FETCH+WRITE) | measured TB/s | algorithmic bytes are |")
print("|---|---|---|---|---|---|---|---|")
for name, key, b, what in rows:
    t = us(key)
    m = tr(key)
    print("| `%s` | %.0f | %.0f | %.2f | %.2f | %.0f | %.2f | %s |" % (name, t, b / 1e6, b / t / 1e6, b / t / 1e6 / 8.0, m / 1e6, m / t / 1e6, what))
