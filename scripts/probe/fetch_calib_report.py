"""FETCH_SIZE / WRITE_SIZE of scripts/probe/fetch_calib against the byte counts it asked for.
usage: python scripts/probe/fetch_calib_report.py <dir with fetch/ and write/ rocprofv3 outputs and calib.log>"""
import collections, csv, glob, os, sys
d = sys.argv[1]
exp = {}
for l in open(os.path.join(d, "calib.log")):
    if l.startswith("EXPECT"):
        _, k, req, lines = l.split()
        exp[k] = (int(req), int(lines))
val = collections.defaultdict(dict)
for counter, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    for f in glob.glob(os.path.join(d, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                k = r["Kernel_Name"].split("(")[0]
                val[k][counter] = val[k].get(counter, 0.0) + float(r["Counter_Value"])
print("%-12s %14s %14s %14s %14s | %9s %9s %9s" % ("kernel", "bytes asked", "64-B lines x64", "FETCH_SIZE KiB", "WRITE_SIZE KiB",
                                                   "F*1024/ask", "F*1024/lin", "W*1024/ask"))
for k, (req, lines) in exp.items():
    F, Wr = val.get(k, {}).get("FETCH_SIZE", 0.0), val.get(k, {}).get("WRITE_SIZE", 0.0)
    print("%-12s %14d %14d %14.0f %14.0f | %9.3f %9.3f %9.3f" % (k, req, lines, F, Wr, F * 1024 / req, F * 1024 / lines, Wr * 1024 / req))
print("reading: F*1024/ask = 0.5 reproduces the guide's half-count for wide coalesced reads; for the gathers the ratio against the "
      "64-B lines touched says what '2 x FETCH_SIZE' over- or under-states")
