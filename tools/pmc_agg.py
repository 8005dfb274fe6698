import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
first = None
for r in rows:
    k = r["Kernel_Name"][:64]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if first is None: first = r["Counter_Name"]
    if r["Counter_Name"] == first: cnt[k] += 1
for k, v in agg.items():
    if any(s in k for s in sys.argv[2:]):
        n = max(cnt[k], 1)
        print(k[-40:], "calls", n, {c: round(x / n / 1e6, 2) for c, x in v.items()})
