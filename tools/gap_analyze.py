import csv, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
name = sys.argv[2] if len(sys.argv) > 2 else "k_triv"
seq = [r for r in rows if name in r["Kernel_Name"]]
gaps = []
for a, b in zip(seq, seq[1:]):
    gaps.append((int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3)
import statistics
print(len(seq), "kernels; median gap", statistics.median(gaps), "us")
big = [(i, g) for i, g in enumerate(gaps) if g > 10]
print("gaps > 10 us:", len(big))
print([ (i, round(g,1)) for i, g in big[:40]])
