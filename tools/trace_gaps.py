"""Development aid: per-round timeline out of a rocprofv3 --kernel-trace CSV (tools/small_timeline.py under the
profiler): for the LAST sweep in the trace, the mean duration of every kernel of a round and the mean idle gap in front
of it.  usage: python tools/trace_gaps.py <kernel_trace.csv> [first-kernel-of-a-round substring]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
first = sys.argv[2] if len(sys.argv) > 2 else "k_dev_step"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ks = [(r["Kernel_Name"].split("(")[0].replace("void bioen::", "").replace("bioen::", ""), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
# rounds: from one `first` kernel to the next
idx = [i for i, k in enumerate(ks) if first in k[0]]
idx = idx[len(idx) // 2:]              # the later half: warm sweeps
dur = collections.defaultdict(list); gap = collections.defaultdict(list); rl = []
for a, b in zip(idx[:-1], idx[1:]):
    rl.append(ks[b][1] - ks[a][1])
    for j in range(a, b):
        name = ks[j][0][:40]
        dur[name].append(ks[j][2] - ks[j][1])
        gap[name].append(ks[j][1] - ks[j - 1][2] if j > 0 else 0)
print("rounds %d, mean round %.1f us" % (len(rl), sum(rl) / len(rl) / 1e3))
tot_d = tot_g = 0
for name in dur:
    n = len(dur[name]); d = sum(dur[name]) / len(rl) / 1e3; g = sum(gap[name]) / len(rl) / 1e3
    tot_d += d; tot_g += g
    print("  %-42s per round: %5.2f launches, busy %6.2f us, idle gap before %6.2f us" % (name, n / len(rl), d, g))
print("  total busy %.1f us, idle %.1f us" % (tot_d, tot_g))
