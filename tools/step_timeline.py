"""Timeline of one PER-DQN step from a rocprofv3 kernel trace (tools/kstats.sh): start / end / duration in us per kernel."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/stats/*/*kernel_trace.csv')[0]
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:60], r['Queue_Id']) for r in csv.DictReader(open(f)))
idx = [i for i, e in enumerate(ev) if 'per_sample_kernel' in e[2]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 40
t0 = ev[idx[k]][0]
for e in ev[idx[k]:idx[k + 1]]:
    print(f"{(e[0]-t0)/1e3:8.2f} {(e[1]-t0)/1e3:8.2f} {(e[1]-e[0])/1e3:7.2f} q{e[3]} {e[2]}")
print("step period", (ev[idx[k + 1]][0] - t0) / 1e3)
