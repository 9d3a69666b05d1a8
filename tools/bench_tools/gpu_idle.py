"""gpu_idle.py <rocprofv3 kernel-trace dir>: wall span of the dispatches, the time at least one kernel was running and the idle gaps between them
(how launch-bound the step is); usage on the GPU box:
  cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-extra
  python3 tools/bench_tools/gpu_idle.py /tmp/kt"""
import csv, glob, sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# the timed steps: the last 60 % of the dispatches (warm-up and set-up come first)
rows = rows[int(len(rows) * 0.4):]
span = rows[-1][1] - rows[0][0]
busy, end = 0, rows[0][0]
gaps = []
for s, e, _ in rows:
    if s > end:
        gaps.append(s - end)
        busy += e - s
        end = e
    elif e > end:
        busy += e - end
        end = e
gaps.sort()
print("dispatches %d  span %.2f ms  busy %.2f ms (%.1f %%)  idle %.2f ms in %d gaps; gaps > 20 us: %d totalling %.2f ms; median gap %.1f us" % (
    len(rows), span / 1e6, busy / 1e6, 100.0 * busy / span, (span - busy) / 1e6, len(gaps), sum(1 for g in gaps if g > 20000),
    sum(g for g in gaps if g > 20000) / 1e6, gaps[len(gaps) // 2] / 1e3 if gaps else 0.0))
if len(sys.argv) > 2:  # list the long gaps with the kernels around them
    end = rows[0][0]
    for i, (s, e, n) in enumerate(rows):
        if s > end + 20000:
            print("gap %.2f ms before dispatch %d  %s   (after %s)" % ((s - end) / 1e6, i, n[:60], rows[i - 1][2][:60]))
        end = max(end, e)
